#!/usr/bin/env python3
"""Known answers on the reference's OWN random test vectors.

The reference's int8 / uint8 kernel tests (tests/unit/test_spaces.cpp:1574-1600, 1658-1690, 1765-1800, 1854-1880, 1939-1965,
2022-2050; dims 32 .. 129) fill v1 / v2 with tests/utils/tests_utils.h:25-49 -- std::mt19937(seed) through
std::uniform_int_distribution<int16_t / uint16_t>, seeds 123 and 1234 -- and ASSERT_EQ every SIMD tier against the scalar
kernel.  This script restates that generator (MT19937 as published; libstdc++'s distribution over a 2^32 engine with a 256-wide
range takes the top byte of each draw: bits/uniform_int_dist.h, Lemire's multiply-shift with a zero rejection threshold) and
writes, per (type, dim), the two vectors' seeds and the EXACT answers computed here in integer arithmetic: sum (a-b)^2,
1 - sum a b, and for Cosine the float32 expression of IP.cpp:264-271.  Nothing from the reference is imported or copied.

    python tests/golden/make_ref_random_kats.py > tests/golden/kat_ref_random_ints.json
"""
import json

import numpy as np


def mt19937(seed, n):
    """n raw 32-bit draws of std::mt19937(seed) (Matsumoto & Nishimura, init_genrand)"""
    mt = [0] * 624
    mt[0] = seed & 0xFFFFFFFF
    for i in range(1, 624):
        mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
    out, idx = [], 624
    while len(out) < n:
        if idx >= 624:
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            idx = 0
        y = mt[idx]
        idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        out.append(y & 0xFFFFFFFF)
    return out


def populate(seed, dim, signed):
    top = np.array([r >> 24 for r in mt19937(seed, dim)], dtype=np.int64)
    return (top - 128) if signed else top


def main():
    cases = []
    for typ, signed in (("i8", True), ("u8", False)):
        for dim in list(range(32, 130)) + [512, 513, 544]:
            a, b = populate(123, dim, signed), populate(1234, dim, signed)
            dot = int(np.sum(a * b))
            na = np.float32(np.sqrt(np.float32(np.sum(a * a))))   # normalize_naive.h:81-88: float norm = sqrt(float(sum))... see test
            nb = np.float32(np.sqrt(np.float32(np.sum(b * b))))
            cases.append({"type": typ, "dim": dim, "seed_a": 123, "seed_b": 1234,
                          "head_a": [int(x) for x in a[:4]], "head_b": [int(x) for x in b[:4]],
                          "l2": int(np.sum((a - b) ** 2)), "ip": 1 - dot, "dot": dot,
                          "sum_sq_a": int(np.sum(a * a)), "sum_sq_b": int(np.sum(b * b))})
    print(json.dumps({"source": "tests/unit/test_spaces.cpp:1574-2050 inputs (tests/utils/tests_utils.h:25-49), exact integer answers",
                      "cases": cases}, indent=0))


if __name__ == "__main__":
    main()
