#!/usr/bin/env python3
"""Generates tests/golden/abi_layout.json by compiling abi_probe.c against the REFERENCE header
(/root/reference/src/VecSim/vec_sim_common.h).  Run in the build container only; the fixture is
committed so tests never need /root/reference."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/VecSim/vec_sim_common.h"


def probe(header):
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "probe")
        subprocess.run(["gcc", "-std=gnu11", "-DVECSIM_COMMON_HEADER=\"%s\"" % header, "-o", exe,
                        os.path.join(HERE, "abi_probe.c")], check=True)
        return subprocess.run([exe], check=True, capture_output=True, text=True).stdout


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference not mounted")
    with open(os.path.join(HERE, "abi_layout.json"), "w") as f:
        f.write(probe(REF))
    print("wrote abi_layout.json")
