"""Makes tests/golden/ref_scalar_random.json: outputs of the REFERENCE'S OWN CODE on seeded random inputs.

    sh oracle/build_ref.sh                       # g++ on /root/reference's L2.cpp, IP.cpp, vecsim_malloc.cpp + our driver
    python tests/golden/make_ref_scalar_random.py

Every value in the fixture left the reference's compiled functions (oracle/ref_driver.cpp only forwards): the scalar
distance kernels of all six types x L2 / IP / Cosine (L2.cpp:76-174, IP.cpp:185-286), the normalisation templates
(normalize_naive.h:24-88), the bf16 / fp16 conversions (types/bfloat16.h, types/float16.h; fp16 -> fp32 and
bf16 -> fp32 over all 65 536 inputs), the scalar SQ8 kernels (IP.cpp:34-183, L2.cpp:30-74,185-201) and the Flat top-k
loop over the reference's containers (vecsim_stl.h:63-83, updatable_heap.h:20-113).  Inputs are seeds
(tests/golden/refgen.py), outputs are IEEE bit patterns / sha256 digests.  This needs /root/reference and therefore
runs in the build container only; the fixture is what travels.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import refgen  # noqa: E402
from oracle import vsref  # noqa: E402


def main():
    vsref.build()
    fx = refgen.compute_all(vsref)
    fx["_made_by"] = "tests/golden/make_ref_scalar_random.py over oracle/_ref/libvsref.so (reference TUs compiled with g++ -O3)"
    path = os.path.join(HERE, "ref_scalar_random.json")
    with open(path, "w") as f:
        json.dump(fx, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes;", {k: len(v) for k, v in fx.items() if isinstance(v, (list, dict))})


if __name__ == "__main__":
    main()
