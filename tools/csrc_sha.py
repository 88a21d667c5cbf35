#!/usr/bin/env python3
"""sha256 (first 16 hex) over the HIP sources + headers the library is built from: what a PMC traffic figure under profiles/ is valid
for.  bench.py prints `traffic` only when the figure's hash equals the tree's."""
import hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def csrc_sha():
    d = os.path.join(ROOT, "unified-unlearning-w-remain-geometry_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)) + ["../../include/sfron.h"]:
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]
if __name__ == "__main__":
    print(csrc_sha())
