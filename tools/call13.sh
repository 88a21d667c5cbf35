#!/bin/bash
set -o pipefail
python3 tools/diag_geluq.py 2>&1 | tail -8
echo "=== soak 12 x 2, product"; python3 tools/soak_repro.py 12 2>&1 | tail -2
echo "=== soak 12 x 2, nogq"; python3 - <<'P' 2>&1 | tail -2
import os, sys
sys.path.insert(0, "tools"); os.environ["SFRON_LIB_NAME"] = "libsfron_nogq.so"
import ab_lib; ab_lib.select()
import soak_repro as m
a = m.run(12); b = m.run(12)
print("nogq:", "BIT-IDENTICAL" if a == b else ("DIFFERENT", a, b))
P
echo "=== soak 12 x 2, product again"; python3 tools/soak_repro.py 12 2>&1 | tail -2
