#!/usr/bin/env python3
"""bench.py against the DEBUG-KNOB build of the library (`make -C .../csrc dbg` -> libsfron_dbg.so), for same-box A-B runs:
    SFRON_ABLATE=32 python tools/bench_dbg.py --steps 10 --no-cpu-baseline     # bit 5: bias gradients by column-sum launches
The product library reads no environment variables; only this build does."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sfron  # noqa: E402,F401
from sfron import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libsfron_dbg.so")
if os.environ.get("SFRON_ATTN_BWD_FORM"):          # 2 = the two-kernel attention backward (process-wide test hook)
    import torch
    torch.zeros(1, device=f"cuda:{os.environ.get('LOCAL_RANK', '0')}")      # HIP runtime up (through torch) before the library loads
    _lib.lib().sfron_attn_bwd_form(int(os.environ["SFRON_ATTN_BWD_FORM"]))
import bench  # noqa: E402
bench.main()
