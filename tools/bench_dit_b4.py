#!/usr/bin/env python3
"""BASELINE config 2 alone (DiT-B/4 256 px, batch 32) through bench.py's own leg: for rocprofv3 runs of that configuration. GPU only."""
import sys, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_lib
ab_lib.select()
import torch, bench
torch.zeros(1, device="cuda:0")
r = bench._dit_leg("DiT-B/4", 32, 32, "cuda:0", False, 20, 5)
print(json.dumps(r))
