#!/usr/bin/env python3
"""Do the four weight-gradient GEMMs of a DiT-XL/2 block (144 / 108 / 36 / 144 workgroups of 192x192 on 256 CUs) scale when they run
CONCURRENTLY on separate streams?  serial vs two pairs vs all four at once; operands rotated through > 256 MB so that they are cold
in the Infinity Cache, as in the step.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"; M, D, F = 8192, 1152, 4608
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)
NSET = 6                                   # 6 x 170 MB of dY + X per set: every launch sees cold operands
sets = []
for _ in range(NSET):
    sets.append(dict(fc2=(rnd(M, D), rnd(M, F), D, F), fc1=(rnd(M, F), rnd(M, D), F, D), proj=(rnd(M, D), rnd(M, D), D, D), qkv=(rnd(M, 3 * D), rnd(M, D), 3 * D, D)))
outs = {k: torch.empty(v[2], v[3], dtype=torch.float32, device=DEV) for k, v in sets[0].items()}
streams = [torch.cuda.Stream() for _ in range(4)]
def wg(s, name):
    dY, X, N, K = s[name]
    ops.gemm(dY, X, N, K, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=outs[name])
def run(groups, reps=12):
    """groups: list of lists of names; the names of a group run concurrently (one stream each), groups one after another"""
    cur = torch.cuda.current_stream()
    def once(i):
        s = sets[i % NSET]
        for grp in groups:
            if len(grp) == 1:
                wg(s, grp[0]); continue
            for st in streams[:len(grp)]: st.wait_stream(cur)
            for st, name in zip(streams, grp):
                with torch.cuda.stream(st): wg(s, name)
            for st in streams[:len(grp)]: cur.wait_stream(st)
    for i in range(3): once(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): once(i + 3)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for label, groups in (("serial fc2, fc1, proj, qkv", [["fc2"], ["fc1"], ["proj"], ["qkv"]]),
                      ("pairs (fc1 + qkv), (fc2 + proj)", [["fc1", "qkv"], ["fc2", "proj"]]),
                      ("pairs (fc1 + fc2), (qkv + proj)", [["fc1", "fc2"], ["qkv", "proj"]]),
                      ("all four at once", [["fc1", "fc2", "qkv", "proj"]]),
                      ("each alone: fc1", [["fc1"]]), ("each alone: qkv", [["qkv"]]), ("each alone: proj", [["proj"]])):
    print(f"{label:40s} {run(groups):8.1f} us per block", flush=True)
