#!/usr/bin/env python3
"""GPU: the per-tensor update cosines / norm ratios of the two reference-trajectory tests (tests/test_gpu_reference_fixtures.py), all
of them, worst first -- what the bounds DIT_UPDATE_COS_MIN / DDPM_UPDATE_COS_MIN are set from."""
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import test_gpu_reference_fixtures as T  # noqa: E402

NROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for name, fn in (("dit", T.test_dit_sfron_trajectory_vs_reference_fixture), ("ddpm", T.test_ddpm_sfron_trajectory_vs_reference_fixture),
                 ("sd_xattn", lambda: T.test_sd_nsfw_removal_trajectory_vs_reference_fixture("xattn")),
                 ("sd_full", lambda: T.test_sd_nsfw_removal_trajectory_vs_reference_fixture("full"))):
    try:
        fn()
        print(f"== {name}: test passed")
    except AssertionError as e:
        print(f"== {name}: assertion: {str(e)[:300]}")
    cs = T._LAST.get(name, {})
    rows = sorted(cs.items(), key=lambda kv: kv[1][0])
    print(f"{name}: {len(rows)} tensors; min cosine {rows[0][1][0]:.4f}; max |norm ratio - 1| {max(abs(v[1] - 1) for v in cs.values()):.4f}; "
          f"min ref rms/lr {min(v[2] for v in cs.values()):.3f}")
    gn = {}
    if name.startswith("sd_"):            # the reference's own gradient norms of the two stages (fixture), relative to the median tensor
        import numpy as np
        U = np.load(os.path.join(T.GOLD, "sd_unet_updates.npz"))
        m = name[3:]
        names = [str(x) for x in U[m + "::names"]]
        gf, gr = U[m + "::gnorm_forget"], U[m + "::gnorm_remain"]
        gn = {n: (a / float(np.median(gf)), b / float(np.median(gr))) for n, a, b in zip(names, gf, gr)}
    shown = set()
    for n, (c, r, rms) in rows[:NROWS]:
        shown.add(n)
        extra = f"  |g_forget| / median {gn[n][0]:.2e}  |g_remain| / median {gn[n][1]:.2e}" if n in gn else ""
        print(f"   {n:50s} cos {c:.4f}  norm ratio {r:.4f}  ref rms/lr {rms:.3f}{extra}")
    for n, (c, r, rms) in sorted(cs.items(), key=lambda kv: -abs(kv[1][1] - 1.0))[:NROWS]:
        if n not in shown and abs(r - 1.0) > 0.03:
            extra = f"  |g_forget| / median {gn[n][0]:.2e}  |g_remain| / median {gn[n][1]:.2e}" if n in gn else ""
            print(f"   {n:50s} cos {c:.4f}  norm ratio {r:.4f}  ref rms/lr {rms:.3f}{extra}   (by norm ratio)")
