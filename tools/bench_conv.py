#!/usr/bin/env python3
"""Micro-benchmark of the convolution / Linear products of the LDM UNet at batch 8 (and the DDPM U-Net at batch 64) through the C ABI:
TFLOP/s per shape for forward, input gradient and weight gradient.   python tools/bench_conv.py [--reps 20]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
from sfron import _lib, unet
if os.environ.get("SFRON_LOADER_WAVES"):            # A-B knob: 10 = convolution tiles in the shared-wave form
    from sfron import _lib as _L; _L.lib().sfron_gemm_loader_waves(int(os.environ["SFRON_LOADER_WAVES"]))
from sfron._lib import check, ptr, stream_ptr
L = _lib.lib()
DEV = "cuda"
torch.zeros(1, device=DEV)


def timed(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3        # us


def conv_case(name, B, H, ci, co, stride=1, up=0):
    ho = H // 2 if stride == 2 else (2 * H if up else H)
    x = torch.randn(B * H * H, ci, device=DEV).to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, device=DEV) * 0.05
    wf = torch.empty(co * 9 * ci, dtype=torch.bfloat16, device=DEV); wd = torch.empty(ci * 9 * co, dtype=torch.bfloat16, device=DEV)
    check(L.sfron_conv_wprep(ptr(w), co, ci, 9, co, ci, ptr(wf), ptr(wd), stream_ptr()), "wprep")
    rows = B * ho * ho
    out = torch.empty(rows, co, dtype=torch.float32, device=DEV)
    pad = 0 if stride == 2 else 1
    d = unet._conv_desc(B, H, H, ci, ho, ho, co, 9, stride, pad, up, 0, out_f32=out, ld_out=co)
    flops = 2.0 * rows * co * 9 * ci
    t_f = timed(lambda: check(L.sfron_conv_fwd(ctypes.byref(d), ptr(x), ptr(wf), stream_ptr()), "fwd"), a.reps)
    dy = torch.randn(rows, co, device=DEV).to(torch.bfloat16)
    if stride == 2:
        ds = torch.empty(B * H * H, ci, dtype=torch.float32, device=DEV)
        dd = unet._conv_desc(B, ho, ho, co, H, H, ci, 9, 1, 2, 0, 1, out_f32=ds, ld_out=ci)
    else:
        ds = torch.empty(rows, ci, dtype=torch.float32, device=DEV)
        dd = unet._conv_desc(B, ho, ho, co, ho, ho, ci, 9, 1, 1, 0, 0, out_f32=ds, ld_out=ci)
    t_d = timed(lambda: check(L.sfron_conv_fwd(ctypes.byref(dd), ptr(dy), ptr(wd), stream_ptr()), "dgrad"), a.reps)
    wdsc = unet._conv_desc(B, H, H, ci, ho, ho, co, 9, stride, pad, up, 0)
    nsl = L.sfron_conv_wgrad_splits(ctypes.byref(wdsc))
    dwg = torch.empty(nsl * co * 9 * ci, dtype=torch.float32, device=DEV)
    dw = torch.empty(co, ci, 3, 3, dtype=torch.float32, device=DEV)
    def wg():
        check(L.sfron_conv_wgrad(ctypes.byref(wdsc), ptr(dy), co, ptr(x), ptr(dwg), stream_ptr()), "wgrad")
        check(L.sfron_conv_wgrad_scatter(ptr(dwg), co, ci, 9, ci, nsl, co * 9 * ci, ptr(dw), stream_ptr()), "scatter")
    t_w = timed(wg, a.reps)
    print(f"conv {name:28s} M={rows:6d} N={co:5d} K={9 * ci:6d}: fwd {t_f:8.1f} us {flops / t_f / 1e6:6.0f} TF | dgrad {t_d:8.1f} us "
          f"{flops * (2 if up else 1) / (1 if stride == 1 else 1) / t_d / 1e6 if stride == 1 else flops * 4 / t_d / 1e6:6.0f} TF(issued) | wgrad {t_w:8.1f} us {flops / t_w / 1e6:6.0f} TF (splits {nsl})",
          flush=True)


def lin_case(name, M, N, K):
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16); W = torch.randn(N, K, device=DEV).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.float32, device=DEV)
    flops = 2.0 * M * N * K
    t_f = timed(lambda: unet.bgemm(A, W, M, N, K, lda=K, ldb=K, c_f32=c, ldc=N), a.reps)
    dY = torch.randn(M, N, device=DEV).to(torch.bfloat16); dx = torch.empty(M, K, dtype=torch.float32, device=DEV)
    t_d = timed(lambda: unet.bgemm(dY, W, M, K, N, lda=N, ldb=K, b_t=True, c_f32=dx, ldc=K), a.reps)
    dW = torch.empty(N, K, dtype=torch.float32, device=DEV)
    t_w = timed(lambda: unet.bgemm(dY, A, N, K, M, lda=N, ldb=K, a_t=True, b_t=True, c_f32=dW, ldc=K), a.reps)
    print(f"lin  {name:28s} M={M:6d} N={N:5d} K={K:6d}: fwd {t_f:8.1f} us {flops / t_f / 1e6:6.0f} TF | dgrad {t_d:8.1f} us {flops / t_d / 1e6:6.0f} TF | "
          f"wgrad {t_w:8.1f} us {flops / t_w / 1e6:6.0f} TF", flush=True)


B = 8
conv_case("sd 320->320 @64", B, 64, 320, 320)
conv_case("sd 320->320 @64 stride2", B, 64, 320, 320, stride=2)
conv_case("sd 320->640 @32", B, 32, 320, 640)
conv_case("sd 640->640 @32", B, 32, 640, 640)
conv_case("sd 640->1280 @16", B, 16, 640, 1280)
conv_case("sd 1280->1280 @16", B, 16, 1280, 1280)
conv_case("sd 1280->1280 @8", B, 8, 1280, 1280)
conv_case("sd 2560->1280 @16", B, 16, 2560, 1280)
conv_case("sd 960->320 @64", B, 64, 960, 320)
conv_case("sd 640->640 up @32->64", B, 32, 640, 640, up=1)
lin_case("sd attn proj 320 @64", B * 4096, 320, 320)
lin_case("sd ff geglu 320->2560 @64", B * 4096, 2560, 320)
lin_case("sd ff out 1280->320 @64", B * 4096, 320, 1280)
lin_case("sd ff geglu 640->5120 @32", B * 1024, 5120, 640)
lin_case("sd ff geglu 1280->10240 @16", B * 256, 10240, 1280)
conv_case("ddpm 128->128 @32 b64", 64, 32, 128, 128)
conv_case("ddpm 256->256 @16 b64", 64, 16, 256, 256)
conv_case("ddpm 512->256 @8 b64", 64, 8, 512, 256)
