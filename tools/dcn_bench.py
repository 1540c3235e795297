#!/usr/bin/env python3
"""Micro-benchmark of the DCNv3 op at the shape of SURVEY a13 (C3_DCNV3 at 256@80x80: N=16, G=4, Gc=64, K=3) and at the shapes
the wired cfg5dcn model runs (group 1): forward gather rate and backward (dev tool)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_dual_amd import _lib as L

dev = torch.device("cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
OSTD = float(os.environ.get("DCN_OFF_STD", "2.0"))        # std of the sampling offsets in pixels (trained models: well below 1)
for (N, H, G, Gc) in [(16, 80, 4, 64), (16, 160, 1, 64), (16, 80, 1, 128), (16, 40, 1, 256), (16, 20, 1, 256)]:
    C, K = G * Gc, 3
    for dt, tdt, es in ((L.YDL_BF16, torch.bfloat16, 2), (L.YDL_F32, torch.float32, 4)):
        R = 3
        xs = [torch.randn(N, H, H, C, device=dev).to(tdt) for _ in range(R)]
        off = (torch.randn(N, H, H, G * K * K * 2, device=dev) * OSTD).to(tdt)
        msk = torch.softmax(torch.randn(N, H, H, G, K * K, device=dev), -1).reshape(N, H, H, G * K * K).to(tdt)
        out = torch.empty(N, H, H, C, device=dev, dtype=tdt)
        go = torch.randn(N, H, H, C, device=dev).to(tdt)
        gin = torch.zeros(N, H, H, C, device=dev); goff = torch.empty(off.shape, device=dev); gmsk = torch.empty(msk.shape, device=dev)
        def fwd(i):
            L.call("ydl_dcnv3_fwd", dt, P(xs[i % R]), P(off), P(msk), P(out), K, K, 1, 1, 1, 1, 1, 1, G, Gc, ctypes.c_float(1.0), N, H, H, H, H, st)
        def bwd(i):
            L.call("ydl_dcnv3_bwd", dt, P(xs[i % R]), P(off), P(msk), P(go), P(gin), P(goff), P(gmsk), K, K, 1, 1, 1, 1, 1, 1, G, Gc,
                   ctypes.c_float(1.0), N, H, H, H, H, st)
        res = []
        for fn in (fwd, bwd):
            for i in range(3): fn(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(10): fn(i)
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 10)
        npix = N * H * H
        gathered = npix * K * K * 4 * C * es          # 4 bilinear corners x 9 points x C channels per output pixel (L1/L2 served)
        alg = npix * C * es * 2 + off.numel() * es + msk.numel() * es          # input once + output once + offsets + masks
        print(f"N{N} {H}x{H} G{G} Gc{Gc} {'bf16' if es == 2 else 'f32 '} | fwd {res[0]*1e3:7.1f}us  gathered {gathered/res[0]/1e9:6.2f} TB/s  algorithmic "
              f"{alg/res[0]/1e9:5.2f} TB/s | bwd {res[1]*1e3:8.1f}us (atomics {npix*K*K*4*C*4/res[1]/1e9:5.2f} TB/s)", flush=True)
