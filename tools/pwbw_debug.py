#!/usr/bin/env python3
"""dev tool (GPU): ydl_conv_bwd_pw against float64, with where-is-it-wrong diagnostics"""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from yolo_dual_amd import _lib as L

def run(N, H, W, ldx, ldy, lddx, ldw, acc):
    rs = np.random.RandomState(1)
    M, C = N * H * W, 128
    x = torch.from_numpy(rs.standard_normal((M, C)).astype(np.float32)).bfloat16()
    dy = torch.from_numpy(rs.standard_normal((M, C)).astype(np.float32)).bfloat16()
    w = torch.from_numpy((rs.standard_normal((C, C)) / np.sqrt(C)).astype(np.float32)).bfloat16()
    dx0 = torch.from_numpy(rs.standard_normal((M, C)).astype(np.float32)).bfloat16()
    dev = torch.device("cuda")
    xg = torch.full((M, ldx), 7.0, dtype=torch.bfloat16, device=dev); xg[:, :C] = x.to(dev)
    dyg = torch.full((M, ldy), 7.0, dtype=torch.bfloat16, device=dev); dyg[:, :C] = dy.to(dev)
    dxg = torch.full((M, lddx), 7.0, dtype=torch.bfloat16, device=dev); dxg[:, :C] = dx0.to(dev)
    ldw_e = ldw or C
    dwg = torch.zeros((C, ldw_e), dtype=torch.float32, device=dev)
    wt = w.t().contiguous().to(dev)
    g = L.ConvGeom(N, H, W, C, H, W, C, 1, 1, 0, ldx, ldy, ldw)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    # poison the LDS first: a kernel that fills its dynamic LDS with NaN patterns is not available; run twice instead
    for rep in range(2):
        dxg[:, :C] = dx0.to(dev); dwg.zero_()
        L.call("ydl_conv_bwd_pw", ctypes.byref(g), L.YDL_BF16, P(xg), P(dyg), P(wt), P(dxg), lddx, acc, P(dwg), st)
        torch.cuda.synchronize()
    ref_dx = (dy.to(dev).double() @ w.to(dev).double()) + (dx0.to(dev).double() if acc else 0.0)
    ref_dw = dy.to(dev).double().t() @ x.to(dev).double()
    got_dx = dxg[:, :C].double(); got_dw = dwg[:, :C].double()
    edx = (got_dx - ref_dx).abs(); edw = (got_dw - ref_dw).abs()
    bad_rows = (edx.max(1).values > 0.05 * ref_dx.abs().max()).nonzero().flatten()
    print(f"M={M} ldx={ldx} ldy={ldy} lddx={lddx} ldw={ldw} acc={acc}: dx nan {int(torch.isnan(got_dx).sum())} max err {float(edx.max()):.3g} "
          f"(scale {float(ref_dx.abs().max()):.3g}) bad rows {bad_rows.numel()} first {bad_rows[:8].tolist()} last {bad_rows[-4:].tolist()} | "
          f"dw nan {int(torch.isnan(got_dw).sum())} max err {float(edw.max()):.3g} (scale {float(ref_dw.abs().max()):.3g})", flush=True)
    if bad_rows.numel():
        r0 = int(bad_rows[0]); print("   row", r0, "got", got_dx[r0, :8].tolist(), "ref", ref_dx[r0, :8].tolist())
        bc = (edx[r0] > 0.05 * ref_dx.abs().max()).nonzero().flatten(); print("   bad cols", bc.tolist()[:40])

for acc in (0, 1):
    run(2, 256, 256, 128, 128, 128, 0, acc)
    run(3, 211, 209, 192, 128, 256, 640, acc)
    run(16, 160, 160, 192, 128, 192, 0, acc)
