#!/usr/bin/env python3
"""dev tool (GPU): ydl_conv_bwd_pw with identity weights and index-valued dy: which element lands where"""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_dual_amd import _lib as L
dev = torch.device("cuda")
N, H, W, C = 2, 256, 256, 128
M = N * H * W
g = L.ConvGeom(N, H, W, C, H, W, C, 1, 1, 0, C, C, 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
x = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
for mode in ("col", "row", "wrow"):
    if mode == "col":
        dy = torch.arange(C, device=dev).float().view(1, C).expand(M, C).contiguous().bfloat16(); wt = torch.eye(C, device=dev).bfloat16()
    elif mode == "row":
        dy = (torch.arange(M, device=dev) % 128).float().view(M, 1).expand(M, C).contiguous().bfloat16(); wt = torch.eye(C, device=dev).bfloat16()
    else:       # dy = one-hot at co 0 scaled 1: dx[m][ci] = wt[ci][0]; wt[ci][co] = ci  -> dx[m][ci] = ci, tests the weight rows
        dy = torch.zeros(M, C, device=dev).bfloat16(); dy[:, 0] = 1
        wt = torch.arange(C, device=dev).float().view(C, 1).expand(C, C).contiguous().bfloat16()
    dx = torch.full((M, C), -1.0, dtype=torch.bfloat16, device=dev)
    dw = torch.zeros(C, C, device=dev)
    L.call("ydl_conv_bwd_pw", ctypes.byref(g), L.YDL_BF16, P(x), P(dy), P(wt), P(dx), C, 0, P(dw), st)
    torch.cuda.synchronize()
    ref = dy.float() @ wt.float().t()
    bad = (dx.float() != ref)
    print(mode, "bad elements", int(bad.sum()), "of", M * C)
    if int(bad.sum()):
        idx = bad.nonzero()
        for (m, c) in idx[:12].tolist() + idx[-4:].tolist():
            print(f"   m={m} (stage row {m % 32}, cta-local {m % 512}) c={c}: got {float(dx[m, c])} want {float(ref[m, c])}")
        cols = bad.any(0).nonzero().flatten().tolist(); print("   bad cols", cols)
        rows32 = torch.zeros(32, dtype=torch.long)
        for m in idx[:, 0].unique().tolist()[:20000]: rows32[m % 32] += 1
        print("   bad rows by (m % 32):", rows32.tolist())
