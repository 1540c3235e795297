#!/usr/bin/env python3
"""Micro-benchmark of the BN entry points (finalize / act_fwd / act_bwd) on layer shapes of BASELINE config 2 (dev tool).
Each op is timed as a back-to-back batch on one stream (events around the batch, not per call), so the figure includes
the real launch-to-launch latency of its internal kernel chain."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_dual_amd import _lib as L
if os.environ.get("YDL_LIB_PATH"):          # timing experiments with a variant build of the library
    L.LIB_PATH = os.environ["YDL_LIB_PATH"]

SHAPES = [(1638400, 64, 12800), (409600, 128, 3200), (409600, 64, 3200), (102400, 256, 800), (25600, 512, 400), (6400, 1024, 100), (6400, 512, 100),
          (409600, 128, 493)]   # (npix, C, conv partial rows)
dev = torch.device("cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
iters = 30


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for npix, C, rows in SHAPES:
    y = torch.randn(npix, C, device=dev).bfloat16()
    dout = torch.randn(npix, C, device=dev).bfloat16()
    out = torch.empty_like(y)
    dy = torch.empty_like(y)
    block_m = (npix + rows - 1) // rows
    ws = torch.rand((rows + rows // 64 + 2) * 2 * C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd, scale, shift = (torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev))
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws2 = torch.empty(L.lib().ydl_bn_bwd_ws_bytes(npix, C) // 4, device=dev)
    t_fin = timeit(lambda: L.call("ydl_bn_finalize", P(ws), rows, block_m, npix, C, P(g), P(b), 1e-5, 0.03, P(rm), P(rv), P(mean), P(invstd),
                                  P(scale), P(shift), 1, st))
    mean.zero_(); invstd.fill_(1.0); scale.fill_(1.0); shift.zero_()
    t_fwd = timeit(lambda: L.call("ydl_bn_act_fwd", L.YDL_BF16, P(y), C, P(scale), P(shift), None, 0, 0, 1, P(out), C, npix, C, st))
    t_bwd = timeit(lambda: L.call("ydl_bn_act_bwd", L.YDL_BF16, P(y), C, P(dout), C, P(out), C, P(g), P(mean), P(invstd), P(scale), P(shift),
                                  0, 1, P(dy), C, None, 0, P(dg), P(db), 1, P(ws2), npix, C, C, st))
    mb = npix * C * 2 / 1e6
    print(f"npix {npix:7d} C {C:5d} rows {rows:5d} | finalize {t_fin:6.1f}us | act_fwd {t_fwd:6.1f}us {2 * mb / t_fwd:5.2f}TB/s"
          f" | act_bwd {t_bwd:6.1f}us {5 * mb / t_bwd:5.2f}TB/s", flush=True)

# throughput-mode forward on replica sums (ydl_bn_act_fwd_sums): the kernel of the training step
print("--- ydl_bn_act_fwd_sums (SiLU, no residual)")
for npix, C, rows in SHAPES:
    y = torch.randn(npix, C, device=dev).bfloat16()
    out = torch.empty_like(y)
    sums = torch.zeros(16, C, device=dev)
    sums[0] = 0.1 * npix; sums[1] = 1.5 * npix
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mean, invstd, scale, shift = (torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev))
    t = timeit(lambda: L.call("ydl_bn_act_fwd_sums", L.YDL_BF16, P(y), C, P(sums), C, npix, P(g), P(b), 1e-5, 0.03, None, None, P(mean), P(invstd),
                              P(scale), P(shift), 1, None, 0, 0, 1, P(out), C, npix, C, C, st))
    mb = npix * C * 2 / 1e6
    print(f"npix {npix:7d} C {C:5d} | act_fwd_sums {t:6.1f}us {2 * mb / t:5.2f}TB/s", flush=True)

# throughput-mode backward on replica sums (ydl_bn_act_bwd_sums: reduce pass + apply pass; the sums row is zeroed per call like the
# tape's slab would be)
print("--- ydl_bn_act_bwd_sums (SiLU, no residual; fill_zero of the sums row included)")
for npix, C, rows in SHAPES:
    y = torch.randn(npix, C, device=dev).bfloat16()
    dout = torch.randn(npix, C, device=dev).bfloat16()
    dy = torch.empty_like(y)
    sums = torch.zeros(16, C, device=dev)
    mean, invstd, scale, shift = (torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev))
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)

    def run():
        L.call("ydl_fill_zero", P(sums), sums.numel() * 4, st)
        L.call("ydl_bn_act_bwd_sums", L.YDL_BF16, P(y), C, P(dout), C, None, 0, P(mean), P(invstd), P(scale), P(shift), 0, 1, P(dy), C, None, 0,
               P(dg), P(db), 0, P(sums), npix, C, C, st)
    t = timeit(run)
    mb = npix * C * 2 / 1e6
    print(f"npix {npix:7d} C {C:5d} | act_bwd_sums {t:6.1f}us {5 * mb / t:5.2f}TB/s (5 passes)", flush=True)
