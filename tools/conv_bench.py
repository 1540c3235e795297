#!/usr/bin/env python3
"""Micro-benchmark of the conv entry points on selected layer geometries of BASELINE config 2 (dev tool).
usage: python tools/conv_bench.py [fwd|dgrad|wgrad|all] [--dtype bf16] [--iters 20] [--only IDX,IDX]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_dual_amd import _lib as L

# (Cin, Cout, k, s, Hout)   bs=16
LAYERS = [(3, 64, 6, 2, 320), (64, 128, 3, 2, 160), (128, 64, 1, 1, 160), (64, 64, 3, 1, 160), (128, 128, 1, 1, 160),
          (128, 256, 3, 2, 80), (256, 128, 1, 1, 80), (128, 128, 3, 1, 80), (256, 256, 1, 1, 80), (256, 512, 3, 2, 40),
          (256, 256, 3, 1, 40), (512, 512, 1, 1, 40), (512, 1024, 3, 2, 20), (512, 512, 3, 1, 20), (2048, 1024, 1, 1, 20),
          (640, 64, 1, 1, 160), (128, 64, 3, 1, 160), (768, 128, 1, 1, 80), (64, 12, 1, 1, 160),
          (64, 64, 1, 1, 160), (64, 128, 1, 1, 160), (128, 256, 1, 1, 80), (256, 64, 1, 1, 80), (512, 128, 1, 1, 40),
          (1024, 1024, 1, 1, 20), (1024, 512, 1, 1, 20), (256, 256, 1, 1, 80),
          (12, 64, 3, 1, 320)]      # [27]: the stem as the model runs it (space-to-depth input, 16 stored channels)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--nostats", action="store_true")
    ap.add_argument("--wg", type=int, default=1)
    ap.add_argument("--pw", type=int, default=1, help="0: tiled kernel for the short-K 1x1 layers")
    ap.add_argument("--rotate", type=int, default=1, help="cycle through this many tensor sets (defeats the 256 MB MALL)")
    ap.add_argument("--acc", type=int, default=0, help="dgrad accumulate flag")
    ap.add_argument("--ring", type=int, default=1, help="0: igemm_kernel instead of the LDS-DMA ring kernel (igemm2)")
    ap.add_argument("--wg3", type=int, default=1, help="0: register-staged wgrad2 instead of the LDS-DMA wgrad3")
    ap.add_argument("--persist", type=int, default=1, help="0: one tile per CTA instead of the persistent ring kernel (igemm2p)")
    ap.add_argument("--halo", type=int, default=1, help="0: ring kernel instead of the patch-form kernel on the 3x3 / stride-1 layers")
    ap.add_argument("--s2", type=int, default=1, help="0: ring kernel instead of the fused-parity kernel on the k3 s2 p1 data gradients")
    ap.add_argument("--sums", action="store_true", help="fwd: BatchNorm statistics as replica sums (ydl_conv_fwd_sums: what the training step calls)")
    ap.add_argument("--det", action="store_true", help="wgrad: deterministic slab + fixed-order reduce instead of f32 atomics")
    ap.add_argument("--check", action="store_true", help="compare fwd/dgrad of the ring kernel with igemm_kernel (max abs diff)")
    a = ap.parse_args()
    L.debug_set(0, a.wg)
    L.debug_set(1, a.pw)
    L.debug_set(3, a.ring)
    L.debug_set(4, a.wg3)
    L.debug_set(6, a.persist)
    L.debug_set(8, a.halo)
    L.debug_set(9, a.s2)
    dt = L.YDL_BF16 if a.dtype == "bf16" else L.YDL_F32
    tdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = torch.device("cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sel = [int(v) for v in a.only.split(",")] if a.only else range(len(LAYERS))
    r8 = lambda v: (v + 7) // 8 * 8
    for li in sel:
        Cin, Cout, k, s, Ho = LAYERS[li]
        p = 2 if k == 6 else k // 2
        Hi = Ho * s
        N = a.bs
        ldx, ldy = r8(Cin), r8(Cout)
        R = a.rotate
        xs = [torch.randn(N, Hi, Hi, ldx, device=dev).to(tdt) for _ in range(R)]
        ys = [torch.empty(N, Ho, Ho, ldy, device=dev, dtype=tdt) for _ in range(R)]
        dys = [torch.randn(N, Ho, Ho, ldy, device=dev).to(tdt) for _ in range(R)]
        dxs = [torch.zeros(N, Hi, Hi, ldx, device=dev, dtype=tdt) for _ in range(R)]
        w = torch.randn(Cout, k * k, ldx, device=dev).to(tdt)
        wt = torch.randn(Cin, k * k, ldy, device=dev).to(tdt)
        dw = torch.zeros(Cout, k * k, ldx, device=dev, dtype=torch.float32)
        g = L.ConvGeom(N, Hi, Hi, Cin, Ho, Ho, Cout, k, s, p, ldx, ldy)
        gp = ctypes.byref(g)
        ws = torch.empty(L.lib().ydl_conv_fwd_stats_ws_bytes(gp, dt) // 4, device=dev)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        wsd = torch.empty(max(L.lib().ydl_conv_wgrad_ws_bytes(gp, dt) // 4, 4), device=dev) if a.det else None
        flops = 2.0 * N * Ho * Ho * Cout * k * k * Cin
        byts = (N * Hi * Hi * Cin + N * Ho * Ho * Cout) * (2 if a.dtype == "bf16" else 4)
        it = [0]

        def nxt():
            it[0] += 1
            return it[0] % R
        sums = torch.zeros(8 * 2 * ldy, device=dev)
        ops = {"fwd": (lambda i: L.call("ydl_conv_fwd_sums", gp, dt, P(xs[i]), P(w), P(ys[i]), P(sums), 0, st)) if a.sums else
                      (lambda i: L.call("ydl_conv_fwd", gp, dt, P(xs[i]), P(w), P(ys[i]), None if a.nostats else P(ws), 0, st)),
               "dgrad": lambda i: L.call("ydl_conv_dgrad", gp, dt, P(dys[i]), P(wt), P(dxs[i]), a.acc, st),
               "wgrad": (lambda i: L.call("ydl_conv_wgrad_det", gp, dt, P(xs[i]), P(dys[i]), P(dw), P(wsd), st)) if a.det else
                        (lambda i: L.call("ydl_conv_wgrad", gp, dt, P(xs[i]), P(dys[i]), P(dw), st))}
        line = f"[{li:2d}] {Cin:5d}->{Cout:5d} k{k}s{s} @{Ho:4d}"
        if a.check:
            outs = []
            for ring in (1, 0):
                L.debug_set(3, ring)
                ws2 = torch.zeros(L.lib().ydl_conv_fwd_stats_ws_bytes(gp, dt) // 4 + 16, device=dev)
                y2 = torch.zeros_like(ys[0]); dx2 = torch.zeros_like(dxs[0])
                L.call("ydl_conv_fwd", gp, dt, P(xs[0]), P(w), P(y2), P(ws2), 0, st)
                kf = L.last_kernel(0)
                L.call("ydl_conv_dgrad", gp, dt, P(dys[0]), P(wt), P(dx2), 0, st)
                kd = L.last_kernel(1)
                gm, bm = L.lib().ydl_conv_fwd_grid_m(gp, dt), L.lib().ydl_conv_fwd_block_m(gp, dt)
                tot = ws2[:gm * 2 * ldy].view(gm, 2, ldy)[:, 0].double().sum(0)      # per-channel sums from the partials
                torch.cuda.synchronize()
                outs.append((y2.float(), dx2.float(), tot, kf, kd))
            L.debug_set(3, a.ring)
            dyv = float((outs[0][0] - outs[1][0]).abs().max()); dxv = float((outs[0][1] - outs[1][1]).abs().max())
            ds = float((outs[0][2] - outs[1][2]).abs().max() / outs[1][2].abs().max().clamp_min(1e-9))
            print(line + f" | check: max|dy| {dyv:.3g} (scale {float(outs[1][0].abs().max()):.3g}) max|ddx| {dxv:.3g} "
                  f"(scale {float(outs[1][1].abs().max()):.3g}) stats rel {ds:.2g} | {outs[0][3]} / {outs[0][4]} vs {outs[1][3]} / {outs[1][4]}", flush=True)
            continue
        for name, fn in ops.items():
            if a.what not in ("all", name):
                continue
            for _ in range(3):
                fn(nxt())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn(nxt())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            kn = L.last_kernel({"fwd": 0, "dgrad": 1, "wgrad": 2}[name]).replace("_kernel", "")
            line += f" | {name} {ms * 1e3:7.1f}us {flops / ms / 1e9:5.0f}TF {byts / ms / 1e9:5.2f}TB/s {kn}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
