#!/usr/bin/env python3
"""What the vendor libraries reach on the same shapes (dev tool, comparison only — nothing here is on the product path):
torch.mm (hipBLASLt / rocBLAS) on the GEMM of each 1x1 layer and F.conv2d (MIOpen, channels_last bf16) on the 3x3 layers of
BASELINE config 2, forward only, timed with events over a rotating set of tensors.
usage: python tools/vendor_bench.py [--iters 20] [--conv 1]"""
import argparse
import sys
import time

import torch
import torch.nn.functional as F

# (Cin, Cout, k, s, Hout)   bs=16
LAYERS = [(64, 128, 3, 2, 160), (64, 64, 3, 1, 160), (128, 128, 1, 1, 160), (128, 256, 3, 2, 80), (128, 128, 3, 1, 80),
          (256, 256, 1, 1, 80), (256, 512, 3, 2, 40), (256, 256, 3, 1, 40), (512, 512, 1, 1, 40), (512, 1024, 3, 2, 20),
          (512, 512, 3, 1, 20), (1024, 1024, 1, 1, 20), (2048, 1024, 1, 1, 20), (1024, 512, 1, 1, 20), (128, 64, 3, 1, 160),
          (640, 512, 1, 1, 40), (512, 256, 1, 1, 40)]


def timed(fn, iters):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--conv", type=int, default=1)
    ap.add_argument("--bs", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    R = 3
    for (ci, co, k, s, ho) in LAYERS:
        M = a.bs * ho * ho
        flops = 2.0 * M * co * ci * k * k
        line = f"{ci:5d}->{co:5d} k{k}s{s} @{ho:4d} | "
        t0 = time.time()
        if k == 1:
            xs = [torch.randn(M, ci, device=dev, dtype=torch.bfloat16) for _ in range(R)]
            w = torch.randn(ci, co, device=dev, dtype=torch.bfloat16)
            ys = [torch.empty(M, co, device=dev, dtype=torch.bfloat16) for _ in range(R)]
            us = timed(lambda i: torch.mm(xs[i % R], w, out=ys[i % R]), a.iters)
            line += f"torch.mm {us:8.1f} us {flops / us / 1e6:6.0f} TF"
        elif a.conv:
            hi = ho * s
            xs = [torch.randn(a.bs, ci, hi, hi, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(R)]
            w = torch.randn(co, ci, k, k, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
            us = timed(lambda i: F.conv2d(xs[i % R], w, None, s, k // 2), a.iters)
            line += f"F.conv2d {us:8.1f} us {flops / us / 1e6:6.0f} TF"
            # the same layer as an explicit GEMM on an im2col matrix that already exists (an upper bound on what any implicit form reaches)
            K = ci * k * k
            if M * K * 2 < 6e9:
                xm = [torch.randn(M, K, device=dev, dtype=torch.bfloat16) for _ in range(2)]
                wm = torch.randn(K, co, device=dev, dtype=torch.bfloat16)
                ym = torch.empty(M, co, device=dev, dtype=torch.bfloat16)
                us2 = timed(lambda i: torch.mm(xm[i % 2], wm, out=ym), a.iters)
                line += f" | mm on a materialised im2col {us2:8.1f} us {flops / us2 / 1e6:6.0f} TF"
                del xm
        print(line + f"   ({time.time() - t0:.1f} s)", flush=True)
        del xs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    sys.exit(main())
