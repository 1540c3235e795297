"""dev check: weight gradients of single Conv layers, bf16 kernels vs f32 kernels at benchmark-like sizes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_dual_amd as ydl
from tests.util import l2_err
cases = [(4, 3, 64, 6, 2, 2, 640), (4, 64, 128, 3, 2, None, 320), (4, 64, 64, 3, 1, None, 160), (16, 64, 64, 3, 1, None, 160),
         (4, 128, 64, 1, 1, None, 160), (16, 640, 64, 1, 1, None, 160), (4, 256, 256, 3, 1, None, 40), (16, 64, 12, 1, 1, None, 160)]
for (N, c1, c2, k, s, p, H) in cases:
    res = {}
    for mode in ("f32", "bf16"):
        ydl.set_compute_dtype(mode)
        torch.manual_seed(0)
        m = ydl.Conv(c1, c2, k, s, p).cuda().train()
        x = torch.randn(N, c1, H, H, device="cuda", generator=torch.Generator("cuda").manual_seed(1)).requires_grad_(True)
        out = m(x)
        out.square().mean().backward()
        res[mode] = (m.conv.weight.grad.float().cpu().clone(), x.grad.float().cpu().clone())
    print((N, c1, c2, k, s, H), "dW l2err", l2_err(res["bf16"][0], res["f32"][0]), "finite", bool(torch.isfinite(res["bf16"][0]).all()),
          "dx l2err", l2_err(res["bf16"][1], res["f32"][1]))
