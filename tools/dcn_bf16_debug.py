#!/usr/bin/env python3
"""dev tool: per-tensor bf16-vs-f32-fixture gradient errors of the DCNv3 module / C3_DCNV3 fixtures"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_dual_amd as ydl
from tests.util import Golden, names, l2_err
from tests.test_gpu_blocks import _load

for name in names("dcnmod_") + names("c3_dcnv3_"):
    g = Golden(name)
    for mode in ("f32", "bf16"):
        ydl.set_compute_dtype(mode)
        if name.startswith("dcnmod"):
            C, k, s, pad, G = [int(v) for v in g.flat["meta"]]
            m = _load(ydl.DCNv3(channels=C, kernel_size=k, stride=s, pad=pad, group=G), g)
        else:
            c1, c2, n = [int(v) for v in g.flat["meta"]]
            m = ydl.C3_DCNV3(c1, c2, n, "noshortcut" not in name)
            m.load_state_dict(g.group("sd"))
            m = m.cuda().train()
        x = g.t("x0").cuda().requires_grad_(True)
        out = m(x)
        (out * g.t("gup").cuda()).sum().backward()
        grads = g.group("grad")
        named = dict(m.named_parameters())
        gs = max(float(v.abs().max()) for v in grads.values())
        errs = {"out": l2_err(out.detach().cpu(), g.t("out")), "x": l2_err(x.grad.cpu(), g.t("gx0"))}
        for kk, v in grads.items():
            if float(v.abs().max()) >= 1e-4 * gs:
                errs[kk] = l2_err(named[kk].grad.detach().float().cpu(), v)
        print(name, mode, x.shape, {k: round(v, 4) for k, v in errs.items()})

# bf16 vs f32 of the HIP path itself at a better-conditioned size
print("---- C3_DCNV3 bf16 vs f32 (HIP), larger sizes")
from oracle.fill import fill_state_dict
for (c, n, short, N, H) in [(64, 2, False, 4, 40), (64, 1, True, 4, 40), (128, 2, False, 8, 40), (24, 2, False, 2, 10), (24, 2, False, 8, 40)]:
    res = {}
    for mode in ("f32", "bf16"):
        ydl.set_compute_dtype(mode)
        m = ydl.C3_DCNV3(c, c, n, short)
        sd = m.state_dict()
        fill_state_dict(sd, 3, bn_stats=False)
        m.load_state_dict(sd)
        m = m.cuda().train()
        g = torch.Generator("cuda").manual_seed(1)
        x = torch.randn(N, c, H, H, device="cuda", generator=g).requires_grad_(True)
        gup = torch.randn(N, c, H, H, device="cuda", generator=g)
        out = m(x)
        (out * gup).sum().backward()
        res[mode] = dict(out=out.detach().float().cpu(), x=x.grad.float().cpu(), **{k: p.grad.detach().float().cpu() for k, p in m.named_parameters()})
    gs = max(float(v.abs().max()) for k, v in res["f32"].items() if k not in ("out", "x"))
    errs = {k: round(l2_err(res["bf16"][k], v), 3) for k, v in res["f32"].items() if float(v.abs().max()) >= 1e-4 * gs}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print((c, n, short, N, H), "out", errs["out"], "x", errs["x"], "worst", worst)
