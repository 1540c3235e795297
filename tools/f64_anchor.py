#!/usr/bin/env python3
"""Dev tool (GPU): how far the HIP f32 path and the f32 CPU oracle each are from a FLOAT64 run of the oracle, per tensor, for a whole
script model at a small size (same state, same batch).  usage: python tools/f64_anchor.py [v5|v9|v9dcn] [size] [bs] [init]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import yaml

import yolo_dual_amd as ydl
from oracle import ref_cpu as R
from oracle.fill import fill_state_dict
from tests.model_shapes import script_model_state_shapes

CW = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "v9dcn"
    H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    bs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    init = sys.argv[4] if len(sys.argv) > 4 else "random"
    cfgf, fam, cls = {"v5": ("yolov5_seg.yaml", "v5", "YOLOv5Seg"), "v9": ("yolov9_seg.yaml", "v9", "YOLOv9Seg"),
                      "v9dcn": ("yolov9_dcnv3_seg.yaml", "v9", "YOLOv9Seg")}[which]
    cfg = yaml.safe_load(open(os.path.join(ROOT, "yolo_dual_amd", "cfg", cfgf)))
    for sec in ("backbone", "head"):
        for l in cfg[sec]:
            l[2] = {"C3_DCN": "C3", "C2f_DCN": "C2f"}.get(l[2], l[2])
    rs = np.random.RandomState(9)
    x = torch.from_numpy(rs.rand(bs, 3, H, W).astype(np.float32))
    t = torch.from_numpy(rs.randint(0, 12, size=(bs, H, W)).astype(np.int64))
    shapes = script_model_state_shapes(cfg)
    sd = {k: (torch.zeros(s) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.int64)) for k, s in shapes.items()}
    fill_state_dict(sd, 31, bn_stats=False)
    if init == "reference":
        for k in sd:
            if k.endswith((".offset.weight", ".offset.bias", ".mask.weight", ".mask.bias")):
                sd[k].zero_()
    pn = [k for k in sd if k.endswith(".weight") or k.endswith(".bias")]

    def oracle(dt):
        ps = {k: sd[k].detach().clone().to(dt).requires_grad_(True) for k in pn}
        run = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
        run.update(ps)
        out = R.script_model_forward(run, cfg, x.to(dt), (H, W), family=fam)
        tot, _, _ = R.seg_loss(out, t, CW.to(dt), "dice")
        tot.backward()
        return out.detach(), {k: p.grad for k, p in ps.items()}

    o32, g32 = oracle(torch.float32)
    o64, g64 = oracle(torch.float64)
    ydl.set_compute_dtype("f32")
    m = getattr(ydl, cls)(cfg)
    m.img_size = [H, W]
    m.load_state_dict(sd)
    m = m.cuda().train()
    crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
    o2 = m(x.cuda())
    tot2, _ = crit(o2, t.cuda())
    tot2.backward()
    named = dict(m.named_parameters())
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    print(f"{which} {H}x{W} bs {bs} init {init}: probabilities vs f64: oracle {rel(o32, o64):.2e} HIP {rel(o2.detach().cpu(), o64):.2e}")
    rows = []
    for k in pn:
        if g64[k] is None or float(g64[k].abs().max()) == 0:
            continue
        rows.append((k, l2(g32[k], g64[k]), l2(named[k].grad.detach().cpu(), g64[k]), l2(named[k].grad.detach().cpu(), g32[k])))
    eo, eh = np.array([r[1] for r in rows]), np.array([r[2] for r in rows])
    print(f"gradients vs f64 (relative L2): oracle median {np.median(eo):.2e}, HIP median {np.median(eh):.2e}, ratio of medians {np.median(eh) / np.median(eo):.2f}")
    for r in rows:
        print(f"  {r[0]:52s} oracle {r[1]:.2e}  HIP {r[2]:.2e}  HIP-vs-oracle32 {r[3]:.2e}  ratio {r[2] / max(r[1], 1e-30):.1f}")


if __name__ == "__main__":
    main()
