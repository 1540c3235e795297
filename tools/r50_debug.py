"""dev tool: per-parameter gradient error of ResNet50Seg (f32 GPU path) against the CPU oracle on the golden input"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yolo_dual_amd as ydl
from oracle import ref_cpu as R
from oracle.fill import fill_state_dict
from tests.util import Golden, l2_err

g = Golden("model_resnet50seg_64")
ydl.set_compute_dtype("f32")
m = ydl.ResNet50Seg({"nc": 12})
sd = m.state_dict()
fill_state_dict(sd, 1234, bn_stats=False)
m.load_state_dict(sd)
x, t = g.t("x"), g.t("target")
# oracle
ps = {k: v.detach().clone().double().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and ("weight" in k or "bias" in k)}
run = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
run.update(ps)
out = R.resnet_seg_forward(run, x.double(), "bottleneck", out_size=(640, 640))
tot, _, _ = R.seg_loss(out, t, None, "dice")
tot.backward()
m = m.cuda().train()
crit = ydl.SegmentationLoss(12, 0.0, None, "dice")
o2 = m(x.cuda())
tt, items = crit(o2, t.cuda())
tt.backward()
print("loss", float(tot), items[0], "out err", l2_err(o2.detach().cpu(), out.detach().float()))
errs = []
for k, p in m.named_parameters():
    if getattr(p, "_ydl_touched", False) and ps[k].grad is not None:
        errs.append((l2_err(p.grad.detach().cpu(), ps[k].grad.float()), k, float(ps[k].grad.norm())))
errs.sort(reverse=True)
for e in errs[:12]:
    print(f"{e[0]:.2e} {e[1]} |g|={e[2]:.3e}")
print("median", sorted(e[0] for e in errs)[len(errs) // 2])
