"""dev tool: eval-mode forward of YOLOv5Seg (running BN statistics) on the HIP path vs the CPU oracle after a few training steps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, yaml
import yolo_dual_amd as ydl
from oracle import ref_cpu as R
from oracle.fill import fill_state_dict
from tests.model_shapes import script_model_state_shapes
from tests.util import rel_err, l2_err
S, BS, STEPS = 128, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 5
CW = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)
cfg = yaml.safe_load(open(os.path.join(ROOT, "yolo_dual_amd", "cfg", "yolov5_seg.yaml")))
for sec in ("backbone", "head"):
    for l in cfg[sec]:
        l[2] = "C3" if l[2] == "C3_DCN" else l[2]
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.rand(BS, 3, S, S).astype(np.float32)); t = torch.from_numpy(rs.randint(0, 12, (BS, S, S)).astype(np.int64))
xv = torch.from_numpy(rs.rand(BS, 3, S, S).astype(np.float32))
shapes = script_model_state_shapes(cfg)
sd = {k: (torch.zeros(s) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.int64)) for k, s in shapes.items()}
fill_state_dict(sd, 77, bn_stats=False)
ydl.set_compute_dtype("f32")
m = ydl.YOLOv5Seg(cfg); m.img_size = [S, S]
m.load_state_dict(sd); m = m.cuda().train()
opt = ydl.FlatSGDEMA(m, lr=0.02, momentum=0.937, weight_decay=0.0, ema=False)
crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
pnames = [k for k in sd if k.endswith(".weight") or k.endswith(".bias")]
bufs = {}
for st in range(STEPS):
    opt.zero_grad(); tot, items = crit(m(x.cuda()), t.cuda()); tot.backward(); opt.step()
    ps = {k: sd[k].detach().clone().requires_grad_(True) for k in pnames}
    run = dict(sd); run.update(ps)
    out = R.script_model_forward(run, cfg, x, (S, S)); total, _, _ = R.seg_loss(out, t, CW, "dice"); total.backward()
    for k in pnames:
        if ps[k].grad is not None:
            bufs[k] = R.sgd_nesterov_step(sd[k], ps[k].grad, bufs.get(k), 0.02, 0.937, 0.0)
    for k in sd:
        if k not in ps: sd[k] = run[k]
    print(st, items[0], float(total))
msd = m.state_dict()
worst = sorted(((rel_err(msd[k].cpu(), sd[k]), k) for k in sd if sd[k].dtype.is_floating_point), reverse=True)[:8]
print("worst state entries:", worst)
m.eval()
with torch.no_grad():
    pe = m(xv.cuda()).cpu()
    pr = R.script_model_forward({k: v.clone() for k, v in sd.items()}, cfg, xv, (S, S), train=False)
print("eval out rel_err", rel_err(pe, pr), "l2", l2_err(pe, pr), "argmax agreement", float((pe.argmax(1) == pr.argmax(1)).float().mean()))
# eval with the ORACLE's state loaded into the HIP model: isolates the eval path from training drift
m.load_state_dict(sd); ydl.config.bump_weight_epoch()
with torch.no_grad():
    pe2 = m(xv.cuda()).cpu()
print("eval (oracle state) rel_err", rel_err(pe2, pr), "argmax agreement", float((pe2.argmax(1) == pr.argmax(1)).float().mean()))
