"""dev tool: how sensitive is the 600-step learning run to a 1e-6 perturbation of the first batch (HIP path, f32)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import yolo_dual_amd as ydl
from oracle.fill import fill_state_dict
from tests.test_gpu_training_parity import _blobby128, _cfg, CW
fx = np.load(os.path.join(ROOT, "tests", "golden", "train_curve_yolov5seg_128.npz"))
S_, BS_, STEPS_, LR_, NB_, LRF_ = int(fx["hyp"][0]), int(fx["hyp"][1]), int(fx["hyp"][2]), float(fx["hyp"][3]), int(fx["hyp"][4]), float(fx["hyp"][5])
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
for eps in (0.0, 1e-6, -1e-6):
    ydl.set_compute_dtype(mode)
    m = ydl.YOLOv5Seg(_cfg()); m.img_size = [S_, S_]
    sd = m.state_dict(); fill_state_dict(sd, 77, bn_stats=False); m.load_state_dict(sd); m = m.cuda().train()
    opt = ydl.FlatSGDEMA(m, lr=LR_, momentum=0.937, weight_decay=0.0, ema=False)
    crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
    batches = [tuple(t.cuda() for t in _blobby128(100 + i, BS_, S_)) for i in range(NB_)]
    batches[0] = (batches[0][0] + eps, batches[0][1])        # same perturbation as oracle/make_train_curve.py
    xv, tv = (t.cuda() for t in _blobby128(2, BS_, S_))
    mious = []
    for st in range(STEPS_):
        x, t = batches[st % NB_]
        for g in opt.param_groups: g["lr"] = LR_ * (1.0 - (1.0 - LRF_) * st / STEPS_)
        opt.zero_grad(); tot, items = crit(m(x), t); tot.backward(); opt.step()
        if st % 25 == 24:
            m.eval()
            with torch.no_grad(): pv = m(xv)
            cm = ydl.ConfusionMatrix(12, ignore_index=11); cm.process_batch(pv, tv); mious.append(cm.compute_iou()[0]); m.train()
    print(mode, "eps", eps, "mIoU curve", " ".join(f"{v:.3f}" for v in mious[3::4]), "final", f"{mious[-1]:.4f}", flush=True)
print("oracle        mIoU curve", " ".join(f"{v:.3f}" for v in fx["mious"][3::4]), "final", f"{fx['mious'][-1]:.4f}")
