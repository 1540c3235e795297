"""dev tool: cProfile of the eager step's HOST side (Python + ctypes enqueue cost), 10 steps of the bench workload"""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
import yolo_dual_amd as ydl

cfg = yaml.safe_load(open(os.path.join(ROOT, "yolo_dual_amd", "cfg", "yolov5_seg.yaml")))
for sec in ("backbone", "head"):
    for l in cfg[sec]:
        l[2] = "C3" if l[2] == "C3_DCN" else l[2]
ydl.set_compute_dtype("bf16")
m = ydl.YOLOv5Seg(cfg).cuda().train()
cw = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)
crit = ydl.SegmentationLoss(12, 0.0, cw, "dice", sync=False)
opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4)
x = torch.rand(16, 3, 640, 640, device="cuda")
t = torch.randint(0, 12, (16, 640, 640), device="cuda")


def step():
    opt.zero_grad()
    loss, _ = crit(m(x), t)
    loss.backward()
    opt.step()


# the backward closures run on autograd's device thread: profile Tape.run_backward there with a second profiler
from yolo_dual_amd import tape as _tape
pr_b = cProfile.Profile()
_orig_rb = _tape.Tape.run_backward
def _rb(self):
    if PROFILE_BW[0]:
        pr_b.enable()
    try:
        return _orig_rb(self)
    finally:
        if PROFILE_BW[0]:
            pr_b.disable()
PROFILE_BW = [False]
_tape.Tape.run_backward = _rb

for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
PROFILE_BW[0] = True
pr.enable()
for _ in range(10):
    step()
pr.disable()
PROFILE_BW[0] = False
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])

s = io.StringIO()
pstats.Stats(pr_b, stream=s).sort_stats("tottime").print_stats(24)
print("---- backward closures (autograd thread) ----")
print(s.getvalue()[:5000])
