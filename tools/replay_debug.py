#!/usr/bin/env python3
"""dev tool: run an eager model and a launch-list model in lockstep, report the first state entries that differ after each step"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_dual_amd as ydl
from yolo_dual_amd import config
from yolo_dual_amd.replay import ReplayedTrainStep
from tests.test_gpu_replay import _setup, _state

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cfgn = sys.argv[2] if len(sys.argv) > 2 else "yolov5_seg.yaml"
cls = sys.argv[3] if len(sys.argv) > 3 else "YOLOv5Seg"
poison = int(sys.argv[4]) if len(sys.argv) > 4 else 1
config.set_deterministic(True)
A = _setup(mode, cfgn, cls)
B = _setup(mode, cfgn, cls)
C = _setup(mode, cfgn, cls)          # a second eager model: run-to-run noise of the eager path itself (atomics)
mC, oC, cC, _, _ = C
mA, oA, cA, xs, ts = A
mB, oB, cB, _, _ = B
x, t = xs[0].clone(), ts[0].clone()


def eager(st):
    x.copy_(xs[st % 3]); t.copy_(ts[st % 3])
    oA.zero_grad()
    tot, it = cA(mA(x), t)
    tot.backward()
    oA.step()
    return float(it[0])


def diff(tag):
    torch.cuda.synchronize()
    sa, sb, sc = _state(mA, oA), _state(mB, oB), _state(mC, oC)
    bad = [(k, float((sa[k].float() - sb[k].float()).abs().max()), bool(torch.isnan(sb[k].float()).any())) for k in sa if not torch.equal(sa[k], sb[k])]
    badc = [(k, float((sa[k].float() - sc[k].float()).abs().max())) for k in sa if not torch.equal(sa[k], sc[k])]
    print(tag, "replay-vs-eager differing entries:", len(bad), bad[:4], "| eager-vs-eager:", len(badc), badc[:4])


def eager_c(st):
    x.copy_(xs[st % 3]); t.copy_(ts[st % 3])
    oC.zero_grad()
    tot, it = cC(mC(x), t)
    tot.backward()
    oC.step()


step_no = [0]
def pre(_m, _i):
    i = step_no[0]
    x.copy_(xs[i % 3]); t.copy_(ts[i % 3])
    step_no[0] += 1
la = [eager(i) for i in range(3)]
[eager_c(i) for i in range(3)]
torch.cuda.synchronize()
h = mB.register_forward_pre_hook(pre)
r = ReplayedTrainStep(mB, cB, oB, x, t, warmup=2)
h.remove()
print("eager losses", la, "record-step loss", float(r.loss_items[0]))
diff("after record step")
if poison:
    print("poisoned bytes", r.poison())
for st in range(3, 7):
    le = eager(st)
    eager_c(st)
    x.copy_(xs[st % 3]); t.copy_(ts[st % 3])
    lr = float(r.step()[0])
    print("step", st, "eager", le, "replay", lr)
    diff(f"after step {st}")

# ---- second experiment: split the replay into fwd/bwd and optimizer, compare gradient arenas in between
print("---- split replay")
st = 7
x.copy_(xs[st % 3]); t.copy_(ts[st % 3])
# resync B to A's state first
with torch.no_grad():
    oB.params_arena.copy_(oA.params_arena); oB.mom_arena.copy_(oA.mom_arena); oB.ema_arena.copy_(oA.ema_arena)
config.bump_weight_epoch()
oA.zero_grad()
tot, it = cA(mA(x), t)
tot.backward()
torch.cuda.synchronize()
oB.prepare_step(1.0)
r.rec.run(0, r._n_fb)
torch.cuda.synchronize()
ga, gb = oA.grads_arena, oB.grads_arena
print("loss", float(it[0]), float(r.loss_items[0]), "grads equal:", torch.equal(ga, gb), float((ga - gb).abs().max()), "hyper_dev", oB._hyper_dev.tolist())
oA.step()
r.rec.run(r._n_fb, r._n_all)
torch.cuda.synchronize()
print("params equal:", torch.equal(oA.params_arena, oB.params_arena), float((oA.params_arena - oB.params_arena).abs().max()))
print("A hyper:", [g["lr"] for g in oA.param_groups], "B:", [g["lr"] for g in oB.param_groups], oA.updates, oB.updates)
