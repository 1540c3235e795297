"""Training-dynamics parity (north_star: "mIoU within +-0.1 of reference").

1. ``test_training_run_reaches_the_oracle_miou``: a run that actually learns — YOLOv5Seg, 128x128 blobby masks,
   batch 8 cycling over 16 batches, CE + 0.5*Dice, SGD-nesterov with a linearly decaying learning rate, 600 steps — replayed on the HIP path in parity (f32) and
   throughput (bf16) mode against the CPU oracle's committed 4-member ensemble (tests/golden/train_curve_yolov5seg_128.npz,
   written by oracle/make_train_curve.py): the oracle reaches mIoUs of 0.77-0.87 on 64 held-out images (val_diceloss.py:37-75 metric,
   eval-mode BN); the run is chaotic, so ensembles are compared, not single trajectories.
2. ``test_short_training_run_tracks_the_oracle``: 24 steps at 96x96 against a live CPU-oracle run, per-step losses.  Parity
   mode is bitwise reproducible (deterministic split-K weight gradients), so its bound is the 2e-3 the test started with."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import ref_cpu as R
from oracle.fill import fill_state_dict
from tests.model_shapes import script_model_state_shapes

pytestmark = pytest.mark.gpu
CFG = os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg", "yolov5_seg.yaml")
CW = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)
S, BS, STEPS, LR = 96, 4, 24, 0.01


def _blobby(seed, n):
    """8x8 random class grid nearest-upsampled to SxS; the image is a class-dependent colour plus noise"""
    rs = np.random.RandomState(seed)
    grid = torch.from_numpy(rs.randint(0, 11, size=(n, 8, 8)).astype(np.int64))
    tgt = grid.repeat_interleave(S // 8, 1).repeat_interleave(S // 8, 2)
    pal = torch.from_numpy(rs.rand(12, 3).astype(np.float32))
    img = pal[tgt].permute(0, 3, 1, 2) * 0.8 + 0.2 * torch.from_numpy(rs.rand(n, 3, S, S).astype(np.float32))
    return img.contiguous(), tgt.contiguous()


def _cfg():
    cfg = yaml.safe_load(open(CFG))
    for sec in ("backbone", "head"):
        for l in cfg[sec]:
            l[2] = "C3" if l[2] == "C3_DCN" else l[2]
    return cfg


def _oracle_run(cfg, x, t, xv, tv):
    shapes = script_model_state_shapes(cfg)
    sd = {k: (torch.zeros(s) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.int64))
          for k, s in shapes.items()}
    fill_state_dict(sd, 77, bn_stats=False)
    pnames = [k for k in sd if k.endswith(".weight") or k.endswith(".bias")]
    bufs, losses = {}, []
    for st in range(STEPS):
        ps = {k: sd[k].detach().clone().requires_grad_(True) for k in pnames}
        run = dict(sd)
        run.update(ps)
        out = R.script_model_forward(run, cfg, x, (S, S))
        total, _, _ = R.seg_loss(out, t, CW, "dice")
        total.backward()
        losses.append(float(total.detach()))
        for k in pnames:
            if ps[k].grad is not None:
                bufs[k] = R.sgd_nesterov_step(sd[k], ps[k].grad, bufs.get(k), LR, 0.937, 0.0)
        for k in sd:
            if k not in ps:
                sd[k] = run[k]
    with torch.no_grad():
        pv = R.script_model_forward(dict(sd), cfg, xv, (S, S), train=False)
    miou, _ = R.miou_from_confusion(R.confusion_matrix(pv.argmax(1), tv, 12))
    return losses, miou


@pytest.fixture(scope="module")
def oracle_result():
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    cfg = _cfg()
    x, t = _blobby(1, BS)
    xv, tv = _blobby(2, BS)
    return cfg, (x, t, xv, tv), _oracle_run(cfg, x, t, xv, tv)


@pytest.mark.parametrize("mode,loss_tol,miou_tol", [("f32", 2e-3, 2e-3), ("bf16", 3e-2, 2e-2)])
def test_short_training_run_tracks_the_oracle(oracle_result, mode, loss_tol, miou_tol):
    import yolo_dual_amd as ydl
    cfg, (x, t, xv, tv), (ref_losses, ref_miou) = oracle_result
    ydl.set_compute_dtype(mode)
    try:
        m = ydl.YOLOv5Seg(cfg)
        m.img_size = [S, S]
        sd = m.state_dict()
        fill_state_dict(sd, 77, bn_stats=False)
        m.load_state_dict(sd)
        m = m.cuda().train()
        opt = ydl.FlatSGDEMA(m, lr=LR, momentum=0.937, weight_decay=0.0, ema=False)
        crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
        xs, ts = x.cuda(), t.cuda()
        losses = []
        for st in range(STEPS):
            opt.zero_grad()
            total, items = crit(m(xs), ts)
            total.backward()
            opt.step()
            losses.append(items[0])
        m.eval()
        with torch.no_grad():
            pv = m(xv.cuda())
        cm = ydl.ConfusionMatrix(12, ignore_index=11)
        cm.process_batch(pv, tv.cuda())
        miou, _ = cm.compute_iou()
    finally:
        ydl.set_compute_dtype("bf16")
    worst = max(abs(a - b) / abs(b) for a, b in zip(losses, ref_losses))
    print(f"[training parity] {mode}: worst per-step loss gap {worst:.2e}, final loss {losses[-1]:.5f} vs oracle "
          f"{ref_losses[-1]:.5f}, mIoU {miou:.5f} vs oracle {ref_miou:.5f}")
    assert worst <= loss_tol, (mode, worst, losses[-3:], ref_losses[-3:])
    assert ref_losses[-1] < ref_losses[0], "the run must actually train"
    assert abs(miou - ref_miou) <= miou_tol, (mode, miou, ref_miou)


def _blobby128(seed, n, S=128):
    """the generator of oracle/make_train_curve.py (same numpy streams)"""
    rs = np.random.RandomState(seed)
    grid = torch.from_numpy(rs.randint(0, 11, size=(n, 8, 8)).astype(np.int64))
    tgt = grid.repeat_interleave(S // 8, 1).repeat_interleave(S // 8, 2)
    pal = torch.from_numpy(np.random.RandomState(1234).rand(12, 3).astype(np.float32))
    img = pal[tgt].permute(0, 3, 1, 2) * 0.8 + 0.2 * torch.from_numpy(rs.rand(n, 3, S, S).astype(np.float32))
    return img.contiguous(), tgt.contiguous()


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_training_run_reaches_the_oracle_miou(mode):
    """The learning run is CHAOTIC: adding 1e-6 to the images of one batch moves the CPU oracle's own final mIoU between 0.77 and
    0.87 (the fixture stores that 4-member ensemble, oracle/make_train_curve.py), and the HIP path behaves the same
    (tools/chaos_probe.py).  A single-run comparison to +-0.01 therefore cannot be met by the oracle against itself; what is
    asserted instead:
      * identical trajectories before the chaos sets in: the first 10 losses of the unperturbed member match the oracle's
        (f32 1e-3 relative — measured 2.5e-6 — / bf16 2e-2), and its whole loss curve stays within 5e-3 / 1e-2 on average;
      * every member learns (mIoU >= 0.3) and none ends more than 0.06 below the oracle ensemble's worst member;
      * the ensemble MEANS agree within the north star's +-0.1.  (A tighter bound is not testable with four members: at the
        measured spread of 0.05 the difference of two 4-member means has a standard error of 0.035, and every change of kernel
        rounding redraws the HIP members — measured means so far: f32 0.812, bf16 0.853 and 0.892 vs the oracle's 0.832.)"""
    import yolo_dual_amd as ydl
    from tests.util import GOLDEN
    fx = np.load(os.path.join(GOLDEN, "train_curve_yolov5seg_128.npz"))
    S_, BS_, STEPS_, LR_, NB_ = (int(fx["hyp"][0]), int(fx["hyp"][1]), int(fx["hyp"][2]), float(fx["hyp"][3]), int(fx["hyp"][4]))
    LRF_ = float(fx["hyp"][5])
    # final evaluation on 64 held-out images (8 batches, one confusion matrix): the oracle's four members end at 0.870 / 0.833 /
    # 0.772 / 0.777 there against 0.878 / 0.856 / 0.795 / 0.781 on the first of those batches alone — the 0.1 spread between
    # members is NOT evaluation noise (an 8x larger validation set moves every member by less than 0.025 and leaves the spread
    # where it was): it is the training run diverging from a 1e-6 perturbation.  Nothing tighter than the ensemble comparison below
    # is testable, whatever the size of the validation set.
    NVAL_ = int(fx["hyp"][6])
    ref_losses, ref_final = fx["losses"], fx["ens_final64"]
    assert ref_final.min() >= 0.3, "the oracle runs must actually learn"
    finals, head, gap = [], None, None
    ydl.set_compute_dtype(mode)
    try:
        base = [_blobby128(100 + i, BS_, S_) for i in range(NB_)]
        held_out = [tuple(t.cuda() for t in _blobby128(2 + i, BS_, S_)) for i in range(NVAL_)]
        for mi, eps in enumerate(fx["ens_eps"].tolist()):
            m = ydl.YOLOv5Seg(_cfg())
            m.img_size = [S_, S_]
            sd = m.state_dict()
            fill_state_dict(sd, 77, bn_stats=False)
            m.load_state_dict(sd)
            m = m.cuda().train()
            opt = ydl.FlatSGDEMA(m, lr=LR_, momentum=0.937, weight_decay=0.0, ema=False)
            crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
            batches = [((x + np.float32(eps)) if i == 0 else x, t) for i, (x, t) in enumerate(base)]
            batches = [(x.cuda(), t.cuda()) for x, t in batches]
            losses = []
            for st in range(STEPS_):
                x, t = batches[st % NB_]
                for gparam in opt.param_groups:
                    gparam["lr"] = LR_ * (1.0 - (1.0 - LRF_) * st / STEPS_)
                opt.zero_grad()
                total, items = crit(m(x), t)
                total.backward()
                opt.step()
                losses.append(items[0])
            m.eval()
            cm = ydl.ConfusionMatrix(12, ignore_index=11)
            with torch.no_grad():
                for xv, tv in held_out:
                    cm.process_batch(m(xv), tv)
            finals.append(cm.compute_iou()[0])
            if mi == 0:
                losses = np.array(losses)
                head = (np.abs(losses[:10] - ref_losses[:10]) / ref_losses[:10]).max()
                gap = (np.abs(losses - ref_losses) / ref_losses).mean()
    finally:
        ydl.set_compute_dtype("bf16")
    finals = np.array(finals)
    print(f"[training parity 128] {mode}: final mIoU per member {np.round(finals, 4).tolist()} (mean {finals.mean():.4f}) vs oracle "
          f"{np.round(ref_final, 4).tolist()} (mean {ref_final.mean():.4f}); member 0 loss gap: first 10 steps {head:.2e}, whole run mean {gap:.2e}")
    assert head <= (1e-3 if mode == "f32" else 2e-2), head
    assert gap <= (5e-3 if mode == "f32" else 1e-2), gap
    assert finals.min() >= 0.3
    assert finals.min() >= ref_final.min() - 0.06, (finals, ref_final)
    assert abs(finals.mean() - ref_final.mean()) <= 0.1, (finals.mean(), ref_final.mean())


def test_full_size_bf16_training_tracks_the_f32_parity_mode():
    """BASELINE config 2 at its full size (640 x 640, bs 16), where the CPU oracle cannot follow a training run: the benchmarked bf16
    throughput mode against the f32 parity mode of the same HIP path (which the tests above pin to the oracle at the sizes the oracle
    reaches).  Same initial weights, same 150 steps on the same blobby batches, three members each (images of the first batch moved
    by 0 / +-1e-6, as in the 128^2 ensemble): both modes learn, the first losses agree before the run turns chaotic, and the ensemble
    mean mIoU on held-out images agrees within the north star's +-0.1."""
    import yolo_dual_amd as ydl
    S_, BS_, STEPS_, LR_, LRF_, NB_, NVAL_ = 640, 16, 150, 0.02, 0.05, 6, 2
    base = [_blobby128(300 + i, BS_, S_) for i in range(NB_)]
    held_out = [tuple(t.cuda() for t in _blobby128(40 + i, BS_, S_)) for i in range(NVAL_)]
    res = {}
    try:
        for mode in ("f32", "bf16"):
            ydl.set_compute_dtype(mode)
            finals, curves = [], []
            for eps in (0.0, 1e-6, -1e-6):
                m = ydl.YOLOv5Seg(_cfg())
                m.img_size = [S_, S_]
                sd = m.state_dict()
                fill_state_dict(sd, 77, bn_stats=False)
                m.load_state_dict(sd)
                m = m.cuda().train()
                opt = ydl.FlatSGDEMA(m, lr=LR_, momentum=0.937, weight_decay=0.0, ema=False)
                crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
                batches = [(((x + np.float32(eps)) if i == 0 else x).cuda(), t.cuda()) for i, (x, t) in enumerate(base)]
                losses = []
                for st in range(STEPS_):
                    x, t = batches[st % NB_]
                    for gparam in opt.param_groups:
                        gparam["lr"] = LR_ * (1.0 - (1.0 - LRF_) * st / STEPS_)
                    opt.zero_grad()
                    total, items = crit(m(x), t)
                    total.backward()
                    opt.step()
                    losses.append(items[0])
                m.eval()
                cm = ydl.ConfusionMatrix(12, ignore_index=11)
                with torch.no_grad():
                    for xv, tv in held_out:
                        cm.process_batch(m(xv), tv)
                finals.append(cm.compute_iou()[0])
                curves.append(np.array(losses))
                del m, opt, crit, batches
                torch.cuda.empty_cache()
            res[mode] = (np.array(finals), curves)
    finally:
        ydl.set_compute_dtype("bf16")
    f32f, bf16f = res["f32"][0], res["bf16"][0]
    head = (np.abs(res["bf16"][1][0][:8] - res["f32"][1][0][:8]) / res["f32"][1][0][:8]).max()
    print(f"[training parity 640] final mIoU per member: f32 {np.round(f32f, 4).tolist()} (mean {f32f.mean():.4f}), bf16 "
          f"{np.round(bf16f, 4).tolist()} (mean {bf16f.mean():.4f}); first 8 losses of member 0 differ by at most {head:.2e}")
    # measured: 7.1e-4 on the first losses; f32 0.669 / 0.653 / 0.652, bf16 0.656 / 0.665 / 0.663 (means 0.658 / 0.661)
    assert head <= 5e-3, head
    assert f32f.min() >= 0.3 and bf16f.min() >= 0.3, (f32f, bf16f)
    assert abs(f32f.mean() - bf16f.mean()) <= 0.05, (f32f, bf16f)        # (the north star allows 0.1)


@pytest.mark.parametrize("hw,bs", [((95, 81), 1), ((64, 96), 3), ((160, 128), 2), ((320, 320), 2)])
def test_ragged_input_sizes_match_the_oracle(hw, bs):
    """odd / non-square inputs and batch 1 through the whole yolov5 model in parity mode: every layer size becomes ragged
    (odd strides-2 outputs, bilinear concat alignment between unequal maps, the stem falls back from the space-to-depth
    form when a side is odd), forward logits, loss and the gradient of every live parameter against the CPU oracle.
    320x320 at batch 2 is the whole model at half the benchmark's linear size: its 160x160 / 80x80 layers run the 128-pixel
    tiles and the point-wise streaming kernel of the benchmark (51 200 ... 204 800 pixels per layer)."""
    import yolo_dual_amd as ydl
    from tests.util import l2_err, rel_err
    cfg = _cfg()
    H, W = hw
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.rand(bs, 3, H, W).astype(np.float32))
    t = torch.from_numpy(rs.randint(0, 12, size=(bs, H, W)).astype(np.int64))
    shapes = script_model_state_shapes(cfg)
    sd = {k: (torch.zeros(s) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.int64))
          for k, s in shapes.items()}
    fill_state_dict(sd, 21, bn_stats=False)
    pnames = [k for k in sd if k.endswith(".weight") or k.endswith(".bias")]
    ps = {k: sd[k].detach().clone().requires_grad_(True) for k in pnames}
    run = dict(sd)
    run.update(ps)
    out = R.script_model_forward(run, cfg, x, (H, W))
    total, _, _ = R.seg_loss(out, t, CW, "dice")
    total.backward()
    ydl.set_compute_dtype("f32")
    try:
        m = ydl.YOLOv5Seg(cfg)
        m.img_size = [H, W]
        m.load_state_dict(sd)
        m = m.cuda().train()
        crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
        o2 = m(x.cuda())
        tot2, items = crit(o2, t.cuda())
        tot2.backward()
        assert list(o2.shape) == list(out.shape)
        assert rel_err(o2.detach().cpu(), out.detach()) < 1e-4
        assert abs(items[0] - float(total)) <= 1e-4 * abs(float(total))
        named = dict(m.named_parameters())
        none_ref = sorted(k for k in pnames if ps[k].grad is None)
        none_got = sorted(k for k, p in named.items() if not getattr(p, "_ydl_touched", False))
        assert none_got == none_ref
        bad = {k: l2_err(named[k].grad.detach().cpu(), ps[k].grad) for k in pnames
               if ps[k].grad is not None and l2_err(named[k].grad.detach().cpu(), ps[k].grad) > 2e-3}
        assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:5]
    finally:
        ydl.set_compute_dtype("bf16")


def _cfg_dcn():
    return yaml.safe_load(open(os.path.join(os.path.dirname(CFG), "yolov9_dcnv3_seg.yaml")))


@pytest.mark.parametrize("init", ["random", "reference"])
@pytest.mark.parametrize("hw,bs", [((64, 64), 2), ((96, 96), 2)])
def test_config5_dcnv3_model_matches_the_oracle(hw, bs, init):
    """BASELINE configs[4] as its string reads — YOLOv9 backbone with C3-DCN from models/ops_dcnv3 (cfg/yolov9_dcnv3_seg.yaml:
    C3_DCNV3 of "common and yolo.py":27-38 around the DCNv3 module, modules/dcnv3.py:50-136) — as a WHOLE model in parity mode
    against the CPU oracle (oracle.ref_cpu.script_model_forward resolves C3_DCNV3 with the reference's pure-PyTorch core), anchored
    on a FLOAT64 run of the same oracle: what fp32 can reproduce of this model is measured, not assumed.
    ``init = "reference"``: the offset / mask projections start at zero as the reference initialises them (modules/dcnv3.py:101-107),
    every sampling point sits ON the integer grid and the probabilities are held to the north star's 1e-4;
    ``init = "random"``: random offset weights put sampling points next to the integer grid, where the bilinear weights have kinks —
    there the f32 oracle itself is 2.4e-4 from its f64 run, and this path is required to be no more than twice as far."""
    import yolo_dual_amd as ydl
    from tests.util import l2_err, rel_err
    cfg = _cfg_dcn()
    H, W = hw
    rs = np.random.RandomState(9)
    x = torch.from_numpy(rs.rand(bs, 3, H, W).astype(np.float32))
    t = torch.from_numpy(rs.randint(0, 12, size=(bs, H, W)).astype(np.int64))
    shapes = script_model_state_shapes(cfg)
    sd = {k: (torch.zeros(s) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.int64))
          for k, s in shapes.items()}
    fill_state_dict(sd, 31, bn_stats=False)
    if init == "reference":
        for k in sd:
            if k.endswith((".offset.weight", ".offset.bias", ".mask.weight", ".mask.bias")):
                sd[k].zero_()
    pnames = [k for k in sd if k.endswith(".weight") or k.endswith(".bias")]

    def oracle(dt):
        ps_ = {k: sd[k].detach().clone().to(dt).requires_grad_(True) for k in pnames}
        run = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
        run.update(ps_)
        out_ = R.script_model_forward(run, cfg, x.to(dt), (H, W), family="v9")
        total_, _, _ = R.seg_loss(out_, t, CW.to(dt), "dice")
        total_.backward()
        return out_.detach(), float(total_.detach()), ps_

    out, total, ps = oracle(torch.float32)
    out64, total64, ps64 = oracle(torch.float64)
    ydl.set_compute_dtype("f32")
    try:
        m = ydl.YOLOv9Seg(cfg)
        m.img_size = [H, W]
        assert sorted(m.state_dict()) == sorted(sd)
        m.load_state_dict(sd)
        m = m.cuda().train()
        crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
        o2 = m(x.cuda())
        tot2, items = crit(o2, t.cuda())
        tot2.backward()
        assert list(o2.shape) == list(out.shape)
        got = o2.detach().cpu()
        e_oracle, e_hip, e_mutual = rel_err(out, out64), rel_err(got, out64), rel_err(got, out)
        print(f"[dcn parity {init} {H}x{W}] probabilities vs f64: oracle f32 {e_oracle:.2e}, HIP f32 {e_hip:.2e}; HIP vs oracle f32 {e_mutual:.2e}")
        # probabilities: no further from the float64 result than three times the f32 oracle's own distance (measured on MI355X:
        # random offsets 1.77e-4 against the oracle's 1.72e-4; reference initialisation 8.8e-5 against 3.5e-5, i.e. 2.5x — the same
        # GAM-block amplification as for the gradients below; floor 2e-5); with the reference's initialisation additionally the
        # north star's 1e-4 against the f32 oracle (measured 7.2e-5)
        assert e_hip <= max(3.0 * e_oracle, 2e-5), (e_hip, e_oracle)
        if init == "reference":
            # 64 x 64: oracle f32 4.9e-5 from its f64 run, this path 6.6e-5 from the f32 oracle -> the 1e-4 holds.  96 x 96: the f32
            # oracle is itself 1.9e-4 from f64 (this path 1.9e-4, mutual 3.3e-4: two f32 roundings of the same ill-conditioned
            # 3 x 3 GAM BatchNorms), so 1e-4 between two f32 runs is not a property the reference has there — twice the oracle's own
            # distance is what is asked instead.
            assert e_mutual < max(1e-4, 2.0 * e_oracle), (e_mutual, e_oracle)
        assert abs(items[0] - total64) <= 1e-4 * abs(total64)
        named = dict(m.named_parameters())
        none_ref = sorted(k for k in pnames if ps[k].grad is None)
        none_got = sorted(k for k, p in named.items() if not getattr(p, "_ydl_touched", False))
        assert none_got == none_ref
        # gradients, per tensor, against the float64 run (a DCNv3 output_proj.bias sits in front of the 1x1 cv3 + train-mode BN of its
        # C3: a per-channel constant there cancels, so its gradient is mathematically zero and any f32 value is rounding noise — such
        # entries are only held to "tiny"; with the reference's zero offset / mask weights the same holds for nothing else)
        gscale = max(float(ps64[k].grad.abs().max()) for k in pnames if ps64[k].grad is not None)
        tiny = [k for k in pnames if ps64[k].grad is not None and float(ps64[k].grad.abs().max()) < 1e-6 * gscale]
        for k in tiny:
            assert float(named[k].grad.abs().max()) < 1e-3 * gscale, k
        live = [k for k in pnames if ps64[k].grad is not None and k not in tiny]
        eo = {k: l2_err(ps[k].grad, ps64[k].grad) for k in live}
        eh = {k: l2_err(named[k].grad.detach().cpu(), ps64[k].grad) for k in live}
        med_o, med_h = float(np.median(list(eo.values()))), float(np.median(list(eh.values())))
        print(f"[dcn parity {init} {H}x{W}] gradients vs f64 (relative L2): oracle f32 median {med_o:.2e} max {max(eo.values()):.2e}, "
              f"HIP f32 median {med_h:.2e} max {max(eh.values()):.2e}")
        # Gradients: FOUR times the f32 oracle's distance from the f64 gradient per tensor (floor: four times the oracle's MEDIAN
        # distance — a tensor the oracle happens to get almost exactly is not a bound), and the medians within a factor of four.
        # Why four and not two (tools/f64_anchor.py prints the per-tensor table): from the loss back to backbone.10 this path and
        # the f32 oracle are equally far from the f64 gradients (ratio 1.1-1.2, 1e-4); through backbone.9 — the GAM block, whose
        # BatchNorms see 2 x 2 x 2 = 8 values at this input size — the oracle's own distance jumps 7x (1e-4 -> 7.5e-4) and this
        # path's 20x (-> 2.3e-3), and both stay there for every layer upstream: a chaotic amplification of f32 rounding at one
        # ill-conditioned block, measured ratio 3.0-3.2 (YOLOv5Seg, no GAM, same size: 1.6-1.8 at 3e-5).
        # The f32 oracle's distance is itself a noisy sample of that amplification: between two hosts (different thread counts ->
        # different summation orders in its convolutions) its median moved 5.9e-4 -> 1.0e-3 and its worst tensor 8.4e-4 -> 2.2e-3 with
        # the reference initialisation, while this path's numbers are the same on every box (median 1.8e-3, worst 3.9e-3).  Hence
        # the absolute floor of 5e-3 under the ratio; a wrong backward is O(0.1-1) here (a sampling point exactly at -1 treated as
        # outside — the .cuh's test instead of the PyTorch core's — showed as 0.25 on the offset biases in exactly this test).
        bad = {k: (eh[k], eo[k]) for k in live if eh[k] > max(4.0 * eo[k], 4.0 * med_o, 5e-3)}
        assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:5]
        assert med_h <= max(4.0 * med_o, 1e-5), (med_h, med_o)
    finally:
        ydl.set_compute_dtype("bf16")


def test_config5_dcnv3_full_size_properties():
    """the same DCN-wired config at BASELINE's full per-GPU size (640x640, bs=16, bf16): size-independent checks — probabilities
    sum to 1, finite loss, every live parameter gets a finite gradient (non-zero where the oracle's is), the loss goes down on
    a fixed batch"""
    import yolo_dual_amd as ydl
    ydl.set_compute_dtype("bf16")
    m = ydl.YOLOv9Seg(_cfg_dcn()).cuda().train()
    opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4)
    crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
    gen = torch.Generator("cuda").manual_seed(0)
    x = torch.rand(16, 3, 640, 640, device="cuda", generator=gen)
    t = torch.randint(0, 12, (16, 640, 640), device="cuda", generator=gen)
    losses = []
    for st in range(3):
        opt.zero_grad()
        out = m(x)
        if st == 0:
            assert out.shape == (16, 12, 640, 640)
            assert float((out.sum(1) - 1).abs().max()) < 1e-4
        total, items = crit(out, t)
        assert np.isfinite(items).all()
        total.backward()
        if st == 0:
            live = {k: p for k, p in m.named_parameters() if getattr(p, "_ydl_touched", False)}
            assert any(".dcnv3." in k for k in live), "the DCNv3 modules received no gradient"
            for k, p in live.items():
                gn = float(p.grad.float().norm())
                assert np.isfinite(gn), k
                # DCNv3's offset / mask projections start at zero (modules/dcnv3.py:101-107): their weight gradients are
                # non-zero, but a zero offset weight leaves nothing else to demand of the first step
                assert gn > 0 or ".dcnv3." in k, k
        opt.step()
        losses.append(float(items[0]))
    assert losses[-1] < losses[0], losses
