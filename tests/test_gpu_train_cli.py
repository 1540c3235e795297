"""GPU side of the checkpoint / CLI surface: train_seg.py end to end on synthetic data (a checkpoint with the reference's keys,
--weights intersect-load, --resume), strip_optimizer on its output, and the BN-folded inference path (Conv.forward_fuse,
model.fuse(); models/common.py:61-64, models/yolo.py:140-148)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_cli_saves_loads_and_resumes(tmp_path):
    import train_seg
    import yolo_dual_amd as ydl
    sd = str(tmp_path / "run")
    common = ["--cfg", os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg", "yolov5_seg.yaml"), "--batch-size", "4",
              "--imgsz", "64", "--steps-per-epoch", "6", "--save-dir", sd, "--dtype", "f32"]
    fit = train_seg.train(train_seg.parse_opt(common + ["--epochs", "2"]))
    last, best = os.path.join(sd, "last.pt"), os.path.join(sd, "best.pt")
    assert os.path.exists(last) and os.path.exists(best) and 0.0 <= fit <= 1.0
    ck = ydl.load_checkpoint(last)
    assert set(ck) >= {"model", "optimizer", "epoch", "best_fitness"} and ck["epoch"] == 1 and ck["optimizer"] is not None
    bk = ydl.load_checkpoint(best)                      # stripped: fp16 weights, no optimizer
    assert bk["optimizer"] is None and bk["epoch"] == -1
    assert all(v.dtype == torch.float16 for v in bk["model"].values() if v.dtype.is_floating_point)
    # --weights (intersect-load) into a fresh model: every entry matches
    m = ydl.YOLOv5Seg(train_seg.build_model(train_seg.parse_opt(common))[0].yaml)
    n, tot = ydl.load_weights(m, best)
    assert n == tot
    # --resume continues from epoch 2 with the saved optimizer state
    fit2 = train_seg.train(train_seg.parse_opt(common + ["--epochs", "3", "--weights", last, "--resume"]))
    assert ydl.load_checkpoint(last)["epoch"] == 2 and fit2 >= 0.0
    ydl.set_compute_dtype("bf16")


def test_train_cli_with_gpu_letterbox(tmp_path):
    """--raw-size: uint8 samples of another size go through LetterboxGPU (the dataset's _resize_and_pad + /255) into the loop"""
    import train_seg
    import yolo_dual_amd as ydl
    fit = train_seg.train(train_seg.parse_opt(["--arch", "resnet18", "--batch-size", "4", "--imgsz", "64", "--steps-per-epoch", "4",
                                               "--epochs", "1", "--save-dir", str(tmp_path / "lb"), "--raw-size", "100x72", "--nosave"]))
    assert 0.0 <= fit <= 1.0
    ydl.set_compute_dtype("bf16")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_fused_inference_matches_eval_mode(mode):
    import yaml
    import yolo_dual_amd as ydl
    from oracle.fill import fill_state_dict
    ydl.set_compute_dtype(mode)
    try:
        cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg", "yolov5_seg.yaml")))
        for sec in ("backbone", "head"):
            for l in cfg[sec]:
                l[2] = "C3" if l[2] == "C3_DCN" else l[2]
        m = ydl.YOLOv5Seg(cfg)
        m.img_size = [96, 96]
        sd = m.state_dict()
        fill_state_dict(sd, 3, bn_stats=True)
        m.load_state_dict(sd)
        m = m.cuda().eval()
        x = torch.rand(2, 3, 96, 96, device="cuda")
        with torch.no_grad():
            ref = m(x)
            m.fuse()
            got = m(x)
        # the checker: the CPU oracle in eval mode (BatchNorm on its running statistics, as validate.run(model=ema.ema) uses the
        # model, seg_diceloss_yolov5.py:1155) — eval-mode kernels AND the BN-folded kernels (utils/torch_utils.py:248-269,
        # models/common.py:61-64) against it, not only against each other
        from oracle import ref_cpu as R
        want = R.script_model_forward({k: v.clone() for k, v in sd.items()}, cfg, x.cpu(), (96, 96), train=False)
        for name, t_ in (("eval", ref), ("fused", got)):
            e = float((t_.cpu() - want).abs().max() / want.abs().max()) if mode == "f32" else float((t_.cpu() - want).norm() / want.norm())
            assert e < (1e-4 if mode == "f32" else 5e-2), (name, e)
        # bf16: the folded weights w * gamma/sigma are rounded to bf16 once more than w itself; compared in relative L2
        err = float((got - ref).abs().max()) if mode == "f32" else float((got - ref).norm() / ref.norm())
        assert err < (2e-5 if mode == "f32" else 5e-2), err
        c = ydl.Conv(8, 16, 3, 1).cuda().eval()
        with torch.no_grad():
            xx = torch.randn(1, 8, 12, 12, device="cuda")
            a = c(xx)
            b = c.forward_fuse(xx)
        assert float((a - b).abs().max()) < (1e-5 if mode == "f32" else 5e-2)
    finally:
        ydl.set_compute_dtype("bf16")


def test_freeze_keeps_frozen_parameters_bitwise_unchanged():
    """``--freeze N`` (seg_diceloss_yolov5.py:955-959: requires_grad = False for backbone.0 .. backbone.N-1): frozen parameters get no
    gradient kernel and are never marked touched, so the fused SGD/EMA step leaves them — weights, BN affine, momentum — bit for
    bit unchanged over two optimizer steps (no weight decay, no momentum), while the layers behind them move.  The frozen layers'
    BatchNorm running statistics still advance (train mode), as in the reference."""
    import train_seg
    import yolo_dual_amd as ydl
    ydl.set_compute_dtype("f32")
    try:
        opt = train_seg.parse_opt(["--batch-size", "2", "--imgsz", "64", "--freeze", "3"])
        model, _ = train_seg.build_model(opt)
        model = model.cuda().train()
        freeze = [f"backbone.{i}." for i in range(3)]
        for k, v in model.named_parameters():
            v.requires_grad = not any(x in k for x in freeze)
        optim = ydl.smart_optimizer(model, "SGD", 0.01, 0.937, 5e-4)
        crit = ydl.SegmentationLoss(12, 0.0, None, "dice")
        before = {k: v.detach().clone() for k, v in model.named_parameters()}
        rm0 = model.state_dict()["backbone.0.bn.running_mean"].clone()
        gen = torch.Generator("cuda").manual_seed(0)
        for _ in range(2):
            x = torch.rand(2, 3, 64, 64, device="cuda", generator=gen)
            t = torch.randint(0, 12, (2, 64, 64), device="cuda", generator=gen)
            optim.zero_grad()
            loss, _ = crit(model(x), t)
            loss.backward()
            frozen_touched = [k for k, p in model.named_parameters() if any(f in k for f in freeze) and getattr(p, "_ydl_touched", False)]
            assert not frozen_touched, frozen_touched
            optim.step()
        moved = 0
        for k, v in model.named_parameters():
            if any(f in k for f in freeze):
                assert torch.equal(v.detach(), before[k]), k
            elif not torch.equal(v.detach(), before[k]):
                moved += 1
        assert moved > 20, moved
        assert not torch.equal(model.state_dict()["backbone.0.bn.running_mean"], rm0)
        # a frozen sibling pair inside a trainable model, and a frozen layer in the middle (its input still needs a gradient)
        for k, v in model.named_parameters():
            v.requires_grad = not k.startswith("backbone.4.cv1.")
        before = {k: v.detach().clone() for k, v in model.named_parameters()}
        optim.zero_grad()
        loss, _ = crit(model(x), t)
        loss.backward()
        optim.step()
        for k, v in model.named_parameters():
            if k.startswith("backbone.4.cv1."):
                assert torch.equal(v.detach(), before[k]), k
        assert not torch.equal(model.backbone[0].conv.weight.detach(), before["backbone.0.conv.weight"])
    finally:
        ydl.set_compute_dtype("bf16")
