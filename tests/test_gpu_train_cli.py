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
