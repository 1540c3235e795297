"""Checkpoint interop, CPU side: name+shape intersection, the {'model','optimizer','epoch','best_fitness'} dict,
strip_optimizer, Conv+BN folding and the --cfg/--weights argument surface — against the reference's own definitions
(utils/general.py:255-257,1004-1018, utils/torch_utils.py:248-269, seg_diceloss_yolov5.py:944-952,1204-1212,1235-1276)."""
import os

import pytest
import torch
import torch.nn as nn


def test_intersect_dicts_matches_keys_shapes_and_excludes():
    from yolo_dual_amd.checkpoint import intersect_dicts
    da = {"a.w": torch.zeros(3, 4), "b.w": torch.zeros(2), "c.w": torch.zeros(5), "anchor.x": torch.zeros(1)}
    db = {"a.w": torch.ones(3, 4), "b.w": torch.ones(3), "c.w": torch.ones(5), "anchor.x": torch.ones(1)}
    out = intersect_dicts(da, db, exclude=["anchor"])
    assert sorted(out) == ["a.w", "c.w"] and out["a.w"] is da["a.w"]


def test_checkpoint_roundtrip_and_strip(tmp_path):
    import yolo_dual_amd as ydl
    m = ydl.C3(16, 16, 1)
    sd = m.state_dict()
    f = str(tmp_path / "last.pt")
    ydl.save_checkpoint(f, sd, None, epoch=3, best_fitness=0.5, ema_state={k: v + 1 if v.dtype.is_floating_point else v for k, v in sd.items()},
                        updates=7)
    ck = ydl.load_checkpoint(f)
    assert set(ck) >= {"model", "optimizer", "epoch", "best_fitness"} and ck["epoch"] == 3 and ck["best_fitness"] == 0.5
    assert list(ck["model"].keys()) == list(sd.keys())
    m2 = ydl.C3(16, 16, 1)
    n, tot = ydl.load_weights(m2, f)
    assert n == tot == len(sd)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k]), k
    mb = ydl.strip_optimizer(f)
    ck2 = ydl.load_checkpoint(f)
    assert mb > 0 and ck2["epoch"] == -1 and ck2["optimizer"] is None and ck2["best_fitness"] is None and ck2["ema"] is None
    w = ck2["model"]["cv1.conv.weight"]
    assert w.dtype == torch.float16 and torch.allclose(w.float(), sd["cv1.conv.weight"] + 1, atol=2e-3)   # EMA replaced the model
    # partial match: a model with a different width only takes the entries whose shapes agree
    m3 = ydl.C3(16, 32, 1)
    n3, tot3 = ydl.load_weights(m3, f)
    assert 0 <= n3 < tot3


def test_pickled_module_checkpoints_are_refused(tmp_path):
    """the reference pickles whole modules ('model': ema.ema); those are never unpickled here"""
    import yolo_dual_amd as ydl
    f = str(tmp_path / "ref_style.pt")
    torch.save({"model": nn.Linear(2, 2), "epoch": 0}, f)
    with pytest.raises(RuntimeError, match="weights_only"):
        ydl.load_checkpoint(f)


def test_fuse_conv_and_bn_equals_eval_bn_of_conv():
    from yolo_dual_amd.checkpoint import fuse_conv_and_bn
    torch.manual_seed(0)
    conv = nn.Conv2d(5, 7, 3, 1, 1, bias=False)
    bn = nn.BatchNorm2d(7)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    bn.eval()
    x = torch.randn(2, 5, 9, 9)
    wf, bf = fuse_conv_and_bn(conv, bn)
    ref = bn(conv(x))
    got = torch.nn.functional.conv2d(x, wf, bf, 1, 1)
    assert torch.allclose(got, ref, atol=1e-5, rtol=1e-5)


def test_cli_surface_has_the_reference_flags():
    import train_seg
    opt = train_seg.parse_opt(["--cfg", "yolo_dual_amd/cfg/yolov5_seg.yaml", "--weights", "x.pt", "--batch-size", "16", "--img", "320",
                               "--epochs", "2", "--cos-lr", "--resume", "--freeze", "3", "--label-smoothing", "0.1"])
    assert opt.cfg.endswith("yolov5_seg.yaml") and opt.weights == "x.pt" and opt.batch_size == 16 and opt.imgsz == 320
    assert opt.cos_lr and opt.resume is True and opt.freeze == [3] and opt.label_smoothing == 0.1
    cw = train_seg.class_weights("", 12)
    assert cw.tolist() == [1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1]
    assert train_seg.class_weights("1,2,3", 3).tolist() == [1, 2, 3]
