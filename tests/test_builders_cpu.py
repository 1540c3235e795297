"""CPU: the yaml / parse_model builders reproduce the reference's parameter names and shapes (no compute)."""
import os

import pytest
import torch
import yaml

CFG = os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg")


def _cfg(name, swap=None):
    cfg = yaml.safe_load(open(os.path.join(CFG, name)))
    for sec in ("backbone", "head"):
        for l in cfg[sec]:
            l[2] = (swap or {}).get(l[2], l[2])
    return cfg


def test_yolov5seg_state_dict_layout_and_param_count():
    import yolo_dual_amd as ydl
    from tests.model_shapes import script_model_state_shapes
    cfg = _cfg("yolov5_seg.yaml", {"C3_DCN": "C3"})
    m = ydl.YOLOv5Seg(cfg)
    sd = m.state_dict()
    sh = script_model_state_shapes(cfg)
    assert list(sd.keys()) == list(sh.keys())
    assert all(tuple(sd[k].shape) == tuple(sh[k]) for k in sh)
    assert sum(p.numel() for p in m.parameters()) == 16894872          # 16.89 M parameters
    # T3: yaml `number` ignored, `C3 [512, False]` -> n = 0 ;  backbone C3 n = 1
    assert [len(l.m) for l in m.backbone if isinstance(l, ydl.C3)] == [1, 1, 1, 1]
    assert [len(l.m) for l in m.head if isinstance(l, ydl.C3)] == [0, 0, 0]
    # T4: concat widths 640 / 768 / 640
    assert [m.head_out_chs[10 + i] for i in (3, 8, 13)] == [640, 768, 640]


def test_dcn_blocks_raise_parity_unpinned():
    import yolo_dual_amd as ydl
    with pytest.raises(NotImplementedError, match="parity unpinned"):
        ydl.YOLOv5Seg(_cfg("yolov5_seg.yaml"))


def test_yolov8_upsample_trap_and_v9_builds():
    import yolo_dual_amd as ydl
    m8 = ydl.YOLOv8Seg(_cfg("yolov8_seg.yaml", {"C2f_DCN": "C2f"}))
    ups = [l for l in m8.head if isinstance(l, ydl.Upsample)]
    assert all(u.size == (256, 256) and u.scale_factor is None for u in ups)      # yolov8 builder: size fallback (256,256)
    m5 = ydl.YOLOv5Seg(_cfg("yolov5_seg.yaml", {"C3_DCN": "C3"}))
    assert [u.scale_factor for u in m5.head if isinstance(u, ydl.Upsample)] == [2.0, 2.0, 2.0, 4.0]
    m9 = ydl.YOLOv9Seg(_cfg("yolov9_seg.yaml"))
    assert isinstance(m9.backbone[9], ydl.GAM) and m9.backbone[9].conv1.k == 1
    assert sum(p.numel() for p in m9.parameters()) > 5e6


def test_parse_model_semantics():
    """models/yolo.py:299-382: width/depth gains, n insertion for C3, save list, layer tags"""
    import yolo_dual_amd as ydl
    d = {"nc": 12, "depth_multiple": 0.33, "width_multiple": 0.5,
         "backbone": [[-1, 1, "Conv", [64, 6, 2, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 3, "C3", [128]],
                      [-1, 1, "Conv", [256, 3, 2]], [-1, 6, "C3", [256]], [-1, 1, "SPPF", [256, 5]]],
         "head": [[-1, 1, "nn.Upsample", ["None", 2, "nearest"]], [[-1, 2], 1, "Concat", [1]], [-1, 3, "C3", [128, False]]]}
    model, save = ydl.parse_model(d, [3])
    assert save == [2]
    assert [m.type for m in model][:3] == ["models.common.Conv", "models.common.Conv", "models.common.C3"]
    assert model[0].conv.weight.shape == (32, 3, 6, 6) and model[1].conv.weight.shape == (64, 32, 3, 3)
    assert len(model[2].m) == 1 and len(model[4].m) == 2                 # n = max(round(n*0.33), 1)
    assert model[8].cv1.conv.weight.shape[1] == 128 + 64                  # concat of 128 (up) + 64 (layer 2)
    assert [m.i for m in model] == list(range(9)) and model[7].f == [-1, 2]
    assert all(hasattr(m, "np") for m in model)


def test_conv_signature_variants():
    import torch.nn as nn
    import yolo_dual_amd as ydl
    assert ydl.Conv(8, 8, 3, 1, None, 1, False).act_code == 0            # script spelling: 7th positional = act
    assert ydl.Conv(8, 8, 3, 1, None, 1, 1, True).act_code == 1           # common.py spelling: (…, g, d, act)
    assert ydl.Conv(8, 8, 1, act=nn.ReLU()).act_code == 2
    with pytest.raises(TypeError):
        ydl.Conv(8.0, 8)
    with pytest.raises(ValueError):
        ydl.Conv(8, 8, 1, 1, None, 3)
    w = ydl.Conv(8, 16, 3).conv.weight
    assert w.shape == (16, 8, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()   # OIHW logical, KRSC physical


def test_resnet50_yaml_builder_state_dict_layout():
    """state_dict names/shapes/order of the yaml-driven ResNet50Seg equal the reference builder's (seg_diceloss_Resnet50.py:570-668):
    the layer-level down-sampling Conv is listed twice (layer and first block share it), C3 [c, False] has no inner convs"""
    import os
    import yaml
    import yolo_dual_amd as ydl
    from tests.model_shapes import resnet50_yaml_state_shapes
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg", "resnet50_seg.yaml")))
    m = ydl.ResNet50SegYaml(cfg)
    shapes, alias = resnet50_yaml_state_shapes(cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
    for k, k0 in alias.items():
        assert sd[k].data_ptr() == sd[k0].data_ptr()
    assert not any(k.startswith("head.4.m.") for k in sd)           # C3 [512, False]: n = int(False) = 0
    assert isinstance(m.head[0].act, __import__("torch").nn.ReLU)


def test_c3_dcnv3_builders():
    """C3_DCNV3 resolves in the script builders (C3_DCNV3(c1, *args)) and in parse_model (n inserted like C3)"""
    import yolo_dual_amd as ydl
    d = {"nc": 12, "depth_multiple": 1.0, "width_multiple": 1.0,
         "backbone": [[-1, 1, "Conv", [16, 3, 2]], [-1, 2, "C3_DCNV3", [16]]], "head": [[-1, 1, "Conv", [12, 1, 1]]]}
    seq, save = ydl.parse_model(d, [3])
    assert type(seq[1]).__name__ == "C3_DCNV3" and len(seq[1].m) == 2 and type(seq[1].m[0].cv2.dcnv3).__name__ == "DCNv3"
