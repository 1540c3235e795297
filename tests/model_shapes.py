"""state_dict name -> shape tables for whole-model fixtures (test helper; independent of the product code)."""
from collections import OrderedDict


def _conv(sh, pre, c1, c2, k):
    sh[pre + ".conv.weight"] = (c2, c1, k, k)
    sh[pre + ".bn.weight"] = (c2,)
    sh[pre + ".bn.bias"] = (c2,)
    sh[pre + ".bn.running_mean"] = (c2,)
    sh[pre + ".bn.running_var"] = (c2,)
    sh[pre + ".bn.num_batches_tracked"] = ()


def script_model_state_shapes(cfg):
    """Shapes as the script builders create them (yaml `number` ignored; C3 [c2, False] -> n=0; head `from`
    absolute): seg_diceloss_yolov5.py:537-628, yolov8/seg_jaccardloss_yolov8.py:528-660."""
    sh = OrderedDict()
    chs = []

    def add(pre, kind, c1, args):
        if kind == "Conv":
            _conv(sh, pre, c1, args[0], args[1] if len(args) > 1 else 1)
            return args[0]
        if kind in ("C3", "C3k2"):
            c2, n = args[0], int(args[1]) if len(args) > 1 else 1
            c_ = int(c2 * 0.5)
            _conv(sh, pre + ".cv1", c1, c_, 1)
            _conv(sh, pre + ".cv2", c1, c_, 1)
            _conv(sh, pre + ".cv3", 2 * c_, c2, 1)
            for i in range(n):
                _conv(sh, f"{pre}.m.{i}", c_, c_, 3)
            return c2
        if kind == "C3_DCNV3":                  # "common and yolo.py":2-38 + modules/dcnv3.py:50-107
            c2, n = args[0], int(args[1]) if len(args) > 1 else 1
            g = int(args[3]) if len(args) > 3 else 1
            c_ = int(c2 * 0.5)
            _conv(sh, pre + ".cv1", c1, c_, 1)
            _conv(sh, pre + ".cv2", c1, c_, 1)
            _conv(sh, pre + ".cv3", 2 * c_, c2, 1)
            for i in range(n):
                b = f"{pre}.m.{i}"
                _conv(sh, b + ".cv1", c_, c_, 1)
                _conv(sh, b + ".cv2.conv", c_, c_, 1)
                d = b + ".cv2.dcnv3"
                sh[d + ".dw_conv.conv.weight"] = (c_, 1, 3, 3)
                for k2, v2 in ((".bn.weight", (c_,)), (".bn.bias", (c_,)), (".bn.running_mean", (c_,)), (".bn.running_var", (c_,)),
                               (".bn.num_batches_tracked", ())):
                    sh[d + ".dw_conv" + k2] = v2
                sh[d + ".offset.weight"] = (g * 9 * 2, c_); sh[d + ".offset.bias"] = (g * 9 * 2,)
                sh[d + ".mask.weight"] = (g * 9, c_); sh[d + ".mask.bias"] = (g * 9,)
                sh[d + ".input_proj.weight"] = (c_, c_); sh[d + ".input_proj.bias"] = (c_,)
                sh[d + ".output_proj.weight"] = (c_, c_); sh[d + ".output_proj.bias"] = (c_,)
            return c2
        if kind == "C2f":
            c2, n = args[0], int(args[1]) if len(args) > 1 else 1
            c = int(c2 * 0.5)
            _conv(sh, pre + ".cv1", c1, 2 * c, 1)
            _conv(sh, pre + ".cv2", (2 + n) * c, c2, 1)
            for i in range(n):
                _conv(sh, f"{pre}.m.{i}", c, c, 3)
            return c2
        if kind == "SPPF":
            c2 = args[0]
            _conv(sh, pre + ".cv1", c1, c1 // 2, 1)
            _conv(sh, pre + ".cv2", (c1 // 2) * 4, c2, 1)
            return c2
        if kind == "GAM":                       # seg_diceloss_yolov9.py:475-490: GAM(c, k=1, s=1, e=0.25)
            k = args[0] if args else 1
            c_ = int(c1 * 0.25)
            _conv(sh, pre + ".conv1", c1, c_, k)
            _conv(sh, pre + ".conv2", c_, c1, k)
            _conv(sh, pre + ".conv3", c_, c1, k)
            return c1
        if kind in ("nn.Upsample", "Upsample", "Concat", "nn.Softmax"):
            return c1
        raise NotImplementedError(kind)

    prev = 3
    for i, (frm, _n, kind, args) in enumerate(cfg["backbone"]):
        c1 = prev if frm == -1 else chs[frm]
        prev = add(f"backbone.{i}", kind, c1, args)
        chs.append(prev)
    for i, (frm, _n, kind, args) in enumerate(cfg["head"]):
        c1 = sum(chs[f] for f in frm) if isinstance(frm, list) else chs[frm]
        chs.append(add(f"head.{i}", kind, c1, args))
    return sh


def resnet_seg_state_shapes(kind, nc):
    """ResNet18Seg / ResNet50Seg (layer4 is constructed although never run): Resnet18:243-375."""
    sh = OrderedDict()
    _conv(sh, "backbone.stem.0", 3, 64, 7)
    inc = 64
    blocks = (2, 2, 2, 2) if kind == "basic" else (3, 4, 6, 3)
    exp = 1 if kind == "basic" else 4
    feat = []
    for li, (mid, nb) in enumerate(zip((64, 128, 256, 512), blocks)):
        for bi in range(nb):
            pre = f"backbone.layer{li + 1}.{bi}"
            stride = 2 if (li > 0 and bi == 0) else 1
            if kind == "basic":
                _conv(sh, pre + ".conv1", inc, mid, 3)
                _conv(sh, pre + ".conv2", mid, mid, 3)
            else:
                _conv(sh, pre + ".conv1", inc, mid, 1)
                _conv(sh, pre + ".conv2", mid, mid, 3)
                _conv(sh, pre + ".conv3", mid, mid * 4, 1)
            if bi == 0 and (stride != 1 or inc != mid * exp):
                _conv(sh, pre + ".downsample", inc, mid * exp, 1)
            inc = mid * exp
        feat.append(inc)
    for i, c in enumerate(feat[:3]):
        _conv(sh, f"head.lateral_convs.{i}", c, 128, 1)
    _conv(sh, "head.final_conv.0", 384, 256, 3)
    _conv(sh, "head.final_conv.1", 256, nc, 1)
    return sh


def resnet50_yaml_state_shapes(cfg):
    """yaml-driven ResNet50Seg (unet-lite/Resnet50/seg_diceloss_Resnet50.py:438-668).  Returns (shapes, aliases): the layer's
    down-sampling Conv appears under ``backbone.i.downsample`` and again under ``backbone.i.layer.0.downsample`` (one module,
    two names); ``aliases`` maps the second name to the first."""
    sh, alias = OrderedDict(), {}
    chs, prev = [], 3
    for i, (frm, _n, kind, args) in enumerate(cfg["backbone"]):
        c1 = prev if frm == -1 else chs[frm]
        pre = f"backbone.{i}"
        if kind == "ResNetStem":
            out = int(args[0])
            _conv(sh, pre + ".stem.0", 3, out, 7)
        else:
            out, nb, stride = int(args[0]), int(args[1]), int(args[2]) if len(args) >= 3 else 1
            mid = out // 4
            has_ds = stride != 1 or c1 != out
            if has_ds:
                _conv(sh, pre + ".downsample", c1, out, 1)
            inc = c1
            for b in range(nb):
                bp = f"{pre}.layer.{b}"
                _conv(sh, bp + ".conv1", inc, mid, 1)
                _conv(sh, bp + ".conv2", mid, mid, 3)
                _conv(sh, bp + ".conv3", mid, out, 1)
                if b == 0 and has_ds:
                    _conv(sh, bp + ".downsample", c1, out, 1)
                    for suf in (".conv.weight", ".bn.weight", ".bn.bias", ".bn.running_mean", ".bn.running_var", ".bn.num_batches_tracked"):
                        alias[bp + ".downsample" + suf] = pre + ".downsample" + suf
                inc = out
        chs.append(out)
        prev = out
    for i, (frm, _n, kind, args) in enumerate(cfg["head"]):
        c1 = sum(chs[f] for f in frm) if isinstance(frm, list) else chs[frm]
        pre = f"head.{i}"
        if kind == "Conv":
            out = int(args[0])
            _conv(sh, pre, c1, out, int(args[1]) if len(args) >= 2 else 1)
        elif kind == "C3":
            out, n = int(args[0]), int(args[1]) if len(args) >= 2 else 1
            c_ = int(out * 0.5)
            _conv(sh, pre + ".cv1", c1, c_, 1)
            _conv(sh, pre + ".cv2", c1, c_, 1)
            _conv(sh, pre + ".cv3", 2 * c_, out, 1)
            for j in range(n):
                _conv(sh, f"{pre}.m.{j}", c_, c_, 3)
        else:
            out = c1
        chs.append(out)
    return sh, alias
