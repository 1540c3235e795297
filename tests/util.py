"""Shared helpers for the test-suite: golden fixture loading."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    """npz fixture with 'group/key' entries exposed as nested dicts of torch tensors."""

    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.flat = {k: z[k] for k in z.files}

    def t(self, key):
        a = self.flat[key]
        return torch.from_numpy(np.array(a))

    def has(self, key):
        return key in self.flat

    def group(self, g):
        pre = g + "/"
        return {k[len(pre):]: torch.from_numpy(np.array(v)) for k, v in self.flat.items() if k.startswith(pre)}

    def strs(self, key):
        return [str(s) for s in self.flat[key].tolist()]

    def rebuild_sd(self):
        """Weights of store_weights=False fixtures: re-created with oracle.fill from the key/shape list."""
        from oracle.fill import fill_state_dict
        keys = self.strs("sd_keys")
        shapes, nd, isf = self.flat["sd_shapes"], self.flat["sd_ndim"], self.flat["sd_isfloat"]
        sd = {}
        for k, s, n, f in zip(keys, shapes, nd, isf):
            shp = tuple(int(v) for v in s[:int(n)])
            sd[k] = torch.zeros(shp, dtype=torch.float32 if f else torch.int64)
        fill_state_dict(sd, int(self.flat["fill_seed"]))
        return sd


def names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def l2_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
