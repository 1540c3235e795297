"""Deterministic, version-independent tensor fills for fixtures (TEST INFRASTRUCTURE).

Uses numpy's legacy ``RandomState`` (bit-stable by specification) so that a golden fixture generated in the
build container and a test running on the GPU box agree on weights without storing them.
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np
import torch


def rs_tensor(seed: int, shape, scale: float = 1.0) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(size=tuple(shape)) * scale).astype(np.float32))


def fill_state_dict(sd: Dict[str, torch.Tensor], seed: int, bn_stats: bool = True) -> None:
    """Fill every float entry of a state_dict in place.  Conv weights ~ N(0, 2/fan_in) (keeps activations
    O(1) through deep stacks), BN gamma in [0.5,1.5], beta in [-0.3,0.3]; running stats non-trivial when
    ``bn_stats`` (per-block fixtures) or torch defaults (whole-model fixtures).  The per-key seed is
    ``seed + crc32(key)`` so the fill does not depend on dict order."""
    with torch.no_grad():
        for k, v in sd.items():
            if not v.dtype.is_floating_point:
                continue
            rs = np.random.RandomState((seed + zlib.crc32(k.encode())) % (2 ** 31))
            if k.endswith("running_mean"):
                a = rs.standard_normal(v.shape) * 0.2 if bn_stats else np.zeros(v.shape)
            elif k.endswith("running_var"):
                a = rs.uniform(0.5, 1.5, v.shape) if bn_stats else np.ones(v.shape)
            elif k.endswith("bn.weight") or (v.dim() == 1 and k.endswith(".weight")):
                a = rs.uniform(0.5, 1.5, v.shape)
            elif k.endswith("bias"):
                a = rs.uniform(-0.3, 0.3, v.shape)
            elif v.dim() >= 2:
                fan_in = int(np.prod(v.shape[1:]))
                a = rs.standard_normal(v.shape) * np.sqrt(2.0 / fan_in)
            else:
                a = rs.standard_normal(v.shape)
            v.copy_(torch.from_numpy(np.asarray(a, dtype=np.float32)).view_as(v))
