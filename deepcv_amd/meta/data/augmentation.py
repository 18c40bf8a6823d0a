""" GPU augmentation: AugMix-style chains of the reference's image operations, applied per sample on the device and fused with
`ToTensor` + `Normalize` (`dcv_images_u8_augment_normalize`).

The reference's `augmentation.py` applies PIL operations on the CPU (`src/deepcv/meta/data/augmentation.py:65-132`) and mixes chains like AugMix
(`augment_and_mix`, `:165-203`); its recipe parser is unfinished (`:152-154`), so the behaviour is DEFINED here (parity unpinned, like the rest of the
oracle): the subset with exact integer semantics — posterize, solarize, translate_x, translate_y, brightness — with the reference's parameter
distributions (`:73-124`), `chains ~ U{chains_count}`, `depth ~ U{chain_depth}`, weights ~ Dirichlet, mix ~ Beta.  All random draws happen on the
host from a seeded `numpy.random.RandomState`, so a run is reproducible and the CPU restatement (`oracle/augment_ref.py`) can replay them. """
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = ['AUG_OPS', 'AUG_SAMPLE_DTYPE', 'sample_recipes', 'DeviceAugmentNormalize']

MAX_CHAINS, MAX_OPS = 3, 3
AUG_OPS: Dict[str, int] = {'posterize': 1, 'solarize': 2, 'translate_x': 3, 'translate_y': 4, 'brightness': 5}
# `dcv_aug_sample` (include/deepcv_hip.h)
AUG_SAMPLE_DTYPE = np.dtype([('op', '<i4', (MAX_CHAINS, MAX_OPS)), ('param', '<i4', (MAX_CHAINS, MAX_OPS)), ('weight', '<f4', (MAX_CHAINS,)), ('mix', '<f4')])


def _draw_param(rs: np.random.RandomState, op: str, severity: float, h: int, w: int) -> int:
    if op == 'posterize':        # augmentation.py:73-75
        bits = rs.uniform(2, 5)
        return int(min(8, max(1, int(bits + (1. - severity) * (8. - bits)))))
    if op == 'solarize':         # :83-85
        return int(rs.uniform((1. - severity) * (220 - 50) + 50, 220))
    if op == 'translate_x':      # :98-100
        return int(rs.uniform(-severity, severity) * (w / 3.))
    if op == 'translate_y':      # :103-105
        return int(rs.uniform(-severity, severity) * (h / 3.))
    if op == 'brightness':       # :120-123, factor in 1/256 steps
        return int(round(rs.uniform(0.3, max(severity * 1.8, 0.3)) * 256))
    raise ValueError(f'unknown augmentation op "{op}" (supported: {sorted(AUG_OPS)})')


def sample_recipes(n: int, h: int, w: int, rs: np.random.RandomState, severity: float = 0.3, ops: Sequence[str] = tuple(AUG_OPS), chains_count: Tuple[int, int] = (1, 3),
                   chain_depth: Tuple[int, int] = (1, 3), transform_chains_dirichlet: float = 1., mix_with_original_beta: float = 1.) -> np.ndarray:
    """ Per-sample AugMix recipes as a `[n]` array of `AUG_SAMPLE_DTYPE` (reference `augment_and_mix`, `augmentation.py:165-203`). """
    out = np.zeros(n, dtype=AUG_SAMPLE_DTYPE)
    ops = list(ops)
    for i in range(n):
        k = int(rs.randint(chains_count[0], min(chains_count[1], MAX_CHAINS) + 1))
        out['weight'][i, :k] = rs.dirichlet([transform_chains_dirichlet] * k).astype(np.float32)
        out['mix'][i] = np.float32(rs.beta(mix_with_original_beta, mix_with_original_beta))
        for c in range(k):
            depth = int(rs.randint(chain_depth[0], min(chain_depth[1], MAX_OPS) + 1))
            for d in range(depth):
                name = ops[int(rs.randint(0, len(ops)))]
                out['op'][i, c, d] = AUG_OPS[name]
                out['param'][i, c, d] = _draw_param(rs, name, severity, h, w)
    return out


class DeviceAugmentNormalize:
    """ Callable for `PrefetchedLoader(device_transform=...)` / `hp['device_transform']`: raw `[N, H, W, C]` uint8 device batch -> augmented, normalised
    `[N, C, H, W]` fp32.  `last_recipes` keeps the recipes of the last batch (for inspection / replay by the oracle). """

    def __init__(self, mean: Sequence[float], std: Sequence[float], severity: float = 0.3, seed: int = 434546, **recipe_kwargs):
        self.mean, self.std = [float(m) for m in mean], [float(s) for s in std]
        self.severity, self.recipe_kwargs = float(severity), recipe_kwargs
        self.rs = np.random.RandomState(seed)
        self.last_recipes: Optional[np.ndarray] = None
        self._dev = {}

    def __call__(self, images_u8: torch.Tensor, recipes: Optional[np.ndarray] = None) -> torch.Tensor:
        from ... import _native as nat
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4:
            raise TypeError(f'DeviceAugmentNormalize expects a [N, H, W, C] uint8 batch, got {images_u8.dtype} {tuple(images_u8.shape)}')
        nat.require_device(images_u8, 'DeviceAugmentNormalize input')
        n, h, w, c = images_u8.shape
        if c != len(self.mean) or c > 4:
            raise ValueError(f'DeviceAugmentNormalize: {c} channels but {len(self.mean)} mean values (at most 4 channels)')
        if recipes is None:
            recipes = sample_recipes(n, h, w, self.rs, self.severity, **self.recipe_kwargs)
        self.last_recipes = recipes
        images_u8 = images_u8.contiguous()
        dev = images_u8.device
        if dev not in self._dev:
            self._dev[dev] = (torch.tensor(self.mean, dtype=torch.float32, device=dev), torch.tensor([1.0 / s for s in self.std], dtype=torch.float32, device=dev))
        mean, inv_std = self._dev[dev]
        rec = torch.from_numpy(recipes.view(np.uint8).reshape(n, AUG_SAMPLE_DTYPE.itemsize).copy()).to(dev, non_blocking=False)
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
        nat.check(nat.lib.dcv_images_u8_augment_normalize(images_u8.data_ptr(), out.data_ptr(), n, h, w, c, nat.LAYOUT_NCHW, rec.data_ptr(), mean.data_ptr(),
                                                          inv_std.data_ptr(), nat.current_stream(dev)), 'dcv_images_u8_augment_normalize')
        return out
