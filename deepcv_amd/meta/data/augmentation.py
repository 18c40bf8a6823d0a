""" GPU augmentation: AugMix-style chains of the reference's image operations, applied per sample on the device and fused with
`ToTensor` + `Normalize` (`dcv_images_u8_augment_normalize`).

The reference applies PIL operations on the CPU (`src/deepcv/meta/data/augmentation.py:65-132`) and mixes chains like AugMix (`augment_and_mix`, `:165-203`);
all 13 operations of its `AUGMENTATION_OPS` list (`:132`) are available here — autocontrast, equalize, posterize, rotate, solarize, shear_x, shear_y,
translate_x, translate_y, color, contrast, brightness, sharpness — each with Pillow's exact arithmetic (the kernels are checked bit for bit against the numpy
restatement `oracle/augment_ref.py`, which is pinned against Pillow itself) and the reference's parameter distributions (`:73-130`).  The reference's recipe
parser is unfinished (`:152-154`), so the RECIPE is defined here: `chains ~ U{chains_count}`, `depth ~ U{chain_depth}`, weights ~ Dirichlet, mix ~ Beta, like
`augment_and_mix`.  All random draws (and the AFFINE matrices Pillow's `Image.rotate` / `transform` derive from them) are made on the host from a seeded
`numpy.random.RandomState`, so a run is reproducible and the oracle can replay it. """
import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = ['AUG_OPS', 'AUGMIX_RECIPE_DTYPE', 'sample_recipes', 'make_op', 'DeviceAugmentNormalize']

MAX_CHAINS, MAX_OPS = 3, 3
# codes DCV_AUG_* of include/deepcv_hip.h, in the order of the reference's AUGMENTATION_OPS list where new
AUG_OPS: Dict[str, int] = {'posterize': 1, 'solarize': 2, 'translate_x': 3, 'translate_y': 4, 'brightness': 5, 'autocontrast': 6, 'equalize': 7, 'rotate': 8,
                           'shear_x': 9, 'shear_y': 10, 'color': 11, 'contrast': 12, 'sharpness': 13}
# `dcv_augmix_recipe` (include/deepcv_hip.h)
AUGMIX_RECIPE_DTYPE = np.dtype({'names': ['op', 'ip', 'fp', 'weight', 'mix', 'm'],
                                'formats': [('<i4', (MAX_CHAINS, MAX_OPS)), ('<i4', (MAX_CHAINS, MAX_OPS)), ('<f4', (MAX_CHAINS, MAX_OPS)), ('<f4', (MAX_CHAINS,)), '<f4',
                                            ('<f8', (MAX_CHAINS, MAX_OPS, 6))],
                                'offsets': [0, 36, 72, 108, 120, 128], 'itemsize': 560})


def _rotate_matrix(w: int, h: int, degrees: float):
    """ Pillow `Image.rotate(degrees)` (expand=False): inverse AFFINE matrix about the image centre. """
    angle = -math.radians(degrees % 360.0)
    m = [round(math.cos(angle), 15), round(math.sin(angle), 15), 0.0, round(-math.sin(angle), 15), round(math.cos(angle), 15), 0.0]
    cx, cy = w / 2.0, h / 2.0
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy
    return m


def make_op(name: str, h: int, w: int, value=None) -> Tuple[int, int, float, Sequence[float]]:
    """ (code, ip, fp, m) of operation `name` with its reference-style parameter `value`: bits (posterize), threshold (solarize), pixels (translate_*),
    degrees (rotate), shear factor (shear_*), enhancement factor (color / contrast / brightness / sharpness); none for autocontrast / equalize. """
    code, ip, fp, m = AUG_OPS[name], 0, 0., [0.] * 6
    if name in ('posterize', 'solarize'):
        ip = int(value)
    elif name == 'translate_x':
        ip, m = int(value), [1., 0., float(int(value)), 0., 1., 0.]          # augmentation.py:100
    elif name == 'translate_y':
        ip, m = int(value), [1., 0., 0., 0., 1., float(int(value))]          # :105
    elif name == 'rotate':
        fp, m = float(value), _rotate_matrix(w, h, float(value))            # :80
    elif name == 'shear_x':
        fp, m = float(value), [1., float(value), 0., 0., 1., 0.]             # :90
    elif name == 'shear_y':
        fp, m = float(value), [1., 0., 0., float(value), 1., 0.]             # :95
    elif name in ('color', 'contrast', 'brightness', 'sharpness'):
        fp = float(np.float32(value))
    return code, ip, fp, m


def _draw_value(rs: np.random.RandomState, op: str, severity: float, h: int, w: int):
    if op == 'posterize':        # augmentation.py:73-75
        bits = rs.uniform(2, 5)
        return int(min(8, max(1, int(bits + (1. - severity) * (8. - bits)))))
    if op == 'rotate':           # :78-80
        return rs.uniform(-severity, severity) * 30.
    if op == 'solarize':         # :83-85
        return int(rs.uniform((1. - severity) * (220 - 50) + 50, 220))
    if op in ('shear_x', 'shear_y'):     # :88-95
        return rs.uniform(-severity, severity) * 0.3
    if op == 'translate_x':      # :98-100
        return int(rs.uniform(-severity, severity) * (w / 3.))
    if op == 'translate_y':      # :103-105
        return int(rs.uniform(-severity, severity) * (h / 3.))
    if op in ('color', 'contrast', 'brightness', 'sharpness'):       # :108-130
        return rs.uniform(0.3, max(severity * 1.8, 0.3))
    if op in ('autocontrast', 'equalize'):                           # :65-70
        return None
    raise ValueError(f'unknown augmentation op "{op}" (supported: {sorted(AUG_OPS)})')


def sample_recipes(n: int, h: int, w: int, rs: np.random.RandomState, severity: float = 0.3, ops: Sequence[str] = tuple(AUG_OPS), chains_count: Tuple[int, int] = (1, 3),
                   chain_depth: Tuple[int, int] = (1, 3), transform_chains_dirichlet: float = 1., mix_with_original_beta: float = 1.) -> np.ndarray:
    """ Per-sample AugMix recipes as a `[n]` array of `AUGMIX_RECIPE_DTYPE` (reference `augment_and_mix`, `augmentation.py:165-203`). """
    out = np.zeros(n, dtype=AUGMIX_RECIPE_DTYPE)
    ops = list(ops)
    for i in range(n):
        k = int(rs.randint(chains_count[0], min(chains_count[1], MAX_CHAINS) + 1))
        out['weight'][i, :k] = rs.dirichlet([transform_chains_dirichlet] * k).astype(np.float32)
        out['mix'][i] = np.float32(rs.beta(mix_with_original_beta, mix_with_original_beta))
        for c in range(k):
            depth = int(rs.randint(chain_depth[0], min(chain_depth[1], MAX_OPS) + 1))
            for d in range(depth):
                name = ops[int(rs.randint(0, len(ops)))]
                code, ip, fp, m = make_op(name, h, w, _draw_value(rs, name, severity, h, w))
                out['op'][i, c, d], out['ip'][i, c, d], out['fp'][i, c, d], out['m'][i, c, d] = code, ip, fp, m
    return out


class DeviceAugmentNormalize:
    """ Callable for `PrefetchedLoader(device_transform=...)` / `hp['device_transform']`: raw `[N, H, W, C]` uint8 device batch (C = 1 or 3) -> augmented,
    normalised `[N, C, H, W]` fp32.  `last_recipes` keeps the recipes of the last batch and `last_chains` the uint8 image each chain ended with
    (`[N, MAX_CHAINS, H, W, C]`, a view of the scratch buffer: for inspection / replay by the oracle). """

    def __init__(self, mean: Sequence[float], std: Sequence[float], severity: float = 0.3, seed: int = 434546, **recipe_kwargs):
        self.mean, self.std = [float(m) for m in mean], [float(s) for s in std]
        self.severity, self.recipe_kwargs = float(severity), recipe_kwargs
        self.rs = np.random.RandomState(seed)
        self.last_recipes: Optional[np.ndarray] = None
        self.last_chains: Optional[torch.Tensor] = None
        self._dev = {}

    def __call__(self, images_u8: torch.Tensor, recipes: Optional[np.ndarray] = None) -> torch.Tensor:
        from ... import _native as nat
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4:
            raise TypeError(f'DeviceAugmentNormalize expects a [N, H, W, C] uint8 batch, got {images_u8.dtype} {tuple(images_u8.shape)}')
        nat.require_device(images_u8, 'DeviceAugmentNormalize input')
        n, h, w, c = images_u8.shape
        if c != len(self.mean) or c not in (1, 3):
            raise ValueError(f'DeviceAugmentNormalize: {c} channels but {len(self.mean)} mean values (grey-level or RGB images)')
        if recipes is None:
            recipes = sample_recipes(n, h, w, self.rs, self.severity, **self.recipe_kwargs)
        self.last_recipes = recipes
        images_u8 = images_u8.contiguous()
        dev = images_u8.device
        if dev not in self._dev:
            self._dev[dev] = (torch.tensor(self.mean, dtype=torch.float32, device=dev), torch.tensor([1.0 / s for s in self.std], dtype=torch.float32, device=dev))
        mean, inv_std = self._dev[dev]
        rec = torch.from_numpy(np.ascontiguousarray(recipes).view(np.uint8).reshape(n, AUGMIX_RECIPE_DTYPE.itemsize).copy()).to(dev, non_blocking=False)
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
        scratch = torch.empty(int(nat.lib.dcv_images_u8_augment_scratch_bytes(n, h, w, c)), dtype=torch.uint8, device=dev)
        nat.check(nat.lib.dcv_images_u8_augment_normalize(images_u8.data_ptr(), out.data_ptr(), n, h, w, c, nat.LAYOUT_NCHW, rec.data_ptr(), mean.data_ptr(),
                                                          inv_std.data_ptr(), scratch.data_ptr(), scratch.numel(), nat.current_stream(dev)), 'dcv_images_u8_augment_normalize')
        self.last_chains = scratch[:n * MAX_CHAINS * 2 * h * w * c].view(n, MAX_CHAINS, 2, h, w, c)[:, :, 0]
        return out
