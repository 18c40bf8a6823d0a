""" Generates tests/golden/augment_pil.npz: outputs of PILLOW ITSELF (the third-party library whose operations the reference calls,
`src/deepcv/meta/data/augmentation.py:65-132`) for every augmentation operation on small seeded images, with the reference-style parameter of each call.
The fixture is DATA (inputs, parameters, Pillow's outputs); `oracle/augment_ref.py` restates the algorithms and `tests/test_augment_cpu.py` compares bit for bit.
usage: python tests/golden/make_augment_golden.py   (needs Pillow; written with Pillow 12.2.0) """
from pathlib import Path

import numpy as np
import PIL
from PIL import Image, ImageEnhance, ImageOps

rs = np.random.RandomState(20261004)


def pil(a):
    return Image.fromarray(a if a.shape[2] == 3 else a[:, :, 0], 'RGB' if a.shape[2] == 3 else 'L')


def arr(im, c):
    a = np.asarray(im)
    return (a if c == 3 else a[:, :, None]).copy()


images = {'rgb': rs.randint(0, 256, (18, 22, 3)).astype(np.uint8), 'grey': rs.randint(0, 256, (13, 9, 1)).astype(np.uint8),
          'narrow': rs.randint(60, 190, (16, 16, 3)).astype(np.uint8),                       # autocontrast / equalize with a reduced range
          'flat': np.full((8, 8, 3), 77, dtype=np.uint8)}                                    # degenerate histograms
CALLS = [('autocontrast', None), ('equalize', None), ('posterize', 2), ('posterize', 5), ('posterize', 8), ('solarize', 50), ('solarize', 171),
         ('rotate', 12.3), ('rotate', -29.75), ('rotate', 0.4), ('shear_x', 0.21), ('shear_x', -0.3), ('shear_y', 0.17), ('shear_y', -0.09),
         ('translate_x', 3), ('translate_x', -5), ('translate_y', 2), ('translate_y', -4),
         ('color', 0.3), ('color', 1.37), ('contrast', 0.55), ('contrast', 1.8), ('brightness', 0.3), ('brightness', 1.71), ('sharpness', 0.42), ('sharpness', 1.66)]


def call(name, im, v):
    w, h = im.size
    if name == 'autocontrast':
        return ImageOps.autocontrast(im)
    if name == 'equalize':
        return ImageOps.equalize(im)
    if name == 'posterize':
        return ImageOps.posterize(im, v)
    if name == 'solarize':
        return ImageOps.solarize(im, v)
    if name == 'rotate':
        return im.rotate(v, resample=Image.BILINEAR)
    if name == 'shear_x':
        return im.transform(im.size, Image.AFFINE, (1, v, 0, 0, 1, 0), resample=Image.BILINEAR)
    if name == 'shear_y':
        return im.transform(im.size, Image.AFFINE, (1, 0, 0, v, 1, 0), resample=Image.BILINEAR)
    if name == 'translate_x':
        return im.transform(im.size, Image.AFFINE, (1, 0, v, 0, 1, 0), resample=Image.BILINEAR)
    if name == 'translate_y':
        return im.transform(im.size, Image.AFFINE, (1, 0, 0, 0, 1, v), resample=Image.BILINEAR)
    enh = {'color': ImageEnhance.Color, 'contrast': ImageEnhance.Contrast, 'brightness': ImageEnhance.Brightness, 'sharpness': ImageEnhance.Sharpness}[name]
    return enh(im).enhance(v)


out = {'pillow_version': np.array(PIL.__version__)}
names, values = [], []
for key, a in images.items():
    out[f'image_{key}'] = a
for k, (name, v) in enumerate(CALLS):
    names.append(name); values.append(np.nan if v is None else float(v))
    for key, a in images.items():
        out[f'out_{k}_{key}'] = arr(call(name, pil(a), v), a.shape[2])
out['call_names'] = np.array(names)
out['call_values'] = np.array(values, dtype=np.float64)
path = Path(__file__).resolve().parent / 'augment_pil.npz'
np.savez_compressed(path, **out)
print(f'wrote {path} ({path.stat().st_size} bytes), {len(CALLS)} calls x {len(images)} images, Pillow {PIL.__version__}')
