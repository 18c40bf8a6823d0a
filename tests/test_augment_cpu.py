""" CPU checks of the augmentation oracle (`oracle/augment_ref.py`) against hand-computed known answers of the PIL operations it restates
(`ImageOps.posterize / solarize`, `Image.transform(AFFINE)`, `ImageEnhance.Brightness`; reference `src/deepcv/meta/data/augmentation.py:73-124`) and of the
host-side recipe sampler (`deepcv_amd/meta/data/augmentation.py`). """
import numpy as np

from deepcv_amd.meta.data.augmentation import AUG_OPS, AUG_SAMPLE_DTYPE, sample_recipes
from oracle import augment_ref as R


def test_pointwise_ops_known_answers():
    img = np.array([[[0], [37], [128], [200], [255]]], dtype=np.uint8)
    assert R.apply_op(img, R.POSTERIZE, 2).ravel().tolist() == [0, 0, 128, 192, 192]            # keep the 2 high bits
    assert R.apply_op(img, R.POSTERIZE, 8).ravel().tolist() == [0, 37, 128, 200, 255]
    assert R.apply_op(img, R.SOLARIZE, 128).ravel().tolist() == [0, 37, 127, 55, 0]              # invert pixels >= threshold
    assert R.apply_op(img, R.BRIGHTNESS, 128).ravel().tolist() == [0, 18, 64, 100, 127]          # x * 0.5, truncated
    assert R.apply_op(img, R.BRIGHTNESS, 461).ravel().tolist() == [0, 66, 230, 255, 255]         # x * 1.8008, clipped


def test_translate_zero_fill():
    img = np.arange(12, dtype=np.uint8).reshape(3, 4, 1) + 1
    tx = R.apply_op(img, R.TRANSLATE_X, 1)[:, :, 0]        # out(x, y) = in(x + 1, y)
    assert tx.tolist() == [[2, 3, 4, 0], [6, 7, 8, 0], [10, 11, 12, 0]]
    ty = R.apply_op(img, R.TRANSLATE_Y, -1)[:, :, 0]       # out(x, y) = in(x, y - 1)
    assert ty.tolist() == [[0, 0, 0, 0], [1, 2, 3, 4], [5, 6, 7, 8]]
    # a pixel shifted out and back in stays zero (sequential semantics)
    back = R.apply_op(R.apply_op(img, R.TRANSLATE_X, 3), R.TRANSLATE_X, -3)[:, :, 0]
    assert back.tolist() == [[0, 0, 0, 4], [0, 0, 0, 8], [0, 0, 0, 12]]


def test_recipe_sampler_ranges_and_reproducibility():
    a = sample_recipes(64, 32, 32, np.random.RandomState(5), severity=0.5)
    b = sample_recipes(64, 32, 32, np.random.RandomState(5), severity=0.5)
    assert a.dtype == AUG_SAMPLE_DTYPE and a.tobytes() == b.tobytes()
    used = a['weight'] > 0
    assert np.allclose(a['weight'].sum(axis=1), 1.0, atol=1e-6) and ((a['mix'] >= 0) & (a['mix'] <= 1)).all()
    assert ((a['op'] != 0).any(axis=2) == used).all()                                   # every weighted chain has at least one op, others none
    for name, code in AUG_OPS.items():
        p = a['param'][a['op'] == code]
        assert p.size > 0
        if name == 'posterize':
            assert ((p >= 2) & (p <= 8)).all()
        elif name == 'solarize':
            assert ((p >= 50) & (p <= 220)).all()
        elif name.startswith('translate'):
            assert (np.abs(p) <= 0.5 * 32 / 3 + 1).all()
        else:
            assert ((p >= int(0.3 * 256) - 1) & (p <= int(1.8 * 256) + 1)).all()


def test_identity_recipe_is_plain_normalisation():
    img = np.random.RandomState(0).randint(0, 256, (2, 5, 6, 3)).astype(np.uint8)
    rec = np.zeros(2, dtype=AUG_SAMPLE_DTYPE)                     # mix = 0: original image
    out = R.augment_normalize(img, rec, [0.5, 0.4, 0.3], [0.2, 0.25, 0.3])
    ref = (np.transpose(img, (0, 3, 1, 2)).astype(np.float64) / 255.0 - np.array([0.5, 0.4, 0.3]).reshape(1, 3, 1, 1)) / np.array([0.2, 0.25, 0.3]).reshape(1, 3, 1, 1)
    assert np.abs(out - ref).max() < 1e-5
