"""GPU: the fused input pipeline through the C ABI, bit-exact against oracle/ref_input.py (itself pinned against
the installed Pillow in tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import ref_input

pytestmark = pytest.mark.gpu


def _images(seed, sizes):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]


@pytest.mark.parametrize("img_type", ["CheXpert-v1.0-small", "CheXpert_Enh"])
def test_training_chain_bit_exact(img_type):
    from mfvit.input_pipeline import NORMALIZE, GpuTransform
    sizes = [(320, 390), (390, 320), (256, 256), (1024, 1024), (100, 80), (224, 300), (257, 511), (320, 390)]
    imgs = _images(11, sizes)
    params = [(False, 0.0, 0, 0), (True, 3.7, 5, 31), (False, -9.99, 32, 0), (True, 10.0, 16, 16), (False, 45.0, 0, 32), (True, 90.0, 7, 9),
              (False, 180.0, 1, 2), (True, 270.0, 32, 32)]                       # incl. Image.rotate's transpose fast paths
    tf = GpuTransform(img_type, img_size=256, crop=224, rotate=10, training=True)
    got = tf(imgs, params).cpu().numpy()
    mean, std = NORMALIZE[img_type]
    for k, (im, (flip, ang, ci, cj)) in enumerate(zip(imgs, params)):
        want = ref_input.transform(im, 256, flip, ang, (ci, cj), 224, mean, std)
        assert got[k].shape == want.shape
        assert np.array_equal(got[k], want), (k, np.abs(got[k] - want).max())   # bit-exact, float32 output included


def test_eval_chain_and_sampled_params():
    from mfvit.input_pipeline import NORMALIZE, GpuTransform
    imgs = _images(12, [(300, 300), (512, 400), (256, 256)])
    ev = GpuTransform("data", img_size=256, crop=224, rotate=10, training=False)
    ps = ev.sample_params(3)
    assert ps == [(False, 0.0, 16, 16)] * 3                                          # CenterCrop offsets round((256-224)/2)
    got = ev(imgs).cpu().numpy()
    mean, std = NORMALIZE["data"]
    for k, im in enumerate(imgs):
        assert np.array_equal(got[k], ref_input.transform(im, 256, False, 0.0, (16, 16), 224, mean, std))
    tr = GpuTransform("data", img_size=256, crop=224, rotate=10, training=True)
    g = torch.Generator().manual_seed(5)
    ps = tr.sample_params(64, g)
    assert all(-10.0 <= a <= 10.0 and 0 <= i <= 32 and 0 <= j <= 32 for _, a, i, j in ps) and {f for f, *_ in ps} == {True, False}
    g2 = torch.Generator().manual_seed(5)
    got = tr(imgs, generator=g2).cpu().numpy()                                       # same generator state -> the first 3 draws
    for k, im in enumerate(imgs):
        f, a, i, j = ps[k]
        assert np.array_equal(got[k], ref_input.transform(im, 256, f, a, (i, j), 224, mean, std))
    # no crop (args.crop == 0) keeps the resized frame; img_size 384 (configs[4])
    big = GpuTransform("CheXpert-v1.0-small", img_size=384, crop=0, rotate=10, training=True)
    got = big(imgs[:1], [(True, -4.25, 0, 0)]).cpu().numpy()
    m, s = NORMALIZE["CheXpert-v1.0-small"]
    assert np.array_equal(got[0], ref_input.transform(imgs[0], 384, True, -4.25, (0, 0), 0, m, s))


def test_mocov3_chain_random_resized_crop():
    """get_transform_type_mocov3 (image_transform.py:86-124): RandomResizedCrop box -> resize -> flip -> rotate, and its eval chain."""
    from mfvit.input_pipeline import NORMALIZE, GpuTransform
    imgs = _images(13, [(320, 390), (390, 320), (512, 512), (90, 400)])
    tf = GpuTransform("data", img_size=224, rotate=10, training=True, mocov3=True, crop_min=0.08)
    g = torch.Generator().manual_seed(9)
    ps = tf.sample_params(4, g, [im.shape[:2] for im in imgs])
    for (f, a, ci, cj, (bi, bj, h, w)), im in zip(ps, imgs):
        assert ci == cj == 0 and 0 <= bi and 0 <= bj and bi + h <= im.shape[0] and bj + w <= im.shape[1]
        assert 0.08 * im.shape[0] * im.shape[1] * 0.9 <= h * w <= im.shape[0] * im.shape[1]
    got = tf(imgs, generator=torch.Generator().manual_seed(9)).cpu().numpy()
    mean, std = NORMALIZE["data"]
    assert got.shape == (4, 3, 224, 224)
    for k, (im, (f, a, _, _, box)) in enumerate(zip(imgs, ps)):
        assert np.array_equal(got[k], ref_input.transform_mocov3(im, box, 224, f, a, mean, std)), k
    # explicit windows incl. one that touches the image border and a 1:1 window (no resample on one axis)
    params = [(True, 5.5, 0, 0, (0, 0, 320, 390)), (False, -3.0, 0, 0, (166, 96, 224, 224)), (True, 0.0, 0, 0, (500, 500, 12, 12)),
              (False, 9.0, 0, 0, (0, 100, 90, 300))]
    got = tf(imgs, params).cpu().numpy()
    for k, (im, (f, a, _, _, box)) in enumerate(zip(imgs, params)):
        assert np.array_equal(got[k], ref_input.transform_mocov3(im, box, 224, f, a, mean, std)), k
    ev = GpuTransform("data", img_size=224, crop=224, training=False, mocov3=True)      # Resize((256, 256)) + CenterCrop(224)
    got = ev(imgs).cpu().numpy()
    for k, im in enumerate(imgs):
        assert np.array_equal(got[k], ref_input.transform(im, 256, False, 0.0, (16, 16), 224, mean, std))
