"""Host-side pieces that need no GPU: Preprocessor contract, resize_by_factor (the reference's own
test cases, tests/test_transforms.py:6-28), take()."""
import numpy as np
import pytest

from empanada_napari_amd.preprocess import Preprocessor, normalize, resize_by_factor


@pytest.mark.parametrize('image,scale,expected', [
    ([[10., 20.], [30., 40.]], 1, [[10., 20.], [30., 40.]]),
    ([[10., 20.], [30., 40.]], 0.5, [[10., 12.5, 17.5, 20.], [15., 17.5, 22.5, 25.], [25., 27.5, 32.5, 35.], [30., 32.5, 37.5, 40.]]),
    ([[10., 20.], [30., 40.]], 2, [[25.]])])
def test_resize_by_factor_reference_cases(image, scale, expected):
    out = resize_by_factor(np.array(image), scale)
    assert out.shape == np.array(expected).shape
    assert np.array_equal(out, expected)


def test_resize_shapes_and_dtype():
    img = (np.arange(35 * 51) % 251).astype(np.uint8).reshape(35, 51)
    out = resize_by_factor(img, 2)
    assert out.shape == (18, 26) and out.dtype == np.uint8
    assert out.shape[0] * 2 >= 35 and out.shape[1] * 2 >= 51      # volume_dataset.py:48-49


def test_preprocessor_contract():
    p = Preprocessor(mean=0.57571, std=0.12765)
    img8 = np.array([[0, 255], [128, 64]], dtype=np.uint8)
    t = p(img8)['image']
    assert tuple(t.shape) == (1, 2, 2) and t.dtype.is_floating_point
    np.testing.assert_allclose(t[0].numpy(), (img8.astype(np.float32) - 0.57571 * 255) / (0.12765 * 255), rtol=1e-6)
    img16 = (img8.astype(np.uint16) * 257)
    np.testing.assert_allclose(p(img16)['image'][0].numpy(), t[0].numpy(), rtol=1e-5, atol=1e-5)   # iinfo(dtype).max scaling (Q14)
    with pytest.raises(Exception, match='cannot be float'):
        p(img8.astype(np.float32))


def test_take():
    from empanada_napari_amd.inference import take
    v = np.arange(24).reshape(2, 3, 4)
    assert np.array_equal(take(v, 1, 0), v[1]) and np.array_equal(take(v, 2, 1), v[:, 2]) and np.array_equal(take(v, 3, 2), v[:, :, 3])
