"""Known-answer tests of the object-mask preprocessing of TSDF.run (gs2mesh_utils/tsdf_utils.py:69-77):
cv2.morphologyEx(mask, MORPH_CLOSE, ones(10,10)) followed by cv2.erode(., ones(10,10)).  cv2 is not in the image; the
expected arrays are worked out BY HAND from OpenCV's documented definition -- erode / dilate take the min / max over
src(x + dx, y + dy) for dx, dy in [-anchor, k - 1 - anchor] with the default anchor k // 2 (= 5 for k = 10, so the window
reaches 5 pixels up/left and 4 down/right), the image border not constraining an erosion (+inf) nor feeding a dilation
(-inf).  With an even kernel the closing therefore moves every shape one pixel towards +x, +y."""
import numpy as np

from gs2mesh_amd.tsdf_utils import _morph, preprocess_object_mask


def rect(H, W, y0, y1, x0, x1):
    m = np.zeros((H, W), bool)
    m[y0:y1 + 1, x0:x1 + 1] = True
    return m


def test_dilate_and_erode_of_one_pixel_with_the_even_kernel():
    m = rect(40, 40, 20, 20, 17, 17)
    d = _morph(m, 10, erode=False)
    # the pixel (x = 17) is seen from every x with x + dx = 17, dx in [-5, 4]  ->  x in [13, 22]; same in y: [16, 25]
    assert np.array_equal(d, rect(40, 40, 16, 25, 13, 22))
    # eroding the 10 x 10 block back: x survives iff [x - 5, x + 4] lies in [13, 22]  ->  x = 18; y = 21
    assert np.array_equal(_morph(d, 10, erode=True), rect(40, 40, 21, 21, 18, 18))
    # odd kernel (3): symmetric, closing a pixel gives the pixel back
    assert np.array_equal(_morph(_morph(m, 3, False), 3, True), m)


def test_close_then_erode_of_a_rectangle():
    H, W = 60, 80
    m = rect(H, W, 10, 40, 20, 60)
    close = _morph(_morph(m, 10, False), 10, True)
    # dilation [20 - 4, 60 + 5] = [16, 65]; erosion keeps x with [x - 5, x + 4] inside -> [21, 61]: the rectangle moved by +1
    assert np.array_equal(close, rect(H, W, 11, 41, 21, 61))
    out = preprocess_object_mask(m, erode=True)
    # a further erosion: [21 + 5, 61 - 4] = [26, 57]; rows [11 + 5, 41 - 4] = [16, 37]
    assert np.array_equal(out, rect(H, W, 16, 37, 26, 57))
    assert out.dtype == bool
    # erode=False (TSDF_erode_mask off) leaves the mask alone, invert flips it first
    assert np.array_equal(preprocess_object_mask(m, erode=False), m)
    assert np.array_equal(preprocess_object_mask(m, invert=True, erode=False), ~m)


def test_closing_fills_gaps_of_up_to_nine_pixels():
    H, W = 30, 100
    for gap, filled in ((9, True), (10, False)):
        m = rect(H, W, 5, 24, 10, 30) | rect(H, W, 5, 24, 31 + gap, 80)
        close = _morph(_morph(m, 10, False), 10, True)
        row = close[15]
        # the two dilated spans [6, 35] and [27 + gap, 85] touch iff 35 >= 27 + gap - 1, i.e. gap <= 9
        assert bool(row[31:31 + gap + 1].all()) == filled, gap
        assert row[11] and row[81] and not row[10] and not row[82]      # outer ends moved by +1 like any closing


def test_the_image_border_does_not_erode():
    H, W = 30, 30
    m = rect(H, W, 0, 29, 0, 12)              # touches the left, top and bottom borders
    out = preprocess_object_mask(m)
    # closing: dilation [0, 17] (clipped), erosion: windows reaching outside the image only test their in-image part:
    # x in [0, 13]; second erosion: x in [0, 13 - 4] = [0, 9]; all rows survive
    assert np.array_equal(out, rect(H, W, 0, 29, 0, 9))


def _brute(m, k, erode):
    """the documented definition, pixel by pixel: min / max over src(y + dy, x + dx), dy, dx in [-k // 2, k - 1 - k // 2],
    positions outside the image skipped (= the +inf / -inf default border)"""
    H, W = m.shape
    a = k // 2
    out = np.zeros_like(m)
    for y in range(H):
        for x in range(W):
            vals = [m[y + dy, x + dx] for dy in range(-a, k - a) for dx in range(-a, k - a)
                    if 0 <= y + dy < H and 0 <= x + dx < W]
            out[y, x] = all(vals) if erode else any(vals)
    return out


def test_random_masks_against_the_definition_and_against_scipy_filters():
    """Two independent restatements on random masks, even and odd kernels, kernels larger than the image: the brute-force
    definition above, and scipy.ndimage's minimum / maximum filters (origin 0: for a size-k window scipy takes
    [i - k // 2, i + k - 1 - k // 2], the same extent as OpenCV's default anchor; cval = the value that never wins)."""
    from scipy import ndimage
    rng = np.random.default_rng(7)
    for H, W, k, p in [(23, 31, 10, 0.5), (17, 9, 3, 0.7), (12, 40, 10, 0.93), (30, 30, 7, 0.1), (8, 8, 10, 0.8), (25, 25, 2, 0.5)]:
        m = rng.random((H, W)) < p
        for erode in (False, True):
            got = _morph(m, k, erode)
            assert got.shape == m.shape and got.dtype == bool
            assert np.array_equal(got, _brute(m, k, erode)), (H, W, k, erode)
            f = ndimage.minimum_filter if erode else ndimage.maximum_filter
            ref = f(m.astype(np.uint8), size=k, mode="constant", cval=1 if erode else 0).astype(bool)
            assert np.array_equal(got, ref), (H, W, k, erode)
        # the composed preprocessing = the composition of the restated steps
        want = _brute(_brute(_brute(m, k, False), k, True), k, True)
        assert np.array_equal(preprocess_object_mask(m, closing_kernel_size=k, erosion_kernel_size=k), want)
