"""oracle/holefill_ref.py -- TEST INFRASTRUCTURE ONLY.  CPU restatement (numpy + scipy.ndimage) of the YCB depth hole
filling: Basic_Utils.fill_missing (ffb6d/utils/basic_utils.py:467-487) -> fill_in_multiscale(extrapolate=False,
blur_type='bilateral', max_depth=3.0) (ffb6d/utils/ip_basic/ip_basic/depth_map_utils_ycb.py:290-445).

PARITY UNPINNED: the reference function is a chain of OpenCV calls and cv2 is not installed in this image, so the
reference cannot be run here and has no golden vectors for this path.  Every cv2 operator is restated from its documented
semantics (footprint max/min with out-of-image pixels ignored; 5x5 median with replicated border; bilateral filter of
radius 2 over the 13 taps with r <= 2, BORDER_REFLECT_101).  OpenCV evaluates the bilateral colour weight through a
4096-bin linearly interpolated table; here it is the exact exponential (difference of order 1e-7 relative)."""
import numpy as np
from scipy import ndimage

EPS = 0.01
CROSS_3 = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)                                         # :19-24
CROSS_5 = np.array([[0, 0, 1, 0, 0]] * 2 + [[1] * 5] + [[0, 0, 1, 0, 0]] * 2, bool)                 # :33-40
CROSS_7 = np.array([[0, 0, 0, 1, 0, 0, 0]] * 3 + [[1] * 7] + [[0, 0, 0, 1, 0, 0, 0]] * 3, bool)      # :53-62


def dilate(img, footprint):
    return ndimage.grey_dilation(img, footprint=footprint, mode='constant', cval=-np.inf).astype(np.float32)


def erode(img, footprint):
    return ndimage.grey_erosion(img, footprint=footprint, mode='constant', cval=np.inf).astype(np.float32)


def full(k):
    return np.ones((k, k), bool)


def median5(img):
    return ndimage.median_filter(img, size=5, mode='nearest').astype(np.float32)


def bilateral5(img, sigma_color, sigma_space):
    H, W = img.shape
    pad = np.pad(img, 2, mode='reflect')                           # BORDER_REFLECT_101
    cc, cs = np.float32(-0.5 / (sigma_color * sigma_color)), np.float32(-0.5 / (sigma_space * sigma_space))
    num = np.zeros_like(img)
    den = np.zeros_like(img)
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            if dy * dy + dx * dx > 4:
                continue
            v = pad[2 + dy:2 + dy + H, 2 + dx:2 + dx + W]
            w = (np.exp(np.float32(dy * dy + dx * dx) * cs) * np.exp((v - img) * (v - img) * cc)).astype(np.float32)
            num = num + w * v
            den = den + w
    return (num / den).astype(np.float32)


def bilateral5_opencv(img, sigma_color, sigma_space):
    """cv2.bilateralFilter(img, 5, sigma_color, sigma_space) for one float32 channel as OpenCV 4's bilateralFilter_32f computes it
    (modules/imgproc/src/bilateral_filter.dispatch.cpp / .simd.hpp): the colour weight comes from a table of
    kExpNumBinsPerChannel = 4096 bins over the image's value range, linearly interpolated; spatial weights exp(r^2 * coeff) over the
    13 taps with r <= 2 in row-major order; BORDER_REFLECT_101; a constant image is returned unchanged.  Restated from the published
    source, NOT executed against cv2 (absent here): `bilateral5` above is the same filter with the exact exponential, and
    tests/test_f4_pin_cpu.py bounds the difference between the two forms."""
    img = np.asarray(img, np.float32)
    H, W = img.shape
    lo, hi = float(img.min()), float(img.max())
    if abs(lo - hi) < np.finfo(np.float32).eps:
        return img.copy()
    gcc, gsc = -0.5 / (float(sigma_color) * float(sigma_color)), -0.5 / (float(sigma_space) * float(sigma_space))
    nbins = 1 << 12
    scale_index = np.float32(nbins / np.float32(hi - lo))
    lut = np.zeros(nbins + 2, np.float32)
    last = 1.0
    for i in range(nbins + 2):
        if last > 0.0:
            val = i / float(scale_index)
            lut[i] = np.float32(np.exp(val * val * gcc))
            last = float(lut[i])
    pad = np.pad(img, 2, mode='reflect')
    num = np.zeros_like(img)
    den = np.zeros_like(img)
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            r = np.sqrt(float(dy * dy + dx * dx))
            if r > 2:
                continue
            sw = np.float32(np.exp(r * r * gsc))
            v = pad[2 + dy:2 + dy + H, 2 + dx:2 + dx + W]
            alpha = (np.abs(v - img) * scale_index).astype(np.float32)
            idx = np.floor(alpha).astype(np.int64)
            alpha = (alpha - idx.astype(np.float32)).astype(np.float32)
            w = (sw * (lut[idx] + alpha * (lut[idx + 1] - lut[idx]))).astype(np.float32)
            num = (num + v * w).astype(np.float32)
            den = (den + w).astype(np.float32)
    return (num / den).astype(np.float32)


def top_mask(img):
    """:366-369 / :384-390: True at and below the first valid pixel of each column (all True when the column has none)."""
    top = np.argmax(img > EPS, axis=0)
    return np.arange(img.shape[0])[:, None] >= top[None, :]


def fill_in_multiscale(depth_map, max_depth=3.0, taps=None):
    depths_in = np.float32(depth_map)                                                   # :310
    far = depths_in > 2.0                                                               # :313-316
    med = (depths_in > 1.0) & (depths_in <= 2.0)
    near = (depths_in > EPS) & (depths_in <= 1.0)
    s1 = depths_in.copy()                                                               # :319-321
    valid = s1 > EPS
    s1[valid] = np.float32(max_depth) - s1[valid]
    d_far = dilate(s1 * far, CROSS_3)                                                   # :324-333
    d_med = dilate(s1 * med, CROSS_5)
    d_near = dilate(s1 * near, CROSS_7)
    s2 = s1.copy()                                                                      # :341-346
    for d in (d_far, d_med, d_near):
        m = d > EPS
        s2[m] = d[m]
    s3 = erode(dilate(s2, full(5)), full(5))                                            # :349-351 MORPH_CLOSE
    s4 = s3.copy()                                                                      # :354-358
    blurred = median5(s3)
    m = s3 > EPS
    s4[m] = blurred[m]
    tm = top_mask(s4)                                                                   # :361-373
    empty = ~(s4 > EPS) & tm
    s5 = s4.copy()
    s5[empty] = dilate(s4, full(9))[empty]
    tm = top_mask(s5)                                                                   # :376-391 (extrapolate=False)
    s7 = s5.copy()
    for _ in range(6):                                                                  # :394-399
        empty = (s7 < EPS) & tm
        s7[empty] = dilate(s7, full(5))[empty]
    blurred = median5(s7)                                                               # :402-405
    valid = (s7 > EPS) & tm
    s7[valid] = blurred[valid]
    blurred = bilateral5(s7, 0.5, 2.0)                                                  # :413-416 reuses `valid`
    s7[valid] = blurred[valid]
    s8 = s7.copy()                                                                      # :419-422
    m = s8 > EPS
    s8[m] = np.float32(max_depth) - s8[m]
    if taps is not None:
        taps.update(s1=s1, s2=s2, s3=s3, s4=s4, s5=s5)
    return s8


def fill_missing(dpt, cam_scale, scale_2_80m=1, max_depth=3.0):
    """basic_utils.py:467-487 with fill_type='multiscale', extrapolate=False."""
    d = np.asarray(dpt) / cam_scale * scale_2_80m
    out = fill_in_multiscale(d, max_depth)
    return out / scale_2_80m * cam_scale
