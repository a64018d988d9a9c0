"""SURVEY section 8f rank 4 (surface normals, YCB depth hole filling): how far the two restatements can be pinned in an image that
has neither `normalSpeed` nor `cv2`.

  * oracle/holefill_ref.py builds fill_in_multiscale (depth_map_utils_ycb.py:290-445) from scipy.ndimage primitives.  Here every
    primitive is checked against an independent pixel-by-pixel statement of the OpenCV operator it stands for (cv2.dilate / erode
    with a structuring element and the default border = out-of-image pixels never win; cv2.medianBlur(5) = replicated border;
    cv2.bilateralFilter(5, sc, ss) = 13 taps, BORDER_REFLECT_101), written with loops and no scipy; the whole chain is then
    re-assembled from the loop primitives and must equal the oracle stage by stage.
  * OpenCV evaluates the bilateral colour weight through an interpolated 4096-bin table; oracle.holefill_ref.bilateral5_opencv
    restates that path and the difference to the exact exponential is bounded (so a kernel within 1e-5 of one is within 1e-5 + that
    bound of the other).
  * oracle/inputs_ref.depth_normal (LINE-MOD normals behind normalSpeed.depth_normal) against ANALYTIC surfaces: planes and
    spheres at several depths and both cameras' intrinsics.
What stays unpinned after this file: that OpenCV's / normalSpeed's compiled code computes what their published sources and papers
say (no binary here to run), i.e. the table-interpolated bilateral weight to the last ulp and normalSpeed's integer rounding.
The inputs and outputs of the cases below are committed as tests/golden/f4_vectors.npz (tests/golden/make_golden_f4.py) and the
GPU kernels are held to them in tests/test_inputs_gpu.py."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import holefill_ref as HR
from oracle import inputs_ref

NEG, POS = -np.inf, np.inf


def loop_morph(img, footprint, dilate):
    """cv2.dilate / cv2.erode with structuring element `footprint` (anchor = centre), default border: pixels outside the image
    take the value that can never win (morphologyDefaultBorderValue)"""
    H, W = img.shape
    fh, fw = footprint.shape
    out = np.empty_like(img)
    for y in range(H):
        for x in range(W):
            best = NEG if dilate else POS
            for j in range(fh):
                for i in range(fw):
                    if not footprint[j, i]:
                        continue
                    yy, xx = y + j - fh // 2, x + i - fw // 2
                    if 0 <= yy < H and 0 <= xx < W:
                        best = max(best, img[yy, xx]) if dilate else min(best, img[yy, xx])
            out[y, x] = best
    return out


def loop_median5(img):
    """cv2.medianBlur(img, 5): BORDER_REPLICATE"""
    H, W = img.shape
    out = np.empty_like(img)
    for y in range(H):
        for x in range(W):
            win = [img[min(max(y + j, 0), H - 1), min(max(x + i, 0), W - 1)] for j in range(-2, 3) for i in range(-2, 3)]
            out[y, x] = sorted(win)[12]
    return out


def reflect101(i, n):
    return -i if i < 0 else (2 * (n - 1) - i if i >= n else i)


def loop_bilateral5(img, sc, ss):
    """cv2.bilateralFilter(img, 5, sc, ss) with the exact exponential, float64 accumulation"""
    H, W = img.shape
    out = np.empty_like(img)
    for y in range(H):
        for x in range(W):
            num = den = 0.0
            for j in range(-2, 3):
                for i in range(-2, 3):
                    if i * i + j * j > 4:
                        continue
                    v = float(img[reflect101(y + j, H), reflect101(x + i, W)])
                    w = np.exp(-0.5 * (i * i + j * j) / (ss * ss)) * np.exp(-0.5 * (v - float(img[y, x])) ** 2 / (sc * sc))
                    num += w * v
                    den += w
            out[y, x] = num / den
    return out


def holes(seed, H=40, W=52):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    d = (0.4 + 0.045 * xx + 0.01 * yy + 0.002 * rng.standard_normal((H, W))).astype(np.float32)       # 0.4 .. 3.1 m: all three bands
    d[rng.rand(H, W) < 0.35] = 0
    d[:4] = 0
    d[10:18, 20:31] = 0
    return d


@pytest.mark.parametrize("seed", [0, 1])
def test_scipy_primitives_of_the_oracle_equal_the_opencv_operators_stated_pixel_by_pixel(seed):
    img = holes(seed)
    for fp in (HR.CROSS_3, HR.CROSS_5, HR.CROSS_7, HR.full(5), HR.full(9)):
        np.testing.assert_array_equal(HR.dilate(img, fp), loop_morph(img, fp, True))
        np.testing.assert_array_equal(HR.erode(img, fp), loop_morph(img, fp, False))
    np.testing.assert_array_equal(HR.median5(img), loop_median5(img))
    filled = HR.fill_in_multiscale(img)
    got, want = HR.bilateral5(filled, 0.5, 2.0), loop_bilateral5(filled, 0.5, 2.0)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_whole_chain_reassembled_from_the_loop_primitives_equals_the_oracle_stage_by_stage(seed):
    """fill_in_multiscale(extrapolate=False, blur_type='bilateral') written a second time, straight from
    depth_map_utils_ycb.py:310-422, on the loop primitives"""
    d = holes(seed)
    taps = {}
    want = HR.fill_in_multiscale(d, 3.0, taps)
    EPS = 0.01
    s1 = d.copy()
    v = s1 > EPS
    s1[v] = np.float32(3.0) - s1[v]                                                    # :319-321
    far, med, near = d > 2.0, (d > 1.0) & (d <= 2.0), (d > EPS) & (d <= 1.0)           # :313-316 (on the un-inverted depths)
    s2 = s1.copy()
    for band, fp in ((far, HR.CROSS_3), (med, HR.CROSS_5), (near, HR.CROSS_7)):        # :324-346
        dil = loop_morph((s1 * band).astype(np.float32), fp, True)
        m = dil > EPS
        s2[m] = dil[m]
    s3 = loop_morph(loop_morph(s2, HR.full(5), True), HR.full(5), False)               # :349-351
    s4 = s3.copy()
    b = loop_median5(s3)
    m = s3 > EPS
    s4[m] = b[m]                                                                       # :354-358

    def top_mask(a):
        tm = np.ones(a.shape, bool)
        for x in range(a.shape[1]):
            col = np.nonzero(a[:, x] > EPS)[0]
            if len(col):
                tm[:col[0], x] = False
        return tm

    tm = top_mask(s4)
    s5 = s4.copy()
    e = ~(s4 > EPS) & tm
    s5[e] = loop_morph(s4, HR.full(9), True)[e]                                        # :361-373
    for k, a in (("s1", s1), ("s2", s2), ("s3", s3), ("s4", s4), ("s5", s5)):
        np.testing.assert_array_equal(a, taps[k], err_msg=k)
    tm = top_mask(s5)
    s7 = s5.copy()
    for _ in range(6):                                                                 # :394-399
        e = (s7 < EPS) & tm
        s7[e] = loop_morph(s7, HR.full(5), True)[e]
    b = loop_median5(s7)
    valid = (s7 > EPS) & tm
    s7[valid] = b[valid]                                                               # :402-405
    b = loop_bilateral5(s7, 0.5, 2.0)
    s7[valid] = b[valid]                                                               # :413-416
    m = s7 > EPS
    s7[m] = np.float32(3.0) - s7[m]                                                    # :419-422
    assert np.abs(s7 - want).max() <= 5e-6


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_table_interpolated_bilateral_weight_is_within_1e6_of_the_exact_exponential(seed):
    """what separates the oracle (exact exp) from OpenCV's float path (4096-bin interpolated table): measured here, on filled maps"""
    filled = HR.fill_in_multiscale(holes(seed, 96, 128))
    exact, table = HR.bilateral5(filled, 0.5, 2.0), HR.bilateral5_opencv(filled, 0.5, 2.0)
    rel = float(np.abs(exact - table).max() / np.abs(exact).max())
    print("bilateral: exact vs table", rel)
    assert rel <= 1e-6
    const = np.full((9, 9), 1.25, np.float32)
    np.testing.assert_array_equal(HR.bilateral5_opencv(const, 0.5, 2.0), const)


CAMS = {"linemod": (572.4114, 573.57043, 325.2611, 242.04899), "ycb": (1066.778, 1067.487, 312.9869, 241.3109)}


def plane_depth(cam, nrm, d0, H=120, W=160):
    """depth image (mm) of the plane through (0, 0, d0) with unit normal nrm, seen by a pinhole camera"""
    fx, fy, cx, cy = CAMS[cam]
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    rx, ry = (xx - cx * W / 640) / (fx * W / 640), (yy - cy * H / 480) / (fy * H / 480)             # ray = (rx, ry, 1) * z
    z = d0 * nrm[2] / (nrm[0] * rx + nrm[1] * ry + nrm[2])
    return z


def sphere_depth(cam, centre, R, H=120, W=160):
    fx, fy, cx, cy = CAMS[cam]
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    ray = np.stack([(xx - cx * W / 640) / (fx * W / 640), (yy - cy * H / 480) / (fy * H / 480), np.ones_like(xx)], -1)
    a = (ray * ray).sum(-1)
    b = -2 * (ray * centre).sum(-1)
    c = float((centre * centre).sum() - R * R)
    disc = b * b - 4 * a * c
    t = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), 0.0)
    pts = ray * t[..., None]
    nrm = (pts - centre) / R
    return np.where(disc > 0, pts[..., 2], 0.0), nrm, disc > 0


def angle_deg(a, b):
    return np.degrees(np.arccos(np.clip((a * b).sum(-1), -1, 1)))


@pytest.mark.parametrize("cam", ["linemod", "ycb"])
@pytest.mark.parametrize("d0", [450.0, 900.0, 1900.0])
def test_depth_normal_recovers_the_normals_of_tilted_planes(cam, d0):
    """LINE-MOD normals of a perspective image of the plane n.X = c: with z = c / (n.ray) the published estimate
    normalize(fx dz/du, fy dz/dv, -z) is normalize(n_x, n_y, n.ray) -- the plane's normal on the optical axis, off it short of the
    true normal by the perspective term the method drops (a few degrees at the image edge).  The restatement must reproduce THAT
    closely (what is left is the 1 mm depth quantisation over a 5-pixel baseline) and stay within 5 degrees of the true normal."""
    fx, fy, cx, cy = CAMS[cam]
    H, W = 120, 160
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    rx, ry = (xx - cx * W / 640) / (fx * W / 640), (yy - cy * H / 480) / (fy * H / 480)
    for tilt in ((0.0, 0.0, -1.0), (0.3, 0.1, -1.0), (-0.2, 0.35, -1.0)):
        n = np.array(tilt) / np.linalg.norm(tilt)
        z = plane_depth(cam, n, d0)
        got = inputs_ref.depth_normal(z.astype(np.float32), fx * W / 640, fy * H / 480, 5, 2000, 20, False)
        near = z.astype(np.float32).astype(np.uint16)[6:-7, 6:-7] < 2000                 # at and beyond the distance threshold the normal is (0, 0, 0)
        inner = got[6:-7, 6:-7][near]
        assert near.mean() > 0.3 and (got[6:-7, 6:-7][~near] == 0).all()
        assert (np.abs(np.linalg.norm(inner, axis=-1) - 1) < 1e-5).all()
        pred = np.stack([np.full_like(rx, n[0]), np.full_like(rx, n[1]), n[0] * rx + n[1] * ry + n[2]], -1)
        pred /= np.linalg.norm(pred, axis=-1, keepdims=True)
        err = angle_deg(inner, pred[6:-7, 6:-7][near])
        quant = np.degrees(np.arctan(1.0 / 10.0 * (fx * W / 640) / d0)) + 0.2       # 1 mm over the 10-pixel baseline, as an angle
        assert err.max() < 1.5 * quant, (cam, d0, tilt, err.max(), quant)
        assert angle_deg(inner, np.broadcast_to(n, inner.shape)).max() < 5.0 + 1.5 * quant


@pytest.mark.parametrize("cam", ["linemod", "ycb"])
def test_depth_normal_on_a_sphere_follows_the_radial_direction(cam):
    fx, fy = CAMS[cam][:2]
    centre, R = np.array([20.0, -10.0, 900.0]), 260.0
    z, nrm, hit = sphere_depth(cam, centre, R)
    got = inputs_ref.depth_normal(z.astype(np.float32), fx * 160 / 640, fy * 120 / 480, 5, 2000, 20, False)
    cosv = -nrm[..., 2]                                                           # facing the camera
    core = hit & (cosv > 0.8)
    core[:6] = core[-7:] = False
    core[:, :6] = core[:, -7:] = False
    assert core.sum() > 500
    err = angle_deg(got[core], nrm[core])
    assert np.median(err) < 3.0 and err.max() < 12.0, (np.median(err), err.max())
    assert (got[~hit][np.linalg.norm(got[~hit], axis=-1) > 0].size) <= 3 * 4 * (hit.sum() ** 0.5) * 6      # only a rim around the silhouette


def test_committed_f4_vectors_are_what_the_oracle_produces():
    g = np.load(os.path.join(GOLDEN, "f4_vectors.npz"))
    for k in [k for k in g.files if k.endswith("/depth_mm")]:
        tag = k.split("/")[0]
        fx, fy = g[tag + "/fxfy"]
        np.testing.assert_array_equal(inputs_ref.depth_normal(g[k], float(fx), float(fy), 5, 2000, 20, False), g[tag + "/normals"])
    for k in [k for k in g.files if k.endswith("/depth_raw")]:
        tag = k.split("/")[0]
        np.testing.assert_array_equal(HR.fill_missing(g[k], float(g[tag + "/cam_scale"]), 1), g[tag + "/filled"])
