"""oracle/inputs_ref.py -- TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's depth
back-projection `dpt_2_pcld` (ffb6d/datasets/linemod/linemod_dataset.py:188-199) and the NaN/Inf
clean-up that follows it (:258-259), with numpy's own dtype promotion (float32 depth, int64 pixel
maps, float64 intrinsics -> float64 result); pinned bit-exactly to the reference's own method
(tests/test_oracle_cpu.py::test_inputs_ref_equals_the_reference_dpt_2_pcld).

`depth_normal`: restatement of the published algorithm behind the third-party `normalSpeed.depth_normal` the
reference calls at linemod_dataset.py:252-254 (OpenCV LINE-MOD bilateral least-squares normals).  PARITY
UNPINNED: normalSpeed is neither vendored in /root/reference nor installed in this image, so neither this
restatement nor the HIP kernel can be compared with its output; they are compared with each other."""
import numpy as np


def dpt_2_pcld(dpt, cam_scale, K):
    h, w = dpt.shape[:2]
    xmap = np.array([[j for i in range(w)] for j in range(h)])      # row index  (linemod_dataset.py:32)
    ymap = np.array([[i for i in range(w)] for j in range(h)])      # col index  (linemod_dataset.py:33)
    if len(dpt.shape) > 2:
        dpt = dpt[:, :, 0]
    dpt = dpt.astype(np.float32) / cam_scale
    msk = (dpt > 1e-8).astype(np.float32)
    row = (ymap - K[0][2]) * dpt / K[0][0]
    col = (xmap - K[1][2]) * dpt / K[1][1]
    dpt_3d = np.concatenate((row[..., None], col[..., None], dpt[..., None]), axis=2)
    dpt_3d = dpt_3d * msk[:, :, None]
    dpt_3d[np.isnan(dpt_3d)] = 0.0
    dpt_3d[np.isinf(dpt_3d)] = 0.0
    return dpt_3d


def depth_normal(depth_mm, fx, fy, k_size=5, distance_threshold=2000, difference_threshold=20, point_into_surface=False):
    """depth_mm [H,W] (cast to uint16 like the call sites do) -> float32 [H,W,3] unit normals, zeros where undefined."""
    assert not point_into_surface
    d = np.asarray(depth_mm).astype(np.uint16).astype(np.int64)
    H, W = d.shape
    r = int(k_size)
    out = np.zeros((H, W, 3), np.float32)
    ys, xs = np.arange(r, H - r - 1), np.arange(r, W - r - 1)
    c = d[np.ix_(ys, xs)]
    a00 = np.zeros_like(c); a01 = np.zeros_like(c); a11 = np.zeros_like(c); b0 = np.zeros_like(c); b1 = np.zeros_like(c)
    for j in (-r, 0, r):
        for i in (-r, 0, r):
            if i == 0 and j == 0:
                continue
            delta = d[np.ix_(ys + j, xs + i)] - c
            f = (np.abs(delta) < difference_threshold).astype(np.int64)
            a00 += f * i * i; a01 += f * i * j; a11 += f * j * j
            b0 += f * i * delta; b1 += f * j * delta
    det = a00 * a11 - a01 * a01
    ddx = a11 * b0 - a01 * b1
    ddy = -a01 * b0 + a00 * b1
    n = np.stack([(float(fx) * ddx.astype(np.float64)).astype(np.float32),
                  (float(fy) * ddy.astype(np.float64)).astype(np.float32), (-det * c).astype(np.float32)], axis=-1)
    length = np.sqrt((n[..., 0] * n[..., 0] + n[..., 1] * n[..., 1]) + n[..., 2] * n[..., 2]).astype(np.float32)
    ok = (c < distance_threshold) & (length > 0)
    inv = np.where(ok, np.float32(1.0) / np.where(ok, length, np.float32(1.0)), np.float32(0.0)).astype(np.float32)
    out[np.ix_(ys, xs)] = n * inv[..., None]
    return out


def sample_order(depth, seed, min_depth=1e-6):
    """The permutation ffb6d_sample_points_f32 documents (csrc/inputs.hip), restated: every pixel of frame b gets the 32-bit key
    mix32(mix32(pixel ^ seed_lo) + 0x9e3779b9 * (b + 1) + seed_hi) (murmur3's finaliser; 0xffffffff is reserved for invalid pixels), the
    frame's pixels are sorted by key, STABLY.  depth [B,H,W] float32 -> (order int64 [B,H*W], n_valid [B]): the first n_valid[b] entries
    of order[b] are the frame's valid pixels in the sampled order (the sample is its prefix).  Test infrastructure."""
    d = np.asarray(depth, np.float32)
    B = d.shape[0]
    HW = d[0].size
    lo, hi = np.uint32(seed & 0xffffffff), np.uint32((seed >> 32) & 0xffffffff)

    def mix32(x):
        x = x.astype(np.uint32)
        x ^= x >> np.uint32(16); x *= np.uint32(0x85ebca6b); x ^= x >> np.uint32(13); x *= np.uint32(0xc2b2ae35); x ^= x >> np.uint32(16)
        return x

    order, n_valid = np.empty((B, HW), np.int64), np.empty(B, np.int64)
    pix = np.arange(HW, dtype=np.uint32)
    with np.errstate(over="ignore"):
        for b in range(B):
            valid = d[b].reshape(-1) > np.float32(min_depth)            # NaN compares false
            h = mix32(mix32(pix ^ lo) + np.uint32((0x9e3779b9 * (b + 1)) & 0xffffffff) + hi)
            key = np.where(valid, np.where(h == np.uint32(0xffffffff), np.uint32(0xfffffffe), h), np.uint32(0xffffffff))
            order[b] = np.argsort(key, kind="stable")
            n_valid[b] = int(valid.sum())
    return order, n_valid

