"""Seeded synthetic RGB-D frames and synthetic weights (numpy only, no GPU, no dataset).

There is no network or dataset on the build/bench machines, so every test, golden vector
and benchmark runs on frames generated here (SURVEY.md section 8d / BASELINE.md section 3):

  depth   z(y,x) = 1.0 + 0.3*sin(x/37)*cos(y/29) + 0.002*N(0,1) metres, 5 % of pixels
          zeroed (invalid depth); the noise keeps point-to-point distances tie-free
  xyz     back-projection with the LineMOD intrinsics exactly as `dpt_2_pcld`
          (ffb6d/datasets/linemod/linemod_dataset.py:188-199; float64 maths, cast to f32
          where the reference casts: knn.pyx:95-96, linemod_dataset.py:325)
  choose  the first N entries of a seeded permutation of the valid pixels (the reference
          samples N valid pixels and shuffles them once, linemod_dataset.py:264-282)
  rgb     uint8 uniform noise; normals: normalised N(0,1)^3
  cld_rgb_nrm = [xyz, rgb at the chosen pixels, normals at the chosen pixels]  [9,N]
                (linemod_dataset.py:284-289)

Seeds follow `1000*config + sample`.
"""
import zlib

import numpy as np

LINEMOD_K = np.array([[572.4114, 0.0, 325.2611],
                      [0.0, 573.57043, 242.04899],
                      [0.0, 0.0, 1.0]])  # ffb6d/common.py:144-146


def frame_seed(config, sample):
    return 1000 * int(config) + int(sample)


def make_frame(seed, n_points=12288, height=480, width=640, invalid_frac=0.05, K=LINEMOD_K):
    """One synthetic RGB-D frame as the dataset would hand it over (numpy, CPU).

    Returns dict: rgb u8 [3,H,W]; dpt_xyz f32 [3,H,W]; cld f32 [N,3];
    cld_rgb_nrm f32 [9,N]; choose i32 [1,N].
    """
    rng = np.random.RandomState(seed)
    ys, xs = np.mgrid[:height, :width]
    z = 1.0 + 0.3 * np.sin(xs / 37.0) * np.cos(ys / 29.0) + 0.002 * rng.standard_normal((height, width))
    dpt = z.astype(np.float32)
    dpt[rng.random_sample((height, width)) < invalid_frac] = 0.0
    # dpt_2_pcld: xmap = row index, ymap = column index (linemod_dataset.py:35-36)
    msk = (dpt > 1e-8).astype(np.float32)
    row = (xs - K[0][2]) * dpt / K[0][0]
    col = (ys - K[1][2]) * dpt / K[1][1]
    dpt_xyz = np.concatenate((row[..., None], col[..., None], dpt[..., None]), axis=2)
    dpt_xyz = (dpt_xyz * msk[:, :, None]).astype(np.float32)  # [H,W,3]

    valid = (dpt.reshape(-1) > 1e-6).nonzero()[0]
    if valid.size < n_points:
        raise ValueError(f"only {valid.size} valid pixels for n_points={n_points}")
    choose = valid[rng.permutation(valid.size)[:n_points]].astype(np.int32)

    rgb = rng.randint(0, 256, size=(height, width, 3)).astype(np.uint8)
    nrm = rng.standard_normal((height, width, 3))
    nrm = (nrm / np.linalg.norm(nrm, axis=2, keepdims=True)).astype(np.float32)

    cld = dpt_xyz.reshape(-1, 3)[choose, :]
    rgb_pt = rgb.reshape(-1, 3)[choose, :].astype(np.float32)
    nrm_pt = nrm.reshape(-1, 3)[choose, :]
    cld_rgb_nrm = np.concatenate((cld, rgb_pt, nrm_pt), axis=1).transpose(1, 0)
    return dict(
        rgb=np.ascontiguousarray(rgb.transpose(2, 0, 1)),
        dpt_xyz=np.ascontiguousarray(dpt_xyz.transpose(2, 0, 1)),
        cld=np.ascontiguousarray(cld, dtype=np.float32),
        cld_rgb_nrm=np.ascontiguousarray(cld_rgb_nrm, dtype=np.float32),
        choose=choose[None, :].copy(),
    )


def make_targets(seed, cld, n_classes=22, n_kps=8, n_objects=3, radius=0.12):
    """Synthetic training targets of one frame with the dataset's shapes and meaning (linemod_dataset.py:291-297,
    ycb_dataset.py:214-252): `n_objects` spheres of `radius` metres around seeded surface points, each with a class id in
    [1, n_classes) and n_kps keypoints; labels [N] int32 = class of the sphere a point falls in (0 = background; the nearest centre
    wins), kp_targ_ofst [N,n_kps,3] / ctr_targ_ofst [N,1,3] = keypoint / centre minus point for the points of an object, 0 elsewhere.
    Its own random stream (seed + 7919): make_frame's values and the golden fixtures built from them do not change."""
    rng = np.random.RandomState(seed + 7919)
    n = cld.shape[0]
    centres = cld[rng.choice(n, size=n_objects, replace=False)].astype(np.float64)
    classes = rng.choice(np.arange(1, max(n_classes, 2)), size=n_objects, replace=n_classes - 1 < n_objects)
    kps = centres[:, None, :] + 0.05 * rng.standard_normal((n_objects, n_kps, 3))
    d = np.linalg.norm(cld[:, None, :].astype(np.float64) - centres[None], axis=2)          # [N, objects]
    owner = d.argmin(axis=1)
    inside = d[np.arange(n), owner] < radius
    labels = np.where(inside, classes[owner], 0).astype(np.int32)
    kp_targ = np.where(inside[:, None, None], kps[owner] - cld[:, None, :], 0.0).astype(np.float32)
    ctr_targ = np.where(inside[:, None, None], centres[owner][:, None, :] - cld[:, None, :], 0.0).astype(np.float32)
    return dict(labels=labels, kp_targ_ofst=kp_targ, ctr_targ_ofst=ctr_targ)


def make_batch(config, batch_size, **kw):
    """Stack `batch_size` frames with seeds 1000*config + sample."""
    frames = [make_frame(frame_seed(config, s), **kw) for s in range(batch_size)]
    return {k: np.stack([f[k] for f in frames], axis=0) for k in frames[0]}


def strided_grids(dpt_xyz):
    """`sr2dptxyz` of linemod_dataset.py:299-311: xyz image sub-sampled at stride 1,2,4,8,
    pixel (y*s, x*s), flattened to [G_s,3].  dpt_xyz: [3,H,W] or [B,3,H,W]."""
    out = {}
    for s in (1, 2, 4, 8):
        h, w = dpt_xyz.shape[-2] // s, dpt_xyz.shape[-1] // s
        g = dpt_xyz[..., : h * s : s, : w * s : s]
        g = g.reshape(*g.shape[:-2], -1)
        out[s] = np.ascontiguousarray(np.swapaxes(g, -1, -2))
    return out


# ------------------------------------------------------------------------------------
# synthetic weights: deterministic per parameter NAME, so the reference model (golden
# generation, build container) and our modules (GPU box) get identical values without
# shipping a 135 MB checkpoint.
# ------------------------------------------------------------------------------------
def _tensor_seed(name, seed):
    return (zlib.crc32(name.encode("utf-8")) ^ (seed * 2654435761)) & 0x7FFFFFFF


def synth_tensor(name, shape, seed=0):
    """float32/int64 numpy array for state-dict entry `name` (shape from the module).

    The values are chosen so that activations stay O(1)-O(100) through the ~110 layers, as a
    trained network's do: He-normal conv weights, BatchNorm statistics close to identity, a
    small gain on the last BatchNorm of every residual block (otherwise the 16 residual adds
    grow the signal by 2^16 and the softmax poolings see logits of 1e6, which makes the
    network chaotic and every fp32 comparison meaningless), and the two input layers scaled
    for raw 0..255 colour values."""
    rng = np.random.RandomState(_tensor_seed(name, seed))
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_var":
        return rng.uniform(0.8, 1.25, size=shape).astype(np.float32)
    if leaf == "running_mean":
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == "bias":
        return (0.05 * rng.standard_normal(shape)).astype(np.float32)
    if len(shape) <= 1:  # BatchNorm gamma / PReLU slope
        if shape == (1,):
            return np.full(shape, 0.25, dtype=np.float32)
        if name.endswith(".bn2.weight"):          # residual-branch gain
            return rng.uniform(0.1, 0.3, size=shape).astype(np.float32)
        return rng.uniform(0.8, 1.2, size=shape).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    w = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    if name == "cnn_pre_stages.0.weight":          # consumes raw 0..255 rgb
        w *= np.float32(0.01)
    if name == "rndla_pre_stages.conv.weight":     # channels 3..5 of cld_rgb_nrm are raw rgb
        w[:, 3:6] *= np.float32(0.01)
    return w


def synth_state_dict(module, seed=0):
    """Deterministic state dict for a torch module with the reference's key names.
    Entries that alias the same storage (the reference shares `cnn.final` between
    cnn_up_stages[2] and [3], ffb6d.py:86-87) get the values of the first alias."""
    import torch

    out, first_name = {}, {}
    for name, t in module.state_dict().items():
        key = (t.data_ptr(), tuple(t.shape)) if t.numel() > 0 else (name, ())
        src = first_name.setdefault(key, name)
        out[name] = torch.from_numpy(synth_tensor(src, t.shape, seed)).to(t.dtype)
    return out


# parameters that alias one tensor in the reference model: `cnn.final` is registered under
# both cnn_up_stages[2] and cnn_up_stages[3] (ffb6d.py:86-87)
REFERENCE_ALIASES = {
    "cnn_up_stages.3.1.0.weight": "cnn_up_stages.2.0.0.weight",
    "cnn_up_stages.3.1.0.bias": "cnn_up_stages.2.0.0.bias",
}


def synth_state_dict_from_shapes(shapes, seed=0, n_classes=None):
    """Same values as synth_state_dict(reference_model), built from a {name: shape} map
    (tests/golden/state_dict_keys.json) instead of a module.  `n_classes` overrides the
    output width of the segmentation head (the only class-count dependent tensors)."""
    import torch

    out = {}
    for name, shape in shapes.items():
        shape = list(shape)
        if n_classes is not None and name.startswith("rgbd_seg_layer.3.conv."):
            shape[0] = n_classes
        src = REFERENCE_ALIASES.get(name, name)
        a = synth_tensor(src, shape, seed)
        out[name] = torch.from_numpy(a)
    return out


# --------------------------------------------------------------------------------------
# Pose-solver cases: what FFB6D.forward hands to
# cal_frame_poses[_lm] (ffb6d/utils/pvn3d_eval_utils_kpls.py:65-158,220-285), synthesised:
# objects = rigidly placed blobs of the cloud, votes = true offsets + noise + outliers.
# --------------------------------------------------------------------------------------
def random_rotation(rng):
    q = rng.randn(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_pose_case(seed, n_pts=2048, n_obj=2, n_kps=8, n_cls=None, noise=0.004, outliers=0.1,
                   label_noise=0.02, bg_frac=0.3, mesh_seed=None):
    """Returns a dict of numpy arrays:
      pcld f32[N,3], mask i64[N] (0 = background, 1..n_obj), ctr_of f32[1,N,3], kp_of f32[n_kps,N,3]
      mesh_kps f32[n_cls,n_kps,3], mesh_ctr f32[n_cls,3] (row = class id, row 0 unused),
      r_lst f32[n_cls-1] object radii, RT f64[n_cls,3,4] the true poses."""
    rng = np.random.RandomState(seed)
    n_cls = n_cls or n_obj + 1
    mrng = rng if mesh_seed is None else np.random.RandomState(mesh_seed)   # frames of one batch share the meshes
    mesh_kps = (mrng.rand(n_cls, n_kps, 3).astype(np.float32) - 0.5) * 0.2
    mesh_ctr = (mrng.rand(n_cls, 3).astype(np.float32) - 0.5) * 0.02
    r_lst = (0.08 + 0.05 * mrng.rand(n_cls - 1)).astype(np.float32)
    RT = np.zeros((n_cls, 3, 4))
    pcld = np.zeros((n_pts, 3), np.float32)
    mask = np.zeros((n_pts,), np.int64)
    ctr_of = np.zeros((1, n_pts, 3), np.float32)
    kp_of = np.zeros((n_kps, n_pts, 3), np.float32)
    owner = rng.choice(n_obj + 1, size=n_pts, p=[bg_frac] + [(1 - bg_frac) / n_obj] * n_obj)
    for c in range(1, n_obj + 1):
        R = random_rotation(rng)
        t = np.array([0.5 * (c - (n_obj + 1) / 2), 0.1 * rng.randn(), 1.0 + 0.2 * rng.rand()])
        RT[c, :, :3], RT[c, :, 3] = R, t
    for i in range(n_pts):
        c = owner[i]
        if c == 0:
            pcld[i] = [rng.uniform(-1, 1), rng.uniform(-0.6, 0.6), rng.uniform(0.6, 1.6)]
            ctr_of[0, i] = 0.2 * rng.randn(3)
            kp_of[:, i] = 0.2 * rng.randn(n_kps, 3)
            continue
        R, t = RT[c, :, :3], RT[c, :, 3]
        local = rng.randn(3)
        local *= 0.07 * rng.rand() ** (1 / 3) / np.linalg.norm(local)
        p = R @ local + t
        pcld[i] = p
        mask[i] = c
        ctr_of[0, i] = p - (R @ mesh_ctr[c] + t) + noise * rng.randn(3)
        kp_of[:, i] = p[None] - (mesh_kps[c] @ R.T + t) + noise * rng.randn(n_kps, 3)
        if rng.rand() < outliers:
            ctr_of[0, i] += 0.15 * rng.randn(3)
            kp_of[:, i] += 0.15 * rng.randn(n_kps, 3)
    flip = rng.rand(n_pts) < label_noise                      # segmentation mistakes
    mask[flip] = rng.randint(0, n_obj + 1, size=int(flip.sum()))
    return dict(pcld=pcld, mask=mask, ctr_of=ctr_of.astype(np.float32), kp_of=kp_of.astype(np.float32),
                mesh_kps=mesh_kps, mesh_ctr=mesh_ctr, r_lst=r_lst, RT=RT)
