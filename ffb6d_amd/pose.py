"""Pose solver on the GPU: the step after FFB6D.forward.

Host-side mirror of the reference's
    MeanShiftTorch                      ffb6d/utils/meanshift_pytorch.py:27-58
    best_fit_transform                  ffb6d/utils/pvn3d_eval_utils_kpls.py:28-61
    cal_frame_poses / cal_frame_poses_lm   ffb6d/utils/pvn3d_eval_utils_kpls.py:65-158,220-285
over the C ABI of include/ffb6d_pose.h.  The reference solves one object at a time (one mean
shift for the centre, one per keypoint, each with host round trips); `solve_poses` runs all
(frame, object, keypoint) vote sets of a batch through the same launches.

The mesh keypoints / centres / radii the reference reads from dataset files through
`Basic_Utils` (pvn3d_eval_utils_kpls.py:149-152,277-280, common.py:89-94) are arguments here.
There is no CPU fallback: tensors must live on a GPU and the HIP library must be built.
"""
import numpy as np
import torch

from . import _lib
from .ops import _need_gpu, _stream

RADIUS = 0.04          # bandwidth used by both flows (pvn3d_eval_utils_kpls.py:76,231)
CHECK_EVERY = 8        # rounds between convergence polls


FIT_SPREAD = 1         # csrc/pose.hip ffb6d_pose_set_fit_spread: 1 = the fits' first two rounds chip-wide when the call has few sets


def set_fit_spread(mode):
    """0 never / 1 when G <= CUs / 2 (default) / 2 always; returns the previous setting.  Results do not depend on it."""
    global FIT_SPREAD
    prev, FIT_SPREAD = FIT_SPREAD, int(mode)
    _lib.load().ffb6d_pose_set_fit_spread(FIT_SPREAD)
    return prev


def _i32(values, device):
    return torch.tensor(list(values), dtype=torch.int32, device=device)


def _mask_bits(mask):
    if mask.dtype == torch.int64:
        return 64
    if mask.dtype == torch.int32:
        return 32
    raise TypeError(f"mask must be int32 or int64, got {mask.dtype}")


def vote_sets(pcld, offsets, mask, frame_of, class_of, keep=None):
    """Votes `pcld - offsets` of the points with mask == class, per (frame, class) pair.
    pcld [B,N,3], offsets [B,S,N,3], mask [B,N] -> sets f32 [P*S, N, 4], counts i32 [P]."""
    _need_gpu(pcld, offsets, mask)
    B, N, _ = pcld.shape
    S = offsets.shape[1]
    P = frame_of.numel()
    pcld, offsets, mask = pcld.contiguous().float(), offsets.contiguous().float(), mask.contiguous()
    sets = torch.empty((P * S, N, 4), dtype=torch.float32, device=pcld.device)
    counts = torch.zeros((P,), dtype=torch.int32, device=pcld.device)
    with torch.cuda.device(pcld.device), _lib.traced("vote_sets", 4 * (3 * B * N * (S + 1)) + 16 * P * S * N, (P, S, N)):
        rc = _lib.load().ffb6d_vote_sets_f32(
            pcld.data_ptr(), offsets.data_ptr(), mask.data_ptr(), _mask_bits(mask),
            keep.data_ptr() if keep is not None else None, frame_of.data_ptr(), class_of.data_ptr(),
            P, B, S, N, N, sets.data_ptr(), counts.data_ptr(), _stream(pcld))
    _lib.check(rc, "ffb6d_vote_sets_f32")
    return sets, counts


def mean_shift(sets, counts, bandwidth, max_iter=300, sets_per_count=1, want_labels=True, check_every=CHECK_EVERY,
               max_count=None):
    """MeanShiftTorch.fit for every set: sets f32 [G,M,4], counts i32 [G/sets_per_count] ->
    centers f32 [G,3], labels bool [G,M] (or None), n_inside i32 [G], rounds i32 [G].
    max_count: bound on the counts (sizes the grids); None = read it back from `counts` when the
    call polls the device anyway (check_every > 0), else unknown."""
    _need_gpu(sets, counts)
    G, M, _ = sets.shape
    dev = sets.device
    centers = torch.empty((G, 3), dtype=torch.float32, device=dev)
    labels = torch.empty((G, M), dtype=torch.uint8, device=dev) if want_labels else None
    n_inside = torch.empty((G,), dtype=torch.int32, device=dev)
    rounds = torch.empty((G,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    if max_count is None:
        max_count = 0            # (round 5: the library sizes its grids itself; no read-back of the counts here)
    wbytes = lib.ffb6d_mean_shift_workspace_bytes(G, M)
    ws = torch.empty((max(wbytes, 1),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _lib.traced("mean_shift", 16 * G * M, (G, M)):
        rc = lib.ffb6d_mean_shift_f32(
            sets.data_ptr(), counts.data_ptr(), sets_per_count, G, M, int(max_count), float(bandwidth), int(max_iter),
            int(check_every), centers.data_ptr(), labels.data_ptr() if labels is not None else None,
            n_inside.data_ptr(), rounds.data_ptr(), ws.data_ptr(), wbytes, _stream(sets))
    _lib.check(rc, "ffb6d_mean_shift_f32")
    return centers, (labels.bool() if labels is not None else None), n_inside, rounds


class MeanShiftTorch:
    """Same constructor and `fit` contract as the reference class (meanshift_pytorch.py:27-58):
    fit(A [N,3] on the GPU) -> (centre [3], labels bool [N])."""

    def __init__(self, bandwidth=0.05, max_iter=300):
        self.bandwidth = bandwidth
        self.stop_thresh = bandwidth * 1e-3
        self.max_iter = max_iter

    def fit_batch(self, clouds):
        """list of [N_i,3] tensors -> (centers [G,3], list of bool [N_i] labels)."""
        dev = clouds[0].device
        M = max(int(c.shape[0]) for c in clouds)
        sets = torch.zeros((len(clouds), M, 4), dtype=torch.float32, device=dev)
        for g, c in enumerate(clouds):
            sets[g, : c.shape[0], :3] = c
        counts = _i32([c.shape[0] for c in clouds], dev)
        centers, labels, _, _ = mean_shift(sets, counts, self.bandwidth, self.max_iter)
        return centers, [labels[g, : c.shape[0]] for g, c in enumerate(clouds)]

    def fit(self, A):
        _need_gpu(A)
        if A.dim() != 2 or A.shape[1] != 3 or A.shape[0] < 1:
            raise ValueError(f"expected a non-empty [N,3] tensor, got {tuple(A.shape)}")
        centers, labels = self.fit_batch([A])
        return centers[0], labels[0]


def best_fit_transform_batch(model, found):
    """model, found f32 [P,n,3] on the GPU -> T f64 [P,3,4]."""
    _need_gpu(model, found)
    if model.shape != found.shape or model.dim() != 3 or model.shape[2] != 3:
        raise ValueError(f"shape mismatch {tuple(model.shape)} / {tuple(found.shape)}")
    P, n, _ = model.shape
    model, found = model.contiguous().float(), found.contiguous().float()
    T = torch.empty((P, 3, 4), dtype=torch.float64, device=model.device)
    with torch.cuda.device(model.device):
        rc = _lib.load().ffb6d_best_fit_transform_f32(model.data_ptr(), found.data_ptr(), P, n, T.data_ptr(),
                                                      _stream(model))
    _lib.check(rc, "ffb6d_best_fit_transform_f32")
    return T


def best_fit_transform(A, B, device=None):
    """numpy in, numpy [3,4] out, like pvn3d_eval_utils_kpls.py:28-61 (A: model points, B: camera)."""
    assert A.shape == B.shape
    device = device or torch.device("cuda", torch.cuda.current_device())
    T = best_fit_transform_batch(torch.from_numpy(np.ascontiguousarray(A, np.float32)).to(device)[None],
                                 torch.from_numpy(np.ascontiguousarray(B, np.float32)).to(device)[None])
    return T[0].cpu().numpy()


def solve_poses(pcld, mask, ctr_of, kp_of, mesh_kps, mesh_ctr, r_lst=None, use_ctr=True,
                use_ctr_clus_flter=True, refine_mask=None, classes=None, radius=RADIUS, max_iter=300, stats=None):
    """All objects of all frames at once.
      pcld f32 [B,N,3]; mask int [B,N] predicted class per point (0 = background);
      ctr_of f32 [B,1,N,3], kp_of f32 [B,n_kps,N,3] (end_points['pred_ctr_ofs'/'pred_kp_ofs']);
      mesh_kps [n_cls,n_kps,3], mesh_ctr [n_cls,3] model-frame keypoints indexed by class id;
      r_lst[cls_id-1] object radii (needed when refine_mask);
      classes: None = every class present in a frame's mask (YCB flow, :82); or a list of class ids
               solved in every frame whether present or not (LineMOD flow: [1], :238-241);
      refine_mask: centre-clustering mask filter (:83-108); default = use_ctr_clus_flter and classes is None.
      stats: optional dict, receives the rounds made per set of each mean-shift pass.
    Returns a list over frames of (class_ids int array, poses [n,3,4] float64, kps [n,n_kps+1,3] float32);
    objects without points get the identity pose and zero keypoints (:114-117 / :239-240)."""
    _need_gpu(pcld, mask, ctr_of, kp_of)
    B, N, _ = pcld.shape
    n_kps = kp_of.shape[1]
    dev = pcld.device
    mesh_kps = torch.as_tensor(np.asarray(mesh_kps, np.float32), device=dev)
    mesh_ctr = torch.as_tensor(np.asarray(mesh_ctr, np.float32), device=dev)
    n_cls = mesh_kps.shape[0]
    if refine_mask is None:
        refine_mask = use_ctr_clus_flter and classes is None
    mask = mask.contiguous()
    lib = _lib.load()

    if classes is None:
        present = torch.zeros((B, n_cls), dtype=torch.bool, device=dev)
        present.scatter_(1, mask.clamp(0, n_cls - 1).long(), True)
        present[:, 0] = False
        present = present.cpu().numpy()
        per_frame = [np.nonzero(present[b])[0] for b in range(B)]
    else:
        per_frame = [np.asarray(classes, np.int64) for _ in range(B)]
    frame_np = np.concatenate([np.full(len(c), b, np.int32) for b, c in enumerate(per_frame)]) if B else np.zeros(0, np.int32)
    class_np = np.concatenate(per_frame).astype(np.int32) if B else np.zeros(0, np.int32)
    P = len(class_np)
    if P == 0:
        return [(np.zeros(0, np.int64), np.zeros((0, 3, 4)), np.zeros((0, n_kps + 1, 3), np.float32)) for _ in range(B)]
    frame_of, class_of = _i32(frame_np, dev), _i32(class_np, dev)
    ctr_of = ctr_of.contiguous().float()
    pcld = pcld.contiguous().float()

    if refine_mask:
        if r_lst is None:
            raise ValueError("refine_mask needs r_lst (object radii, common.py:89-94)")
        sets, counts = vote_sets(pcld, ctr_of, mask, frame_of, class_of)
        centers, _, _, rounds = mean_shift(sets, counts, radius, max_iter, want_labels=False)
        if stats is not None:
            stats["rounds_refine"], stats["counts_refine"] = rounds, counts
        begin = np.zeros(B + 1, np.int32)
        np.add.at(begin, frame_np + 1, 1)
        pair_begin = _i32(np.cumsum(begin), dev)
        max_dist = torch.tensor([np.float32(float(r_lst[c - 1]) * 0.8) for c in class_np], dtype=torch.float32, device=dev)
        new_mask = torch.empty_like(mask)
        with torch.cuda.device(dev):
            rc = lib.ffb6d_refine_mask_by_center(pcld.data_ptr(), ctr_of.data_ptr(), mask.data_ptr(), _mask_bits(mask),
                                                 centers.data_ptr(), class_of.data_ptr(), pair_begin.data_ptr(),
                                                 max_dist.data_ptr(), B, N, new_mask.data_ptr(), _stream(pcld))
        _lib.check(rc, "ffb6d_refine_mask_by_center")
        mask = new_mask

    # object centres (+ the labels of the winning centre cluster)
    sets, counts = vote_sets(pcld, ctr_of, mask, frame_of, class_of)
    centers, labels, _, rounds = mean_shift(sets, counts, radius, max_iter, want_labels=use_ctr_clus_flter)
    if stats is not None:
        stats["rounds_ctr"], stats["counts_ctr"] = rounds, counts
    keep = None
    if use_ctr_clus_flter:
        keep = torch.zeros((B, N), dtype=torch.uint8, device=dev)
        lab8 = labels.to(torch.uint8)
        with torch.cuda.device(dev):
            rc = lib.ffb6d_set_labels_to_points(sets.data_ptr(), lab8.data_ptr(), counts.data_ptr(), frame_of.data_ptr(),
                                                P, N, N, keep.data_ptr(), _stream(pcld))
        _lib.check(rc, "ffb6d_set_labels_to_points")
    # keypoints: n_kps sets per pair
    ksets, kcounts = vote_sets(pcld, kp_of, mask, frame_of, class_of, keep=keep)
    kcenters, _, _, rounds = mean_shift(ksets, kcounts, radius, max_iter, sets_per_count=n_kps, want_labels=False)
    if stats is not None:
        stats["rounds_kps"], stats["counts_kps"] = rounds, kcounts
    found = torch.cat([kcenters.view(P, n_kps, 3), centers.view(P, 1, 3)], dim=1)          # cls_kps rows (:124,136)
    cls_long = class_of.long()
    model = torch.cat([mesh_kps[cls_long], mesh_ctr[cls_long].view(P, 1, 3)], dim=1)
    if use_ctr:
        T = best_fit_transform_batch(model, found)
    else:
        T = best_fit_transform_batch(model[:, :n_kps], found[:, :n_kps])

    empty = (counts == 0).cpu().numpy()
    T = T.cpu().numpy()
    found = found.cpu().numpy()
    T[empty] = np.identity(4)[:3, :]
    found[empty] = 0
    out = []
    for b in range(B):
        sel = frame_np == b
        out.append((class_np[sel].astype(np.int64), T[sel], found[sel]))
    return out


def cal_frame_poses_lm(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter, obj_id,
                       debug=False, mesh_kps=None, mesh_ctr=None):
    """Signature of pvn3d_eval_utils_kpls.py:220-285 plus the mesh keypoints of `obj_id`
    (mesh_kps [n_kps,3], mesh_ctr [3]) the reference loads from its dataset files (:277-280).
    pcld [N,3], mask [N], ctr_of [1,N,3], pred_kp_of [n_kps,N,3] -> [pose [3,4]]."""
    if mesh_kps is None or (use_ctr and mesh_ctr is None):
        raise ValueError("mesh_kps / mesh_ctr of the object are required")
    n_kps = pred_kp_of.shape[0]
    kps = np.zeros((2, n_kps, 3), np.float32)
    kps[1] = mesh_kps
    ctr = np.zeros((2, 3), np.float32)
    if mesh_ctr is not None:
        ctr[1] = np.asarray(mesh_ctr).reshape(3)
    res = solve_poses(pcld[None], mask[None], ctr_of[None], pred_kp_of[None], kps, ctr, use_ctr=use_ctr,
                      use_ctr_clus_flter=use_ctr_clus_flter, refine_mask=False, classes=[1])
    return [res[0][1][0]]


def cal_frame_poses(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter, gt_kps=None, gt_ctrs=None,
                    debug=False, kp_type='farthest', mesh_kps=None, mesh_ctr=None, r_lst=None):
    """Signature of pvn3d_eval_utils_kpls.py:65-158 plus mesh_kps [n_cls,n_kps,3], mesh_ctr [n_cls,3]
    (row = class id) and r_lst.  Returns (pred_cls_ids, pred_pose_lst, pred_kps_lst)."""
    if mesh_kps is None or mesh_ctr is None:
        raise ValueError("mesh_kps / mesh_ctr are required")
    ids, poses, kps = solve_poses(pcld[None], mask[None], ctr_of[None], pred_kp_of[None], mesh_kps, mesh_ctr,
                                  r_lst=r_lst, use_ctr=use_ctr, use_ctr_clus_flter=use_ctr_clus_flter)[0]
    return ids, [p for p in poses], [k for k in kps]
