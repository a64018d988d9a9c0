"""oracle/pose_ref.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement (torch fp32 / numpy) of the reference's pose solver, the step that follows
FFB6D.forward:

  mean_shift_fit        MeanShiftTorch.fit         ffb6d/utils/meanshift_pytorch.py:27-58
  best_fit_transform    best_fit_transform          ffb6d/utils/pvn3d_eval_utils_kpls.py:28-61
  frame_poses_lm        cal_frame_poses_lm          ffb6d/utils/pvn3d_eval_utils_kpls.py:220-285
  frame_poses_ycb       cal_frame_poses             ffb6d/utils/pvn3d_eval_utils_kpls.py:65-158

Differences from the reference are interface only: the mesh keypoints / centres / radii the
reference reads from dataset files through `Basic_Utils` are arguments here, and nothing
is moved to a GPU.  Pinned against the reference itself by tests/golden/make_golden_pose.py
(fixture tests/golden/pose_small.npz) and tests/test_oracle_cpu.py.
"""
import math

import numpy as np
import torch


def mean_shift_fit(votes, bandwidth=0.05, max_iter=300):
    """votes f32 [M,3] -> (centre f32[3], labels bool[M], iterations).
    Every point is moved to the Gaussian-weighted mean of all points until the largest move is
    below bandwidth*1e-3 (or max_iter+1 rounds, :35-49); the answer is the converged point with
    the most neighbours inside `bandwidth`, labels = membership of that ball (:50-55)."""
    M, c = votes.shape
    pts = votes.clone()
    norm_c = bandwidth * math.sqrt(2 * np.pi)
    rounds = 0
    while True:
        rounds += 1
        dist = torch.norm(pts.reshape(1, M, c) - pts.reshape(M, 1, c), dim=2)
        wgt = (torch.exp(-0.5 * ((dist / bandwidth)) ** 2) / norm_c).reshape(M, M, 1)
        moved = torch.sum(wgt * pts, dim=1) / torch.sum(wgt, dim=1)
        step = torch.norm(moved - pts, dim=1)
        pts = moved
        if torch.max(step) < bandwidth * 1e-3 or rounds > max_iter:
            break
    dist = torch.norm(pts.view(M, 1, c) - pts.view(1, M, c), dim=2)
    inside = torch.sum(dist < bandwidth, dim=1)
    _, best = torch.max(inside, 0)
    return pts[best, :], dist[best] < bandwidth, rounds


def best_fit_transform(model_pts, cam_pts):
    """Least-squares rigid transform model -> camera (Kabsch), [3,4] = [R|t] (:28-61).
    Runs in the dtype of the inputs (float32 in the reference's callers), result float64."""
    assert model_pts.shape == cam_pts.shape
    m = model_pts.shape[1]
    ca = np.mean(model_pts, axis=0)
    cb = np.mean(cam_pts, axis=0)
    H = np.dot((model_pts - ca).T, cam_pts - cb)
    U, _, Vt = np.linalg.svd(H)
    R = np.dot(Vt.T, U.T)
    if np.linalg.det(R) < 0:          # reflection: flip the axis of the smallest singular value
        Vt[m - 1, :] *= -1
        R = np.dot(Vt.T, U.T)
    T = np.zeros((3, 4))
    T[:, :3] = R
    T[:, 3] = cb.T - np.dot(R, ca.T)
    return T


def _votes(pcld, ctr_of, kp_of):
    n_kps, n_pts, _ = kp_of.shape
    return pcld - ctr_of[0], pcld.view(1, n_pts, 3).repeat(n_kps, 1, 1) - kp_of


def _object_keypoints(ctr_votes, kp_votes, sel, use_ctr_clus_flter, radius):
    """centre + keypoints of one object from the votes of the points in `sel` (:262-275 / :120-136)."""
    ctr, labels, _ = mean_shift_fit(ctr_votes[sel, :], radius)
    if labels.sum() < 1:
        labels[0] = 1
    cand = kp_votes[:, sel, :]
    if use_ctr_clus_flter:
        cand = cand[:, labels, :]
    kps = [mean_shift_fit(v, radius)[0] for v in cand]
    return torch.stack(kps + [ctr])


def frame_poses_lm(pcld, mask, ctr_of, kp_of, use_ctr, use_ctr_clus_flter, mesh_kps, mesh_ctr,
                   radius=0.04):
    """LineMOD: one object, class id 1 (:220-285).  Returns ([3,4] pose list, kps [n_kps+1,3])."""
    ctr_votes, kp_votes = _votes(pcld, ctr_of, kp_of)
    n_kps = kp_of.shape[0]
    sel = mask == 1
    if sel.sum() < 1:
        return [np.identity(4)[:3, :]], np.zeros((n_kps + 1, 3), np.float32)
    kps = _object_keypoints(ctr_votes, kp_votes, sel, use_ctr_clus_flter, radius)
    model = mesh_kps
    if use_ctr:
        model = np.concatenate((mesh_kps, mesh_ctr.reshape(1, 3)), axis=0)
        found = kps
    else:
        found = kps[:n_kps]
    return [best_fit_transform(model, found.contiguous().numpy())], kps.numpy()


def frame_poses_ycb(pcld, mask, ctr_of, kp_of, use_ctr, use_ctr_clus_flter, mesh_kps, mesh_ctr, r_lst,
                    radius=0.04):
    """YCB: every class present in `mask` (:65-158).  mesh_kps [n_cls,n_kps,3], mesh_ctr [n_cls,3]
    indexed by class id, r_lst[cls_id-1] = object radius.  Returns (class ids, poses, kps)."""
    ctr_votes, kp_votes = _votes(pcld, ctr_of, kp_of)
    n_kps, n_pts, _ = kp_of.shape
    cls_ids = np.unique(mask[mask > 0].contiguous().numpy())
    if use_ctr_clus_flter and len(cls_ids):
        # reassign every foreground point to the class whose voted centre is closest, if that
        # centre is within 0.8 object radii (:85-108)
        ctrs = torch.stack([mean_shift_fit(ctr_votes[mask == c, :], radius)[0] for c in cls_ids])
        d = torch.norm(ctr_votes.view(n_pts, 1, 3) - ctrs.view(1, -1, 3), dim=2)
        min_dis, min_idx = torch.min(d, dim=1)
        closest = torch.from_numpy(cls_ids.astype(np.int64))[min_idx]
        new_mask = mask.clone()
        for c in cls_ids:
            upd = (mask > 0) & (closest == c) & (min_dis < r_lst[c - 1] * 0.8)
            new_mask[upd] = closest[upd].to(new_mask.dtype)
        mask = new_mask
    poses, kps_out = [], []
    for c in cls_ids:
        sel = mask == c
        if sel.sum() < 1:
            poses.append(np.identity(4)[:3, :])
            kps_out.append(np.zeros((n_kps + 1, 3)))
            continue
        kps = _object_keypoints(ctr_votes, kp_votes, sel, use_ctr_clus_flter, radius)
        if not use_ctr:
            kps = torch.cat([kps[:n_kps], torch.zeros(1, 3)])[:n_kps]
        model = mesh_kps[c]
        if use_ctr:
            model = np.concatenate((model, mesh_ctr[c].reshape(1, 3)), axis=0)
        found = kps.contiguous().numpy()
        poses.append(best_fit_transform(model, found))
        kps_out.append(found)
    return cls_ids, poses, kps_out
