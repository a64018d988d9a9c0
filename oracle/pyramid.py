"""oracle/pyramid.py -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the index-pyramid
builder inside the reference's dataset (ffb6d/datasets/linemod/linemod_dataset.py:299-353,
identical block in ycb_dataset.py:253-309), parameterised by the knn_search callable so it
can run on our C restatement (oracle.knn.knn_search) or on the reference's own kd-tree
(oracle.ref_harness -> DataProcessing.knn_search)."""
import numpy as np

RGB_DS_SR = [4, 8, 8, 8]      # linemod_dataset.py:313
RGB_UP_SR = [4, 2, 2]         # linemod_dataset.py:342
PCLD_SUB_SR = [4, 4, 4, 4]    # linemod_dataset.py:315
N_DS, N_UP, K_NEI = 4, 3, 16


def strided_grids(dpt_xyz_chw):
    """linemod_dataset.py:299-311 for one frame: {stride: [G,3]}."""
    c, h, w = dpt_xyz_chw.shape
    out = {1: dpt_xyz_chw.reshape(3, -1).transpose(1, 0)}
    for i in range(3):
        s = 2 ** (i + 1)
        nh, nw = h // s, w // s
        ys, xs = np.mgrid[:nh, :nw]
        out[s] = dpt_xyz_chw[:, ys * s, xs * s].reshape(3, -1).transpose(1, 0)
    return out


def build_pyramid(cld, dpt_xyz_chw, knn_search):
    """One frame.  cld [N,3], dpt_xyz_chw [3,H,W] -> dict of numpy arrays with the
    reference's key names and dtypes (int32 indices, float32 xyz)."""
    sr2dptxyz = strided_grids(dpt_xyz_chw)
    inputs = {}
    cld = np.asarray(cld)
    for i in range(N_DS):
        nei_idx = knn_search(cld[None], cld[None], K_NEI).astype(np.int32).squeeze(0)
        n_sub = cld.shape[0] // PCLD_SUB_SR[i]
        sub_pts = cld[:n_sub, :]
        pool_i = nei_idx[:n_sub, :]
        up_i = knn_search(sub_pts[None], cld[None], 1).astype(np.int32).squeeze(0)
        inputs['cld_xyz%d' % i] = cld.astype(np.float32).copy()
        inputs['cld_nei_idx%d' % i] = nei_idx.astype(np.int32).copy()
        inputs['cld_sub_idx%d' % i] = pool_i.astype(np.int32).copy()
        inputs['cld_interp_idx%d' % i] = up_i.astype(np.int32).copy()
        grid = sr2dptxyz[RGB_DS_SR[i]]
        inputs['r2p_ds_nei_idx%d' % i] = knn_search(grid[None], sub_pts[None], K_NEI).astype(np.int32).squeeze(0)
        inputs['p2r_ds_nei_idx%d' % i] = knn_search(sub_pts[None], grid[None], 1).astype(np.int32).squeeze(0)
        cld = sub_pts
    for i in range(N_UP):
        grid = sr2dptxyz[RGB_UP_SR[i]]
        pts = inputs['cld_xyz%d' % (N_DS - i - 1)]
        inputs['r2p_up_nei_idx%d' % i] = knn_search(grid[None], pts[None], K_NEI).astype(np.int32).squeeze(0)
        inputs['p2r_up_nei_idx%d' % i] = knn_search(pts[None], grid[None], 1).astype(np.int32).squeeze(0)
    return inputs


def build_batch(frames, knn_search):
    """frames: dict of batched numpy arrays from ffb6d_amd.synth.make_batch."""
    per = [build_pyramid(frames['cld'][b], frames['dpt_xyz'][b], knn_search)
           for b in range(frames['cld'].shape[0])]
    return {k: np.stack([p[k] for p in per], axis=0) for k in per[0]}


def knn_calls(pyr, dpt_xyz_chw):
    """The (support, query) pair behind every index tensor of one frame's pyramid:
    {key: (support [S,3], query [Q,3])}, for tie canonicalisation (oracle.knn.canonical_ties)."""
    grids = strided_grids(dpt_xyz_chw)
    calls = {}
    for i in range(N_DS):
        cld = pyr['cld_xyz%d' % i]
        n_sub = cld.shape[0] // PCLD_SUB_SR[i]
        sub = cld[:n_sub]
        g = np.ascontiguousarray(grids[RGB_DS_SR[i]], dtype=np.float32)
        calls['cld_nei_idx%d' % i] = (cld, cld)
        calls['cld_sub_idx%d' % i] = (cld, sub)
        calls['cld_interp_idx%d' % i] = (sub, cld)
        calls['r2p_ds_nei_idx%d' % i] = (g, sub)
        calls['p2r_ds_nei_idx%d' % i] = (sub, g)
    for i in range(N_UP):
        g = np.ascontiguousarray(grids[RGB_UP_SR[i]], dtype=np.float32)
        pts = pyr['cld_xyz%d' % (N_DS - i - 1)]
        calls['r2p_up_nei_idx%d' % i] = (g, pts)
        calls['p2r_up_nei_idx%d' % i] = (pts, g)
    return calls
