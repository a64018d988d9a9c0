"""On-device index-pyramid builder: the 22 KNN searches per frame that the reference runs
on the CPU inside `Dataset.get_item` (ffb6d/datasets/linemod/linemod_dataset.py:299-353,
same block in ycb_dataset.py:253-309), batched over B frames and executed by the gfx950
KNN kernel without leaving the GPU.

Per encoder level i (N_i = N / 4^i):
    cld_nei_idx{i}     = knn(cld_i, cld_i, 16)
    cld_sub_idx{i}     = cld_nei_idx{i}[:N_i/4]          ("random sampling" = prefix of the
    sub_pts            = cld_i[:N_i/4]                      once-shuffled cloud, :322-323)
    cld_interp_idx{i}  = knn(sub_pts, cld_i, 1)
    r2p_ds_nei_idx{i}  = knn(grid[sr_i], sub_pts, 16)     sr = [4,8,8,8]
    p2r_ds_nei_idx{i}  = knn(sub_pts, grid[sr_i], 1)
Per decoder level i:  r2p_up_nei_idx{i} = knn(grid[sr_i], cld_xyz{3-i}, 16),
                      p2r_up_nei_idx{i} = knn(cld_xyz{3-i}, grid[sr_i], 1)   sr = [4,2,2]
grid[s] = xyz image at stride s, pixel (y*s, x*s), flattened row-major (:299-311).
"""
import torch

from .nearest_neighbors import PreparedPoints, knn_batch_device, knn_prepared, uses_pruning

RGB_DS_SR = (4, 8, 8, 8)
RGB_UP_SR = (4, 2, 2)
SUB_RATIO = (4, 4, 4, 4)
K_NEI = 16


def strided_grid(dpt_xyz, s):
    """dpt_xyz [B,3,H,W] -> [B,(H//s)*(W//s),3] contiguous."""
    B, _, H, W = dpt_xyz.shape
    h, w = H // s, W // s
    g = dpt_xyz[:, :, : h * s : s, : w * s : s]
    return g.reshape(B, 3, h * w).transpose(1, 2).contiguous()


class PyramidBuilder:
    """The pyramid level by level, so that a consumer can start on level 0 while the later levels are still being
    searched (forward_pm.forward enqueues the levels on their own HIP stream): `encoder_level(i)` for i = 0..3 in
    order, then `decoder_level(i)` for i = 0..2.  Each returns the keys of that level."""

    def __init__(self, cld, dpt_xyz, index_dtype=torch.int64):
        if cld.dim() != 3 or cld.shape[2] != 3 or dpt_xyz.dim() != 4 or dpt_xyz.shape[1] != 3:
            raise ValueError(f"bad shapes {tuple(cld.shape)} / {tuple(dpt_xyz.shape)}")
        self.B = cld.shape[0]
        self.index_dtype = index_dtype
        self.dpt_xyz = dpt_xyz
        self.grids = {}
        self.prepared = {}
        self.cur = cld.contiguous()
        self.xyz = []

    def grid(self, s):
        if s not in self.grids:
            self.grids[s] = strided_grid(self.dpt_xyz, s)
        return self.grids[s]

    def search(self, support, query, k):
        """Route big supports through Morton-prepared sets (each set is sorted once and reused by
        every search that touches it), small ones through the brute-force scan."""
        prepared = self.prepared
        if not uses_pruning(self.B, support.shape[1], query.shape[1], k):
            return knn_batch_device(support, query, k, dtype=self.index_dtype)
        if id(support) not in prepared:
            prepared[id(support)] = PreparedPoints(support)
        if k >= 2 and id(query) not in prepared:
            # 16-lane rows work on one query each: unsorted queries are fine, skip their sort
            return knn_prepared(prepared[id(support)], query, k, dtype=self.index_dtype)
        if id(query) not in prepared:
            prepared[id(query)] = PreparedPoints(query)
        return knn_prepared(prepared[id(support)], prepared[id(query)], k, dtype=self.index_dtype)

    def encoder_level(self, i):
        assert i == len(self.xyz), "levels are built in order"
        cur = self.cur
        n_sub = cur.shape[1] // SUB_RATIO[i]
        nei = self.search(cur, cur, K_NEI)
        sub = cur[:, :n_sub, :].contiguous()
        g = self.grid(RGB_DS_SR[i])
        out = {'cld_xyz%d' % i: cur, 'cld_nei_idx%d' % i: nei, 'cld_sub_idx%d' % i: nei[:, :n_sub, :].contiguous(),
               'cld_interp_idx%d' % i: self.search(sub, cur, 1),
               'r2p_ds_nei_idx%d' % i: self.search(g, sub, K_NEI), 'p2r_ds_nei_idx%d' % i: self.search(sub, g, 1)}
        self.xyz.append(cur)
        self.cur = sub
        return out

    def decoder_level(self, i):
        assert len(self.xyz) == 4, "decoder levels come after the four encoder levels"
        g = self.grid(RGB_UP_SR[i])
        pts = self.xyz[3 - i]
        return {'r2p_up_nei_idx%d' % i: self.search(g, pts, K_NEI), 'p2r_up_nei_idx%d' % i: self.search(pts, g, 1)}


def build_index_pyramid(cld, dpt_xyz, index_dtype=torch.int64):
    """cld [B,N,3] f32, dpt_xyz [B,3,H,W] f32 (both on the GPU) -> dict with the reference's
    key names: cld_xyz{i}, cld_nei_idx{i}, cld_sub_idx{i}, cld_interp_idx{i},
    r2p_ds_nei_idx{i}, p2r_ds_nei_idx{i} (i=0..3), r2p_up_nei_idx{i}, p2r_up_nei_idx{i}
    (i=0..2).  Index dtype int64 is what `model_fn` feeds the network (train_lm.py:236-237);
    int32 is what the dataset stores (halves the index traffic of every gather)."""
    b = PyramidBuilder(cld, dpt_xyz, index_dtype)
    out = {}
    for i in range(4):
        out.update(b.encoder_level(i))
    for i in range(3):
        out.update(b.decoder_level(i))
    return out


def frames_to_device(frames, device, with_pyramid=True, index_dtype=torch.int64):
    """numpy batch from ffb6d_amd.synth.make_batch -> the model's input dict on `device`,
    converted like `model_fn` (train_lm.py:233-241: float/uint8 -> float32, int32 -> int64)."""
    inputs = {
        'rgb': torch.from_numpy(frames['rgb']).to(device).float(),
        'cld_rgb_nrm': torch.from_numpy(frames['cld_rgb_nrm']).to(device),
        'choose': torch.from_numpy(frames['choose']).to(device).long(),
    }
    if with_pyramid:
        cld = torch.from_numpy(frames['cld']).to(device)
        dpt_xyz = torch.from_numpy(frames['dpt_xyz']).to(device)
        inputs.update(build_index_pyramid(cld, dpt_xyz, index_dtype=index_dtype))
    return inputs
