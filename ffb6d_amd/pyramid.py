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

from .nearest_neighbors import prepare_many, search_many, uses_pruning

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


def point_sets(cloud, dpt_xyz, channel_major=False, with_table=False):
    """Every point set the pyramid's searches read, in ONE launch (ffb6d_pyramid_sets_f32; 13 ATen launches before): returns
    (sets, table) with sets[('c', i)] = cloud level i as [B,N_i,3] rows (i = 0..4: the cloud, its prefixes of a quarter each --
    linemod_dataset.py:322-323 -- and the prefix below the last level), sets[('g', s)] = strided_grid(dpt_xyz, s) for the strides of
    RGB_DS_SR / RGB_UP_SR, table = the [B,N,4] coordinate rows of the fused local feature aggregation (ops_pm.xyz_table) or None.
    cloud: [B,N,3], or with channel_major the tensor whose first three channels are the coordinates ([B,C,N]: cld_rgb_nrm)."""
    import ctypes

    from . import _lib
    lib = _lib.load()
    if cloud.dtype != torch.float32 or dpt_xyz.dtype != torch.float32 or cloud.dim() != 3 or dpt_xyz.dim() != 4 or dpt_xyz.shape[1] != 3:
        raise TypeError("point_sets: float32 cloud [B,N,3] (or [B,C,N]) and xyz image [B,3,H,W]")
    if cloud.shape[1 if channel_major else 2] < 3 or cloud.shape[0] != dpt_xyz.shape[0] or cloud.device != dpt_xyz.device:
        raise ValueError(f"bad shapes {tuple(cloud.shape)} / {tuple(dpt_xyz.shape)}")
    dev = cloud.device
    B, N = int(cloud.shape[0]), int(cloud.shape[2 if channel_major else 1])
    H, W = int(dpt_xyz.shape[2]), int(dpt_xyz.shape[3])
    dpt_xyz = dpt_xyz.contiguous()
    strides = sorted(set(RGB_DS_SR) | set(RGB_UP_SR))
    ns = [N]
    for r in SUB_RATIO:
        ns.append(ns[-1] // r)
    sets = {('c', i): torch.empty((B, n, 3), dtype=torch.float32, device=dev) for i, n in enumerate(ns)}
    for s in strides:
        sets[('g', s)] = torch.empty((B, (H // s) * (W // s), 3), dtype=torch.float32, device=dev)
    table = torch.empty((B, N, 4), dtype=torch.float32, device=dev) if with_table else None
    fs, cs, ps = (cloud.stride(0), cloud.stride(1), cloud.stride(2)) if channel_major else (cloud.stride(0), cloud.stride(2), cloud.stride(1))
    level_n = (ctypes.c_int64 * len(ns))(*ns)
    level_out = (ctypes.c_void_p * len(ns))(*[sets[('c', i)].data_ptr() for i in range(len(ns))])
    st = (ctypes.c_int * len(strides))(*strides)
    grid_out = (ctypes.c_void_p * len(strides))(*[sets[('g', s)].data_ptr() for s in strides])
    nbytes = 4 * (sum(3 * B * n for n in ns) + 3 * B * N + (4 * B * N if with_table else 0)
                  + 2 * sum(3 * B * (H // s) * (W // s) for s in strides))
    with torch.cuda.device(dev), _lib.traced("pyramid_sets", nbytes, (B, N, H, W)):
        rc = lib.ffb6d_pyramid_sets_f32(cloud.data_ptr(), fs, cs, ps, B, N, len(ns), level_n, level_out,
                                        table.data_ptr() if with_table else None, dpt_xyz.data_ptr(), H, W, len(strides), st, grid_out,
                                        torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "ffb6d_pyramid_sets_f32")
    return sets, table


class PyramidBuilder:
    """The pyramid with a level-by-level interface -- `encoder_level(i)` for i = 0..3 in order, then `decoder_level(i)` for
    i = 0..2, each returning the keys of that level -- but TWO batches of searches behind it (`search_batch`): the eleven K = 16
    searches, then the eleven K = 1 searches (nearest_neighbors.search_many: one launch per kernel involved, 0.6 ms against 1.41 ms
    for 22 separate launches).  forward_pm.forward runs the builder on its own HIP stream and takes the batches one by one
    (`neighbour_keys`, `nearest_keys`): the point branch starts after the first.

    All 22 searches read only the cloud and the xyz image, so every point set is known up front: the four cloud levels
    (prefixes of the cloud, linemod_dataset.py:322-323), the prefix below the last one, and the image grids at strides 2, 4
    and 8.  The sets that go through the Morton-ordered search are prepared TOGETHER when the builder is created
    (nearest_neighbors.prepare_many: one sort, a handful of launches), before the first level is searched."""

    def __init__(self, cld, dpt_xyz, index_dtype=torch.int64, sets=None):
        if cld.dim() != 3 or cld.shape[2] != 3 or dpt_xyz.dim() != 4 or dpt_xyz.shape[1] != 3:
            raise ValueError(f"bad shapes {tuple(cld.shape)} / {tuple(dpt_xyz.shape)}")
        self.B = cld.shape[0]
        self.index_dtype = index_dtype
        # ('c', i) = cloud level i (i = 4: the prefix below level 3), ('g', s) = image grid at stride s (strided_grid); `sets`: the
        # caller has them already (forward_pm.StreamedPyramid builds them together with the cloud rows and the coordinate table)
        self.sets = sets if sets is not None else point_sets(cld.float(), dpt_xyz.float())[0]
        # the searches, level by level: (output key, support set, query set, K)
        self.plan = []
        for i in range(4):
            c, sub, g = ('c', i), ('c', i + 1), ('g', RGB_DS_SR[i])
            self.plan.append([('cld_nei_idx%d' % i, c, c, K_NEI), ('cld_interp_idx%d' % i, sub, c, 1),
                              ('r2p_ds_nei_idx%d' % i, g, sub, K_NEI), ('p2r_ds_nei_idx%d' % i, sub, g, 1)])
        for i in range(3):
            g, pts = ('g', RGB_UP_SR[i]), ('c', 3 - i)
            self.plan.append([('r2p_up_nei_idx%d' % i, g, pts, K_NEI), ('p2r_up_nei_idx%d' % i, pts, g, 1)])
        # sets the pruned search wants in Morton order: every support of such a search, and its queries when K = 1 (the
        # 16-lane rows of the K >= 2 kernel take unsorted queries)
        need = []
        for level in self.plan:
            for _, sup, qry, k in level:
                if uses_pruning(self.B, self.sets[sup].shape[1], self.sets[qry].shape[1], k):
                    for key in (sup,) + ((qry,) if k < 2 else ()):
                        if key not in need:
                            need.append(key)
        self.prepared = dict(zip(need, prepare_many([self.sets[k] for k in need])))
        self.n_levels = 0
        self.found = None
        self.done = set()

    def search_batch(self, k16):
        """One batch of searches, none of which depends on another, in ONE call (nearest_neighbors.search_many: one launch per
        kernel involved): the eleven K = 16 searches (the 16-lane row kernel + the scan at K = 16) or the eleven K = 1 searches
        (the K = 1 kernel + the scan at K = 1).  Two batches since the end of round 5: the point branch's first layers need only
        K = 16 indices (neighbours and the sub-sampling prefix), so the forward lets it start after the first batch while the
        second one runs (forward_pm.StreamedPyramid); the K = 1 indices are first read by the first fusion stage."""
        flat = [srch for level in self.plan for srch in level if (srch[3] == K_NEI) == bool(k16)]
        args = []
        for _, sup, qry, k in flat:
            pruned = uses_pruning(self.B, self.sets[sup].shape[1], self.sets[qry].shape[1], k)
            args.append((self.prepared[sup] if pruned else self.sets[sup],
                         self.prepared.get(qry, self.sets[qry]) if pruned else self.sets[qry], k))
        out = search_many(args, dtype=self.index_dtype)
        if self.found is None:
            self.found = {}
        self.found.update({name: idx for (name, _, _, _), idx in zip(flat, out)})
        self.done.add(bool(k16))

    def neighbour_keys(self, sub_levels=(0, 1, 2, 3)):
        """everything the K = 16 batch produces, for all levels: cld_xyz{i}, cld_nei_idx{i}, r2p_ds_nei_idx{i}, r2p_up_nei_idx{i}
        (runs the batch if it has not run), and the sub-sampling prefixes cld_sub_idx{i} of `sub_levels`"""
        if True not in self.done:
            self.search_batch(True)
        out = {name: self.found[name] for level in self.plan for name, _, _, k in level if k == K_NEI}
        for i in range(4):
            out['cld_xyz%d' % i] = self.sets[('c', i)]
        out.update(self.sub_index_keys(sub_levels))
        return out

    def sub_index_keys(self, levels):
        """cld_sub_idx{i} = the neighbour rows of the points that survive the sub-sampling (a prefix, linemod_dataset.py:322-323)"""
        return {'cld_sub_idx%d' % i: self.found['cld_nei_idx%d' % i][:, :self.sets[('c', i + 1)].shape[1], :].contiguous()
                for i in levels}

    def nearest_keys(self):
        """everything the K = 1 batch produces: cld_interp_idx{i}, p2r_ds_nei_idx{i}, p2r_up_nei_idx{i}"""
        if False not in self.done:
            self.search_batch(False)
        return {name: self.found[name] for level in self.plan for name, _, _, k in level if k != K_NEI}

    def _level(self, j):
        assert j == self.n_levels, "levels are built in order"
        self.n_levels += 1
        for k16 in (True, False):
            if k16 not in self.done:
                self.search_batch(k16)
        return {name: self.found[name] for name, _, _, _ in self.plan[j]}

    def encoder_level(self, i):
        out = self._level(i)
        nei = out['cld_nei_idx%d' % i]
        n_sub = self.sets[('c', i + 1)].shape[1]
        out['cld_xyz%d' % i] = self.sets[('c', i)]
        out['cld_sub_idx%d' % i] = nei[:, :n_sub, :].contiguous()
        return out

    def decoder_level(self, i):
        assert self.n_levels >= 4, "decoder levels come after the four encoder levels"
        return self._level(4 + i)


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
