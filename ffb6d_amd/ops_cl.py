"""The neighbour operators of the TRAINING step on channels-last tensors (autograd-capable, float32 or bfloat16).

Same operators and signatures as ffb6d_amd.ops (= the reference's: FFB6D.random_sample / nearest_interpolation ffb6d.py:159-194,
Building_block.gather_neighbour / relative_pos_encoding RandLANet.py:216-234, the softmax-pool of Att_pooling.forward
RandLANet.py:245-248; plus the LogSoftmax of the colour decoder's `final` layer, pspnet.py:108-112), but every [B,C,...] tensor they take or return is read and written as ROWS of C contiguous channels --
torch's channels_last memory of a [B,C,N,K] tensor, which is what MIOpen's NHWC convolutions produce and consume -- in the
activation dtype (bf16 under torch.autocast).  A training step built from them has no transposing copy and no bf16 <-> fp32 cast
between a convolution and a neighbour operator (round 3's profile: 15 + 5 ms of an 82 ms step).

Forward gathers and the max-pool are the inference kernels (csrc/ops_pm.hip); the backward bodies are csrc/train_rows.hip.
Tensors of other layouts are accepted (one copy into rows); channel counts that are not a multiple of the 16-byte unit
(4 float32 / 8 bfloat16) fall back to the channel-major operators of ffb6d_amd.ops."""
import torch

from . import _lib, ops, ops_pm

_need_gpu = ops._need_gpu
_stream = ops._stream
_idx = ops._idx


def _act(t):
    """float32 / bfloat16 as they are; float16 (an fp16 autocast) is computed in float32"""
    return t.float() if t.dtype == torch.float16 else t


def _vl(t):
    return 8 if t.dtype == torch.bfloat16 else 4


def _dt(t):
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"float32 or bfloat16 rows expected, got {t.dtype}")
    return 1 if t.dtype == torch.bfloat16 else 0


def to_rows(x):
    """[B,C,*S] -> [B, prod(S), C] contiguous: a view when x is channels_last (or [B,C,N,1] written by a channels_last
    convolution), one transposing copy otherwise."""
    B, C = x.shape[:2]
    r = x.reshape(B, C, -1).transpose(1, 2)
    return r if r.is_contiguous() else r.contiguous()


def from_rows(r, spatial):
    """[B,M,C] contiguous rows -> [B,C,*spatial] view with channels_last strides"""
    B, M, C = r.shape
    return r.transpose(1, 2).reshape(B, C, *spatial)


def _rows_ld(x):
    """rows of a [B,C,*S] tensor WITHOUT copying channel slices of a channels_last tensor (the gradient of one half of a
    torch.cat): returns ([B,M,C] tensor, row stride in elements)."""
    B, C = x.shape[:2]
    r = x.reshape(B, C, -1).transpose(1, 2)
    M = r.shape[1]
    ld = r.stride(1) if M > 1 else C
    regular = r.stride(2) == 1 and ld >= C and (B == 1 or r.stride(0) == M * ld) and ld % _vl(x) == 0 and r.data_ptr() % 16 == 0 \
        and (M > 1 or r.is_contiguous())
    if not regular:
        r = r.contiguous()
        ld = C
    return r, ld


def _covers(*ts):
    """the row kernels cover these tensors' channel counts (axis 1)"""
    return all(t.shape[1] % _vl(t) == 0 for t in ts)


class GatherPlan:
    """The index tensor of a row gather, idx [B,U] into M source rows per frame, and -- built on the first backward, shared by
    every gather through the same indices (the two neighbour gathers of a Building_block) -- its inverse as CSR lists: `order` =
    the gather's flat output rows sorted by the source row they read, `start[r] .. start[r+1]` = the slice of `order` that reads
    source row r.  One radix sort + one searchsorted (torch / rocprim); lives as long as the autograd graph of one step."""

    def __init__(self, idx, m):
        self.idx, self.m = idx, int(m)
        self._csr = None

    def csr(self):
        if self._csr is None:
            B, U = self.idx.shape
            R = B * self.m
            kd = torch.int32 if R < 2 ** 31 - 1 else torch.int64
            key = self.idx.to(kd)
            if B > 1:
                key = key + (torch.arange(B, device=key.device, dtype=kd) * self.m).unsqueeze(1)
            skey, order = torch.sort(key.reshape(-1), stable=True)
            start = torch.searchsorted(skey, torch.arange(R + 1, device=key.device, dtype=kd))
            self._csr = (order, start)
        return self._csr


class _GatherRows(torch.autograd.Function):
    """rows [B,M,C], plan.idx [B,U] -> [B,U,C]; backward: every source row sums the output rows that read it (fp32), stored once
    in the activation dtype"""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, rows, plan):
        ctx.plan = plan
        return ops_pm.gather_rows(rows, plan.idx)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        plan = ctx.plan
        B, U, C = g.shape
        g = _act(g)
        if U == 0:                                           # an empty gather read nothing
            return torch.zeros((B, plan.m, C), dtype=g.dtype, device=g.device), None
        gr, ld = _rows_ld(g.transpose(1, 2))
        order, start = plan.csr()
        out = torch.empty((B, plan.m, C), dtype=g.dtype, device=g.device)
        nbytes = g.element_size() * C * B * (U + plan.m) + 8 * B * (U + plan.m)
        with torch.cuda.device(g.device), _lib.traced("gather_sum_rows", nbytes, (C, plan.m, U)):
            rc = _lib.load().ffb6d_gather_sum_rows(_dt(gr), gr.data_ptr(), ld, order.data_ptr(), start.data_ptr(), out.data_ptr(),
                                                   B * plan.m, C, _stream(g))
        _lib.check(rc, "ffb6d_gather_sum_rows")
        return out, None


class _RandomSampleRows(torch.autograd.Function):
    """rows [B,M,C], pool_idx [B,Np,K] -> [B,Np,C] (max over the K gathered rows)"""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, rows, pool_idx):
        ctx.save_for_backward(rows, pool_idx)
        return ops_pm.random_sample(rows, pool_idx)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        rows, pool_idx = ctx.saved_tensors
        B, M, C = rows.shape
        Np, K = pool_idx.shape[1], pool_idx.shape[2]
        g = _act(g)
        if g.dtype != rows.dtype:
            g = g.to(rows.dtype)
        gr, ld = _rows_ld(g.transpose(1, 2))
        i, bits = _idx(pool_idx)
        acc = torch.zeros((B, M, C), dtype=torch.float32, device=g.device)
        nbytes = rows.element_size() * B * C * (Np * K + Np) + (bits // 8) * B * Np * K + 8 * B * M * C
        with torch.cuda.device(g.device), _lib.traced("random_sample_rows_bwd", nbytes, (C, M, Np)):
            rc = _lib.load().ffb6d_random_sample_rows_bwd(_dt(rows), rows.data_ptr(), i.data_ptr(), bits, gr.data_ptr(), ld,
                                                          acc.data_ptr(), B, M, C, Np, K, _stream(g))
        _lib.check(rc, "ffb6d_random_sample_rows_bwd")
        return acc.to(rows.dtype), None


class _AttPoolRows(torch.autograd.Function):
    """feat, scores: [B,C,N,K] (rows (n,k) of C channels); -> [B,N,C] rows"""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, feat, scores):
        B, C, N, K = feat.shape
        fr, ldf = _rows_ld(feat)
        sr, lds = _rows_ld(scores)
        out = torch.empty((B, N, C), dtype=feat.dtype, device=feat.device)
        nbytes = feat.element_size() * B * C * N * (2 * K + 1)
        with torch.cuda.device(feat.device), _lib.traced("att_pool_rows", nbytes, (C, N)):
            rc = _lib.load().ffb6d_att_pool_rows(_dt(feat), fr.data_ptr(), ldf, sr.data_ptr(), lds, out.data_ptr(), B * N, K, C,
                                                 _stream(feat))
        _lib.check(rc, "ffb6d_att_pool_rows")
        ctx.save_for_backward(fr, sr)
        ctx.meta = (B, C, N, K, ldf, lds)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        fr, sr = ctx.saved_tensors
        B, C, N, K, ldf, lds = ctx.meta
        g = _act(g)
        if g.dtype != fr.dtype:
            g = g.to(fr.dtype)
        gr, ldg = _rows_ld(g.transpose(1, 2))
        gf = torch.empty((B, N * K, C), dtype=fr.dtype, device=g.device)
        gs = torch.empty((B, N * K, C), dtype=fr.dtype, device=g.device)
        nbytes = fr.element_size() * B * C * N * (4 * K + 1)
        with torch.cuda.device(g.device), _lib.traced("att_pool_rows_bwd", nbytes, (C, N)):
            rc = _lib.load().ffb6d_att_pool_rows_bwd(_dt(fr), gr.data_ptr(), ldg, fr.data_ptr(), ldf, sr.data_ptr(), lds, gf.data_ptr(),
                                                     gs.data_ptr(), B * N, K, C, _stream(g))
        _lib.check(rc, "ffb6d_att_pool_rows_bwd")
        return from_rows(gf, (N, K)), from_rows(gs, (N, K))


class _LogSoftmaxRows(torch.autograd.Function):
    """x [B,C,*S] -> log_softmax over C, same dtype and memory"""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x):
        rows = to_rows(x)                                                    # [B,M,C]
        B, M, C = rows.shape
        y = torch.empty_like(rows)
        with torch.cuda.device(x.device), _lib.traced("log_softmax_rows", 2 * rows.element_size() * rows.numel(), (C, B * M)):
            rc = _lib.load().ffb6d_log_softmax_rows(_dt(rows), rows.data_ptr(), y.data_ptr(), B * M, C, _stream(x))
        _lib.check(rc, "ffb6d_log_softmax_rows")
        ctx.save_for_backward(rows)                                          # the input: the backward recomputes the softmax in fp32
        ctx.spatial = tuple(x.shape[2:])
        return from_rows(y, ctx.spatial)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        B, M, C = x.shape
        g = _act(g)
        gr = to_rows(g if g.dtype == x.dtype else g.to(x.dtype))
        gx = torch.empty_like(x)
        with torch.cuda.device(x.device), _lib.traced("log_softmax_rows_bwd", 3 * x.element_size() * x.numel(), (C, B * M)):
            rc = _lib.load().ffb6d_log_softmax_rows_bwd(_dt(x), gr.data_ptr(), x.data_ptr(), gx.data_ptr(), B * M, C, _stream(x))
        _lib.check(rc, "ffb6d_log_softmax_rows_bwd")
        return from_rows(gx, ctx.spatial)


def channel_log_softmax(x):
    """nn.LogSoftmax over dim 1 of a [B,C,H,W] map (pspnet.py:108-112) in the map's own dtype (fp32 arithmetic).  torch.autocast
    lists log_softmax as a float32 operator; every consumer of this map is a convolution that rounds its input to the autocast
    dtype, a row gather or a max over gathered rows -- all of which commute with that (monotonic) rounding -- so rounding once
    here gives them the same values and keeps the two largest maps of the decoder out of fp32."""
    _need_gpu(x)
    x = _act(x)
    q, r = divmod(x.shape[1], _vl(x))
    if x.dim() < 3 or r or q > 64 or q & (q - 1) or x.dtype not in (torch.float32, torch.bfloat16):
        return torch.log_softmax(x, dim=1)
    return _LogSoftmaxRows.apply(x)


def nearest_interpolation(feature, interp_idx, spatial=None, plan=None):
    """FFB6D.nearest_interpolation (ffb6d.py:179-194): feature [B,C,M,1] (or any [B,C,*S]), interp_idx [B,U,1] -> [B,C,U,1]
    (or [B,C,*spatial] with prod(spatial) == U: the point -> pixel fusion reshapes to the map right away).
    `plan`: a GatherPlan of the same indices to share their inverse with other gathers."""
    _need_gpu(feature, interp_idx)
    feature = _act(feature)
    B = feature.shape[0]
    idx = interp_idx.reshape(B, -1)
    spatial = (idx.shape[1], 1) if spatial is None else tuple(spatial)
    if not _covers(feature):
        return ops.nearest_interpolation(feature.reshape(B, feature.shape[1], -1), idx.unsqueeze(2)).reshape(B, -1, *spatial)
    rows = to_rows(feature)
    if plan is None:
        plan = GatherPlan(idx, rows.shape[1])
    elif plan.m != rows.shape[1] or plan.idx.shape != idx.shape:
        raise ValueError("nearest_interpolation: the GatherPlan belongs to another index tensor")
    return from_rows(_GatherRows.apply(rows, plan), spatial)


def neighbour_plan(neigh_idx):
    """GatherPlan of a [B,N,K] neighbour index tensor (sources = the N points themselves)"""
    B, N, K = neigh_idx.shape
    return GatherPlan(neigh_idx.reshape(B, N * K), N)


def gather_neighbour(feature, neigh_idx, plan=None):
    """Building_block.gather_neighbour (RandLANet.py:225-234) on a [B,C,N,1] map: -> [B,C,N,K], rows (n,k) of C channels."""
    B, N, K = neigh_idx.shape
    return nearest_interpolation(feature, neigh_idx.reshape(B, N * K, 1), (N, K), plan)


def gather_neighbour_rows(pc, neighbor_idx):
    """Building_block.gather_neighbour with the reference's own signature (RandLANet.py:225-234): pc [B,N,C] rows, neighbor_idx
    [B,N,K] -> [B,N,K,C] in pc's dtype -- the reference's layout IS rows; its permute(0,3,1,2) afterwards is a channels_last view."""
    _need_gpu(pc, neighbor_idx)
    pc = _act(pc)
    if pc.dim() != 3 or neighbor_idx.dim() != 3 or pc.shape[0] != neighbor_idx.shape[0]:
        raise ValueError(f"bad shapes {tuple(pc.shape)} / {tuple(neighbor_idx.shape)}")
    B, N, K = neighbor_idx.shape
    if pc.shape[2] % _vl(pc):
        return ops.gather_neighbour(pc, neighbor_idx)
    rows = pc if pc.is_contiguous() else pc.contiguous()
    return _GatherRows.apply(rows, GatherPlan(neighbor_idx.reshape(B, N * K), rows.shape[1])).reshape(B, N, K, -1)


def choose_gather(rgb_emb, choose):
    """the per-point pixel pick of ffb6d.py:309-312: rgb_emb [B,C,H,W], choose [B,1,N] -> [B,C,N]"""
    return nearest_interpolation(rgb_emb, choose.reshape(choose.shape[0], -1, 1)).squeeze(3)


def random_sample(feature, pool_idx):
    """FFB6D.random_sample (ffb6d.py:159-177): feature [B,C,M,1] / [B,C,H,W] / [B,C,M], pool_idx [B,N',K] -> [B,C,N',1]"""
    _need_gpu(feature, pool_idx)
    feature = _act(feature)
    if not _covers(feature):
        return ops.random_sample(feature.reshape(feature.shape[0], feature.shape[1], -1), pool_idx)
    return from_rows(_RandomSampleRows.apply(to_rows(feature), pool_idx), (pool_idx.shape[1], 1))


def att_pool(feature_set, att_activation):
    """sum_K(feature_set * softmax_K(att_activation)) (RandLANet.py:245-248): both [B,C,N,K] -> [B,C,N,1]"""
    _need_gpu(feature_set, att_activation)
    if feature_set.shape != att_activation.shape or feature_set.dim() != 4:
        raise ValueError(f"bad shapes {tuple(feature_set.shape)} / {tuple(att_activation.shape)}")
    f, a = _act(feature_set), _act(att_activation)
    if f.dtype != a.dtype:
        dt = torch.promote_types(f.dtype, a.dtype)
        f, a = f.to(dt), a.to(dt)
    if not _covers(f):
        return ops.att_pool(f, a)
    return from_rows(_AttPoolRows.apply(f, a), (f.shape[2], 1))


def relative_pos_encoding(xyz, neigh_idx, dtype=torch.float32):
    """relative_pos_encoding (RandLANet.py:216-223), no gradient (xyz is an input): xyz [B,N,3], neigh_idx [B,N,K] ->
    [B,16,N,K] in channels_last memory: channels [dis, p-q, p, q] + 6 zero channels (the rows are 16 wide so that they stay
    16-byte units; multiply with the 10-column weight padded by 6 zero columns -- padded_mlp1_weight)."""
    enc = ops_pm.relative_pos_encoding(xyz, neigh_idx, dtype=dtype)                    # [B,N,K,16]
    return enc.permute(0, 3, 1, 2)
