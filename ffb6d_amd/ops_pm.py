"""Operators of the point-major / pixel-major ("channels last") inference path: every activation is a matrix
[rows, C] with one contiguous row of C floats per point or pixel, rows running over all frames of the batch.
Thin wrappers around the C ABI (include/ffb6d_ops.h); inference only (no autograd), GPU only (no CPU fallback).

Reference functions these implement are cited per wrapper; the layout itself is this package's choice (every gather
of the hot path moves whole rows, both GEMM operands are K-contiguous -- csrc/mlp_pm.hip)."""
import torch

from . import _lib
from .ops import ACT_LEAKY, ACT_NONE, ACT_RELU, _idx, _need_gpu, _stream  # noqa: F401


def _dt(*ts):
    """dtype code of the row elements (0 = float32, 1 = bfloat16), the same for every tensor given."""
    dts = {t.dtype for t in ts if t is not None}
    if len(dts) != 1 or next(iter(dts)) not in (torch.float32, torch.bfloat16):
        raise TypeError(f"row operands must all be float32 or all bfloat16, got {sorted(map(str, dts))}")
    return 1 if next(iter(dts)) == torch.bfloat16 else 0


def rows_view(x):
    """[..., C] tensor whose leading dims are row-regular -> (2-d view [rows, C], row stride in floats)."""
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    if x2.stride(1) != 1 or x2.data_ptr() % 16 or (x2.shape[0] > 1 and (x2.stride(0) * x2.element_size()) % 16):
        x2 = x2.contiguous()
    return x2, (x2.stride(0) if x2.shape[0] > 1 else C)


ACT_LOG_SOFTMAX = 3
# the tile-sequence form of the long-row fp32 GEMMs (csrc/mlp_pm.hip: mlp_pm_seq_kernel, round 5); False = the LDS-tiled form everywhere
# (bit-identical results; forward_pm.GEMM_SEQ_FORM sets it per forward for in-process A/B)
MLP_SEQ_FORM = True
# the 256 x 256 bf16 tile with LDS-DMA operand loads (csrc/mlp_pm_big.hip, round 6); False = the LDS-tiled 128 x 128 form on those launches
# (bit-identical results; forward_pm.GEMM_BIG_FORM sets it per forward)
MLP_BIG_FORM = True
# schedule of the tile-sequence form (csrc/mlp_pm.hip: LIN): 0 = the round-5 plans (whole-point-tile groups in three regions), 1 = balanced
# contiguous sequences per XCD with at least two per workgroup slot, 2 = one per slot; forward_pm.GEMM_SEQ_LIN sets it per forward
MLP_SEQ_LIN = 0
_big_form_set = _seq_lin_set = None


def _sync_big_form(lib):
    global _big_form_set, _seq_lin_set
    if _big_form_set is not MLP_BIG_FORM:
        lib.ffb6d_mlp_pm_set_big_form(int(MLP_BIG_FORM))
        _big_form_set = MLP_BIG_FORM
    if _seq_lin_set != MLP_SEQ_LIN:
        lib.ffb6d_mlp_pm_set_seq_lin(int(MLP_SEQ_LIN))
        _seq_lin_set = MLP_SEQ_LIN


def mlp(x1, w, bias=None, act=ACT_NONE, x2=None, add=None, gather=None, x1_gather=None, out=None, tile_hint=0, role="path"):
    """Shared MLP (1x1 conv + folded BatchNorm + activation; pytorch_utils.py:75-129, RandLA/pytorch_utils.py:35-111)
    on row-major activations:  out[r,:] = act(w @ cat(x1[r], x2[r]) + bias + extra[r])
        x1 [..., K1], x2 [..., K2] or None, w [Cout, K1+K2] (conv weight layout, BN folded), bias [Cout] or None
        add       = Y [..., Cout] with the output's leading shape: extra[r] = Y[r]          (residual / broadcast sums)
        gather    = (Y [B, Py, Cout], idx [B, P]): extra[b, p] = Y[b, idx[b, p]]            (conv(cat(a, interp(b))))
        x1_gather = idx [B, P]: x1 is [B, M, K1] and row (b, p) of the GEMM reads x1[b, idx[b, p]]   (`choose`, :309-312)
    act: ACT_NONE / ACT_RELU / ACT_LEAKY / ACT_LOG_SOFTMAX (over the Cout <= 64 channels, pspnet.py:108-112).
    Returns [..., Cout] (or writes `out`, which may be a channel slice of a wider row buffer).
    `role` only labels the launch in traces (bench.py reports the north-star path's GEMMs and the colour decoder's separately)."""
    _need_gpu(x1, w)
    lib = _lib.load()
    _sync_big_form(lib)
    dt = _dt(x1, w, x2, add, gather[0] if gather is not None else None, out)
    tdt = x1.dtype
    esz = x1.element_size()
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("bias stays float32 in both precisions")
    a, ld1 = rows_view(x1.detach())
    K1 = a.shape[1]
    xi, px, bits = None, 0, 0
    lead = x1.shape[:-1]
    if x1_gather is not None:
        if x1.dim() != 3 or x1_gather.dim() != 2 or x1_gather.shape[0] != x1.shape[0]:
            raise ValueError("x1_gather needs x1 [B,M,K1] and idx [B,P]")
        xi, bits = _idx(x1_gather.reshape(-1))
        px = x1.shape[1]
        lead = tuple(x1_gather.shape)
    rows = 1
    for n in lead:
        rows *= int(n)
    P = rows // lead[0] if (x1_gather is not None or gather is not None) and len(lead) else rows
    b, ld2, K2 = None, 0, 0
    if x2 is not None:
        b, ld2 = rows_view(x2.detach())
        K2 = b.shape[1]
        if b.shape[0] != rows:
            raise ValueError(f"x2 {tuple(x2.shape)} does not match the {rows} output rows")
    if w.dim() != 2 or w.shape[1] != K1 + K2 or not w.is_contiguous():
        raise ValueError(f"w must be contiguous [Cout, {K1 + K2}], got {tuple(w.shape)}")
    Cout = w.shape[0]
    y = gi = None
    ldy = py = 0
    if gather is not None:
        yv, gidx = gather
        B = yv.shape[0]
        y, ldy = rows_view(yv.detach())
        py = yv.shape[1]
        gi, gbits = _idx(gidx.reshape(-1))
        if xi is not None and gbits != bits:
            raise TypeError("x1_gather and gather indices must have the same dtype")
        bits = gbits
        if gi.numel() != rows or rows % B:
            raise ValueError("gather index must have one entry per output row")
        P = rows // B
    elif add is not None:
        y, ldy = rows_view(add.detach())
        if y.shape[0] != rows or y.shape[1] != Cout:
            raise ValueError(f"add {tuple(add.shape)} does not match the output [{rows}, {Cout}]")
    if out is None:
        out = torch.empty(tuple(lead) + (Cout,), dtype=tdt, device=x1.device)
    if out.stride(-1) != 1 or out.numel() != rows * Cout:
        raise ValueError("out must be [..., Cout] with contiguous channels")
    ldo = out.stride(-2) if out.dim() >= 2 and rows > 1 else Cout
    if out.dim() > 2 and any(out.stride(i) != out.stride(i + 1) * out.shape[i + 1] for i in range(out.dim() - 2)):
        raise ValueError("out must be row-regular (a channel slice of a contiguous row buffer is fine)")
    nbytes = esz * ((K1 + K2) * Cout + rows * (K1 + K2) + rows * Cout) + rows * (bits // 8) * ((gi is not None) + (xi is not None)) + \
        (esz * (y.shape[0] if gi is None else rows) * Cout if y is not None else 0)
    if tile_hint == 0 and not MLP_SEQ_FORM and not dt and \
            (lib.ffb6d_mlp_pm_choice(rows, Cout, K1, K2, int(act), 0, int(xi is not None)) & 255) == 8:
        tile_hint = 7                # A/B: the LDS-tiled form where the automatic choice is the tile-sequence form
    if _lib.TRACER is not None:      # tag = the kernel instantiation a profile lists this launch under (8 = the tile-sequence form, any plan)
        tile = (int(tile_hint) if tile_hint > 0 else lib.ffb6d_mlp_pm_choice(rows, Cout, K1, K2, int(act), int(dt), int(xi is not None))) & 255
    else:
        tile = 0
    with torch.cuda.device(x1.device), _lib.traced("mlp_pm", nbytes, (K1 + K2, Cout, rows, tile, dt, role)):
        rc = (lib.ffb6d_mlp_pm_bf16 if dt else lib.ffb6d_mlp_pm_f32)(w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                  a.data_ptr(), K1, ld1, xi.data_ptr() if xi is not None else None, px,
                                  b.data_ptr() if b is not None else None, K2, ld2,
                                  y.data_ptr() if y is not None else None, ldy,
                                  gi.data_ptr() if gi is not None else None, py, bits, P,
                                  out.data_ptr(), ldo, rows, Cout, int(act), int(tile_hint), _stream(x1))
    _lib.check(rc, "ffb6d_mlp_pm_f32")
    return out


def att_pool(f, nei_idx, g, w_fc, out=None):
    """Att_pooling.forward up to the pooled tensor (RandLANet.py:243-248) with the neighbour gather fused in.
    f [B,N,C1] point features, nei_idx [B,N,16], g [B,N,16,C2] per-pair features, w_fc [C1+C2, C1+C2] (fc conv weight)
    -> [B,N,C1+C2] = sum_k S * softmax_k(S w_fc^T), S[(n,k)] = cat(f[nei[n,k]], g[n,k])."""
    _need_gpu(f, g, w_fc)
    lib = _lib.load()
    B, N, C1 = f.shape
    K, C2 = g.shape[2], g.shape[3]
    if g.shape[:2] != (B, N) or nei_idx.shape != (B, N, K) or w_fc.shape != (C1 + C2, C1 + C2) or not w_fc.is_contiguous():
        raise ValueError(f"bad shapes {tuple(f.shape)} / {tuple(nei_idx.shape)} / {tuple(g.shape)} / {tuple(w_fc.shape)}")
    dt = _dt(f, g, w_fc, out)
    f2, ldf = rows_view(f.detach())
    g2, ldg = rows_view(g.detach())
    idx, bits = _idx(nei_idx)
    d = C1 + C2
    if out is None:
        out = torch.empty((B, N, d), dtype=f.dtype, device=f.device)
    nbytes = f.element_size() * (d * d + B * N * C1 + B * N * K * C2 + B * N * d) + (bits // 8) * B * N * K
    with torch.cuda.device(f.device), _lib.traced("att_pool_pm", nbytes, (d, N, dt)):
        rc = (lib.ffb6d_att_pool_pm_bf16 if dt else lib.ffb6d_att_pool_pm_f32)(w_fc.data_ptr(), f2.data_ptr(), C1, ldf, idx.data_ptr(), bits, g2.data_ptr(), C2, ldg,
                                       out.data_ptr(), out.stride(-2), B, N, K, _stream(f))
    _lib.check(rc, "ffb6d_att_pool_pm_f32")
    return out


def random_sample(feature, pool_idx):
    """FFB6D.random_sample (ffb6d.py:159-177) on rows: feature [B,M,C], pool_idx [B,Np,K] -> [B,Np,C] (max over K rows)."""
    _need_gpu(feature, pool_idx)
    lib = _lib.load()
    f = feature.detach()
    f = f if f.is_contiguous() else f.contiguous()
    B, M, C = f.shape
    Np, K = pool_idx.shape[1], pool_idx.shape[2]
    idx, bits = _idx(pool_idx)
    out = torch.empty((B, Np, C), dtype=f.dtype, device=f.device)
    nbytes = f.element_size() * (B * C * M + B * C * Np) + (bits // 8) * B * Np * K
    with torch.cuda.device(f.device), _lib.traced("random_sample_pm", nbytes, (C, M, Np)):
        rc = lib.ffb6d_random_sample_pm(_dt(f), f.data_ptr(), idx.data_ptr(), bits, out.data_ptr(), B, M, C, Np, K, _stream(f))
    _lib.check(rc, "ffb6d_random_sample_pm_f32")
    return out


def gather_rows(feature, idx):
    """FFB6D.nearest_interpolation / the `choose` pick (ffb6d.py:179-194,309-312) on rows:
    feature [B,M,C], idx [B,U] -> [B,U,C]."""
    _need_gpu(feature, idx)
    lib = _lib.load()
    f = feature.detach()
    f = f if f.is_contiguous() else f.contiguous()
    B, M, C = f.shape
    i, bits = _idx(idx.reshape(B, -1))
    U = i.shape[1]
    out = torch.empty((B, U, C), dtype=f.dtype, device=f.device)
    nbytes = f.element_size() * (B * C * M + B * C * U) + (bits // 8) * B * U
    with torch.cuda.device(f.device), _lib.traced("gather_rows_pm", nbytes, (C, M, U)):
        rc = lib.ffb6d_gather_rows_pm(_dt(f), f.data_ptr(), i.data_ptr(), bits, out.data_ptr(), B, M, C, U, _stream(f))
    _lib.check(rc, "ffb6d_gather_rows_pm_f32")
    return out


def relative_pos_encoding(xyz, neigh_idx, dtype=torch.float32):
    """relative_pos_encoding (RandLANet.py:216-223) as rows of 16 channels [dis, p-q, p, q, 0*6]: xyz [B,N,3] float32,
    neigh_idx [B,N,K] -> [B,N,K,16] of `dtype` (the zero padding makes the row a legal K of the point-major shared MLP)."""
    _need_gpu(xyz, neigh_idx)
    lib = _lib.load()
    x = xyz.detach().contiguous()
    idx, bits = _idx(neigh_idx)
    B, N, K = idx.shape
    out = torch.empty((B, N, K, 16), dtype=dtype, device=x.device)
    nbytes = 12 * B * N + (bits // 8) * B * N * K + 16 * out.element_size() * B * N * K
    with torch.cuda.device(x.device), _lib.traced("relative_pos_encoding_pm", nbytes, (N,)):
        rc = lib.ffb6d_relative_pos_encoding_pm(_dt(out), x.data_ptr(), idx.data_ptr(), bits, out.data_ptr(), B, N, K, _stream(x))
    _lib.check(rc, "ffb6d_relative_pos_encoding_pm_f32")
    return out


def posenc_mlp(xyz, neigh_idx, w, bias, act, dtype=torch.float32):
    """mlp1(relative_pos_encoding(xyz, neigh_idx)) of Building_block.forward (RandLANet.py:196-199,216-223) in one pass:
    xyz [B,N,3] float32, neigh_idx [B,N,K], w [cout, >=10] float32 (BatchNorm folded, columns past 10 ignored), bias [cout]
    float32 or None -> [B,N,K,cout] rows of `dtype`; the 10-channel encoding stays in registers (csrc/posenc.hip)."""
    _need_gpu(xyz, neigh_idx, w)
    lib = _lib.load()
    x = xyz.detach().contiguous()
    idx, bits = _idx(neigh_idx)
    B, N, K = idx.shape
    if w.dtype != torch.float32 or w.dim() != 2 or w.shape[1] < 10 or w.stride(1) != 1 or (bias is not None and bias.dtype != torch.float32):
        raise ValueError("posenc_mlp: w must be float32 [cout, >=10] with contiguous rows, bias float32")
    cout = w.shape[0]
    out = torch.empty((B, N, K, cout), dtype=dtype, device=x.device)
    nbytes = 12 * B * N + (bits // 8) * B * N * K + out.element_size() * out.numel()
    with torch.cuda.device(x.device), _lib.traced("posenc_mlp_pm", nbytes, (N, cout)):
        rc = lib.ffb6d_posenc_mlp_pm(_dt(out), x.data_ptr(), idx.data_ptr(), bits, w.data_ptr(), w.stride(0),
                                     bias.data_ptr() if bias is not None else None, int(act), out.data_ptr(), B, N, K, cout, _stream(x))
    _lib.check(rc, "ffb6d_posenc_mlp_pm")
    return out


def xyz_table(xyz):
    """[B,N,3] float32 coordinates -> [B,N,4] table of 16-byte rows {x, y, z, 0}: what csrc/lfa_pm.hip gathers neighbour
    coordinates from (one aligned 16-byte load per point instead of three 4-byte ones)."""
    return torch.nn.functional.pad(xyz.detach().float(), (0, 1)).contiguous()


def k_chunked(w, perm=False):
    """[cout, d] weight -> [d / VL, cout, VL] (VL = 16 bytes of consecutive k): the layout csrc/lfa_pm.hip reads the output
    MLP's weight in (one coalesced 16-byte load per channel and k-chunk).
    perm (bfloat16, d a multiple of 32): the chunks in the order the accumulators of a 32x32x16 MFMA layer supply k in
    (csrc/mlp_chain.hip): chunk 2m + h holds the input channels base + (0..3), base + 8 + (0..3), base = 32 (m >> 1) + 16 (m & 1) + 4 h."""
    vl = 16 // w.element_size()
    cout, d = w.shape
    w = w.detach()
    if perm:
        if vl != 8 or d % 32:
            raise ValueError("the permuted chunk order is the bfloat16 chain's: 8 k per chunk, d a multiple of 32")
        q = torch.arange(d // 8, device=w.device)
        m, h = q // 2, q % 2
        base = 32 * (m // 2) + 16 * (m % 2) + 4 * h
        off = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11], device=w.device)
        w = w[:, (base[:, None] + off[None, :]).reshape(-1)]
    return w.reshape(cout, d // vl, vl).permute(1, 0, 2).contiguous()


def mlp_chain3(x, layer1, layer2, layer3, cout3):
    """Three shared MLPs in a row as one launch (csrc/mlp_chain.hip; the layers after the first of a prediction head,
    ffb6d.py:135-157): x [..., 128] float32 or bfloat16 rows (a channel slice of a wider row buffer is fine); layer_i = (W k-chunked, bias,
    act) with W1, W2 = k_chunked([128,128]), W3 = k_chunked([32,128]) (rows >= cout3 zero) of x's dtype -- in bfloat16 W2 and W3 with
    perm=True --, biases float32 [128], [128], [32] -> [..., cout3] of x's dtype."""
    _need_gpu(x)
    lib = _lib.load()
    a, ldx = rows_view(x.detach())
    if x.dtype not in (torch.float32, torch.bfloat16) or a.shape[1] != 128:
        raise ValueError("mlp_chain3 takes float32 / bfloat16 rows of 128 channels")
    vl = 16 // x.element_size()
    (w1, b1, a1), (w2, b2, a2), (w3, b3, a3) = layer1, layer2, layer3
    for w, b, shape, n in ((w1, b1, (128 // vl, 128, vl), 128), (w2, b2, (128 // vl, 128, vl), 128), (w3, b3, (128 // vl, 32, vl), 32)):
        if tuple(w.shape) != shape or w.dtype != x.dtype or not w.is_contiguous() or b.dtype != torch.float32 or b.numel() != n:
            raise ValueError(f"k-chunked {x.dtype} weight {shape} and float32 bias [{n}] expected, got {tuple(w.shape)} {w.dtype} / {tuple(b.shape)}")
    rows = a.shape[0]
    out = torch.empty(tuple(x.shape[:-1]) + (int(cout3),), dtype=x.dtype, device=x.device)
    nbytes = x.element_size() * (rows * (128 + int(cout3)) + 2 * 128 * 128 + 32 * 128)
    fn = lib.ffb6d_mlp_chain3_pm_bf16 if x.dtype == torch.bfloat16 else lib.ffb6d_mlp_chain3_pm_f32
    with torch.cuda.device(x.device), _lib.traced("mlp_chain3_pm", nbytes, (128, int(cout3), rows)):
        rc = fn(a.data_ptr(), ldx, w1.data_ptr(), b1.data_ptr(), int(a1), w2.data_ptr(), b2.data_ptr(), int(a2),
                w3.data_ptr(), b3.data_ptr(), int(a3), out.data_ptr(), int(cout3), rows, int(cout3), _stream(x))
    _lib.check(rc, "ffb6d_mlp_chain3_pm")
    return out


def lfa_half(mode, xyz, neigh_idx, f, w1, b1, act1, wfc, wm, bm, actm, w2=None, b2=None, act2=ACT_NONE, out=None, p_hint=0):
    """One half of Building_block.forward (RandLANet.py:196-214) in one launch (csrc/lfa_pm.hip): neighbour gather +
    relative_pos_encoding + mlp1 (+ mlp2 for mode 2) + Att_pooling (fc, softmax over the 16 neighbours, weighted sum, mlp).
    xyz [B,N,3] float32 or a coordinate table [B,N,4] (xyz_table; may be the first N rows of every frame of a finer level's
    table: stride(0) is honoured), neigh_idx [B,N,16], f [B,N,d/2] rows (float32 / bfloat16); w1 [d/2, >=10] / b1 float32
    (BatchNorm folded), w2 [d/2,d/2], wfc [d,d], wm [cout,d] of f's dtype, b2 / bm float32 -> [B,N,cout], cout = d/2 (mode 1)
    or d (mode 2).  The per-pair tensors of the reference ([B,N,16,10], [B,N,16,d/2], [B,N,16,d]) exist only in LDS."""
    _need_gpu(xyz, neigh_idx, f, wfc, wm)
    lib = _lib.load()
    idx, bits = _idx(neigh_idx)
    B, N, K = idx.shape
    x = xyz if xyz.shape[-1] == 4 else xyz_table(xyz)
    if x.dtype != torch.float32 or x.shape[:2] != (B, N) or x.stride(2) != 1 or x.stride(1) != 4 or x.data_ptr() % 16 or \
            (B > 1 and x.stride(0) < 4 * N) or x.stride(0) % 4:
        x = x.contiguous()
    xfs = x.stride(0) // 4 if B > 1 else N
    h = f.shape[-1]
    d = 2 * h
    cout = h if mode == 1 else d
    dt = _dt(f, wfc, wm, w2, out)
    if w1.dtype != torch.float32 or w1.dim() != 2 or w1.shape[0] != h or w1.shape[1] < 10 or w1.stride(1) != 1:
        raise ValueError("lfa_half: w1 must be float32 [d/2, >=10] with contiguous rows")
    if wm.dim() == 2:
        wm = k_chunked(wm)
    vl = 16 // f.element_size()
    for name, t, shape in (("wfc", wfc, (d, d)), ("wm", wm, (d // vl, cout, vl)), ("w2", w2, (h, h))):
        if t is not None and (tuple(t.shape) != shape or not t.is_contiguous()):
            raise ValueError(f"lfa_half: {name} must be contiguous {shape}, got {tuple(t.shape)}")
    if (mode == 2) != (w2 is not None) or any(b is not None and b.dtype != torch.float32 for b in (b1, b2, bm)):
        raise ValueError("lfa_half: mode 2 needs w2 / b2; biases stay float32")
    f2, ldf = rows_view(f.detach())
    if out is None:
        out = torch.empty((B, N, cout), dtype=f.dtype, device=f.device)
    esz = f.element_size()
    nbytes = 16 * B * N + (bits // 8) * B * N * K + esz * B * N * (h + cout) + esz * (d * d + cout * d + (h * h if mode == 2 else 0)) + 40 * h
    flops = 2 * K * B * N * (d * d + 10 * h + (h * h if mode == 2 else 0)) + 2 * B * N * d * cout
    # SURVEY section 8d "reference-equivalent" bytes: what the unfused reference operators this launch replaces move at their own
    # boundaries -- relative_pos_encoding (half 1), gather_neighbour, the attentive pooling pass (feature set + activation in)
    ib = bits // 8
    ref_bytes = (12 * B * N + ib * B * N * K + 40 * B * N * K if mode == 1 else 0) + (esz * B * N * h + ib * B * N * K + esz * B * N * K * h) \
        + (2 * esz * B * N * K * d + esz * B * N * d)
    with torch.cuda.device(f.device), _lib.traced("lfa_pm", nbytes, (mode, d, N, dt, flops, ref_bytes)):
        rc = lib.ffb6d_lfa_pm(dt, int(mode), x.data_ptr(), xfs, idx.data_ptr(), bits, f2.data_ptr(), ldf, w1.data_ptr(), w1.stride(0),
                              b1.data_ptr(), int(act1), w2.data_ptr() if w2 is not None else None,
                              b2.data_ptr() if b2 is not None else None, int(act2), wfc.data_ptr(), wm.data_ptr(), bm.data_ptr(),
                              int(actm), out.data_ptr(), out.stride(-2), B, N, K, d, int(p_hint), _stream(f))
    _lib.check(rc, "ffb6d_lfa_pm")
    return out


def affine_act_(x, scale, shift, act=ACT_NONE, slope=0.0, residual=None, res_affine=None):
    """In-place per-channel affine + optional (affine) residual + activation on [..., C] rows (the eval-mode
    BatchNorm / ReLU / PReLU / residual glue of the colour branch, extractors.py:49-63, pspnet.py:34-45)."""
    _need_gpu(x)
    lib = _lib.load()
    if not x.is_contiguous():
        raise ValueError("affine_act_ needs a contiguous tensor")
    dt = _dt(x, residual)
    C = x.shape[-1]
    rows = x.numel() // C
    r = rs = rb = None
    if residual is not None:
        r = residual if residual.is_contiguous() else residual.contiguous()
        if r.shape != x.shape:
            raise ValueError("residual shape mismatch")
        if res_affine is not None:
            rs, rb = res_affine
    nbytes = x.element_size() * x.numel() * (3 if r is not None else 2)
    with torch.cuda.device(x.device), _lib.traced("affine_act_pm", nbytes, (C, rows)):
        rc = lib.ffb6d_affine_act_pm(dt, x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                         r.data_ptr() if r is not None else None,
                                         rs.data_ptr() if rs is not None else None,
                                         rb.data_ptr() if rb is not None else None,
                                         x.data_ptr(), rows, C, int(act), float(slope), _stream(x))
    _lib.check(rc, "ffb6d_affine_act_pm_f32")
    return x


def affine_relu_maxpool(x, scale, shift):
    """BatchNorm (eval, folded to scale / shift) + ReLU + MaxPool2d(3, 2, 1) of the colour stem (extractors.py; ffb6d.py:222)
    in one pass: x [B,H,W,C] -> [B,(H-1)//2+1,(W-1)//2+1,C]; the normalised full-resolution map is never written."""
    _need_gpu(x)
    lib = _lib.load()
    xc = x.detach()
    xc = xc if xc.is_contiguous() else xc.contiguous()
    B, H, W, C = xc.shape
    out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
    nbytes = x.element_size() * (xc.numel() + out.numel())
    with torch.cuda.device(x.device), _lib.traced("affine_relu_maxpool_pm", nbytes, (C, H, W)):
        rc = lib.ffb6d_affine_relu_maxpool_pm(_dt(xc), xc.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), B, H, W, C,
                                              _stream(xc))
    _lib.check(rc, "ffb6d_affine_relu_maxpool_pm")
    return out


def bilinear_resize(x, size, align_corners):
    """x [B,IH,IW,C] -> [B,OH,OW,C], bilinear (pspnet.py:24-28 align_corners=False; :37-42 align_corners=True)."""
    _need_gpu(x)
    lib = _lib.load()
    xc = x.detach()
    xc = xc if xc.is_contiguous() else xc.contiguous()
    B, IH, IW, C = xc.shape
    OH, OW = int(size[0]), int(size[1])
    out = torch.empty((B, OH, OW, C), dtype=x.dtype, device=x.device)
    nbytes = x.element_size() * B * C * (IH * IW + OH * OW)
    with torch.cuda.device(x.device), _lib.traced("bilinear_resize_pm", nbytes, (C, OH, OW)):
        rc = lib.ffb6d_bilinear_resize_pm(_dt(xc), xc.data_ptr(), out.data_ptr(), B, IH, IW, OH, OW, C,
                                              1 if align_corners else 0, _stream(xc))
    _lib.check(rc, "ffb6d_bilinear_resize_pm_f32")
    return out


def upsampled_patch_rows(x, idx, size):
    """x [B,IH,IW,C], idx [B,P] (flat pixel Y*OW + X of the up-sampled map) -> [B,P,9*C]: the 3x3 patches (tap-major) of the
    align_corners bilinear up-sampling of x to `size` around the picked pixels, zeros outside the map -- PSPUpsample's convolution
    (pspnet.py:34-45) as a K = 9*C GEMM over the picked pixels only (the `choose` pick, ffb6d.py:309-312, moved in front of it)."""
    _need_gpu(x, idx)
    lib = _lib.load()
    xc = x.detach()
    xc = xc if xc.is_contiguous() else xc.contiguous()
    B, IH, IW, C = xc.shape
    if idx.dim() != 2 or idx.shape[0] != B:
        raise ValueError("idx must be [B,P]")
    P = idx.shape[1]
    ii, bits = _idx(idx.reshape(-1))
    OH, OW = int(size[0]), int(size[1])
    out = torch.empty((B, P, 9 * C), dtype=x.dtype, device=x.device)
    nbytes = x.element_size() * B * C * (IH * IW + 9 * P) + B * P * (bits // 8)
    with torch.cuda.device(x.device), _lib.traced("upsampled_patch_rows_pm", nbytes, (C, OH, OW, P)):
        rc = lib.ffb6d_upsampled_patch_rows_pm(_dt(xc), xc.data_ptr(), ii.data_ptr(), bits, out.data_ptr(), B, IH, IW, OH, OW, C, P,
                                               _stream(xc))
    _lib.check(rc, "ffb6d_upsampled_patch_rows_pm")
    return out


def upconv_combine(z, shift, slope, size):
    """Second half of the folded PSPUpsample (pspnet.py:34-45; csrc/upconv.hip): z [B,IH,IW,9*C] = per-tap channel mixing
    at the low resolution (tap-major blocks of C channels), shift [C] float32, one PReLU slope -> [B,OH,OW,C]:
    prelu(shift + sum over the 3x3 taps of the align_corners bilinear up-sampling of their plane, zero padding outside)."""
    _need_gpu(z)
    lib = _lib.load()
    zc = z.detach()
    zc = zc if zc.is_contiguous() else zc.contiguous()
    B, IH, IW, C9 = zc.shape
    if C9 % 9 or shift.dtype != torch.float32 or shift.numel() * 9 != C9 or not shift.is_contiguous():
        raise ValueError(f"z {tuple(z.shape)} must hold 9 tap blocks of the {shift.numel()} channels of a contiguous float32 shift")
    C = C9 // 9
    OH, OW = int(size[0]), int(size[1])
    out = torch.empty((B, OH, OW, C), dtype=z.dtype, device=z.device)
    nbytes = z.element_size() * (zc.numel() + out.numel())
    with torch.cuda.device(z.device), _lib.traced("upconv_combine_pm", nbytes, (C, OH, OW)):
        rc = lib.ffb6d_upconv_combine_pm(_dt(zc), zc.data_ptr(), shift.data_ptr(), float(slope), out.data_ptr(), B, IH, IW, OH, OW,
                                         C, _stream(zc))
    _lib.check(rc, "ffb6d_upconv_combine_pm")
    return out


def _int_array(values):
    import ctypes
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


def psp_pool(x, sizes):
    """All adaptive average pools of `sizes` of x [B,H,W,C] (float32 or bfloat16) -> float32 [B, sum(s*s), C]."""
    _need_gpu(x)
    lib = _lib.load()
    xc = x.detach()
    xc = xc if xc.is_contiguous() else xc.contiguous()
    B, H, W, C = xc.shape
    nb = sum(int(s) * int(s) for s in sizes)
    out = torch.empty((B, nb, C), dtype=torch.float32, device=x.device)
    sz = _int_array(sizes)
    wbytes = lib.ffb6d_psp_pool_pm_workspace_bytes(B, H, C, sz, len(sizes))
    ws = torch.empty((wbytes,), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device), _lib.traced("psp_pool_pm", xc.element_size() * xc.numel() + 4 * out.numel(), (C, H * W)):
        rc = lib.ffb6d_psp_pool_pm(_dt(xc), xc.data_ptr(), out.data_ptr(), B, H, W, C, sz, len(sizes), ws.data_ptr(), wbytes,
                                       _stream(xc))
    _lib.check(rc, "ffb6d_psp_pool_pm_f32")
    return out


def psp_prior_sum(z, sizes, size, dtype=torch.float32):
    """float32 z [B, sum(s*s), M] -> [B,H,W,M] of `dtype` = sum over levels of the bilinear (align_corners=False)
    up-sampling to (H,W)."""
    _need_gpu(z)
    lib = _lib.load()
    zc = z.detach().float()
    zc = zc if zc.is_contiguous() else zc.contiguous()
    B, _, M = zc.shape
    H, W = int(size[0]), int(size[1])
    out = torch.empty((B, H, W, M), dtype=dtype, device=z.device)
    with torch.cuda.device(z.device), _lib.traced("psp_prior_sum_pm", 4 * zc.numel() + out.element_size() * out.numel(), (M, H * W)):
        rc = lib.ffb6d_psp_prior_sum_pm(_dt(out), zc.data_ptr(), out.data_ptr(), B, H, W, M, _int_array(sizes), len(sizes), _stream(zc))
    _lib.check(rc, "ffb6d_psp_prior_sum_pm_f32")
    return out
