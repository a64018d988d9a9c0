"""Torch-facing operators with the reference's own call signatures, each one a thin,
autograd-capable wrapper around one C-ABI entry point of libffb6d_amd.so.

    random_sample(feature, pool_idx)            == FFB6D.random_sample          ffb6d.py:159-177
    nearest_interpolation(feature, interp_idx)  == FFB6D.nearest_interpolation  ffb6d.py:179-194
    gather_neighbour(pc, neighbor_idx)          == Building_block.gather_neighbour  RandLANet.py:225-234
    relative_pos_encoding(xyz, neigh_idx)       == Building_block.relative_pos_encoding RandLANet.py:216-223
    att_pool(feature_set, att_activation)       == the softmax/mul/sum of Att_pooling.forward RandLANet.py:245-248
    choose_gather(rgb_emb, choose)              == the final per-point pixel pick ffb6d.py:309-312
    upsample_align / prelu                      == PSPUpsample's up-sampling and PReLU with hand-written backward (pspnet.py:34-45)
    bn_fold(bn)                                 == eval-mode BatchNorm as (scale, shift), for the fused inference path

Tensors: float32, contiguous (made so), on a ROCm device; indices int64 or int32.
There is no CPU path: a CPU tensor raises FFB6DNativeError."""
import torch

from . import _lib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.FFB6DNativeError(
                "ffb6d_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _f32(t):
    """float32 and contiguous.  Under torch.autocast (mixed-precision training: the convolutions hand over bfloat16 / float16
    activations) the neighbour operators keep their fp32 kernels: reduced-precision floats are cast up (differentiably);
    anything else is a caller error."""
    if t.dtype in (torch.bfloat16, torch.float16) and torch.is_autocast_enabled():
        t = t.float()
    if t.dtype != torch.float32:
        raise TypeError(f"float32 expected, got {t.dtype}")
    return t.contiguous()


def _idx(t):
    if t.dtype == torch.int64:
        return t.contiguous(), 64
    if t.dtype == torch.int32:
        return t.contiguous(), 32
    raise TypeError(f"index tensor must be int64 or int32, got {t.dtype}")


class _RandomSample(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)      # under autocast: the fp32 kernels, fp32 operands
    def forward(ctx, feature, pool_idx):
        # feature [B,C,M], pool_idx [B,Np,K] -> [B,C,Np]
        lib = _lib.load()
        B, C, M = feature.shape
        Np, K = pool_idx.shape[1], pool_idx.shape[2]
        idx, bits = _idx(pool_idx)
        out = torch.empty((B, C, Np), dtype=torch.float32, device=feature.device)
        need_arg = feature.requires_grad
        arg = torch.empty((B, C, Np), dtype=torch.int32, device=feature.device) if need_arg else None
        nbytes = 4 * B * C * M + (bits // 8) * B * Np * K + 4 * B * C * Np
        with torch.cuda.device(feature.device), _lib.traced("random_sample", nbytes, (C, M, Np)):
            rc = lib.ffb6d_random_sample_f32(feature.data_ptr(), idx.data_ptr(), bits, out.data_ptr(),
                                             arg.data_ptr() if need_arg else None,
                                             B, C, M, Np, K, _stream(feature))
        _lib.check(rc, "ffb6d_random_sample_f32")
        if need_arg:
            ctx.save_for_backward(arg)
            ctx.shape = (B, C, M, Np)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        lib = _lib.load()
        (arg,) = ctx.saved_tensors
        B, C, M, Np = ctx.shape
        g = _f32(grad_out)
        grad_feat = torch.empty((B, C, M), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            rc = lib.ffb6d_random_sample_bwd_f32(g.data_ptr(), arg.data_ptr(), grad_feat.data_ptr(),
                                                 B, C, M, Np, _stream(g))
        _lib.check(rc, "ffb6d_random_sample_bwd_f32")
        return grad_feat, None


def random_sample(feature, pool_idx):
    """
    :param feature: [B, C, M] or [B, C, M, 1] input features
    :param pool_idx: [B, N', K] neighbour indices into M
    :return: [B, C, N', 1]  max over the K gathered columns (ffb6d.py:159-177)
    """
    if feature.dim() > 3:
        feature = feature.squeeze(dim=3)
    _need_gpu(feature, pool_idx)
    if feature.dim() != 3 or pool_idx.dim() != 3 or feature.shape[0] != pool_idx.shape[0]:
        raise ValueError(f"bad shapes {tuple(feature.shape)} / {tuple(pool_idx.shape)}")
    return _RandomSample.apply(_f32(feature), pool_idx).unsqueeze(3)


class _NearestInterp(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)      # under autocast: the fp32 kernels, fp32 operands
    def forward(ctx, feature, interp_idx):
        # feature [B,C,M], interp_idx [B,U] -> [B,C,U]
        lib = _lib.load()
        B, C, M = feature.shape
        U = interp_idx.shape[1]
        idx, bits = _idx(interp_idx)
        out = torch.empty((B, C, U), dtype=torch.float32, device=feature.device)
        nbytes = 4 * B * C * M + (bits // 8) * B * U + 4 * B * C * U
        with torch.cuda.device(feature.device), _lib.traced("nearest_interpolation", nbytes, (C, M, U)):
            rc = lib.ffb6d_nearest_interpolation_f32(feature.data_ptr(), idx.data_ptr(), bits,
                                                     out.data_ptr(), B, C, M, U, _stream(feature))
        _lib.check(rc, "ffb6d_nearest_interpolation_f32")
        if feature.requires_grad:
            ctx.save_for_backward(idx)
            ctx.meta = (B, C, M, U, bits)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        B, C, M, U, bits = ctx.meta
        g = _f32(grad_out)
        grad_feat = torch.empty((B, C, M), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            rc = lib.ffb6d_nearest_interpolation_bwd_f32(g.data_ptr(), idx.data_ptr(), bits,
                                                         grad_feat.data_ptr(), B, C, M, U, _stream(g))
        _lib.check(rc, "ffb6d_nearest_interpolation_bwd_f32")
        return grad_feat, None


def nearest_interpolation(feature, interp_idx):
    """
    :param feature: [B, C, M, 1] (or [B, C, M]) input features
    :param interp_idx: [B, U, 1] nearest neighbour index
    :return: [B, C, U, 1] (ffb6d.py:179-194)
    """
    if feature.dim() > 3:
        feature = feature.squeeze(dim=3)
    _need_gpu(feature, interp_idx)
    B = interp_idx.shape[0]
    U = interp_idx.shape[1]
    if feature.dim() != 3 or feature.shape[0] != B:
        raise ValueError(f"bad shapes {tuple(feature.shape)} / {tuple(interp_idx.shape)}")
    return _NearestInterp.apply(_f32(feature), interp_idx.reshape(B, U)).unsqueeze(3)


def choose_gather(rgb_emb, choose):
    """rgb_emb [B,C,H*W] (or [B,C,H,W]), choose [B,1,N] -> [B,C,N]: the per-point pixel
    feature pick of ffb6d.py:309-312 (same gather as nearest_interpolation)."""
    B, C = rgb_emb.shape[:2]
    return nearest_interpolation(rgb_emb.reshape(B, C, -1), choose.reshape(B, -1, 1)).squeeze(3)


class _GatherNeighbour(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)      # under autocast: the fp32 kernels, fp32 operands
    def forward(ctx, pc, neighbor_idx):
        lib = _lib.load()
        B, M, C = pc.shape
        N, K = neighbor_idx.shape[1], neighbor_idx.shape[2]
        idx, bits = _idx(neighbor_idx)
        out = torch.empty((B, N, K, C), dtype=torch.float32, device=pc.device)
        nbytes = 4 * B * M * C + (bits // 8) * B * N * K + 4 * B * N * K * C
        with torch.cuda.device(pc.device), _lib.traced("gather_neighbour", nbytes, (M, C)):
            rc = lib.ffb6d_gather_neighbour_f32(pc.data_ptr(), idx.data_ptr(), bits, out.data_ptr(),
                                                B, M, C, N, K, _stream(pc))
        _lib.check(rc, "ffb6d_gather_neighbour_f32")
        if pc.requires_grad:
            ctx.save_for_backward(idx)
            ctx.meta = (B, M, C, N, K, bits)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        B, M, C, N, K, bits = ctx.meta
        g = _f32(grad_out)
        grad_pc = torch.empty((B, M, C), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            rc = lib.ffb6d_gather_neighbour_bwd_f32(g.data_ptr(), idx.data_ptr(), bits,
                                                    grad_pc.data_ptr(), B, M, C, N, K, _stream(g))
        _lib.check(rc, "ffb6d_gather_neighbour_bwd_f32")
        return grad_pc, None


def gather_neighbour(pc, neighbor_idx):
    """pc: batch*npoint*channel, neighbor_idx: batch*npoint*nsamples
    -> batch*npoint*nsamples*channel (RandLANet.py:225-234)."""
    _need_gpu(pc, neighbor_idx)
    if pc.dim() != 3 or neighbor_idx.dim() != 3 or pc.shape[0] != neighbor_idx.shape[0]:
        raise ValueError(f"bad shapes {tuple(pc.shape)} / {tuple(neighbor_idx.shape)}")
    return _GatherNeighbour.apply(_f32(pc), neighbor_idx)


def relative_pos_encoding(xyz, neigh_idx):
    """xyz [B,N,3], neigh_idx [B,N,K] -> [B,N,K,10] = [dis, p-q, p, q] (RandLANet.py:216-223).
    xyz is input data (no gradient flows to it in the reference model either)."""
    _need_gpu(xyz, neigh_idx)
    if xyz.dim() != 3 or xyz.shape[2] != 3 or neigh_idx.dim() != 3:
        raise ValueError(f"bad shapes {tuple(xyz.shape)} / {tuple(neigh_idx.shape)}")
    lib = _lib.load()
    xyz_c = _f32(xyz.detach())
    idx, bits = _idx(neigh_idx)
    B, N, _ = xyz_c.shape
    K = idx.shape[2]
    out = torch.empty((B, N, K, 10), dtype=torch.float32, device=xyz.device)
    nbytes = 12 * B * N + (bits // 8) * B * N * K + 40 * B * N * K
    with torch.cuda.device(xyz.device), _lib.traced("relative_pos_encoding", nbytes, (N,)):
        rc = lib.ffb6d_relative_pos_encoding_f32(xyz_c.data_ptr(), idx.data_ptr(), bits, out.data_ptr(),
                                                 B, N, K, _stream(xyz_c))
    _lib.check(rc, "ffb6d_relative_pos_encoding_f32")
    return out


def relative_pos_encoding_cm(xyz, neigh_idx):
    """Channel-major position encoding [B,10,N,K] (= relative_pos_encoding(...).permute(0,3,1,2),
    RandLANet.py:197-198) written directly in that layout."""
    _need_gpu(xyz, neigh_idx)
    lib = _lib.load()
    xyz_c = _f32(xyz.detach())
    idx, bits = _idx(neigh_idx)
    B, N, _ = xyz_c.shape
    K = idx.shape[2]
    out = torch.empty((B, 10, N, K), dtype=torch.float32, device=xyz.device)
    nbytes = 12 * B * N + (bits // 8) * B * N * K + 40 * B * N * K
    with torch.cuda.device(xyz.device), _lib.traced("relative_pos_encoding", nbytes, (N,)):
        rc = lib.ffb6d_relative_pos_encoding_cm_f32(xyz_c.data_ptr(), idx.data_ptr(), bits, out.data_ptr(),
                                                    B, N, K, _stream(xyz_c))
    _lib.check(rc, "ffb6d_relative_pos_encoding_cm_f32")
    return out


class _AttPool(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)      # under autocast: the fp32 kernels, fp32 operands
    def forward(ctx, feature_set, att_activation):
        lib = _lib.load()
        B, C, N, K = feature_set.shape
        out = torch.empty((B, C, N), dtype=torch.float32, device=feature_set.device)
        nbytes = 2 * 4 * B * C * N * K + 4 * B * C * N
        with torch.cuda.device(feature_set.device), _lib.traced("att_pool", nbytes, (C, N)):
            rc = lib.ffb6d_att_pool_f32(feature_set.data_ptr(), att_activation.data_ptr(),
                                        out.data_ptr(), B, C, N, K, _stream(feature_set))
        _lib.check(rc, "ffb6d_att_pool_f32")
        if feature_set.requires_grad or att_activation.requires_grad:
            ctx.save_for_backward(feature_set, att_activation)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        lib = _lib.load()
        feat, act = ctx.saved_tensors
        B, C, N, K = feat.shape
        g = _f32(grad_out)
        gf = torch.empty_like(feat)
        ga = torch.empty_like(act)
        with torch.cuda.device(g.device):
            rc = lib.ffb6d_att_pool_bwd_f32(g.data_ptr(), feat.data_ptr(), act.data_ptr(),
                                            gf.data_ptr(), ga.data_ptr(), B, C, N, K, _stream(g))
        _lib.check(rc, "ffb6d_att_pool_bwd_f32")
        return gf, ga


def att_pool(feature_set, att_activation):
    """feature_set, att_activation [B,C,N,K] -> [B,C,N,1]:
    sum_K(feature_set * softmax_K(att_activation))  (RandLANet.py:245-248)."""
    _need_gpu(feature_set, att_activation)
    if feature_set.shape != att_activation.shape or feature_set.dim() != 4:
        raise ValueError(f"bad shapes {tuple(feature_set.shape)} / {tuple(att_activation.shape)}")
    return _AttPool.apply(_f32(feature_set), _f32(att_activation)).unsqueeze(3)


ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


# ----------------------------------------------------------------------------------------------------------------------
# training-side replacements for the two slowest non-convolution backward passes of the colour decoder (PSPUpsample,
# pspnet.py:34-45; csrc/train_ops.hip).  Forward results are torch's own; what changes is the backward.
# ----------------------------------------------------------------------------------------------------------------------
def _rows_dt(t):
    return {torch.float32: 0, torch.bfloat16: 1}.get(t.dtype)


# the forward of upsample_align on channels_last maps: the inference path's row kernel (True) or ATen's interpolate (False: A/B, and the
# operator autocast would run in float32); bench.py records it
UPSAMPLE_ROWS = True


class _UpsampleAlign(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) whose backward is a gather (ffb6d_bilinear_bwd_pm) instead
    of ATen's atomic scatter."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, size):
        ctx.in_shape = tuple(x.shape)
        # autocast lists the up-sampling kernels as float32 operators (a bf16 map would be cast up, interpolated and handed to the
        # next convolution's cast down: three passes over the largest maps of the decoder); here it stays in the map's own dtype.
        # On pixel-major rows (channels_last maps: what the convolutions of the training step read and write) the forward is the
        # inference path's row kernel -- ATen's source index and blend order (bilinear_pm_kernel), 2.8-3.8 TB/s where ATen's NHWC
        # bf16 kernel moves the decoder's largest map at 0.5 TB/s (646 us per call, profiles/r04_rocprofv3_kernel_stats_train_bf16.txt)
        B, C, IH, IW = x.shape
        vl = 8 if x.dtype == torch.bfloat16 else 4
        if UPSAMPLE_ROWS and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and C % vl == 0 and \
                x.is_contiguous(memory_format=torch.channels_last):
            from . import ops_pm
            rows = ops_pm.bilinear_resize(x.permute(0, 2, 3, 1), size, True)       # [B,OH,OW,C]
            return rows.permute(0, 3, 1, 2)                                         # channels_last [B,C,OH,OW]
        with torch.autocast("cuda", enabled=False):
            return torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=True)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        B, C, IH, IW = ctx.in_shape
        OH, OW = g.shape[2], g.shape[3]
        dt = _rows_dt(g)
        vl = 8 if dt == 1 else 4
        if g.is_cuda and dt is not None and C % vl == 0 and IH > 1 and IW > 1 and IH <= OH <= 4 * IH - 3 and IW <= OW <= 4 * IW - 3 \
                and B * IH < 65536:
            g = g.contiguous(memory_format=torch.channels_last)              # pixel-major rows [B,OH,OW,C]
            gin = torch.empty((B, C, IH, IW), dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
            nbytes = g.element_size() * (g.numel() + gin.numel())
            with torch.cuda.device(g.device), _lib.traced("bilinear_bwd_pm", nbytes, (C, OH, OW)):
                rc = _lib.load().ffb6d_bilinear_bwd_pm(dt, g.data_ptr(), gin.data_ptr(), B, IH, IW, OH, OW, C, _stream(g))
            _lib.check(rc, "ffb6d_bilinear_bwd_pm")
            return gin, None
        return torch.ops.aten.upsample_bilinear2d_backward(g, [OH, OW], list(ctx.in_shape), True, None, None), None


def upsample_align(x, size):
    """bilinear, align_corners=True (pspnet.py:37-42) with the gather backward; x [B,C,IH,IW]."""
    return _UpsampleAlign.apply(x, (int(size[0]), int(size[1])))


class _PReLU(torch.autograd.Function):
    """Single-slope PReLU on a dense tensor of any memory format: the slope's gradient is reduced inside the backward kernel
    (ATen materialises a gradient tensor as large as the map and reduces it afterwards)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight):
        y = torch.empty_like(x)                                            # same strides as x
        with torch.cuda.device(x.device), _lib.traced("prelu_fwd", 2 * x.element_size() * x.numel(), (x.numel(),)):
            rc = _lib.load().ffb6d_prelu_fwd(_rows_dt(x), x.data_ptr(), weight.data_ptr(), y.data_ptr(), x.numel(), _stream(x))
        _lib.check(rc, "ffb6d_prelu_fwd")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        if g.dtype != x.dtype or g.stride() != x.stride():
            g = torch.empty_like(x).copy_(g)
        gx = torch.empty_like(x)
        ga = torch.empty(1, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device), _lib.traced("prelu_bwd", 3 * x.element_size() * x.numel(), (x.numel(),)):
            rc = _lib.load().ffb6d_prelu_bwd(_rows_dt(x), x.data_ptr(), g.data_ptr(), weight.data_ptr(), gx.data_ptr(), ga.data_ptr(),
                                             x.numel(), _stream(x))
        _lib.check(rc, "ffb6d_prelu_bwd")
        return gx, ga.to(weight.dtype).reshape(weight.shape)


def prelu(x, weight):
    """torch.nn.functional.prelu(x, weight) for a single float32 slope, with the in-kernel slope gradient; anything the
    kernel does not cover (per-channel slopes, other dtypes, non-dense or unaligned tensors) goes to torch."""
    dt = _rows_dt(x)
    dense = x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))
    if not (x.is_cuda and dt is not None and weight.numel() == 1 and weight.dtype == torch.float32 and dense
            and x.numel() % (8 if dt else 4) == 0 and x.data_ptr() % 16 == 0):
        return torch.nn.functional.prelu(x, weight.to(x.dtype) if weight.dtype != x.dtype else weight)
    return _PReLU.apply(x, weight)


def bn_fold(bn):
    """(scale, shift) of an eval-mode BatchNorm: y = scale*x + shift.  Cached on the module and
    recomputed whenever one of its tensors was modified in place (load_state_dict, training)."""
    ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.device)
    cache = getattr(bn, "_ffb6d_fold", None)
    if cache is None or cache[0] != ver:
        with torch.no_grad():
            scale = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
            shift = (bn.bias - bn.running_mean * scale).contiguous()
        cache = (ver, scale, shift)
        bn._ffb6d_fold = cache
    return cache[1], cache[2]


def check_index_range(idx, M):
    """Number of entries of `idx` outside [0, M) (debug aid; the kernels do not bounds-check)."""
    _need_gpu(idx)
    lib = _lib.load()
    i, bits = _idx(idx)
    bad = torch.zeros((1,), dtype=torch.int32, device=idx.device)
    with torch.cuda.device(idx.device):
        rc = lib.ffb6d_check_index_range(i.data_ptr(), bits, i.numel(), int(M), bad.data_ptr(), _stream(i))
    _lib.check(rc, "ffb6d_check_index_range")
    return int(bad.item())
