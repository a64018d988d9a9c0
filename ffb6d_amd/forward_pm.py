"""Point-major / pixel-major ("channels last") inference forward of FFB6D on the gfx950 kernels.

Same module tree, parameters and dataflow as `model.FFB6D` / the reference (ffb6d/models/ffb6d.py:203-337,
RandLANet.py:170-250, cnn/pspnet.py, cnn/extractors.py); what changes is the memory layout of every activation:
one row of C contiguous floats per point or pixel ([B,N,C], [B,H,W,C]) instead of the reference's [B,C,N,1] / [B,C,H,W].

Why (MI355X): every gather of the hot path -- neighbour features of the local feature aggregation, the 16-pixel max
pooling of the pixel->point fusion, the 1-NN interpolation of the point->pixel fusion and of the decoder, `choose` --
moves whole contiguous rows instead of one 4-byte element per 64-byte sector; both operands of every shared-MLP GEMM
are K-contiguous and stream straight into MFMA operand registers (csrc/mlp_pm.hip, no LDS, no barriers); MIOpen's
fastest fp32 convolutions on this chip are its NHWC implicit-GEMM kernels, which then run without the NCHW<->NHWC
transposes MIOpen otherwise inserts around them.

Fusions (reference ops -> here), all inference only:
    conv1x1 + BN + act                                    one GEMM, BN folded into W/b            (ops_pm.mlp)
    cat(a, b) -> conv                                     two K ranges of one GEMM
    conv(cat(a, interp(b)))                               W_a a + gather(W_b b) in the epilogue   (p2r fusion, decoder)
    leaky(mlp2(f) + shortcut(x))                          one GEMM over K = [f | x]
    gather_neighbour + cat + fc + softmax + mul + sum     ops_pm.att_pool: gather = operand load, softmax in-lane
    choose gather + cat + head conv                       operand gather of the head GEMM
    final conv1x1 + LogSoftmax                            log-softmax epilogue of the GEMM
    pyramid pooling                                       W_x x + b + sum_i up_i((W_b,i W_i) pool_i(x))
Dense 3x3 / 7x7 convolutions stay on MIOpen, fed channels_last tensors and weights.
"""
import torch
import torch.nn.functional as F

from . import _lib, ops, ops_pm, pyramid


def cached(mod, name, sources, build):
    """Per-module cache of inference-time derived tensors (folded / split / padded weights), keyed on the version
    counter, storage and device of every source tensor: load_state_dict, optimizer steps, .to(device) and in-place
    edits all invalidate it.  (Edits through `.data` bypass version counters; FFB6D.train() drops every cache.)"""
    key = tuple((t._version, t.data_ptr(), t.device) for t in sources)
    store = mod.__dict__.setdefault("_pm_cache", {})
    hit = store.get(name)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            hit = (key, build())
        store[name] = hit
    return hit[1]


def _pad_k(w, k_to):
    if w.shape[1] == k_to:
        return w.contiguous()
    out = w.new_zeros(w.shape[0], k_to)
    out[:, :w.shape[1]] = w
    return out


def mlp_sources(m):
    src = [m.conv.weight]
    if m.has_bn:
        bn = m._bn_module()
        src += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
    else:
        src.append(m.conv.bias)
    return src


F32 = torch.float32


def folded(m, pad_k=None, dt=F32):
    """(W [Cout, Cin(padded)] of dtype `dt`, b [Cout] float32) of a model.SharedMLP: conv weight layout, eval-mode
    BatchNorm absorbed (folded in fp32, rounded once when dt is bfloat16)."""
    def build():
        w = m.conv.weight.detach().reshape(m.conv.weight.shape[0], -1)
        if m.has_bn:
            bn = m._bn_module()
            scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
            w = w * scale[:, None]
            b = bn.bias.detach() - bn.running_mean * scale
        else:
            b = m.conv.bias.detach()
        return _pad_k(w, pad_k or w.shape[1]).to(dt), b.float().contiguous()
    return cached(m, "folded%s%s" % (pad_k or "", dt), mlp_sources(m), build)


def stacked(owner, ms, dt=F32):
    """folded() of several SharedMLPs over the same input, stacked along the output channels: (W [sum Cout, Cin], b [sum Cout])"""
    def build():
        parts = [folded(m, dt=dt) for m in ms]
        return torch.cat([w for w, _ in parts]).contiguous(), torch.cat([b for _, b in parts]).contiguous()
    return cached(owner, "stacked%s%s" % ("_".join(str(id(m)) for m in ms), dt), [t for m in ms for t in mlp_sources(m)], build)


def out_padded(m, cout_to, dt=F32):
    """folded(m) with zero rows appended to the weight (and zeros to the bias) up to cout_to output channels"""
    def build():
        w, b = folded(m, dt=dt)
        wp, bp = w.new_zeros(cout_to, w.shape[1]), b.new_zeros(cout_to)
        wp[:w.shape[0]], bp[:b.shape[0]] = w, b
        return wp, bp
    return cached(m, "outpad%d%s" % (cout_to, dt), mlp_sources(m), build)


def kc_padded(m, cout_to, dt=F32, perm=False):
    """out_padded(m, cout_to) with the weight k-chunked (ops_pm.k_chunked): the last layer of csrc/mlp_chain.hip"""
    def build():
        w, b = out_padded(m, cout_to, dt)
        return ops_pm.k_chunked(w, perm), b
    return cached(m, "kcpad%d%s%d" % (cout_to, dt, perm), mlp_sources(m), build)


def split(m, k1, dt=F32):
    """(W_a [Cout,k1], W_b [Cout,Cin-k1], b) for conv(cat(a, gather(b))) == W_a a + gather(W_b b)."""
    def build():
        w, b = folded(m, dt=dt)
        return w[:, :k1].contiguous(), w[:, k1:].contiguous(), b
    return cached(m, "split%d%s" % (k1, dt), mlp_sources(m), build)


def mlp(m, x1, x2=None, pad_k=None, **kw):
    w, b = folded(m, pad_k, x1.dtype)
    return ops_pm.mlp(x1, w, b, m.act_code, x2=x2, **kw)


def fc_weight(att, dt=F32):
    w = att.fc.weight
    return cached(att, "fc%s" % dt, [w], lambda: w.detach().reshape(w.shape[0], -1).to(dt).contiguous())


# ----------------------------------------------------------------------------------------------------
# point branch
# ----------------------------------------------------------------------------------------------------
# Module attributes, not environment switches: the measured forms are the defaults; tests and A/B scripts flip an attribute to
# reach the chain a fused kernel replaced (profiles/r02_opt_in_forms_ab.json, profiles/r03_lfa_levels.txt).
# relative_pos_encoding + lfa.mlp1 as one vector-ALU pass (csrc/posenc.hip) instead of the padded encoding tensor + a K = 16 GEMM
POSENC_FUSED = True
# BatchNorm + ReLU + MaxPool2d(3,2,1) of the colour stem as one kernel (False: affine_act + torch's max pooling)
STEM_FUSED = True
# One launch per half of the local feature aggregation (csrc/lfa_pm.hip): the per-pair tensors live in LDS only.
# False: the round-2 chain posenc_mlp -> att_pool -> mlp (six launches, [B,N,16,d/2] through HBM).
LFA_FUSED = True
# The first layers of the three prediction heads (ffb6d.py:316-318: same input rows, same shape) as one GEMM over stacked weights.
HEADS_SHARE_FIRST = True
# ... and their last layers (22 / 24 / 3 channels) with the output rows padded to whole 16 bytes: the stream form instead of a 128-wide tile
HEADS_ALIGN_LAST = True
# ... and the keypoint head on the side stream while the other two run on main (only with two_streams)
HEADS_ON_BOTH_STREAMS = True
# ... and the three layers after the first of each head (128 -> 128 -> 128 -> c) as one launch with the hidden activations in
# registers (csrc/mlp_chain.hip; fp32, and bf16 since round 5)
HEADS_CHAIN_FUSED = True
# The long-row fp32 GEMMs (p2r fusion, PSP bottleneck, z GEMMs, the heads' stacked first layer) in the tile-sequence form
# (csrc/mlp_pm.hip: mlp_pm_seq_kernel -- a finished tile's epilogue rides between the MFMAs of the next tile of the same workgroup);
# False: the LDS-tiled form (one tile per workgroup) for those launches.  Equal bits.
GEMM_SEQ_FORM = True
# The long-row bf16 GEMMs with K >= 256 on the 256 x 256 tile with LDS-DMA operand loads (csrc/mlp_pm_big.hip); False: the LDS-tiled
# 128 x 128 form for those launches.  Equal bits.
GEMM_BIG_FORM = True
# Schedule of the tile-sequence form: balanced contiguous sequences per XCD (round 6; csrc/mlp_pm.hip: LIN) instead of whole-point-tile groups
# (round 5).  GEMM_SEQ_LIN_ONE: one sequence per workgroup slot instead of at least two.  Equal bits.  Measured and left OFF: alone on the
# chip the seven launches of a step take 3337 / 3295 us against 3361 (+0.7 / +2 %: profiles/r06_seq_lin_probe.txt -- the dispatcher
# back-fills the round-5 groups well enough that the quantisation the tile counts suggest, 6 tile times against 5, does not materialise),
# and inside the three-stream step both are 0.07 ms SLOWER (profiles/r06_seq_lin_in_step_ab.json: 19.82 -> 19.89 ms; longer resident
# sequences get in the way of the side streams' kernels, as the persistent forms of round 4 did).
GEMM_SEQ_LIN = False
GEMM_SEQ_LIN_ONE = False
LFA_WIDTHS = (32, 64, 128, 256)


def folded_kc(m, dt, perm=False):
    """folded(m) with the weight in the k-chunked layout of the fused LFA kernel's output MLP (ops_pm.k_chunked)"""
    def build():
        w, b = folded(m, dt=dt)
        return ops_pm.k_chunked(w, perm), b
    return cached(m, "kc%s%d" % (dt, perm), mlp_sources(m), build)


def building_block(bb, xyz, f_pc, nei):
    """RandLANet.py:196-214 (Building_block.forward): f_pc [B,N,d/2] -> [B,N,d]."""
    dt = f_pc.dtype
    if LFA_FUSED and 2 * f_pc.shape[-1] in LFA_WIDTHS and nei.shape[-1] == 16:
        w1, b1 = folded(bb.mlp1)                                                   # fp32 [d/2, 10] in both precisions
        a1, a2 = bb.att_pooling_1, bb.att_pooling_2
        wm1, bm1 = folded_kc(a1.mlp, dt)
        f_agg = ops_pm.lfa_half(1, xyz, nei, f_pc, w1, b1, bb.mlp1.act_code, fc_weight(a1, dt), wm1, bm1, a1.mlp.act_code)
        w2, b2 = folded(bb.mlp2, dt=dt)
        wm2, bm2 = folded_kc(a2.mlp, dt)
        return ops_pm.lfa_half(2, xyz, nei, f_agg, w1, b1, bb.mlp1.act_code, fc_weight(a2, dt), wm2, bm2, a2.mlp.act_code,
                               w2=w2, b2=b2, act2=bb.mlp2.act_code)
    if xyz.shape[-1] == 4:
        xyz = xyz[..., :3].contiguous()
    if POSENC_FUSED:                                                               # encoding generated in registers
        w, b = folded(bb.mlp1)                                                     # fp32 [d/2, 10] in both precisions
        f_xyz = ops_pm.posenc_mlp(xyz, nei, w, b, bb.mlp1.act_code, dtype=dt)      # [B,N,16,d/2]
    else:
        enc = ops_pm.relative_pos_encoding(xyz, nei, dtype=dt)                    # [B,N,16,16] (10 used)
        f_xyz = mlp(bb.mlp1, enc, pad_k=16)                                        # [B,N,16,d/2]
    pooled = ops_pm.att_pool(f_pc, nei, f_xyz, fc_weight(bb.att_pooling_1, dt))    # [B,N,d]
    f_agg = mlp(bb.att_pooling_1.mlp, pooled)                                      # [B,N,d/2]
    f_xyz = mlp(bb.mlp2, f_xyz)
    pooled = ops_pm.att_pool(f_agg, nei, f_xyz, fc_weight(bb.att_pooling_2, dt))
    return mlp(bb.att_pooling_2.mlp, pooled)                                  # [B,N,d]


def dilated_res_block(rb, feature, xyz, nei):
    """RandLANet.py:179-184: leaky(mlp2(lfa(mlp1(f))) + shortcut(f)) with the sum as ONE GEMM over K = [lfa | f].
    `feature` may carry zero-padded channels (the 8-channel stem output is stored 16 wide): weights are padded to match."""
    kin, dt = feature.shape[-1], feature.dtype
    f = building_block(rb.lfa, xyz, mlp(rb.mlp1, feature, pad_k=kin), nei)

    def build():
        (w2, b2), (ws, bs) = folded(rb.mlp2), folded(rb.shortcut, pad_k=kin)
        return torch.cat([w2, ws], dim=1).to(dt).contiguous(), (b2 + bs).contiguous()
    w, b = cached(rb, "res%d%s" % (kin, dt), mlp_sources(rb.mlp2) + mlp_sources(rb.shortcut), build)
    return ops_pm.mlp(f, w, b, ops.ACT_LEAKY, x2=feature)


def decode(stage, skip, p_emb, interp_idx):
    """conv(cat(skip, interp(p))) (ffb6d.py:273-279,302-307) = W_a skip + gather(W_b p)."""
    wa, wb, bias = split(stage, skip.shape[-1], skip.dtype)
    y = ops_pm.mlp(p_emb, wb)
    return ops_pm.mlp(skip, wa, bias, stage.act_code, gather=(y, interp_idx.reshape(interp_idx.shape[0], -1)))


# ----------------------------------------------------------------------------------------------------
# colour branch (tensors kept as [B,H,W,C]; MIOpen sees them as channels_last NCHW views)
# ----------------------------------------------------------------------------------------------------
def conv(x, c):
    """Dense convolution of a [B,H,W,C] map through MIOpen's NHWC path; returns [B,H',W',C']."""
    w = cached(c, "cl%s" % x.dtype, [c.weight],
               lambda: c.weight.detach().to(x.dtype).contiguous(memory_format=torch.channels_last))
    y = F.conv2d(x.permute(0, 3, 1, 2), w, None, c.stride, c.padding, c.dilation, c.groups)
    y = y.permute(0, 2, 3, 1)
    return y if y.is_contiguous() else y.contiguous()


def res_block(rb, x):
    """extractors.py:49-63 (BasicBlock): BN+ReLU and BN+residual(+BN of the projection)+ReLU as one pass each."""
    y = ops_pm.affine_act_(conv(x, rb.conv1), *ops.bn_fold(rb.bn1), act=ops.ACT_RELU)
    y = conv(y, rb.conv2)
    if rb.downsample is not None:
        return ops_pm.affine_act_(y, *ops.bn_fold(rb.bn2), act=ops.ACT_RELU, residual=conv(x, rb.downsample[0]),
                                  res_affine=ops.bn_fold(rb.downsample[1]))
    return ops_pm.affine_act_(y, *ops.bn_fold(rb.bn2), act=ops.ACT_RELU, residual=x)


def pyramid_pooling(pp, x):
    """pspnet.py:7-31 as W_x x + b + sum_i up_i((W_b,i W_i) pool_i(x)): no 2560-channel cat, bottleneck K = 512."""
    B, h, w_, C = x.shape
    sizes = [st[0].output_size[0] for st in pp.stages]
    bw = pp.bottleneck.weight

    def build():
        ch = pp.stages[0][1].weight.shape[0]
        wb = bw.detach().reshape(bw.shape[0], -1)                                            # [1024, 2560]
        prods = [(wb[:, i * ch:(i + 1) * ch] @ st[1].weight.detach().reshape(ch, ch)).contiguous()
                 for i, st in enumerate(pp.stages)]                                           # each [1024, 512]
        return prods, wb[:, len(pp.stages) * ch:].contiguous()
    prods, wx = cached(pp, "fold", [bw] + [st[1].weight for st in pp.stages], build)
    pooled = ops_pm.psp_pool(x, sizes)                                                        # [B,50,512] fp32
    zs, off = [], 0
    for s, wl in zip(sizes, prods):                                                           # 50 columns: fp32 in both precisions
        zs.append(ops_pm.mlp(pooled[:, off:off + s * s].contiguous(), wl, role="cnn"))
        off += s * s
    prior = ops_pm.psp_prior_sum(torch.cat(zs, dim=1), sizes, (h, w_), dtype=x.dtype)         # [B,h,w,1024]
    wx = cached(pp, "wx%s" % x.dtype, [bw], lambda: wx.to(x.dtype))
    return ops_pm.mlp(x, wx, pp.bottleneck.bias.detach().float(), ops.ACT_RELU, add=prior, role="cnn")


# Which PSPUpsample blocks run in the folded form (csrc/upconv.hip): "auto" = every block in fp32 (measured, profiles/r02_opt_in_forms_ab.json:
# step 28.8 -> 23.8 ms) and, since round 6, in bf16 (UPCONV_FOLD_BF16; profiles/r06_upconv_fold_bf16_ab_start.json: configuration 5
# 11.19 -> 10.16 ms in one process -- the round-2 record that kept it off, 17.2 -> 18.1 ms, was three GEMM generations old); None = every
# block in both precisions; a frozenset of input widths = those blocks (tests / A/B set the attribute).
UPCONV_FOLD = "auto"
# ... and what "auto" means in bf16 (a boolean so that scripts/ab_forms.py / bench.py --form can flip it)
UPCONV_FOLD_BF16 = True


def _fold_block(cin, dtype):
    if UPCONV_FOLD == "auto":
        return dtype == torch.float32 or UPCONV_FOLD_BF16
    return UPCONV_FOLD is None or cin in UPCONV_FOLD


def upconv_folded(ub, dt=F32):
    """PSPUpsample's convolution regrouped for the low resolution: (W9 [9*cout, cin] of dtype `dt` with rows
    (ky*3+kx)*cout + co = BatchNorm scale[co] * conv.weight[co, :, ky, kx], shift [cout] float32 = BatchNorm shift +
    scale * conv bias, PReLU slope).  Pure torch: also what the host simulation of the kernel is checked with on CPU."""
    cv, bn, prelu = ub.conv[1], ub.conv[2], ub.conv[3]
    if prelu.weight.numel() != 1:
        raise NotImplementedError("per-channel PReLU in PSPUpsample")

    def build():
        scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
        shift = bn.bias.detach() - bn.running_mean * scale
        if cv.bias is not None:
            shift = shift + scale * cv.bias.detach()
        w = cv.weight.detach() * scale[:, None, None, None]                                  # [cout, cin, 3, 3]
        w9 = w.permute(2, 3, 0, 1).reshape(9 * w.shape[0], w.shape[1])
        return w9.to(dt).contiguous(), shift.float().contiguous(), float(prelu.weight.detach().item())
    src = [cv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, prelu.weight] + ([cv.bias] if cv.bias is not None else [])
    return cached(ub, "fold9%s" % dt, src, build)


def up_block(ub, x):
    """pspnet.py:34-45 (PSPUpsample): bilinear x2 (align_corners) -> conv3x3 -> [BN + PReLU in one pass].
    Folded form (UPCONV_FOLD): up-sampling and channel mixing commute, so the nine taps of the convolution are mixed at
    the LOW resolution by one GEMM with 9*cout output channels (a quarter of the convolution's flops, on csrc/mlp_pm.hip)
    and `upconv_combine` blends the tap planes, adds the BatchNorm shift and applies the PReLU in one pass."""
    B, h, w_, cin = x.shape
    cv, bn, prelu = ub.conv[1], ub.conv[2], ub.conv[3]
    if _fold_block(cin, x.dtype) and cv.kernel_size == (3, 3) and cv.padding == (1, 1) \
            and cv.stride == (1, 1) and cv.dilation == (1, 1) and cv.groups == 1:
        w9, shift, slope = upconv_folded(ub, x.dtype)
        z = ops_pm.mlp(x, w9, role="cnn")                                                     # [B,h,w,9*cout]
        return ops_pm.upconv_combine(z, shift, slope, (2 * h, 2 * w_))
    y = ops_pm.bilinear_resize(x, (2 * h, 2 * w_), align_corners=True)
    y = conv(y, cv)
    if prelu.weight.numel() != 1:
        raise NotImplementedError("per-channel PReLU in PSPUpsample")
    slope = cached(ub, "slope", [prelu.weight], lambda: float(prelu.weight.detach().item()))
    scale, shift = ops.bn_fold(bn)
    # conv bias rides in the BatchNorm shift: BN(conv(y)+b) = scale*conv(y) + (shift + scale*b)
    shift = cached(ub, "shift", [cv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var],
                   lambda: (shift + scale * cv.bias.detach()).contiguous())
    return ops_pm.affine_act_(y, scale, shift, act=ops.ACT_LEAKY, slope=slope)


def final_head(fh, x):
    """pspnet.py:108-112 `final`: Conv2d(64,64,1) + LogSoftmax(dim=1) as one GEMM with a log-softmax epilogue."""
    cv = fh[0]
    w = cached(fh, "w%s" % x.dtype, [cv.weight], lambda: cv.weight.detach().reshape(cv.out_channels, -1).to(x.dtype).contiguous())
    return ops_pm.mlp(x, w, cv.bias.detach().float(), ops_pm.ACT_LOG_SOFTMAX, role="cnn")


# The last colour stage (PSPUpsample 64 -> 64 to full resolution, then `final`) is read by the heads only through the `choose` pick
# (ffb6d.py:302-312): evaluate it at the picked pixels -- 3x3 patches of the up-sampled map as operand rows of a K = 9*cin GEMM, then
# BatchNorm + PReLU and `final` on [B,n,64] rows -- instead of on all 480 x 640 pixels of which 4 % are read.  Same arithmetic per
# picked pixel (the patch elements ARE the up-sampled map's; the convolution's sum runs over the same 9*cin products), no full map.
LAST_STAGE_AT_CHOSEN = True


def last_stage_at_chosen(stage, x, choose):
    """cnn_up_stages[-1] = Sequential(UpBlock, FinalHead) on x [B,h,w,cin] -> the rows [B,n,C] the heads' `choose` pick would read
    from its [B,2h,2w,C] output; None when the stage is not of that form (the caller runs it densely)."""
    from . import model
    mods = [m for m in stage if not isinstance(m, torch.nn.Dropout2d)] if isinstance(stage, torch.nn.Sequential) else []
    if len(mods) != 2 or not isinstance(mods[0], model.UpBlock) or not isinstance(mods[1], model.FinalHead):
        return None
    ub, fh = mods
    cv, bn, prelu = ub.conv[1], ub.conv[2], ub.conv[3]
    if not (cv.kernel_size == (3, 3) and cv.padding == (1, 1) and cv.stride == (1, 1) and cv.dilation == (1, 1) and cv.groups == 1) \
            or prelu.weight.numel() != 1:
        return None
    B, h, w_, cin = x.shape
    wk = cached(ub, "taps%s" % x.dtype, [cv.weight],
                lambda: cv.weight.detach().permute(0, 2, 3, 1).reshape(cv.out_channels, 9 * cin).to(x.dtype).contiguous())
    patches = ops_pm.upsampled_patch_rows(x, choose, (2 * h, 2 * w_))                        # [B,n,9*cin]
    y = ops_pm.mlp(patches, wk, role="cnn")
    slope = cached(ub, "slope", [prelu.weight], lambda: float(prelu.weight.detach().item()))
    scale, shift = ops.bn_fold(bn)
    if cv.bias is not None:
        shift = cached(ub, "shift", [cv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var],
                       lambda: (shift + scale * cv.bias.detach()).contiguous())
    y = ops_pm.affine_act_(y, scale, shift, act=ops.ACT_LEAKY, slope=slope)
    return final_head(fh, y)


def cnn_stage(stage, x):
    """Run one entry of cnn_ds_stages / cnn_up_stages on a [B,H,W,C] map."""
    from . import model
    for m in (stage if isinstance(stage, torch.nn.Sequential) else [stage]):
        if isinstance(m, model.ResBlock):
            x = res_block(m, x)
        elif isinstance(m, model.PyramidPooling):
            x = pyramid_pooling(m, x)
        elif isinstance(m, model.UpBlock):
            x = up_block(m, x)
        elif isinstance(m, model.FinalHead):
            x = final_head(m, x)
        elif isinstance(m, torch.nn.Dropout2d):
            pass                                                  # eval mode
        elif isinstance(m, torch.nn.Sequential):
            x = cnn_stage(m, x)
        else:
            raise NotImplementedError(type(m).__name__)
    return x


def supported(net):
    """Can the row kernels run this module tree?  Every shared-MLP width must be a legal row: output channels a multiple of 8
    (or <= 64: padded), input channels (the K of the GEMM) a multiple of 8 in fp32 / 16 in bf16 -- except the two stems, whose
    9- and 8-channel rows are stored 16 wide.  Every width of the reference's configuration is; anything else falls back to
    the channel-major path instead of failing inside a kernel.  Memoised per precision (dropped with the weight caches)."""
    prec = getattr(net, "precision", "fp32")
    memo = net.__dict__.setdefault("_pm_supported", {})
    if prec not in memo:
        kmul = 16 if prec == "bf16" else 8
        ok = True
        for name, m in net.named_modules():
            if hasattr(m, "conv") and hasattr(m, "act_code"):
                cout, cin = m.conv.weight.shape[0], m.conv.weight.shape[1]
                stem = name == "rndla_pre_stages" or name.endswith("rndla_ds_stages.0.mlp1") or name.endswith("rndla_ds_stages.0.shortcut") \
                    or name.endswith("lfa.mlp1")
                ok = ok and (cout % 8 == 0 or cout <= 64) and (cin % kmul == 0 or stem)
        memo[prec] = ok
    return memo[prec]


# ----------------------------------------------------------------------------------------------------
# whole forward (ffb6d.py:203-337)
# ----------------------------------------------------------------------------------------------------
PYRAMID_KEYS = ([k % i for i in range(4) for k in ('cld_xyz%d', 'cld_nei_idx%d', 'cld_sub_idx%d', 'cld_interp_idx%d',
                                                   'r2p_ds_nei_idx%d', 'p2r_ds_nei_idx%d')]
                + [k % i for i in range(3) for k in ('r2p_up_nei_idx%d', 'p2r_up_nei_idx%d')])


# A/B: True = every consumer waits for the whole pyramid (the schedule of rounds 3-5: the point branch starts when all 22 searches are done)
PYRAMID_ONE_EVENT = False


class StreamedPyramid(dict):
    """The input dict of a forward whose index pyramid (linemod_dataset.py:299-353) is not built yet: it is built on a third
    (high-priority) HIP stream under the colour stem.  The builder launches the 22 searches as two batches (pyramid.PyramidBuilder):
    the K = 16 searches, whose indices the point branch's first layers need -- `level(i, stream)` / `up_level(i, stream)` make
    `stream` wait for them -- then the K = 1 searches, first read by the first fusion stage (`nearest(stream)`).  With ONE batch
    (rounds 3-5) the point branch started 1.5 ms into the step and the colour stream waited 0.28 ms for it at the first fusion
    (profiles/r05_step_timeline.txt)."""

    def __init__(self, net, inputs, main, side):
        super().__init__(inputs)
        dev = inputs['rgb'].device
        self.events = []
        idx = net._index_stream(dev) if side is not main else main
        idx.wait_stream(main)
        with torch.cuda.stream(idx):
            # the cloud as rows (linemod_dataset.py:285,318), every coarser level (a prefix of it, :322-323), the image grids and the
            # coordinate table of the fused local feature aggregation (16-byte rows; ONE table serves all four levels through its
            # frame stride) in one launch
            sets, self.table0 = pyramid.point_sets(inputs['cld_rgb_nrm'].float(), inputs['dpt_xyz'].float(), channel_major=True,
                                                   with_table=True)
            if idx is not main:
                self.table0.record_stream(side)
            b = pyramid.PyramidBuilder(sets[('c', 0)], inputs['dpt_xyz'], getattr(net, 'index_dtype', torch.int64), sets=sets)
            # first event: the K = 16 indices and level 0's sub-sampling prefix (what the first encoder level of the point branch
            # reads); second: the other prefixes (small copies, kept off the first level's critical chain) and the K = 1 indices
            for keys in (lambda: b.neighbour_keys(sub_levels=(0,)), lambda: {**b.sub_index_keys((1, 2, 3)), **b.nearest_keys()}):
                d = keys()
                ev = torch.cuda.Event()
                ev.record(idx)
                self.events.append(ev)
                for t in d.values():
                    if idx is not main:
                        t.record_stream(main)
                        t.record_stream(side)
                self.update(d)
        self.waited = set()
        self.single = idx is main

    def _wait(self, which, stream):
        if PYRAMID_ONE_EVENT:
            which = 1
        if not self.single and (which, stream) not in self.waited:
            stream.wait_event(self.events[which])
            self.waited.add((which, stream))

    def level(self, i, stream):
        """neighbour indices (K = 16) of every level and level 0's sub-sampling prefix: the first batch of searches; the prefixes of
        the coarser levels come with the second event"""
        self._wait(0, stream)
        if i > 0:
            self._wait(1, stream)

    def up_level(self, i, stream):
        self._wait(0, stream)

    def nearest(self, stream):
        """the K = 1 indices (interpolation, pixel <- point): the second batch"""
        self._wait(1, stream)

    def xyz_table(self, i):
        """coordinate table of encoder level i: the first N_i rows of every frame of the level-0 table (valid after level(i))"""
        return self.table0[:, :self['cld_xyz%d' % i].shape[1]]


def forward(net, inputs, end_points, two_streams=True, taps=None):
    """taps: optional dict that receives the two embeddings after every fusion stage (`rgb_emb_ds{i}`, `p_emb_ds{i}`,
    `rgb_emb_up{i}`, `p_emb_up{i}`), converted to the reference layout -- diagnostics / stage-level parity tests."""
    dev = inputs['rgb'].device
    dt = torch.bfloat16 if getattr(net, "precision", "fp32") == "bf16" else torch.float32
    ops_pm.MLP_SEQ_FORM = GEMM_SEQ_FORM
    ops_pm.MLP_BIG_FORM = GEMM_BIG_FORM
    ops_pm.MLP_SEQ_LIN = (2 if GEMM_SEQ_LIN_ONE else 1) if GEMM_SEQ_LIN else 0
    main = torch.cuda.current_stream(dev)
    side = net._side_stream(dev) if two_streams else main
    if two_streams:
        side.wait_stream(main)                      # inputs (and the index pyramid) come from `main`

    lazy = 'cld_nei_idx0' not in inputs             # no index pyramid in the dict: build it here, level by level
    if lazy:
        inputs = StreamedPyramid(net, inputs, main, side)

    def need(i, up=False):
        """indices of encoder / decoder level i are about to be used on the side stream (main follows through handover)"""
        if lazy:
            (inputs.up_level if up else inputs.level)(i, side)
            if not two_streams:
                return
            (inputs.up_level if up else inputs.level)(i, main)

    def need_nearest():
        """the K = 1 indices are about to be used"""
        if lazy:
            inputs.nearest(side)
            if two_streams:
                inputs.nearest(main)

    def on_side():
        return torch.cuda.stream(side)

    probe = getattr(net, "_stall_probe", None)      # diagnostics: list that receives (tag, event, event) around every wait

    def handover(t, producer, consumer, tag=None):
        """tensor produced on `producer`, about to be read on `consumer`"""
        if producer is not consumer:
            ev = torch.cuda.Event()
            ev.record(producer)
            if probe is not None:
                a = torch.cuda.Event(enable_timing=True)
                a.record(consumer)
            consumer.wait_event(ev)
            if probe is not None:
                b = torch.cuda.Event(enable_timing=True)
                b.record(consumer)
                probe.append((tag, a, b))
            t.record_stream(consumer)
        return t

    def fuse(i, pre_p2r, fuse_p2r, pre_r2p, fuse_r2p, rgb0, p0, p2r_idx, r2p_idx):
        """One bidirectional fusion step (ffb6d.py:245-263 / 281-298); both directions read the pre-fusion tensors."""
        B, h, w_, c = rgb0.shape
        st = ("ds%d" if pre_p2r is net.ds_fuse_p2r_pre_layers else "up%d") % i
        handover(p0, side, main, "main waits for point stage " + st)
        handover(rgb0, main, side, "side waits for colour stage " + st)
        if two_streams:
            p2r_idx.record_stream(main)
        # p2r on main: conv(cat(rgb0, interp(e))) = W_a rgb0 + gather(W_b e)
        e = mlp(pre_p2r[i], p0)
        wa, wb, bias = split(fuse_p2r[i], c, dt)
        y = ops_pm.mlp(e, wb)
        rgb = ops_pm.mlp(rgb0, wa, bias, fuse_p2r[i].act_code, gather=(y, p2r_idx.reshape(B, -1)))
        with on_side():     # r2p: max over the 16 nearest pixels, then conv(cat(p0, pre(.))) as a two-source GEMM
            r2p = mlp(pre_r2p[i], ops_pm.random_sample(rgb0.view(B, h * w_, c), r2p_idx))
            p = mlp(fuse_r2p[i], p0, x2=r2p)
        return rgb, p

    # ---- stems ----
    rgb = inputs['rgb'].to(dt).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)   # [B,H,W,3] view
    pool = net.cnn_pre_stages[3]
    if STEM_FUSED and (pool.kernel_size, pool.stride, pool.padding, pool.dilation, pool.ceil_mode) == (3, 2, 1, 1, False):
        # BatchNorm + ReLU + max pooling in one pass: the normalised full-resolution map is never written
        rgb_emb = ops_pm.affine_relu_maxpool(conv(rgb, net.cnn_pre_stages[0]), *ops.bn_fold(net.cnn_pre_stages[1]))
    else:
        y = ops_pm.affine_act_(conv(rgb, net.cnn_pre_stages[0]), *ops.bn_fold(net.cnn_pre_stages[1]), act=ops.ACT_RELU)
        rgb_emb = pool(y.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)                                 # max pool, stays NHWC
    with on_side():
        raw = inputs['cld_rgb_nrm']                                                              # [B,9,N]
        x0 = torch.zeros(raw.shape[0], raw.shape[2], (raw.shape[1] + 15) // 16 * 16, dtype=dt, device=dev)
        x0[..., :raw.shape[1]] = raw.transpose(1, 2)
        # the 8-channel stem output is stored 16 wide (zero padding): a legal K for both precisions
        c_pre = net.rndla_pre_stages.conv.weight.shape[0]
        p_emb = torch.zeros(raw.shape[0], raw.shape[2], (c_pre + 15) // 16 * 16, dtype=dt, device=dev)
        mlp(net.rndla_pre_stages, x0, pad_k=x0.shape[-1], out=p_emb[..., :c_pre])

    # ---- encoder ----
    ds_emb = []
    for i in range(4):
        rgb0 = cnn_stage(net.cnn_ds_stages[i], rgb_emb)
        need(i)
        with on_side():
            xyz = inputs['cld_xyz%d' % i]
            if LFA_FUSED:
                xyz = inputs.xyz_table(i) if lazy else ops_pm.xyz_table(xyz)
            f_enc = dilated_res_block(net.rndla_ds_stages[i], p_emb, xyz, inputs['cld_nei_idx%d' % i])
            p0 = ops_pm.random_sample(f_enc, inputs['cld_sub_idx%d' % i])
        if i == 0:
            ds_emb.append(f_enc)
        need_nearest()
        rgb_emb, p_emb = fuse(i, net.ds_fuse_p2r_pre_layers, net.ds_fuse_p2r_fuse_layers, net.ds_fuse_r2p_pre_layers,
                              net.ds_fuse_r2p_fuse_layers, rgb0, p0, inputs['p2r_ds_nei_idx%d' % i],
                              inputs['r2p_ds_nei_idx%d' % i])
        ds_emb.append(p_emb)
        if taps is not None:
            torch.cuda.synchronize(dev)
            taps['rgb_emb_ds%d' % i], taps['p_emb_ds%d' % i] = rgb_emb.permute(0, 3, 1, 2).float(), p_emb.transpose(1, 2).unsqueeze(3).float()

    # ---- decoder ----
    n_up = len(net.rndla_up_stages)
    for i in range(n_up - 1):
        rgb0 = cnn_stage(net.cnn_up_stages[i], rgb_emb)
        need(i, up=True)
        need_nearest()
        with on_side():
            p0 = decode(net.rndla_up_stages[i], ds_emb[-i - 2], p_emb, inputs['cld_interp_idx%d' % (n_up - i - 1)])
        rgb_emb, p_emb = fuse(i, net.up_fuse_p2r_pre_layers, net.up_fuse_p2r_fuse_layers, net.up_fuse_r2p_pre_layers,
                              net.up_fuse_r2p_fuse_layers, rgb0, p0, inputs['p2r_up_nei_idx%d' % i],
                              inputs['r2p_up_nei_idx%d' % i])
        if taps is not None:
            torch.cuda.synchronize(dev)
            taps['rgb_emb_up%d' % i], taps['p_emb_up%d' % i] = rgb_emb.permute(0, 3, 1, 2).float(), p_emb.transpose(1, 2).unsqueeze(3).float()
    B = rgb_emb.shape[0]
    choose = inputs['choose'].reshape(B, -1)
    img = last_stage_at_chosen(net.cnn_up_stages[n_up - 1], rgb_emb, choose) if LAST_STAGE_AT_CHOSEN else None
    if img is not None:
        choose = None                                                               # img [B,n,c]: already the picked rows
    else:
        rgb_emb = cnn_stage(net.cnn_up_stages[n_up - 1], rgb_emb)
        img = rgb_emb.view(B, -1, rgb_emb.shape[-1])
    with on_side():
        p_emb = decode(net.rndla_up_stages[n_up - 1], ds_emb[0], p_emb, inputs['cld_interp_idx0'])
    handover(p_emb, side, main, "main waits for the last decoder")                 # also the final join: main is behind all side work

    # ---- heads: conv(cat(rgb[choose], p_emb)) with the pick as the operand gather of the first GEMM ----

    seqs = [net.rgbd_seg_layer, net.kp_ofst_layer, net.ctr_ofst_layer]
    firsts = [seq[0] for seq in seqs]
    if HEADS_SHARE_FIRST and len({(m.conv.weight.shape, m.act_code) for m in firsts}) == 1:
        # the three first layers read the same rows (ffb6d.py:316-318): one GEMM over the stacked weights -- each output channel's
        # dot product is the one the separate launch computes -- and the heads continue on channel slices of its output
        w, b = stacked(net, firsts, img.dtype)
        y0 = ops_pm.mlp(img, w, b, firsts[0].act_code, x2=p_emb, x1_gather=choose)
        step = firsts[0].conv.weight.shape[0]
        starts = [y0[..., h * step:(h + 1) * step] for h in range(3)]
    else:
        starts = [mlp(m, img, x2=p_emb, x1_gather=choose) for m in firsts]

    def head_plan(seq, y):
        """the derived weights of a head's remaining layers (folded / k-chunked / padded: cached on the modules) -- built HERE, on the
        main stream, so that cached tensors always belong to main's allocator pool even when the head itself runs on the side stream"""
        layers = list(seq)[1:]
        if HEADS_CHAIN_FUSED and y.dtype in (torch.float32, torch.bfloat16) and len(layers) == 3 and y.shape[-1] == 128 and \
                [tuple(m.conv.weight.shape[:2]) for m in layers[:2]] == [(128, 128)] * 2 and layers[2].conv.weight.shape[1] == 128 \
                and layers[2].conv.weight.shape[0] <= 32 and all(m.act_code in (0, 1, 2) for m in layers):
            # the three remaining layers as one launch, hidden activations in registers (csrc/mlp_chain.hip)
            cout = layers[2].conv.weight.shape[0]
            perm = y.dtype == torch.bfloat16        # bf16: layers 2 and 3 take k in the order the accumulators supply it (csrc/mlp_chain.hip)
            parts = [folded_kc(layers[0], y.dtype) + (layers[0].act_code,), folded_kc(layers[1], y.dtype, perm) + (layers[1].act_code,),
                     kc_padded(layers[2], 32, y.dtype, perm) + (layers[2].act_code,)]
            return ("chain", layers, parts, cout)
        last, cout = layers[-1], layers[-1].conv.weight.shape[0]
        mult = 16 // y.element_size()
        plain = [(m, folded(m, None, y.dtype)) for m in layers[:-1]]
        if HEADS_ALIGN_LAST and cout % mult:
            # 22 / 3 output channels: zero weight rows up to whole 16-byte output rows, which the stream form needs
            # (ffb6d_mlp_pm_choice); the extra channels are never read
            return ("layers", plain, last, out_padded(last, -(-cout // mult) * mult, y.dtype), cout)
        return ("layers", plain, last, folded(last, None, y.dtype), cout)

    def head(plan, y):
        if plan[0] == "chain":
            _, layers, parts, cout = plan
            try:
                return ops_pm.mlp_chain3(y, parts[0], parts[1], parts[2], -(-cout // 4) * 4)[..., :cout]
            except _lib.FFB6DNativeError as e:           # e.g. a device that refuses 145 KB of dynamic LDS: the three launches below
                if "hipFuncSetAttribute" not in str(e):
                    raise
            for layer in layers:
                y = mlp(layer, y)
            return y
        _, plain, last, (w, b), cout = plan
        for layer, (wl, bl) in plain:
            y = ops_pm.mlp(y, wl, bl, layer.act_code)
        return ops_pm.mlp(y, w, b, last.act_code)[..., :cout]

    plans = [head_plan(seq, y) for seq, y in zip(seqs, starts)]
    n = p_emb.shape[1]
    # the three chains are independent and each leaves CUs idle (768 workgroups of short-K GEMMs): the keypoint head runs on the
    # side stream, idle since the last decoder, under the other two
    kp_on = side if HEADS_ON_BOTH_STREAMS else main
    handover(starts[1], main, kp_on, "side waits for the heads' first layer")
    with torch.cuda.stream(kp_on):
        kp = head(plans[1], starts[1]).float().reshape(B, n, net.n_kps, 3).permute(0, 2, 1, 3).contiguous()
    end_points['pred_rgbd_segs'] = head(plans[0], starts[0]).float().transpose(1, 2).contiguous()           # [B,n_cls,N]
    ctr = head(plans[2], starts[2]).float().reshape(B, n, 1, 3).permute(0, 2, 1, 3).contiguous()
    end_points['pred_kp_ofs'] = handover(kp, kp_on, main, "main waits for the keypoint head")
    end_points['pred_ctr_ofs'] = ctr
    return end_points
