"""FFB6D network with the reference's parameter names (checkpoints load unchanged) whose
point branch and pixel<->point fusion run on the gfx950 kernels of this package.

Reference: ffb6d/models/ffb6d.py:16-337 (FFB6D), ffb6d/models/RandLA/RandLANet.py:170-250
(Dilated_res_block / Building_block / Att_pooling), the two conv+BN wrappers
(ffb6d/models/pytorch_utils.py:75-129, ffb6d/models/RandLA/pytorch_utils.py:35-111) and the
ResNet34-PSPNet colour branch (ffb6d/models/cnn/extractors.py, pspnet.py).  The module
tree reproduces the reference's `state_dict()` keys and shapes exactly
(tests/golden/state_dict_keys.json); `FFB6D.forward(inputs)` takes the same input dict and
returns the same `end_points`.

Two ways through this module tree:
  * `eval()` + `torch.no_grad()` on a GPU: the fused point-major inference path (ffb6d_amd/forward_pm.py) -- activations as
    rows, BatchNorm folded into the GEMMs, one launch per half of the local feature aggregation, three HIP streams;
  * everything else (training, gradients, train() under no_grad, widths the row kernels do not cover): the stock modules below
    -- conv -> BatchNorm -> activation exactly as upstream, so gradients and running statistics behave like the reference --
    with every gather / pooling / encoding step as one HIP kernel call with an autograd body instead of index.repeat +
    torch.gather + permute().contiguous() chains.  Those operators work on channels-last rows in the activation dtype
    (ffb6d_amd.ops_cl), i.e. on what MIOpen's NHWC convolutions read and write: with the model in channels_last memory format
    and torch.autocast(bfloat16) no transposing copy and no cast sits between a convolution and a neighbour operator.
The dense 3x3/7x7 convolutions of the colour branch stay on MIOpen (out of scope for hand-written kernels, SURVEY.md section 2
row 8).  (Rounds 1-2 also carried a channel-major fused inference path, `layout="cm"`; it was removed in round 3.)
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import forward_pm, ops, ops_cl, pyramid

D_OUT = (32, 64, 128, 256)   # ConfigRandLA.d_out (ffb6d/common.py:26)
IN_C = 9                     # ConfigRandLA.in_c
DS_RGB_OC = (64, 128, 512, 1024)
UP_RGB_OC = (256, 64, 64)


# --------------------------------------------------------------------------------------
# shared MLP = 1x1 conv (+BN) (+activation), two naming flavours
# --------------------------------------------------------------------------------------
class _BN(nn.Sequential):
    def __init__(self, ch, dims, eps, momentum):
        super().__init__()
        cls = nn.BatchNorm1d if dims == 1 else nn.BatchNorm2d
        self.add_module("bn", cls(ch, eps=eps, momentum=momentum))


class SharedMLP(nn.Module):
    """`flavour='randla'`: children conv / bn.bn, BN(eps=1e-6, momentum=0.99), LeakyReLU(0.2)
       (RandLA/pytorch_utils.py:35-111);
       `flavour='pvn'`: children conv / normlayer.bn, default BN, ReLU (models/pytorch_utils.py:75-129).
    The conv has a bias only without BN, as upstream."""

    def __init__(self, cin, cout, dims=2, bn=True, act=True, flavour="randla"):
        super().__init__()
        conv_cls = nn.Conv1d if dims == 1 else nn.Conv2d
        ks = 1 if dims == 1 else (1, 1)
        self.conv = conv_cls(cin, cout, ks, bias=not bn)
        nn.init.kaiming_normal_(self.conv.weight)
        if not bn:
            nn.init.constant_(self.conv.bias, 0)
        self.flavour, self.has_bn, self.act = flavour, bn, act
        if bn:
            if flavour == "randla":
                self.bn = _BN(cout, dims, 1e-6, 0.99)
            else:
                self.normlayer = _BN(cout, dims, 1e-5, 0.1)

    def _bn_module(self):
        return (self.bn if self.flavour == "randla" else self.normlayer).bn

    def activation(self, y):
        if not self.act:
            return y
        return F.leaky_relu_(y, 0.2) if self.flavour == "randla" else F.relu_(y)

    @property
    def act_code(self):
        return ops.ACT_NONE if not self.act else (ops.ACT_LEAKY if self.flavour == "randla" else ops.ACT_RELU)

    def forward(self, x, pad_k=0):  # conv -> BN -> act as separate modules (the fused inference path folds them: forward_pm.folded)
        if pad_k:                   # input rows carry pad_k zero channels behind the layer's own (ops_cl.relative_pos_encoding)
            y = F.conv2d(x, F.pad(self.conv.weight, (0, 0, 0, 0, 0, pad_k)), self.conv.bias)
        else:
            y = self.conv(x)
        if self.has_bn:
            y = (self.bn if self.flavour == "randla" else self.normlayer)(y)
        return self.activation(y)


def _activation_dtype(x):
    """dtype the convolutions compute in: the autocast dtype when it is bfloat16, else the tensor's own (float32)"""
    if x.is_cuda and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return torch.bfloat16
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32


def _autograd_path(x, mod=None):
    """True when the stock-torch layers (conv -> BN -> activation as separate modules) must be used instead of the
    fused inference path: gradients enabled, tensors not on a GPU, or the module in train() mode -- the fused
    kernels fold BatchNorm with its RUNNING statistics and skip dropout, which is eval() semantics only; train() under
    torch.no_grad() (BN re-calibration, a validation loop that forgot eval()) must keep batch statistics, running-stat
    updates and dropout exactly like the reference.
    This is no CPU forward: the neighbour operators (ops.random_sample, gather_neighbour, ...) have no
    CPU implementation and raise FFB6DNativeError on the first CPU tensor they see
    (tests/test_model_cpu.py::test_forward_on_cpu_tensors_fails_loudly)."""
    return torch.is_grad_enabled() or not x.is_cuda or (mod is not None and mod.training)


# --------------------------------------------------------------------------------------
# RandLA-Net local feature aggregation (point branch)
# --------------------------------------------------------------------------------------
class AttPooling(nn.Module):
    """RandLANet.py:237-250."""

    def __init__(self, d_in, d_out):
        super().__init__()
        self.fc = nn.Conv2d(d_in, d_in, (1, 1), bias=False)
        self.mlp = SharedMLP(d_in, d_out)

    def forward(self, feature_set):
        att = self.fc(feature_set)
        return self.mlp(ops_cl.att_pool(feature_set, att))


class BuildingBlock(nn.Module):
    """RandLANet.py:187-214 (local spatial encoding + two attentive poolings)."""

    def __init__(self, d_out):
        super().__init__()
        self.mlp1 = SharedMLP(10, d_out // 2)
        self.att_pooling_1 = AttPooling(d_out, d_out // 2)
        self.mlp2 = SharedMLP(d_out // 2, d_out // 2)
        self.att_pooling_2 = AttPooling(d_out, d_out)

    def forward(self, xyz, feature, neigh_idx):
        # RandLANet.py:196-214, same arithmetic; every tensor is rows of channels (channels_last memory) in the activation dtype,
        # written by one kernel each -- none of the reference's permutes / contiguous copies, forward or backward (ops_cl)
        enc = ops_cl.relative_pos_encoding(xyz, neigh_idx, dtype=_activation_dtype(feature))      # [B,16,N,K]: 10 channels + 6 zeros
        f_xyz = self.mlp1(enc, pad_k=6)
        plan = ops_cl.neighbour_plan(neigh_idx)          # both gathers read through the same indices: one inverse for their backward
        f_nei = ops_cl.gather_neighbour(feature, neigh_idx, plan)
        f_agg = self.att_pooling_1(torch.cat([f_nei, f_xyz], dim=1))
        f_xyz = self.mlp2(f_xyz)
        f_nei = ops_cl.gather_neighbour(f_agg, neigh_idx, plan)
        return self.att_pooling_2(torch.cat([f_nei, f_xyz], dim=1))


class DilatedResBlock(nn.Module):
    """RandLANet.py:170-184."""

    def __init__(self, d_in, d_out):
        super().__init__()
        self.mlp1 = SharedMLP(d_in, d_out // 2)
        self.lfa = BuildingBlock(d_out)
        self.mlp2 = SharedMLP(d_out, d_out * 2, act=False)
        self.shortcut = SharedMLP(d_in, d_out * 2, act=False)

    def forward(self, feature, xyz, neigh_idx):
        f = self.lfa(xyz, self.mlp1(feature), neigh_idx)
        return F.leaky_relu(self.mlp2(f) + self.shortcut(feature), negative_slope=0.2)


# --------------------------------------------------------------------------------------
# colour branch: ResNet34 + pyramid pooling + up-sampling (dense convs -> MIOpen)
# --------------------------------------------------------------------------------------
class ResBlock(nn.Module):
    """extractors.py:34-63 (BasicBlock)."""

    def __init__(self, cin, cout, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu_(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu_(y + x)


def res_layer(cin, cout, blocks, stride):
    # extractors.py:133-156: with output_stride=32 never reached, layers 3 and 4 keep stride 1
    # and dilation 1 (the `dilation=` argument upstream is ignored)
    layers = [ResBlock(cin, cout, stride, project=(stride != 1 or cin != cout))]
    layers += [ResBlock(cout, cout, 1, False) for _ in range(blocks - 1)]
    return nn.Sequential(*layers)


def _psp_operators(h, w, sizes, device):
    """Constant operators of the pyramid pooling module on an h x w map, as matrices over the pixel rows:
         ind  [nbins, h*w]  0/1 membership of every adaptive-average-pool bin (ATen: rows floor(i*h/s) .. ceil((i+1)*h/s)),
         inv  [nbins]       1 / bin area,
         up   [h*w, nbins]  bilinear (align_corners = False) up-sampling weights from the s x s grids back to h x w -- taken from
                            F.interpolate itself on the s*s unit maps, so they are ATen's weights by construction;
       bins of sizes[0] first, row-major (the packing of ops_pm.psp_pool)."""
    key = (h, w, tuple(sizes), str(device))
    hit = _psp_operators.cache.get(key)
    if hit is not None:
        return hit
    ind, inv, up = [], [], []
    for s in sizes:
        for i in range(s):
            y0, y1 = (i * h) // s, -((-(i + 1) * h) // s)
            for j in range(s):
                x0, x1 = (j * w) // s, -((-(j + 1) * w) // s)
                m = torch.zeros(h, w)
                m[y0:y1, x0:x1] = 1.0
                ind.append(m.reshape(-1))
                inv.append(1.0 / ((y1 - y0) * (x1 - x0)))
        unit = torch.eye(s * s).view(s * s, 1, s, s)
        up.append(F.interpolate(unit, size=(h, w), mode="bilinear", align_corners=False).reshape(s * s, h * w).t())
    out = (torch.stack(ind).to(device), torch.tensor(inv, device=device), torch.cat(up, dim=1).contiguous().to(device))
    _psp_operators.cache[key] = out
    return out


_psp_operators.cache = {}


class PyramidPooling(nn.Module):
    """pspnet.py:7-31."""

    def __init__(self, ch=512, out_ch=1024, sizes=(1, 2, 3, 6)):
        super().__init__()
        self.sizes = tuple(sizes)
        self.stages = nn.ModuleList(
            nn.Sequential(nn.AdaptiveAvgPool2d((s, s)), nn.Conv2d(ch, ch, 1, bias=False)) for s in sizes)
        self.bottleneck = nn.Conv2d(ch * (len(sizes) + 1), out_ch, 1)

    fold_in_training = True      # class attribute (tests flip it): the module's linear maps folded into matrix products (forward_folded)

    def forward(self, x):
        if x.is_cuda and self.fold_in_training:
            return self.forward_folded(x)
        h, w = x.shape[2:]
        pri = [F.interpolate(st(x), size=(h, w), mode="bilinear", align_corners=False) for st in self.stages]
        return F.relu_(self.bottleneck(torch.cat(pri + [x], 1)))

    def forward_folded(self, x):
        """The same module with the linear steps reordered (training path on the GPU; autograd through plain matrix products):
        pooling, the stage convolution, the up-sampling and the bottleneck convolution are all linear, and a 1x1 convolution
        commutes with the up-sampling, so
            bottleneck(cat(up(conv_i(pool_i x)), x)) = sum_i up( pool_i(x) (Wb_i Wc_i)^T ) + x Wb_x^T + b.
        The 2560-channel concatenation and 4/5 of the bottleneck GEMM disappear; the four adaptive pools become one product with
        the 0/1 bin-membership matrix (exact in bf16; the 1/area scaling is done in fp32 on the 50 bins) and the four up-samplings
        one product with the bilinear weight matrix, accumulated onto the x term (baddbmm).  Under autocast that matrix is split
        into two bf16 terms (hi + lo: weights exact to 2^-17) so that the interpolation weights still sum to one."""
        B, C, h, w = x.shape
        ind, inv, up = _psp_operators(h, w, self.sizes, x.device)
        n_s = len(self.sizes)
        wb = self.bottleneck.weight.view(self.bottleneck.out_channels, -1)                 # [out, (n_s + 1) C]
        y = F.conv2d(x, wb[:, n_s * C:].reshape(-1, C, 1, 1), self.bottleneck.bias)        # x term (+ bias), autocast dtype
        dt = y.dtype
        yr = y.permute(0, 2, 3, 1).reshape(B, h * w, -1)                                    # pixel rows; no copy when channels-last
        xr = x.permute(0, 2, 3, 1).reshape(B, h * w, C)
        with torch.autocast(x.device.type, enabled=False):
            sums = torch.bmm(ind.to(xr.dtype).unsqueeze(0).expand(B, -1, -1), xr)           # [B, nbins, C] bin sums
            pooled = sums.float() * inv.view(1, -1, 1)
            z, off = [], 0
            for i, s in enumerate(self.sizes):
                w_eff = wb[:, i * C:(i + 1) * C].float() @ self.stages[i][1].weight.view(C, C).float()   # [out, C]
                z.append(pooled[:, off:off + s * s] @ w_eff.t())
                off += s * s
            z = torch.cat(z, dim=1)                                                         # [B, nbins, out] fp32
            if dt == torch.float32:
                out = torch.baddbmm(yr, up.unsqueeze(0).expand(B, -1, -1), z)
            else:
                hi = up.to(dt)
                lo = (up - hi.float()).to(dt)
                zt = z.to(dt)
                out = torch.baddbmm(yr, torch.cat([hi, lo], dim=1).unsqueeze(0).expand(B, -1, -1), torch.cat([zt, zt], dim=1))
        return F.relu_(out).view(B, h, w, -1).permute(0, 3, 1, 2)


class UpBlock(nn.Module):
    """pspnet.py:34-45 (PSPUpsample)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
                                  nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.PReLU())

    def forward(self, x):
        if not x.is_cuda:
            return self.conv(x)
        # same modules and arithmetic, but the up-sampling and the PReLU carry hand-written backward passes
        # (csrc/train_ops.hip: gather instead of ATen's atomic scatter; slope gradient reduced in the kernel)
        _, conv, bn, prelu = self.conv
        y = ops.upsample_align(x, (2 * x.shape[2], 2 * x.shape[3]))
        return ops.prelu(bn(conv(y)), prelu.weight)


class FinalHead(nn.Sequential):
    """pspnet.py:108-112 `final`: Conv2d(64,64,1) + LogSoftmax (implicit dim = 1 on a 4-d map); the fused inference path runs it
    as one GEMM with a log-softmax epilogue (forward_pm.final_head)."""

    # True: the LogSoftmax runs on rows in the map's own dtype (under bf16 autocast the output is rounded to bf16; every consumer is a
    # convolution that rounds its input to bf16, a row gather or a max over gathered rows, all of which commute with that rounding).
    # False: ATen's operator, which autocast runs and stores in float32 as apex amp does for the reference.  bench.py records it.
    rows_log_softmax = True

    def __init__(self, ch=64):
        super().__init__(nn.Conv2d(ch, ch, 1), nn.LogSoftmax(dim=1))

    def forward(self, x):
        y = self[0](x)
        if y.is_cuda and self.rows_log_softmax:
            # the same operator on rows, in the map's own dtype: under autocast ATen's log_softmax is an fp32 operator that writes the two
            # largest maps of the decoder as fp32 (history: profiles/r03_r04_training_step_history_notes.md; measured with the multi-lane gather backward: 61.2 -> 54.0 ms
            # per bf16 training step, profiles/r04_start_bench_train_bf16_{default,optin}.json)
            return ops_cl.channel_log_softmax(y)
        return self[1](y)


def _head(cin, cout):
    """ffb6d.py:135-157: three conv1d+BN+ReLU then a plain conv1d (pt_utils.Seq numbering)."""
    seq = nn.Sequential()
    for i in range(3):
        seq.add_module(str(i), SharedMLP(cin if i == 0 else 128, 128, dims=1, flavour="pvn"))
    seq.add_module("3", SharedMLP(128, cout, dims=1, bn=False, act=False, flavour="pvn"))
    return seq


class FFB6D(nn.Module):
    def __init__(self, n_classes, n_pts, rndla_cfg=None, n_kps=8):
        super().__init__()
        self.n_cls, self.n_pts, self.n_kps = n_classes, n_pts, n_kps
        d_out = tuple(getattr(rndla_cfg, "d_out", D_OUT))
        in_c = getattr(rndla_cfg, "in_c", IN_C)

        # ---- colour branch (names follow ffb6d.py:30-47,82-88) ----
        self.cnn_pre_stages = nn.Sequential(
            nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, 2, 1))
        self.cnn_ds_stages = nn.ModuleList([
            res_layer(64, 64, 3, 1),
            res_layer(64, 128, 4, 2),
            nn.Sequential(res_layer(128, 256, 6, 1), res_layer(256, 512, 3, 1)),
            nn.Sequential(PyramidPooling(512, 1024), nn.Dropout2d(p=0.3)),
        ])
        final = FinalHead(64)  # shared by stages 2 and 3 (ffb6d.py:86-87)
        self.cnn_up_stages = nn.ModuleList([
            nn.Sequential(UpBlock(1024, 256), nn.Dropout2d(p=0.15)),
            nn.Sequential(UpBlock(256, 64), nn.Dropout2d(p=0.15)),
            nn.Sequential(final),
            nn.Sequential(UpBlock(64, 64), final),
        ])

        # ---- point branch (RandLANet.py:16-36 pieces that FFB6D reuses) ----
        self.rndla_pre_stages = SharedMLP(in_c, 8, dims=1)
        self.rndla_ds_stages = nn.ModuleList()
        d_in = 8
        for d in d_out:
            self.rndla_ds_stages.append(DilatedResBlock(d_in, d))
            d_in = 2 * d
        ds_rndla_oc = [2 * d for d in d_out]
        self.rndla_up_stages = nn.ModuleList()
        d_cur = d_in
        for j in range(4):
            if j < 3:
                cin, d_cur = d_cur + 2 * d_out[-j - 2], 2 * d_out[-j - 2]
            else:
                cin, d_cur = 4 * d_out[-4], 2 * d_out[-4]
            self.rndla_up_stages.append(SharedMLP(cin, d_cur))
        up_rndla_oc = [ds_rndla_oc[-j - 2] if j < 3 else ds_rndla_oc[0] for j in range(4)]

        # ---- bidirectional fusion layers (ffb6d.py:51-80,100-129) ----
        def fuse_lists(rgb_oc, pt_oc, n):
            r2p_pre, r2p_fuse, p2r_pre, p2r_fuse = (nn.ModuleList() for _ in range(4))
            for i in range(n):
                r2p_pre.append(SharedMLP(rgb_oc[i], pt_oc[i], flavour="pvn"))
                r2p_fuse.append(SharedMLP(pt_oc[i] * 2, pt_oc[i], flavour="pvn"))
                p2r_pre.append(SharedMLP(pt_oc[i], rgb_oc[i], flavour="pvn"))
                p2r_fuse.append(SharedMLP(rgb_oc[i] * 2, rgb_oc[i], flavour="pvn"))
            return r2p_pre, r2p_fuse, p2r_pre, p2r_fuse

        (self.ds_fuse_r2p_pre_layers, self.ds_fuse_r2p_fuse_layers,
         self.ds_fuse_p2r_pre_layers, self.ds_fuse_p2r_fuse_layers) = fuse_lists(DS_RGB_OC, ds_rndla_oc, 4)
        (self.up_fuse_r2p_pre_layers, self.up_fuse_r2p_fuse_layers,
         self.up_fuse_p2r_pre_layers, self.up_fuse_p2r_fuse_layers) = fuse_lists(UP_RGB_OC, up_rndla_oc, 3)

        # ---- per-point heads (ffb6d.py:135-157) ----
        c = up_rndla_oc[-1] + UP_RGB_OC[-1]
        self.rgbd_seg_layer = _head(c, n_classes)
        self.ctr_ofst_layer = _head(c, 3)
        self.kp_ofst_layer = _head(c, n_kps * 3)

    def train(self, mode=True):
        # drop every cached inference-time fold (BatchNorm scale/shift, split weights): they are
        # also version-checked, this covers edits made through `.data`
        self.__dict__.pop("_pm_supported", None)
        for m in self.modules():
            m.__dict__.pop("_pm_cache", None)
        return super().train(mode)

    # the reference exposes these two as static methods of the model (ffb6d.py:159-194)
    random_sample = staticmethod(ops.random_sample)
    nearest_interpolation = staticmethod(ops.nearest_interpolation)

    def _fuse(self, i, pre_p2r, fuse_p2r, pre_r2p, fuse_r2p, rgb_emb0, p_emb0, p2r_idx, r2p_idx):
        """One bidirectional fusion step (ffb6d.py:245-263 / 281-298); both directions read
        the pre-fusion tensors, so they are independent."""
        hr, wr = rgb_emb0.shape[2:]
        p2r = ops_cl.nearest_interpolation(pre_p2r[i](p_emb0), p2r_idx, (hr, wr))
        rgb_emb = fuse_p2r[i](torch.cat((rgb_emb0, p2r), dim=1))
        r2p = ops_cl.random_sample(rgb_emb0, r2p_idx)
        p_emb = fuse_r2p[i](torch.cat((p_emb0, pre_r2p[i](r2p)), dim=1))
        return rgb_emb, p_emb

    def _decode(self, stage, skip, p_emb, interp_idx):
        """RandLA decoder step conv(cat(skip, interp(p))) (ffb6d.py:273-279,302-307)."""
        return stage(torch.cat([skip, ops_cl.nearest_interpolation(p_emb, interp_idx)], dim=1))

    # Fused inference (forward_pm.forward): the point branch on a second HIP stream under the colour branch's convolutions,
    # the index pyramid on a third; two_streams = False keeps everything on the caller's stream (bit-identical results).
    two_streams = True
    # arithmetic of the fused point-major path: "fp32" (default, BASELINE configurations 2-4) or "bf16" (configuration 5:
    # bfloat16 activations and weights, fp32 accumulation / BatchNorm / softmax arithmetic, fp32 end_points)
    precision = "fp32"

    # dtype of the indices when the forward builds the index pyramid itself (inputs carry 'dpt_xyz' instead of the 26
    # index tensors): int64 is what model_fn feeds the reference network (train_lm.py:236-237)
    index_dtype = torch.int64

    def _index_stream(self, device):
        st = getattr(self, "_idx_stream", None)
        if st is None or st.device != device:
            st = torch.cuda.Stream(device=device, priority=-1)
            self._idx_stream = st
        return st

    def _side_stream(self, device):
        st = getattr(self, "_side", None)
        if st is None or st.device != device:
            st = torch.cuda.Stream(device=device)
            self._side = st
        return st

    def check_indices(self, inputs):
        """Debug aid (FFB6D_CHECK_INDICES=1 runs it on every forward): every index tensor of the input dict must address
        its source set.  The gather kernels do not bounds-check (torch.gather device-asserts there); an out-of-range
        index is an out-of-bounds read.  One device counter per tensor, one host sync at the end."""
        B, _, H, W = inputs['rgb'].shape
        n = [inputs['cld_xyz%d' % i].shape[1] for i in range(4)]
        n_sub = [inputs['cld_sub_idx%d' % i].shape[1] for i in range(4)]

        def hw(idx_key):            # pixels of the map a p2r index tensor has one entry per
            return inputs[idx_key].shape[1]
        limits = {'choose': H * W}
        for i in range(4):
            limits['cld_nei_idx%d' % i] = n[i]
            limits['cld_sub_idx%d' % i] = n[i]
            limits['cld_interp_idx%d' % i] = n_sub[i]
            limits['r2p_ds_nei_idx%d' % i] = hw('p2r_ds_nei_idx%d' % i)
            limits['p2r_ds_nei_idx%d' % i] = n_sub[i]
        for i in range(3):
            limits['r2p_up_nei_idx%d' % i] = hw('p2r_up_nei_idx%d' % i)
            limits['p2r_up_nei_idx%d' % i] = inputs['r2p_up_nei_idx%d' % i].shape[1]
        bad = {k: ops.check_index_range(inputs[k], m) for k, m in limits.items()}
        bad = {k: v for k, v in bad.items() if v}
        if bad:
            raise IndexError(f"index tensors with out-of-range entries (tensor: count): {bad}")

    def forward(self, inputs, end_points=None, scale=1, taps=None):
        if not end_points:
            end_points = {}
        if os.environ.get("FFB6D_CHECK_INDICES") == "1" and inputs['rgb'].is_cuda:
            if 'cld_nei_idx0' in inputs:
                self.check_indices(inputs)
            else:        # the forward builds the pyramid itself: `choose` is the one index tensor that still comes from the caller
                bad = ops.check_index_range(inputs['choose'], inputs['rgb'].shape[2] * inputs['rgb'].shape[3])
                if bad:  # (the patch-row kernel of the last colour stage would clamp it silently: forward_pm.last_stage_at_chosen)
                    raise IndexError(f"choose: {bad} entries outside the {inputs['rgb'].shape[2]} x {inputs['rgb'].shape[3]} image")
        rgb = inputs['rgb']
        fused = not _autograd_path(rgb, self)
        if 'cld_nei_idx0' not in inputs:
            # no index pyramid in the dict (the dataset's 22 knn_search calls, linemod_dataset.py:299-353): build it on
            # the device from the xyz image.  The fused point-major path streams it level by level under the network;
            # every other path builds it up front.
            if 'dpt_xyz' not in inputs:
                raise KeyError("inputs carry neither the index pyramid ('cld_nei_idx0', ...) nor 'dpt_xyz' to build it from")
            if not (fused and forward_pm.supported(self)) or taps is not None:
                inputs = dict(inputs)
                inputs.update(pyramid.build_index_pyramid(inputs['cld_rgb_nrm'][:, :3, :].transpose(1, 2).contiguous(),
                                                          inputs['dpt_xyz'], index_dtype=self.index_dtype))
        if fused and forward_pm.supported(self):
            return forward_pm.forward(self, inputs, end_points, two_streams=self.two_streams, taps=taps)
        # stock modules + the neighbour operators on channels-last rows (training; or widths the fused kernels do not cover)
        rgb_emb = self.cnn_pre_stages(inputs['rgb'])
        p_emb = self.rndla_pre_stages(inputs['cld_rgb_nrm']).unsqueeze(3)

        ds_emb = []
        for i in range(4):
            rgb_emb0 = self.cnn_ds_stages[i](rgb_emb)
            f_enc = self.rndla_ds_stages[i](p_emb, inputs['cld_xyz%d' % i], inputs['cld_nei_idx%d' % i])
            p_emb0 = ops_cl.random_sample(f_enc, inputs['cld_sub_idx%d' % i])
            if i == 0:
                ds_emb.append(f_enc)
            rgb_emb, p_emb = self._fuse(
                i, self.ds_fuse_p2r_pre_layers, self.ds_fuse_p2r_fuse_layers,
                self.ds_fuse_r2p_pre_layers, self.ds_fuse_r2p_fuse_layers, rgb_emb0, p_emb0,
                inputs['p2r_ds_nei_idx%d' % i], inputs['r2p_ds_nei_idx%d' % i])
            ds_emb.append(p_emb)

        n_up = len(self.rndla_up_stages)
        for i in range(n_up - 1):
            rgb_emb0 = self.cnn_up_stages[i](rgb_emb)
            p_emb0 = self._decode(self.rndla_up_stages[i], ds_emb[-i - 2], p_emb,
                                  inputs['cld_interp_idx%d' % (n_up - i - 1)])
            rgb_emb, p_emb = self._fuse(
                i, self.up_fuse_p2r_pre_layers, self.up_fuse_p2r_fuse_layers,
                self.up_fuse_r2p_pre_layers, self.up_fuse_r2p_fuse_layers, rgb_emb0, p_emb0,
                inputs['p2r_up_nei_idx%d' % i], inputs['r2p_up_nei_idx%d' % i])

        rgb_emb = self.cnn_up_stages[n_up - 1](rgb_emb)
        p_emb = self._decode(self.rndla_up_stages[n_up - 1], ds_emb[0], p_emb,
                             inputs['cld_interp_idx0']).squeeze(-1)

        bs = rgb_emb.shape[0]
        rgb_emb_c = ops_cl.choose_gather(rgb_emb, inputs['choose'])

        rgbd_emb = torch.cat([rgb_emb_c, p_emb], dim=1)          # ffb6d.py:318-323: one concatenation feeds the three heads

        def head(seq):
            return seq(rgbd_emb)

        end_points['pred_rgbd_segs'] = head(self.rgbd_seg_layer)
        end_points['pred_kp_ofs'] = head(self.kp_ofst_layer).view(
            bs, self.n_kps, 3, -1).permute(0, 1, 3, 2).contiguous()
        end_points['pred_ctr_ofs'] = head(self.ctr_ofst_layer).view(
            bs, 1, 3, -1).permute(0, 1, 3, 2).contiguous()
        return end_points
