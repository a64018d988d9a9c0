"""oracle/forward_ref.py -- TEST INFRASTRUCTURE ONLY (also the `cpu_baseline` leg of bench.py).

Functional, plain-PyTorch fp32 CPU restatement of the reference's whole forward pass
    FFB6D.forward                      ffb6d/models/ffb6d.py:203-337
    Dilated_res_block / Building_block / Att_pooling   ffb6d/models/RandLA/RandLANet.py:170-250
    conv+BN+act wrappers               ffb6d/models/pytorch_utils.py:75-129 (ReLU, BN eps 1e-5)
                                       ffb6d/models/RandLA/pytorch_utils.py:35-111 (LeakyReLU 0.2, BN eps 1e-6)
    ResNet34 / PSP / up-sampling       ffb6d/models/cnn/extractors.py:34-200, pspnet.py:7-45,93-112
driven directly by a reference-format state dict (no nn.Module tree), eval-mode semantics
(BatchNorm uses running statistics, dropout is the identity).  It exists to check the HIP
product path and to time the CPU baseline on hosts where /root/reference is absent; it is
pinned to the reference by tests/golden/forward_small.npz (tests/test_oracle_cpu.py).
"""
import torch
import torch.nn.functional as F

from . import ops_ref


class _W:
    """state-dict view with a name prefix."""

    def __init__(self, sd, prefix=""):
        self.sd, self.p = sd, prefix

    def sub(self, name):
        return _W(self.sd, f"{self.p}{name}.")

    def __getitem__(self, name):
        return self.sd[self.p + name]

    def has(self, name):
        return (self.p + name) in self.sd


def _bn(x, w, eps):
    return F.batch_norm(x, w["running_mean"], w["running_var"], w["weight"], w["bias"], False, 0.0, eps)


def mlp_randla(x, w, act=True):
    """RandLA shared MLP: 1x1 conv (no bias when BN) + BN(eps=1e-6) + LeakyReLU(0.2)."""
    cw = w["conv.weight"]
    bias = w["conv.bias"] if w.has("conv.bias") else None
    y = F.conv2d(x, cw, bias) if cw.dim() == 4 else F.conv1d(x, cw, bias)
    if w.has("bn.bn.weight"):
        y = _bn(y, w.sub("bn.bn"), 1e-6)
    return F.leaky_relu(y, 0.2) if act else y


def mlp_pvn(x, w, act=True):
    """PVN3D-style shared MLP: 1x1 conv + BN(eps=1e-5) + ReLU."""
    cw = w["conv.weight"]
    bias = w["conv.bias"] if w.has("conv.bias") else None
    y = F.conv2d(x, cw, bias) if cw.dim() == 4 else F.conv1d(x, cw, bias)
    if w.has("normlayer.bn.weight"):
        y = _bn(y, w.sub("normlayer.bn"), 1e-5)
    return F.relu(y) if act else y


def att_pooling(feature_set, w):
    act = F.conv2d(feature_set, w["fc.weight"])
    return mlp_randla(ops_ref.att_pool(feature_set, act), w.sub("mlp"))


def building_block(xyz, feature, neigh_idx, w):
    f_xyz = ops_ref.relative_pos_encoding(xyz, neigh_idx).permute(0, 3, 1, 2)
    f_xyz = mlp_randla(f_xyz, w.sub("mlp1"))
    f_nei = ops_ref.gather_neighbour(feature.squeeze(-1).permute(0, 2, 1), neigh_idx).permute(0, 3, 1, 2)
    f_agg = att_pooling(torch.cat([f_nei, f_xyz], dim=1), w.sub("att_pooling_1"))
    f_xyz = mlp_randla(f_xyz, w.sub("mlp2"))
    f_nei = ops_ref.gather_neighbour(f_agg.squeeze(-1).permute(0, 2, 1), neigh_idx).permute(0, 3, 1, 2)
    return att_pooling(torch.cat([f_nei, f_xyz], dim=1), w.sub("att_pooling_2"))


def dilated_res_block(feature, xyz, neigh_idx, w):
    f = mlp_randla(feature, w.sub("mlp1"))
    f = building_block(xyz, f, neigh_idx, w.sub("lfa"))
    f = mlp_randla(f, w.sub("mlp2"), act=False)
    return F.leaky_relu(f + mlp_randla(feature, w.sub("shortcut"), act=False), 0.2)


def basic_block(x, w, stride):
    y = F.conv2d(x, w["conv1.weight"], None, stride, 1)
    y = F.relu(_bn(y, w.sub("bn1"), 1e-5))
    y = F.conv2d(y, w["conv2.weight"], None, 1, 1)
    y = _bn(y, w.sub("bn2"), 1e-5)
    if w.has("downsample.0.weight"):
        x = _bn(F.conv2d(x, w["downsample.0.weight"], None, stride), w.sub("downsample.1"), 1e-5)
    return F.relu(y + x)


def res_layer(x, w, n_blocks, stride):
    for i in range(n_blocks):
        x = basic_block(x, w.sub(str(i)), stride if i == 0 else 1)
    return x


def psp_module(x, w, sizes=(1, 2, 3, 6)):
    h, wd = x.shape[2], x.shape[3]
    priors = []
    for i, s in enumerate(sizes):
        p = F.conv2d(F.adaptive_avg_pool2d(x, (s, s)), w[f"stages.{i}.1.weight"])
        priors.append(F.interpolate(p, size=(h, wd), mode="bilinear", align_corners=False))
    y = F.conv2d(torch.cat(priors + [x], 1), w["bottleneck.weight"], w["bottleneck.bias"])
    return F.relu(y)


def psp_upsample(x, w):
    y = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    y = F.conv2d(y, w["conv.1.weight"], w["conv.1.bias"], 1, 1)
    y = _bn(y, w.sub("conv.2"), 1e-5)
    return F.prelu(y, w["conv.3.weight"])


def final_head(x, w):
    # nn.Sequential(Conv2d(64,64,1), LogSoftmax()) -- implicit dim on a 4-d input is 1
    return F.log_softmax(F.conv2d(x, w["0.weight"], w["0.bias"]), dim=1)


def head(x, w):
    for i in range(3):
        x = mlp_pvn(x, w.sub(str(i)))
    return mlp_pvn(x, w.sub("3"), act=False)


def ffb6d_forward(state_dict, inputs, n_kps=8, taps=None):
    """inputs: the reference's input dict (torch CPU tensors, int64 indices).
    taps: optional dict that receives the two embeddings after every fusion stage (`rgb_emb_ds{i}`, `p_emb_ds{i}`,
    `rgb_emb_up{i}`, `p_emb_up{i}`, reference layout [B,C,H,W] / [B,C,N,1]) for stage-level parity checks."""
    w = _W(state_dict)
    rgb_emb = F.conv2d(inputs["rgb"], w["cnn_pre_stages.0.weight"], None, 2, 3)
    rgb_emb = F.relu(_bn(rgb_emb, w.sub("cnn_pre_stages.1"), 1e-5))
    rgb_emb = F.max_pool2d(rgb_emb, 3, 2, 1)
    p_emb = mlp_randla(inputs["cld_rgb_nrm"], w.sub("rndla_pre_stages")).unsqueeze(3)

    ds_emb = []
    for i in range(4):
        cw = w.sub(f"cnn_ds_stages.{i}")
        if i == 0:
            rgb_emb0 = res_layer(rgb_emb, cw, 3, 1)
        elif i == 1:
            rgb_emb0 = res_layer(rgb_emb, cw, 4, 2)
        elif i == 2:
            rgb_emb0 = res_layer(res_layer(rgb_emb, cw.sub("0"), 6, 1), cw.sub("1"), 3, 1)
        else:
            rgb_emb0 = psp_module(rgb_emb, cw.sub("0"))
        bs, c, hr, wr = rgb_emb0.shape
        f_enc = dilated_res_block(p_emb, inputs[f"cld_xyz{i}"], inputs[f"cld_nei_idx{i}"],
                                  w.sub(f"rndla_ds_stages.{i}"))
        p_emb0 = ops_ref.random_sample(f_enc, inputs[f"cld_sub_idx{i}"])
        if i == 0:
            ds_emb.append(f_enc)
        p2r = mlp_pvn(p_emb0, w.sub(f"ds_fuse_p2r_pre_layers.{i}"))
        p2r = ops_ref.nearest_interpolation(p2r, inputs[f"p2r_ds_nei_idx{i}"]).view(bs, -1, hr, wr)
        rgb_emb = mlp_pvn(torch.cat((rgb_emb0, p2r), dim=1), w.sub(f"ds_fuse_p2r_fuse_layers.{i}"))
        r2p = ops_ref.random_sample(rgb_emb0.reshape(bs, c, hr * wr, 1), inputs[f"r2p_ds_nei_idx{i}"])
        r2p = mlp_pvn(r2p, w.sub(f"ds_fuse_r2p_pre_layers.{i}"))
        p_emb = mlp_pvn(torch.cat((p_emb0, r2p), dim=1), w.sub(f"ds_fuse_r2p_fuse_layers.{i}"))
        ds_emb.append(p_emb)
        if taps is not None:
            taps[f"rgb_emb_ds{i}"], taps[f"p_emb_ds{i}"] = rgb_emb, p_emb

    for i in range(3):
        cw = w.sub(f"cnn_up_stages.{i}")
        rgb_emb0 = psp_upsample(rgb_emb, cw.sub("0")) if i < 2 else final_head(rgb_emb, cw.sub("0"))
        bs, c, hr, wr = rgb_emb0.shape
        f_interp = ops_ref.nearest_interpolation(p_emb, inputs[f"cld_interp_idx{3 - i}"])
        p_emb0 = mlp_randla(torch.cat([ds_emb[-i - 2], f_interp], dim=1), w.sub(f"rndla_up_stages.{i}"))
        p2r = mlp_pvn(p_emb0, w.sub(f"up_fuse_p2r_pre_layers.{i}"))
        p2r = ops_ref.nearest_interpolation(p2r, inputs[f"p2r_up_nei_idx{i}"]).view(bs, -1, hr, wr)
        rgb_emb = mlp_pvn(torch.cat((rgb_emb0, p2r), dim=1), w.sub(f"up_fuse_p2r_fuse_layers.{i}"))
        r2p = ops_ref.random_sample(rgb_emb0.reshape(bs, c, hr * wr), inputs[f"r2p_up_nei_idx{i}"])
        r2p = mlp_pvn(r2p, w.sub(f"up_fuse_r2p_pre_layers.{i}"))
        p_emb = mlp_pvn(torch.cat((p_emb0, r2p), dim=1), w.sub(f"up_fuse_r2p_fuse_layers.{i}"))
        if taps is not None:
            taps[f"rgb_emb_up{i}"], taps[f"p_emb_up{i}"] = rgb_emb, p_emb

    cw = w.sub("cnn_up_stages.3")
    rgb_emb = final_head(psp_upsample(rgb_emb, cw.sub("0")), cw.sub("1"))
    f_interp = ops_ref.nearest_interpolation(p_emb, inputs["cld_interp_idx0"])
    p_emb = mlp_randla(torch.cat([ds_emb[0], f_interp], dim=1), w.sub("rndla_up_stages.3")).squeeze(-1)
    bs, di = rgb_emb.shape[:2]
    rgb_c = ops_ref.nearest_interpolation(rgb_emb.view(bs, di, -1, 1), inputs["choose"].view(bs, -1, 1)).squeeze(3)
    rgbd = torch.cat([rgb_c, p_emb], dim=1)
    segs = head(rgbd, w.sub("rgbd_seg_layer"))
    kp = head(rgbd, w.sub("kp_ofst_layer")).view(bs, n_kps, 3, -1).permute(0, 1, 3, 2).contiguous()
    ctr = head(rgbd, w.sub("ctr_ofst_layer")).view(bs, 1, 3, -1).permute(0, 1, 3, 2).contiguous()
    return {"pred_rgbd_segs": segs, "pred_kp_ofs": kp, "pred_ctr_ofs": ctr}
