"""A reference-SHAPED model for measuring and testing the operator-level drop-in where the reference tree itself is not at hand
(the GPU box): the class interface `patch.patch_classes` swaps operators into --

    FFB6D.random_sample / FFB6D.nearest_interpolation        static methods      ffb6d/models/ffb6d.py:159-194
    Building_block.gather_neighbour                          static method       RandLA/RandLANet.py:225-234
    Building_block.relative_pos_encoding                     method              RandLANet.py:216-223
    Att_pooling.forward                                      method              RandLANet.py:243-250

-- around the parameters of a `model.FFB6D` (same state_dict as the reference), executing the reference's dataflow
(ffb6d.py:203-337, RandLANet.py:170-250) in the reference's tensor layout: channel-major [B,C,N,1] / [B,C,H,W] activations through
stock conv -> BatchNorm -> activation modules, and the five operators as PLAIN TORCH by default (index expansion + torch.gather +
permutes, softmax / multiply / sum), i.e. what an unpatched reference model executes.  After

    undo = patch.patch_classes(dropin.FFB6D, dropin.Building_block, dropin.Att_pooling)

the same forward runs the channel-major HIP operators of `ffb6d_amd.ops` -- exactly what `patch.patch_reference` makes an unmodified
reference model execute.  `bench.py --path dropin` times both; tests/test_dropin_gpu.py compares them on the device.
(The real reference classes are patched and run in tests/test_model_cpu.py / tests/test_simt_cpu.py where /root/reference exists.)
"""
import torch
import torch.nn.functional as F


class Att_pooling:
    def __init__(self, mod):
        self.fc, self.mlp = mod.fc, mod.mlp

    def forward(self, feature_set):                     # [B,d,N,K] -> [B,d_out,N,1]
        scores = torch.softmax(self.fc(feature_set), dim=3)
        return self.mlp(torch.sum(feature_set * scores, dim=3, keepdim=True))

    def __call__(self, feature_set):
        return self.forward(feature_set)                # looked up on the class at call time: patch_classes swaps `forward`


class Building_block:
    def __init__(self, mod):
        self.mlp1, self.mlp2 = mod.mlp1, mod.mlp2
        self.att_pooling_1, self.att_pooling_2 = Att_pooling(mod.att_pooling_1), Att_pooling(mod.att_pooling_2)

    @staticmethod
    def gather_neighbour(pc, neighbor_idx):             # pc [B,N,d], idx [B,N,K] -> [B,N,K,d]
        B, N, K = neighbor_idx.shape
        d = pc.shape[2]
        flat = neighbor_idx.reshape(B, N * K, 1).expand(B, N * K, d).long()
        return torch.gather(pc, 1, flat).reshape(B, N, K, d)

    def relative_pos_encoding(self, xyz, neigh_idx):    # [B,N,3], [B,N,K] -> [B,N,K,10] = [distance, p - q, p, q]
        q = self.gather_neighbour(xyz, neigh_idx)
        p = xyz.unsqueeze(2).expand(-1, -1, neigh_idx.shape[-1], -1)
        rel = p - q
        dis = torch.sqrt(torch.sum(rel * rel, dim=-1, keepdim=True))
        return torch.cat([dis, rel, p, q], dim=-1)

    def _neighbours(self, feature, neigh_idx):          # [B,d,N,1] -> [B,d,N,K]
        rows = feature.squeeze(-1).permute(0, 2, 1)
        return self.gather_neighbour(rows, neigh_idx).permute(0, 3, 1, 2)

    def forward(self, xyz, feature, neigh_idx):
        f_xyz = self.mlp1(self.relative_pos_encoding(xyz, neigh_idx).permute(0, 3, 1, 2))
        f_agg = self.att_pooling_1(torch.cat([self._neighbours(feature, neigh_idx), f_xyz], dim=1))
        f_xyz = self.mlp2(f_xyz)
        return self.att_pooling_2(torch.cat([self._neighbours(f_agg, neigh_idx), f_xyz], dim=1))

    def __call__(self, xyz, feature, neigh_idx):
        return self.forward(xyz, feature, neigh_idx)


class _ResBlock:                                        # Dilated_res_block, RandLANet.py:170-184
    def __init__(self, mod):
        self.mlp1, self.mlp2, self.shortcut, self.lfa = mod.mlp1, mod.mlp2, mod.shortcut, Building_block(mod.lfa)

    def __call__(self, feature, xyz, neigh_idx):
        f = self.mlp2(self.lfa(xyz, self.mlp1(feature), neigh_idx))
        return F.leaky_relu(f + self.shortcut(feature), negative_slope=0.2)


class FFB6D:
    """net: ffb6d_amd.model.FFB6D (its parameters and stock modules are used as they are; its own forward is not)."""

    def __init__(self, net):
        self.net = net
        self.ds = [_ResBlock(m) for m in net.rndla_ds_stages]

    @staticmethod
    def random_sample(feature, pool_idx):               # [B,C,M(,1)], [B,N',K] -> [B,C,N',1]: max over the K gathered columns
        if feature.dim() > 3:
            feature = feature.squeeze(dim=3)
        B, C, _ = feature.shape
        Np, K = pool_idx.shape[1], pool_idx.shape[2]
        flat = pool_idx.reshape(B, 1, Np * K).expand(B, C, Np * K).long()
        return torch.gather(feature, 2, flat).reshape(B, C, Np, K).max(dim=3, keepdim=True)[0]

    @staticmethod
    def nearest_interpolation(feature, interp_idx):     # [B,C,M(,1)], [B,U,1] -> [B,C,U,1]
        if feature.dim() > 3:
            feature = feature.squeeze(dim=3)
        B, C, _ = feature.shape
        U = interp_idx.shape[1]
        flat = interp_idx.reshape(B, 1, U).expand(B, C, U).long()
        return torch.gather(feature, 2, flat).unsqueeze(3)

    def _fuse(self, i, pre_p2r, fuse_p2r, pre_r2p, fuse_r2p, rgb0, p0, p2r_idx, r2p_idx):
        B, c, h, w = rgb0.shape
        p2r = self.nearest_interpolation(pre_p2r[i](p0), p2r_idx).reshape(B, -1, h, w)
        rgb = fuse_p2r[i](torch.cat([rgb0, p2r], dim=1))
        r2p = pre_r2p[i](self.random_sample(rgb0.reshape(B, c, h * w, 1), r2p_idx))
        return rgb, fuse_r2p[i](torch.cat([p0, r2p], dim=1))

    def forward(self, inputs):
        net = self.net
        rgb_emb = net.cnn_pre_stages(inputs['rgb'])
        p_emb = net.rndla_pre_stages(inputs['cld_rgb_nrm']).unsqueeze(3)
        ds_emb = []
        for i in range(4):
            rgb0 = net.cnn_ds_stages[i](rgb_emb)
            f_enc = self.ds[i](p_emb, inputs['cld_xyz%d' % i], inputs['cld_nei_idx%d' % i])
            p0 = self.random_sample(f_enc, inputs['cld_sub_idx%d' % i])
            if i == 0:
                ds_emb.append(f_enc)
            rgb_emb, p_emb = self._fuse(i, net.ds_fuse_p2r_pre_layers, net.ds_fuse_p2r_fuse_layers, net.ds_fuse_r2p_pre_layers,
                                        net.ds_fuse_r2p_fuse_layers, rgb0, p0, inputs['p2r_ds_nei_idx%d' % i], inputs['r2p_ds_nei_idx%d' % i])
            ds_emb.append(p_emb)
        n_up = len(net.rndla_up_stages)
        for i in range(n_up - 1):
            rgb0 = net.cnn_up_stages[i](rgb_emb)
            interp = self.nearest_interpolation(p_emb, inputs['cld_interp_idx%d' % (n_up - i - 1)])
            p0 = net.rndla_up_stages[i](torch.cat([ds_emb[-i - 2], interp], dim=1))
            rgb_emb, p_emb = self._fuse(i, net.up_fuse_p2r_pre_layers, net.up_fuse_p2r_fuse_layers, net.up_fuse_r2p_pre_layers,
                                        net.up_fuse_r2p_fuse_layers, rgb0, p0, inputs['p2r_up_nei_idx%d' % i], inputs['r2p_up_nei_idx%d' % i])
        rgb_emb = net.cnn_up_stages[n_up - 1](rgb_emb)
        interp = self.nearest_interpolation(p_emb, inputs['cld_interp_idx0'])
        p_emb = net.rndla_up_stages[n_up - 1](torch.cat([ds_emb[0], interp], dim=1)).squeeze(-1)
        B, di = rgb_emb.shape[:2]
        pick = inputs['choose'].reshape(B, 1, -1).expand(B, di, -1).long()
        rgbd = torch.cat([torch.gather(rgb_emb.reshape(B, di, -1), 2, pick), p_emb], dim=1)
        n = rgbd.shape[2]
        return {'pred_rgbd_segs': net.rgbd_seg_layer(rgbd),
                'pred_kp_ofs': net.kp_ofst_layer(rgbd).view(B, net.n_kps, 3, n).permute(0, 1, 3, 2).contiguous(),
                'pred_ctr_ofs': net.ctr_ofst_layer(rgbd).view(B, 1, 3, n).permute(0, 1, 3, 2).contiguous()}

    def __call__(self, inputs):
        return self.forward(inputs)
