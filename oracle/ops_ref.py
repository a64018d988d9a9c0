"""oracle/ops_ref.py -- TEST INFRASTRUCTURE ONLY.  Plain PyTorch fp32 (CPU) restatements of
the reference's neighbour operators, written with advanced indexing instead of the
reference's gather+repeat so they are an independent statement of the same maths.  Each
is pinned against the reference's own function in tests/test_oracle_cpu.py."""
import torch


def _bidx(B, *rest, device=None):
    return torch.arange(B, device=device).view(B, *([1] * len(rest)))


def random_sample(feature, pool_idx):
    """ffb6d/models/ffb6d.py:159-177: [B,C,M(,1)], [B,Np,K] -> [B,C,Np,1] max over K."""
    if feature.dim() > 3:
        feature = feature.squeeze(3)
    B, C, M = feature.shape
    idx = pool_idx.long()
    # feature[b, :, idx[b,n,k]] -> [B,C,Np,K]
    g = feature[_bidx(B, 1, 1, 1), torch.arange(C).view(1, C, 1, 1), idx.unsqueeze(1)]
    return g.max(dim=3, keepdim=True)[0]


def nearest_interpolation(feature, interp_idx):
    """ffb6d/models/ffb6d.py:179-194: [B,C,M,1], [B,U,1] -> [B,C,U,1]."""
    if feature.dim() > 3:
        feature = feature.squeeze(3)
    B, C, M = feature.shape
    idx = interp_idx.long().reshape(B, 1, -1)
    g = feature[_bidx(B, 1, 1), torch.arange(C).view(1, C, 1), idx]
    return g.unsqueeze(3)


def gather_neighbour(pc, neighbor_idx):
    """RandLANet.py:225-234: [B,M,C], [B,N,K] -> [B,N,K,C]."""
    B = pc.shape[0]
    return pc[_bidx(B, 1, 1), neighbor_idx.long()]


def relative_pos_encoding(xyz, neigh_idx):
    """RandLANet.py:216-223: [B,N,3], [B,N,K] -> [B,N,K,10] = [dis, p-q, p, q]."""
    q = gather_neighbour(xyz, neigh_idx)
    p = xyz.unsqueeze(2).expand_as(q)
    rel = p - q
    dis = torch.sqrt(torch.sum(torch.pow(rel, 2), dim=-1, keepdim=True))
    return torch.cat([dis, rel, p, q], dim=-1)


def att_pool(feature_set, att_activation):
    """RandLANet.py:245-248: softmax over K of the activation, weighted sum of the features."""
    scores = torch.softmax(att_activation, dim=3)
    return torch.sum(feature_set * scores, dim=3, keepdim=True)


def _act(v, code):
    """0 = none, 1 = ReLU, 2 = LeakyReLU(0.2) (RandLA/pytorch_utils.py:35-111, the wrappers' activations)"""
    return v if code == 0 else (torch.relu(v) if code == 1 else torch.nn.functional.leaky_relu(v, 0.2))


def lfa_half(mode, xyz, neigh_idx, f, w1, b1, act1, wfc, wm, bm, actm, w2=None, b2=None, act2=0, dtype=torch.float64, store=None):
    """One half of Building_block.forward (RandLANet.py:196-214) on ROW tensors, in `dtype` arithmetic (float64 = the bar the
    fused kernel is held to): xyz [B,N,3], neigh_idx [B,N,K], f [B,N,d/2] -> [B,N,cout].
      mode 1: :196-206  mlp1(relative_pos_encoding) | gather_neighbour(f) -> att_pooling_1 (fc, softmax over K, sum, mlp)
      mode 2: :208-213  mlp2(mlp1(...))             | gather_neighbour(f) -> att_pooling_2
    BatchNorm is folded into (w, b) by the caller.  `store`: optional rounding applied wherever the kernel stores rows of the
    activation dtype (the pair rows, the pooled rows) -- identity for float32, a bfloat16 round trip for the bf16 path."""
    r = (lambda t: t) if store is None else store
    enc = relative_pos_encoding(xyz.to(dtype), neigh_idx)                         # [B,N,K,10]
    g = r(_act(enc @ w1[:, :10].to(dtype).t() + b1.to(dtype), act1))              # lfa.mlp1
    if mode == 2:
        g = r(_act(g @ w2.to(dtype).t() + b2.to(dtype), act2))                    # lfa.mlp2
    s = torch.cat([gather_neighbour(f.to(dtype), neigh_idx), g], dim=3)          # feature set [B,N,K,d]
    pooled = r((s * torch.softmax(s @ wfc.to(dtype).t(), dim=2)).sum(dim=2))      # Att_pooling :245-248
    return _act(pooled @ wm.to(dtype).t() + bm.to(dtype), actm)                   # Att_pooling.mlp :249
