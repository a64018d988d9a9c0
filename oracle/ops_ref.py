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
