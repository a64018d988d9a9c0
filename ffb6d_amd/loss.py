"""The training objective of the reference, restated (ffb6d/models/loss.py:12-86, assembled as in train_lm.py:245-259):

    loss = 2 * focal(pred_rgbd_segs, labels) + sum_k L1(pred_kp_ofs, kp_targ_ofst | labels) + L1(pred_ctr_ofs, ctr_targ_ofst | labels)

  focal (gamma = 2, no class weights, mean over all points): -(1 - p_t)^gamma * log p_t with p_t the soft-max probability of the
        point's label; the modulating factor is a constant of the graph (the reference takes it from `.data`), so the gradient is
        (1 - p_t)^gamma * d(-log p_t).
  L1 offsets: per frame and keypoint, the absolute offset error summed over the points of an object (label > 0) and the three
        coordinates, divided by (number of object points + 1e-3); the caller sums the [B, n_kps] table.

Plain torch (these are O(B*N) reductions on the heads' outputs, 0.1 % of the step); bench.py --mode train times the step with it."""
import torch


def focal_loss(logits, labels, gamma=2.0):
    """logits [B,C,N] (or [M,C]), labels int [B,N] / [B*N] -> scalar (loss.py:21-44)"""
    if logits.dim() > 2:
        logits = logits.reshape(logits.shape[0], logits.shape[1], -1).transpose(1, 2).reshape(-1, logits.shape[1])
    logpt = torch.log_softmax(logits.float(), dim=1).gather(1, labels.reshape(-1, 1).long()).reshape(-1)      # fp32 as under amp
    pt = logpt.detach().exp()
    return (-((1.0 - pt) ** gamma) * logpt).mean()


def offset_l1_loss(pred_ofsts, targ_ofsts, labels):
    """pred_ofsts [B,K,N,3], targ_ofsts [B,N,K,3], labels [B,N] (or [B,N,1]) -> [B,K] (loss.py:47-77, normalize=True)"""
    B, K, N, c = pred_ofsts.shape
    w = (labels.reshape(B, 1, N, 1) > 1e-8).to(torch.float32)
    err = (pred_ofsts - targ_ofsts.reshape(B, N, K, c).permute(0, 2, 1, 3)).abs() * w
    return err.reshape(B, K, -1).sum(2) / (w.expand(B, K, N, 1).reshape(B, K, -1).sum(2) + 1e-3)


def training_loss(end_points, labels, kp_targ_ofst, ctr_targ_ofst):
    """train_lm.py:245-259: (total, dict of the three terms)"""
    seg = focal_loss(end_points['pred_rgbd_segs'], labels.reshape(-1))
    kp = offset_l1_loss(end_points['pred_kp_ofs'], kp_targ_ofst, labels).sum()
    ctr = offset_l1_loss(end_points['pred_ctr_ofs'], ctr_targ_ofst, labels).sum()
    return 2.0 * seg + kp + ctr, {"loss_rgbd_seg": seg, "loss_kp_of": kp, "loss_ctr_of": ctr}
