"""oracle/inputs_ref.py -- TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's depth
back-projection `dpt_2_pcld` (ffb6d/datasets/linemod/linemod_dataset.py:188-199) and the NaN/Inf
clean-up that follows it (:258-259), with numpy's own dtype promotion (float32 depth, int64 pixel
maps, float64 intrinsics -> float64 result)."""
import numpy as np


def dpt_2_pcld(dpt, cam_scale, K):
    h, w = dpt.shape[:2]
    xmap = np.array([[j for i in range(w)] for j in range(h)])      # row index  (linemod_dataset.py:32)
    ymap = np.array([[i for i in range(w)] for j in range(h)])      # col index  (linemod_dataset.py:33)
    if len(dpt.shape) > 2:
        dpt = dpt[:, :, 0]
    dpt = dpt.astype(np.float32) / cam_scale
    msk = (dpt > 1e-8).astype(np.float32)
    row = (ymap - K[0][2]) * dpt / K[0][0]
    col = (xmap - K[1][2]) * dpt / K[1][1]
    dpt_3d = np.concatenate((row[..., None], col[..., None], dpt[..., None]), axis=2)
    dpt_3d = dpt_3d * msk[:, :, None]
    dpt_3d[np.isnan(dpt_3d)] = 0.0
    dpt_3d[np.isinf(dpt_3d)] = 0.0
    return dpt_3d
