"""On-device input pipeline in front of the index pyramid: the
parts of the reference's `Dataset.get_item` that turn a depth image into the network's geometric
inputs, so a frame needs no host-side geometry at all.

    depth_to_cloud   == dpt_2_pcld                 linemod_dataset.py:188-199 (+ NaN/Inf -> 0, :258-259)
    sample_choose    == valid-pixel sampling       linemod_dataset.py:262-282 (N valid pixels, shuffled;
                                                    'wrap' padding when fewer than N are valid)
    assemble_inputs  == cld / cld_rgb_nrm / choose + build_index_pyramid   linemod_dataset.py:284-353

Sampling uses torch's generator instead of numpy's global RNG, so it is distribution-equivalent,
not bit-identical, to the reference (any N-subset of the valid pixels in uniformly random order).
"""
import torch

from . import _lib
from .pyramid import build_index_pyramid


def depth_to_cloud(depth, K, cam_scale=1.0):
    """depth [B,H,W] float32 GPU tensor (raw units), K [B,3,3] (or [3,3]) intrinsics,
    -> dpt_xyz [B,3,H,W] float32: x = (col - K[0,2]) * d / K[0,0], y = (row - K[1,2]) * d / K[1,1], z = d
    with d = depth / cam_scale, zero where d <= 1e-8."""
    if not depth.is_cuda:
        raise _lib.FFB6DNativeError("depth_to_cloud needs a GPU tensor (no CPU fallback)")
    if depth.dim() != 3 or depth.dtype != torch.float32:
        raise TypeError("depth must be float32 [B,H,W]")
    lib = _lib.load()
    d = depth.contiguous()
    B, H, W = d.shape
    Kd = torch.as_tensor(K, dtype=torch.float64, device=d.device)
    if Kd.dim() == 2:
        Kd = Kd.unsqueeze(0).expand(B, 3, 3)
    Kd = Kd.contiguous()
    if Kd.shape != (B, 3, 3):
        raise ValueError(f"K must be [3,3] or [{B},3,3]")
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device), _lib.traced("depth_to_cloud", 16 * d.numel(), (H, W)):
        rc = lib.ffb6d_depth_to_cloud_f32(d.data_ptr(), Kd.data_ptr(), float(cam_scale), out.data_ptr(), B, H, W,
                                          torch.cuda.current_stream(d.device).cuda_stream)
    _lib.check(rc, "ffb6d_depth_to_cloud_f32")
    return out


def sample_choose(depth, n_points, generator=None, min_points=400):
    """Per frame: indices (into H*W) of `n_points` valid-depth pixels in uniformly random order;
    frames with fewer valid pixels are padded by wrapping around (np.pad(..., 'wrap'),
    linemod_dataset.py:276-277).  Returns int64 [B,1,n_points] (the model's `choose`).
    Raises if a frame has fewer than `min_points` valid pixels (the reference skips such frames)."""
    B = depth.shape[0]
    flat = depth.reshape(B, -1)
    out = torch.empty((B, 1, n_points), dtype=torch.int64, device=depth.device)
    for b in range(B):
        valid = torch.nonzero(flat[b] > 1e-6, as_tuple=False).squeeze(1)
        n = valid.numel()
        if n < min_points:
            raise ValueError(f"frame {b}: only {n} valid depth pixels")
        if n >= n_points:
            pick = valid[torch.randperm(n, device=depth.device, generator=generator)[:n_points]]
        else:
            wrapped = valid[torch.arange(n_points, device=depth.device) % n]
            pick = wrapped[torch.randperm(n_points, device=depth.device, generator=generator)]
        out[b, 0] = pick
    return out


def assemble_inputs(rgb, depth, normals, K, n_points, cam_scale=1.0, generator=None, index_dtype=torch.int64):
    """rgb [B,3,H,W] (uint8 or float), depth [B,H,W] f32, normals [B,3,H,W] f32, K intrinsics ->
    the complete input dict of FFB6D.forward, everything computed on the device."""
    dpt_xyz = depth_to_cloud(depth, K, cam_scale)
    choose = sample_choose(depth / cam_scale, n_points, generator)
    B, _, H, W = dpt_xyz.shape
    idx = choose.expand(B, 3, n_points)
    cld_c = torch.gather(dpt_xyz.reshape(B, 3, H * W), 2, idx)                      # [B,3,N]
    rgb_f = rgb.float()
    rgb_pt = torch.gather(rgb_f.reshape(B, 3, H * W), 2, idx)
    nrm_pt = torch.gather(normals.reshape(B, 3, H * W), 2, idx)
    inputs = {
        'rgb': rgb_f,
        'cld_rgb_nrm': torch.cat([cld_c, rgb_pt, nrm_pt], dim=1).contiguous(),     # [B,9,N]
        'choose': choose,
        'dpt_xyz': dpt_xyz,
    }
    inputs.update(build_index_pyramid(cld_c.transpose(1, 2).contiguous(), dpt_xyz, index_dtype=index_dtype))
    return inputs
