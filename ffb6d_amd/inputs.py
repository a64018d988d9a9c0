"""On-device input pipeline in front of the index pyramid: the
parts of the reference's `Dataset.get_item` that turn a depth image into the network's geometric
inputs, so a frame needs no host-side geometry at all.

    depth_to_cloud   == dpt_2_pcld                 linemod_dataset.py:188-199 (+ NaN/Inf -> 0, :258-259)
    sample_choose    == valid-pixel sampling       linemod_dataset.py:262-282 (N valid pixels, shuffled;
                                                    'wrap' padding when fewer than N are valid)
    assemble_inputs  == cld / cld_rgb_nrm / choose + build_index_pyramid   linemod_dataset.py:284-353

Sampling hashes (seed, frame, pixel) into sort keys instead of drawing from numpy's global RNG, so it is
distribution-equivalent, not bit-identical, to the reference (any N-subset of the valid pixels in uniformly random
order); csrc/inputs.hip has the details.
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


def depth_normal(depth_mm, fx, fy, k_size=5, distance_threshold=2000, difference_threshold=20, point_into_surface=False):
    """Surface normals of a depth image, the algorithm of `normalSpeed.depth_normal` with its argument order
    (linemod_dataset.py:252-254: `normalSpeed.depth_normal(dpt_mm, K[0][0], K[1][1], 5, 2000, 20, False)`).
    depth_mm [B,H,W] (or [H,W]) in millimetres: float32 (truncated to uint16 like `.astype(np.uint16)`), int16/uint16 bit
    patterns are taken as uint16.  Returns float32 [B,3,H,W] ([3,H,W]) -- the channel-major planes `assemble_inputs`
    consumes (the reference's map is [H,W,3]: `.permute(0, 2, 3, 1)`).  normalSpeed is not available here: parity is
    checked against the CPU restatement of the published algorithm only (csrc/inputs.hip)."""
    if not depth_mm.is_cuda:
        raise _lib.FFB6DNativeError("depth_normal needs a GPU tensor (no CPU fallback)")
    lib = _lib.load()
    single = depth_mm.dim() == 2
    d = (depth_mm.unsqueeze(0) if single else depth_mm).contiguous()
    if d.dim() != 3:
        raise ValueError("depth_mm must be [B,H,W] or [H,W]")
    if d.dtype in (torch.int16, torch.uint16):
        is_u16 = 1
    elif d.dtype == torch.float32:
        is_u16 = 0
    else:
        raise TypeError(f"depth_mm must be float32 or (u)int16, got {d.dtype}")
    B, H, W = d.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device), _lib.traced("depth_normal", d.element_size() * d.numel() + 12 * d.numel(), (H, W)):
        rc = lib.ffb6d_depth_normal(d.data_ptr(), is_u16, float(fx), float(fy), int(k_size), int(distance_threshold),
                                    int(difference_threshold), 1 if point_into_surface else 0, out.data_ptr(), B, H, W,
                                    torch.cuda.current_stream(d.device).cuda_stream)
    _lib.check(rc, "ffb6d_depth_normal")
    return out[0] if single else out


def fill_missing(dpt, cam_scale, scale_2_80m=1.0, max_depth=3.0):
    """Basic_Utils.fill_missing(dpt, cam_scale, scale_2_80m) with its default fill_type='multiscale', extrapolate=False,
    blur_type='bilateral' (utils/basic_utils.py:467-487 -> fill_in_multiscale, depth_map_utils_ycb.py:290-445): dense depth
    from a depth image with holes, same unit in and out.  dpt [B,H,W] (or [H,W]) float32 on the GPU.  OpenCV is not
    available here: parity is checked against the CPU restatement only (oracle/holefill_ref.py)."""
    if not dpt.is_cuda:
        raise _lib.FFB6DNativeError("fill_missing needs a GPU tensor (no CPU fallback)")
    lib = _lib.load()
    single = dpt.dim() == 2
    d = (dpt.unsqueeze(0) if single else dpt).float().contiguous()
    if d.dim() != 3:
        raise ValueError("dpt must be [B,H,W] or [H,W]")
    B, H, W = d.shape
    out = torch.empty_like(d)
    wbytes = lib.ffb6d_fill_missing_workspace_bytes(B, H, W)
    ws = torch.empty((wbytes,), dtype=torch.uint8, device=d.device)
    with torch.cuda.device(d.device), _lib.traced("fill_missing", 8 * d.numel(), (H, W)):
        rc = lib.ffb6d_fill_missing_f32(d.data_ptr(), float(cam_scale), float(scale_2_80m), float(max_depth), out.data_ptr(),
                                        B, H, W, ws.data_ptr(), wbytes, torch.cuda.current_stream(d.device).cuda_stream)
    _lib.check(rc, "ffb6d_fill_missing_f32")
    return out[0] if single else out


def _seed(seed, generator):
    if seed is not None:
        return int(seed) & 0xFFFFFFFFFFFFFFFF
    # a draw from a CPU generator (torch's default one when none is given) never touches the device queue
    return int(torch.randint(0, 2 ** 62, (1,), generator=generator,
                             device=generator.device if generator is not None else "cpu").item())


def sample_points(depth, n_points, dpt_xyz=None, rgb=None, normals=None, seed=None, generator=None, min_depth=1e-6):
    """Valid-pixel sampling (+ point assembly) of Dataset.get_item, linemod_dataset.py:262-289, on the device and without a
    host round trip: per frame a uniformly random `n_points`-subset of the pixels with depth > min_depth in uniformly random
    order ('wrap'-style repetition when a frame has fewer valid pixels).  depth [B,H,W] float32.
    Returns a dict: choose int64 [B,1,N] (the model's `choose`), n_valid int32 [B], and -- when dpt_xyz [B,3,H,W],
    rgb [B,3,H,W] (uint8 or float32) and normals [B,3,H,W] are given -- cld [B,N,3] and cld_rgb_nrm [B,9,N]."""
    if not depth.is_cuda:
        raise _lib.FFB6DNativeError("sample_points needs GPU tensors (no CPU fallback)")
    if depth.dim() != 3 or depth.dtype != torch.float32:
        raise TypeError("depth must be float32 [B,H,W]")
    lib = _lib.load()
    d = depth.contiguous()
    B, H, W = d.shape
    N = int(n_points)
    dev = d.device
    out = {"choose": torch.empty((B, 1, N), dtype=torch.int64, device=dev),
           "n_valid": torch.empty((B,), dtype=torch.int32, device=dev)}
    gather = dpt_xyz is not None
    xyz = rgb_c = nrm = None
    is_u8 = 0
    if gather:
        if rgb is None or normals is None:
            raise ValueError("dpt_xyz, rgb and normals come together")
        xyz, nrm = dpt_xyz.contiguous(), normals.contiguous()
        if rgb.dtype == torch.uint8:
            rgb_c, is_u8 = rgb.contiguous(), 1
        else:
            rgb_c = rgb.float().contiguous()
        for t in (xyz, rgb_c, nrm):
            if tuple(t.shape) != (B, 3, H, W):
                raise ValueError(f"image sources must be [B,3,H,W] = {(B, 3, H, W)}, got {tuple(t.shape)}")
        if xyz.dtype != torch.float32 or nrm.dtype != torch.float32:
            raise TypeError("dpt_xyz and normals must be float32")
        out["cld"] = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        out["cld_rgb_nrm"] = torch.empty((B, 9, N), dtype=torch.float32, device=dev)
    wbytes = lib.ffb6d_sample_points_workspace_bytes(B, H, W)
    ws = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None       # noqa: E731
    with torch.cuda.device(dev), _lib.traced("sample_points", 4 * d.numel() + 48 * B * N, (H, W, N)):
        rc = lib.ffb6d_sample_points_f32(d.data_ptr(), float(min_depth), ptr(xyz), ptr(rgb_c), is_u8, ptr(nrm),
                                         _seed(seed, generator), out["choose"].data_ptr(), ptr(out.get("cld")),
                                         ptr(out.get("cld_rgb_nrm")), out["n_valid"].data_ptr(), B, H, W, N,
                                         ws.data_ptr(), wbytes, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "ffb6d_sample_points_f32")
    return out


def sample_choose(depth, n_points, generator=None, min_points=400, seed=None):
    """Per frame: indices (into H*W) of `n_points` valid-depth pixels in uniformly random order (see sample_points).
    Returns int64 [B,1,n_points] (the model's `choose`).  Raises if a frame has fewer than `min_points` valid pixels (the
    reference skips such frames, linemod_dataset.py:264-268); pass min_points=0 to skip that check and its host sync."""
    res = sample_points(depth, n_points, seed=seed, generator=generator)
    if min_points:
        nv = res["n_valid"].cpu()
        if int(nv.min()) < min_points:
            raise ValueError(f"frame {int(nv.argmin())}: only {int(nv.min())} valid depth pixels")
    return res["choose"]


def assemble_inputs(rgb, depth, normals, K, n_points, cam_scale=1.0, generator=None, index_dtype=torch.int64, seed=None,
                    min_points=0):
    """rgb [B,3,H,W] (uint8 or float), depth [B,H,W] f32, normals [B,3,H,W] f32, K intrinsics ->
    the complete input dict of FFB6D.forward, everything computed on the device.
    Frames with too few valid depth pixels: the reference's dataset returns None for fewer than 400 (linemod_dataset.py:264-268);
    a frame with none at all would otherwise yield N copies of pixel 0.  min_points=400 reproduces the reference's rule (raises;
    costs one host synchronisation); with the default 0 nothing synchronises and the CALLER must drop the frames whose
    `n_valid` [B] (returned in the dict) is below its threshold before trusting their outputs."""
    dpt_xyz = depth_to_cloud(depth, K, cam_scale)
    pts = sample_points(depth / cam_scale, n_points, dpt_xyz, rgb, normals, seed=seed, generator=generator)
    if min_points:
        nv = pts["n_valid"].cpu()
        if int(nv.min()) < min_points:
            raise ValueError(f"frame {int(nv.argmin())}: only {int(nv.min())} valid depth pixels (< {min_points})")
    inputs = {
        'rgb': rgb.float(),
        'cld_rgb_nrm': pts["cld_rgb_nrm"],     # [B,9,N]
        'choose': pts["choose"],
        'dpt_xyz': dpt_xyz,
        'n_valid': pts["n_valid"],
    }
    inputs.update(build_index_pyramid(pts["cld"], dpt_xyz, index_dtype=index_dtype))
    return inputs
