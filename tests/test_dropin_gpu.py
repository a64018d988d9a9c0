"""The operator-level drop-in on the device, inside a WHOLE model (the reference tree is not on the GPU box; its classes are patched and
run in tests/test_model_cpu.py / tests/test_simt_cpu.py where /root/reference exists): the reference-shaped stand-in of
ffb6d_amd/dropin.py -- pinned to the reference's own end_points on the CPU, tests/test_dropin_cpu.py -- with `patch.patch_classes`
applied runs the channel-major HIP operators (ffb6d_amd.ops) and must agree with its unpatched plain-torch self on the same device."""
import numpy as np
import pytest
import torch

from ffb6d_amd import _lib, dropin, patch, pyramid, synth
from test_forward_gpu import build, assert_close_scaled

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config,B,n_pts,H,W,n_cls", [(7, 2, 1024, 120, 160, 5), (2, 2, 12288, 480, 640, 22), (3, 1, 1100, 136, 168, 3)])
def test_patched_stand_in_equals_its_plain_torch_self(device, config, B, n_pts, H, W, n_cls):
    net = build(n_cls, n_pts, device)
    fr = synth.make_batch(config, B, n_points=n_pts, height=H, width=W)
    inputs = {"rgb": torch.from_numpy(fr["rgb"]).to(device).float(), "cld_rgb_nrm": torch.from_numpy(fr["cld_rgb_nrm"]).to(device),
              "choose": torch.from_numpy(fr["choose"]).to(device).long()}
    inputs.update(pyramid.build_index_pyramid(torch.from_numpy(fr["cld"]).to(device), torch.from_numpy(fr["dpt_xyz"]).to(device)))
    stand = dropin.FFB6D(net)
    with torch.no_grad():
        want = stand(inputs)
        undo = patch.patch_classes(dropin.FFB6D, dropin.Building_block, dropin.Att_pooling)
        tracer = _lib.Tracer(None)
        _lib.TRACER = tracer
        try:
            got = stand(inputs)
            torch.cuda.synchronize() if device.type == "cuda" else None
        finally:
            _lib.TRACER = None
            undo()
    launched = tracer.summary()
    for op in ("random_sample", "nearest_interpolation", "gather_neighbour", "relative_pos_encoding", "att_pool"):
        assert launched.get(op, {}).get("launches", 0) > 0, (op, sorted(launched))
    # 8 gathers of neighbour features + 4 position encodings + 8 poolings + 11 max-pool gathers + 11 interpolations per forward
    assert launched["att_pool"]["launches"] == 8 and launched["random_sample"]["launches"] == 11
    for k in want:
        assert_close_scaled(got[k].cpu().numpy(), want[k].cpu().numpy(), 1e-5, k)
