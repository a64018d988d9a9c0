"""Wave-level kernels of the product library executed on the CPU by the SIMT emulator of tests/simt (fibers for threads,
rendezvous at MFMA / shuffle / barrier / buffer instructions): the point-major GEMM family of csrc/mlp_pm.hip in all of its
forms (tile kernels, K-split, stream form, LDS-tiled form; fp32 and bf16) and the fused attentive pooling, driven through the
package's own host wrappers (ffb6d_amd/ops_pm.py: argument marshalling, strides, per-frame gather bookkeeping) and checked
against float64.  The same cases run on the GPU in tests/test_pm_gpu.py; here they need no GPU."""
import os

import pytest
import torch

from ffb6d_amd import _lib, ops, ops_pm

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs ROCm's clang++ to build the emulated library")


def _ref(x1, w, bias, act, x2=None, add=None, gather=None, x1_gather=None):
    if x1_gather is not None:
        x1 = torch.gather(x1, 1, x1_gather.long().unsqueeze(2).expand(-1, -1, x1.shape[2]))
    x = x1 if x2 is None else torch.cat([x1, x2], dim=-1)
    y = x.double() @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    if gather is not None:
        Y, idx = gather
        y = y + torch.gather(Y.double(), 1, idx.long().unsqueeze(2).expand(-1, -1, Y.shape[2]))
    if add is not None:
        y = y + add.double()
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = torch.nn.functional.leaky_relu(y, 0.2)
    elif act == 3:
        y = torch.log_softmax(y, dim=-1)
    return y


# (B, P, K1, K2, Cout, act, py [>0 gathered epilogue rows, <0 added rows], tile_hint); sizes an emulated run finishes in a moment
CASES = [(2, 130, 24, 8, 37, 1, 13, h) for h in (1, 2, 3, 4, 5)]                      # every tile kernel, ragged rows / channels
CASES += [(1, 300, 64, 0, 64, 1, 0, 6), (2, 201, 24, 0, 16, 0, 0, 6), (1, 260, 32, 96, 100, 2, 70, 6), (1, 130, 128, 0, 128, 1, -1, 6),
          (2, 300, 64, 0, 64, 1, 37, 6), (3, 50, 32, 0, 24, 2, 9, 6)]  # stream form (the last two: Y rows fetched ahead of the tile prefetch)
CASES += [(1, 300, 96, 0, 40, 0, 0, 7), (2, 140, 32, 32, 72, 2, 50, 7), (1, 257, 256, 0, 200, 1, -1, 7), (8, 24, 512, 256, 256, 2, 0, 7)]  # LDS-tiled form
CASES += [(8, 48, 1024, 0, 64, 1, 0, 5), (1, 33, 8, 8, 5, 1, 0, 0), (1, 1, 8, 0, 8, 0, 0, 0), (2, 64, 128, 0, 22, 0, 0, 0)]      # K split, tiny, automatic choice


# tile-sequence form (hint 8 + 256 * tiles per workgroup): 2 / 3 / 5 tiles in a sequence, ragged last group, ragged rows and channels,
# exactly four steps per tile (K = 128) and more, two sources, gathered / added / no epilogue rows, every activation
SEQ = lambda t: 8 + 256 * t                                                             # noqa: E731
CASES += [(1, 300, 128, 0, 200, 1, 0, SEQ(2)), (2, 140, 64, 64, 384, 2, 50, SEQ(3)), (1, 257, 160, 0, 520, 1, -1, SEQ(2)),
          (1, 130, 256, 0, 520, 0, 33, SEQ(3)), (3, 70, 128, 32, 600, 1, 0, SEQ(5)), (1, 129, 128, 0, 264, 2, -1, SEQ(8)),
          # guided schedule: 29 point tiles x 3 channel tiles = regions of 8 (T = 3), 8 (T = 2) and 13 point tiles (single tiles)
          (1, 3650, 128, 0, 300, 1, 17, SEQ(3 | 2 << 4 | 1 << 8 | 1 << 16))]


@pytest.mark.parametrize("B,P,K1,K2,Cout,act,py,hint", CASES)
def test_mlp_pm_fp32_on_the_emulator_matches_fp64(emu, B, P, K1, K2, Cout, act, py, hint):
    g = torch.Generator().manual_seed(K1 + Cout + P)
    x1 = torch.randn(B, P, K1, generator=g)
    x2 = torch.randn(B, P, K2, generator=g) if K2 else None
    w = torch.randn(Cout, K1 + K2, generator=g) / (K1 + K2) ** 0.5
    bias = torch.randn(Cout, generator=g)
    gather = (torch.randn(B, py, Cout, generator=g), torch.randint(0, py, (B, P), generator=g)) if py > 0 else None
    add = torch.randn(B, P, Cout, generator=g) if py < 0 else None
    want = _ref(x1, w, bias, act, x2=x2, add=add, gather=gather)
    got = ops_pm.mlp(x1, w, bias, act, x2=x2, add=add, gather=gather, tile_hint=hint)
    assert got.shape == want.shape
    assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("hint", [2, 3, 6])
def test_log_softmax_epilogue_on_the_emulator(emu, hint):
    """pspnet.py:108-112 `final`: the wave's tile spans all channels, a point's channels sit in lanes l and l ^ 32"""
    cout = 32 if hint == 3 else 64
    g = torch.Generator().manual_seed(hint)
    x = torch.randn(1, 300, 64, generator=g)
    w = torch.randn(cout, 64, generator=g) / 8
    bias = torch.randn(cout, generator=g)
    want = _ref(x, w, bias, 3)
    got = ops_pm.mlp(x, w, bias, ops_pm.ACT_LOG_SOFTMAX, tile_hint=hint)
    assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("hint", [1, 2, 7])
@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
def test_operand_gather_on_the_emulator(emu, hint, idt):
    """`choose` (ffb6d.py:309-312): the gather of the picked pixel rows IS the operand load of the head GEMM"""
    g = torch.Generator().manual_seed(5)
    B, M, P, K1, K2, Cout = 2, 500, 150, 32, 32, 48
    img = torch.randn(B, M, K1, generator=g)
    pts = torch.randn(B, P, K2, generator=g)
    pick = torch.randint(0, M, (B, P), generator=g).to(idt)
    w = torch.randn(Cout, K1 + K2, generator=g) / 8
    bias = torch.randn(Cout, generator=g)
    want = _ref(img, w, bias, 1, x2=pts, x1_gather=pick)
    got = ops_pm.mlp(img, w, bias, 1, x2=pts, x1_gather=pick, tile_hint=hint)
    assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("rows,cout,acts,sliced", [(517, 22, (1, 1, 0), True), (128, 3, (1, 2, 0), False), (33, 24, (0, 1, 1), True)])
def test_mlp_chain3_on_the_emulator(emu, rows, cout, acts, sliced):
    """csrc/mlp_chain.hip: tests/test_pm_gpu.py's own check with CPU tensors"""
    import test_pm_gpu as TPM
    TPM.test_mlp_chain3_matches_the_separate_layers(torch.device("cpu"), rows, cout, acts, sliced)


@pytest.mark.parametrize("rows,cout,acts,sliced", [(517, 22, (1, 1, 0), True), (128, 3, (1, 2, 0), False), (33, 24, (0, 1, 1), True)])
def test_mlp_chain3_bf16_on_the_emulator(emu, rows, cout, acts, sliced):
    """csrc/mlp_chain.hip: tests/test_pm_gpu.py's own check with CPU tensors"""
    import test_pm_gpu as TPM
    TPM.test_mlp_chain3_bf16_matches_the_separate_layers(torch.device("cpu"), rows, cout, acts, sliced)


@pytest.mark.parametrize("B,h,w,C,P,idt", [(2, 6, 8, 8, 40, torch.int64), (1, 5, 7, 16, 70, torch.int32), (3, 1, 1, 8, 4, torch.int64)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_upsampled_patch_rows_on_the_emulator(emu, B, h, w, C, P, idt, dt):
    """forward_pm.LAST_STAGE_AT_CHOSEN: tests/test_pm_gpu.py's own check with CPU tensors"""
    import test_pm_gpu as TPM
    TPM.test_upsampled_patch_rows_are_the_unfolded_upsampled_map(torch.device("cpu"), B, h, w, C, P, idt, dt)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_stacked_head_gemm_on_the_emulator(emu, dt):
    """forward_pm.HEADS_SHARE_FIRST: tests/test_pm_gpu.py's own check with CPU tensors"""
    import test_pm_gpu as TPM
    TPM.test_stacked_head_gemm_equals_the_separate_launches(torch.device("cpu"), dt)


def test_kernel_forms_are_bit_identical_on_the_emulator(emu):
    """DESIGN.md 4a (forms 2 and 3): stream form and LDS-tiled form feed every accumulator the same products in the same k order as the
    tile kernels -- equal bits (the emulated MFMA is deterministic, so any difference would be one of operand order)"""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 300, 128, generator=g)
    w = torch.randn(96, 128, generator=g) / 11
    bias = torch.randn(96, generator=g)
    outs = [ops_pm.mlp(x, w, bias, 1, tile_hint=h) for h in (1, 2, 4, 6, 7)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # the tile-sequence form (round 5): deferred epilogue pieces, same sums -- with gathered rows, without a bias, identity activation
    # (negative zeros must survive the additions of "nothing")
    x = torch.randn(2, 150, 160, generator=g)
    w = torch.randn(392, 160, generator=g) / 12
    bias = torch.randn(392, generator=g)
    Y, idx = torch.randn(2, 40, 392, generator=g), torch.randint(0, 40, (2, 150), generator=g)
    for kw in ({"gather": (Y, idx)}, {"add": Y[:, idx[0]]}, {}):
        for b, act in ((bias, 2), (None, 0)):
            want = ops_pm.mlp(x, w, b, act, tile_hint=7, **kw)
            for t in (2, 3, 4):
                got = ops_pm.mlp(x, w, b, act, tile_hint=8 + 256 * t, **kw)
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (list(kw), act, t)
    # round 6: balanced contiguous sequences per XCD (plan 0xF0 | rounds | sequences-per-XCD << 8): 10 point tiles -> XCDs 0 and 1 hold two
    # point tiles = 8 tiles each, cut into 3 sequences of 3 + 3 + 2 (they run from one point tile into the next) or 5 of 2 + 2 + 2 + 1 + 1
    x = torch.randn(2, 620, 160, generator=g)
    Y, idx = torch.randn(2, 40, 392, generator=g), torch.randint(0, 40, (2, 620), generator=g)
    for kw in ({"gather": (Y, idx)}, {}):
        for b, act in ((bias, 2), (None, 0)):
            want = ops_pm.mlp(x, w, b, act, tile_hint=7, **kw)
            for per_xcd in (3, 5, 0):
                got = ops_pm.mlp(x, w, b, act, tile_hint=8 + 256 * (0xF0 | 2 | per_xcd << 8), **kw)
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (list(kw), act, per_xcd)


@pytest.mark.parametrize("hint", [1, 2, 5, 6, 7])
def test_mlp_pm_bf16_on_the_emulator(emu, hint):
    """bf16 rows, fp32 accumulation (v_mfma_f32_32x32x16_bf16): against float64 on the bf16-rounded operands, bar = one bf16
    rounding of the output"""
    g = torch.Generator().manual_seed(hint)
    BF = torch.bfloat16
    x1 = torch.randn(2, 140, 64, generator=g).to(BF)
    x2 = torch.randn(2, 140, 64, generator=g).to(BF)
    w = (torch.randn(80, 128, generator=g) / 11).to(BF)
    bias = torch.randn(80, generator=g)
    gather = (torch.randn(2, 30, 80, generator=g).to(BF), torch.randint(0, 30, (2, 140), generator=g))
    want = _ref(x1, w, bias, 2, x2=x2, gather=gather)
    got = ops_pm.mlp(x1, w, bias, 2, x2=x2, gather=gather, tile_hint=hint)
    assert got.dtype == BF
    assert float((got.double() - want).abs().max()) <= 1e-2 * float(want.abs().max())


@pytest.mark.parametrize("K1,K2,cout,rows", [(256, 0, 72, 300), (128, 64, 130, 200), (64, 0, 128, 129), (1024, 0, 40, 64)])
def test_lds_tiled_bf16_prefetch_past_the_last_step(emu, K1, K2, cout, rows):
    """mlp_pm_lds_kernel<bf16> fetches three steps ahead with branch-free loads: steps past the last one read at an
    out-of-range offset; one to sixteen steps, one or two sources, ragged rows and channels; also equal to the tile kernel"""
    g = torch.Generator().manual_seed(K1 + cout)
    BF = torch.bfloat16
    x1 = torch.randn(1, rows, K1, generator=g).to(BF)
    x2 = torch.randn(1, rows, K2, generator=g).to(BF) if K2 else None
    w = (torch.randn(cout, K1 + K2, generator=g) / (K1 + K2) ** 0.5).to(BF)
    bias = torch.randn(cout, generator=g)
    want = _ref(x1, w, bias, 1, x2=x2)
    got = ops_pm.mlp(x1, w, bias, 1, x2=x2, tile_hint=7)
    assert float((got.double() - want).abs().max()) <= 1e-2 * float(want.abs().max())
    assert torch.equal(got, ops_pm.mlp(x1, w, bias, 1, x2=x2, tile_hint=1))


@pytest.mark.parametrize("K1,K2,cout,rows,extra", [(256, 0, 784, 300, "gather"), (128, 64, 144, 520, "add"), (64, 64, 528, 257, None),
                                                   (192, 0, 80, 64, "gather"), (128, 128, 48, 700, "xgather")])
def test_big_tile_bf16_form_on_the_emulator(emu, K1, K2, cout, rows, extra):
    """mlp_pm_big_kernel (csrc/mlp_pm_big.hip, tile_hint 9): 256 x 256 tile, LDS-DMA operand loads into chunk-permuted images; ragged
    rows and channels (zero rows past the end), one or two sources, gathered / added epilogue rows, gathered operand rows; against
    float64 and bit-identical to the LDS-tiled form (same products in the same k order per accumulator)"""
    g = torch.Generator().manual_seed(K1 + cout + rows)
    BF = torch.bfloat16
    B = 2 if rows % 2 == 0 else 1
    P = rows // B
    kw = {}
    if extra == "xgather":
        x1 = torch.randn(B, 90, K1, generator=g).to(BF)
        kw["x1_gather"] = torch.randint(0, 90, (B, P), generator=g)
    else:
        x1 = torch.randn(B, P, K1, generator=g).to(BF)
    x2 = torch.randn(B, P, K2, generator=g).to(BF) if K2 else None
    w = (torch.randn(cout, K1 + K2, generator=g) / (K1 + K2) ** 0.5).to(BF)
    bias = torch.randn(cout, generator=g)
    if extra == "gather":
        kw["gather"] = (torch.randn(B, 30, cout, generator=g).to(BF), torch.randint(0, 30, (B, P), generator=g))
    elif extra == "add":
        kw["add"] = torch.randn(B, P, cout, generator=g).to(BF)
    got = ops_pm.mlp(x1, w, bias, 2, x2=x2, tile_hint=9, **kw)
    assert torch.equal(got.view(torch.int16), ops_pm.mlp(x1, w, bias, 2, x2=x2, tile_hint=7, **kw).view(torch.int16))
    for tpg in ((1, 2, 3, 4) if rows <= 300 else (2,)):   # tiles per workgroup (csrc/mlp_pm_big.hip: a sequence of channel tiles of one point tile)
        assert torch.equal(got.view(torch.int16), ops_pm.mlp(x1, w, bias, 2, x2=x2, tile_hint=9 + 256 * tpg, **kw).view(torch.int16)), tpg
    if rows == 64:
        assert torch.equal(got.view(torch.int16), ops_pm.mlp(x1, w, None, 0, x2=x2, tile_hint=9, **kw).view(torch.int16)) is False
        nob = ops_pm.mlp(x1, w, None, 0, x2=x2, tile_hint=9)         # no bias, no Y, identity: negative zeros survive the additions of "nothing"
        assert torch.equal(nob.view(torch.int16), ops_pm.mlp(x1, w, None, 0, x2=x2, tile_hint=7).view(torch.int16))
    want = _ref(x1, w, bias, 2, x2=x2, **kw)
    assert float((got.double() - want).abs().max()) <= 1e-2 * float(want.abs().max())


@pytest.mark.parametrize("B,N,C1,C2,idt", [(2, 50, 16, 16, torch.int64), (1, 37, 32, 32, torch.int32), (1, 20, 64, 64, torch.int64),
                                           (3, 9, 8, 24, torch.int64)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_att_pool_pm_on_the_emulator(emu, B, N, C1, C2, idt, dt):
    """Att_pooling.forward up to the pooled tensor (RandLANet.py:243-248) with the neighbour gather fused in: the softmax over
    the 16 neighbours is in-lane arithmetic on the accumulator registers (slot permutation of csrc/mlp_pm.hip)"""
    if dt == torch.bfloat16 and (C1 % 16 or C2 % 16):
        pytest.skip("bf16 rows: 16-channel operands")
    g = torch.Generator().manual_seed(N + C1)
    f = torch.randn(B, N, C1, generator=g).to(dt)
    nei = torch.randint(0, N, (B, N, 16), generator=g).to(idt)
    gp = torch.randn(B, N, 16, C2, generator=g).to(dt)
    d = C1 + C2
    w = (torch.randn(d, d, generator=g) / d ** 0.5).to(dt)
    got = ops_pm.att_pool(f, nei, gp, w)
    fn = torch.gather(f.double().unsqueeze(1).expand(-1, N, -1, -1), 2, nei.long().unsqueeze(3).expand(-1, -1, -1, C1))   # [B,N,16,C1]
    s = torch.cat([fn, gp.double()], dim=3)                                                                                 # feature set
    want = (s * torch.softmax(s @ w.double().t(), dim=2)).sum(dim=2)
    tol = 1e-5 if dt == torch.float32 else 2e-2
    assert float((got.double() - want).abs().max()) <= tol * float(want.abs().max())


def _lfa_case(B, N, d, mode, dt, idt, seed):
    g = torch.Generator().manual_seed(seed)
    h, cout = d // 2, (d // 2 if mode == 1 else d)
    a = dict(xyz=torch.rand(B, N, 3, generator=g), nei=torch.randint(0, N, (B, N, 16), generator=g).to(idt),
             f=torch.randn(B, N, h, generator=g).to(dt), w1=torch.randn(h, 10, generator=g) / 2, b1=torch.randn(h, generator=g) / 2,
             wfc=(torch.randn(d, d, generator=g) / d ** 0.5 * 2).to(dt), wm=(torch.randn(cout, d, generator=g) / d ** 0.5).to(dt),
             bm=torch.randn(cout, generator=g) / 2)
    if mode == 2:
        a.update(w2=(torch.randn(h, h, generator=g) / h ** 0.5).to(dt), b2=torch.randn(h, generator=g) / 2)
    return a


# one launch per half of the local feature aggregation (csrc/lfa_pm.hip): both halves, both group sizes of every width, both
# index types, ragged tails (N not a multiple of the points per workgroup, groups that straddle two frames)
@pytest.mark.parametrize("B,N,d,p_hint", [(2, 70, 32, 1), (1, 37, 32, 2), (2, 41, 64, 1), (1, 19, 64, 2), (2, 13, 128, 1), (1, 9, 128, 2),
                                          (1, 7, 256, 1), (2, 3, 256, 2), (2, 70, 32, 9), (2, 41, 64, 10), (2, 45, 32, 3), (1, 23, 64, 11),
                                          (2, 45, 32, 4), (1, 23, 64, 4)])     # + 8: weights resident in LDS; 4: one wave per workgroup
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_fused_lfa_half_on_the_emulator(emu, B, N, d, p_hint, mode, dt):
    """Building_block.forward, RandLANet.py:196-214: gather + position encoding + mlp1 (+ mlp2) + attentive pooling + mlp in
    one kernel, pair rows in LDS only -- against the float64 restatement (oracle/ops_ref.lfa_half)"""
    from oracle import ops_ref
    if dt == torch.bfloat16 and (p_hint in (2, 3, 10, 11) or (d == 256 and mode == 1)):
        pytest.skip("bf16: one group size per width is enough on the emulator (the GPU suite runs them all)")
    idt = torch.int64 if (N + mode) % 2 else torch.int32
    a = _lfa_case(B, N, d, mode, dt, idt, seed=N + d + mode)
    kw = dict(w2=a["w2"], b2=a["b2"], act2=2) if mode == 2 else {}
    got = ops_pm.lfa_half(mode, a["xyz"], a["nei"], a["f"], a["w1"], a["b1"], 2, a["wfc"], a["wm"], a["bm"], 2, p_hint=p_hint, **kw)
    store = None if dt == torch.float32 else (lambda t: t.to(torch.bfloat16).to(torch.float64))
    want = ops_ref.lfa_half(mode, a["xyz"], a["nei"], a["f"], a["w1"], a["b1"], 2, a["wfc"], a["wm"], a["bm"], 2, store=store, **kw)
    assert got.shape == want.shape and got.dtype == dt
    tol = 1e-5 if dt == torch.float32 else 2e-2
    assert float((got.double() - want).abs().max()) <= tol * float(want.abs().max())


@pytest.mark.parametrize("B,N,d,p_hint,mode", [(2, 300, 32, 2, 1), (2, 300, 32, 2, 2), (1, 45, 256, 2, 2), (3, 50, 64, 1, 1), (2, 150, 32, 4, 2),
                                               (2, 70, 64, 4, 1)])
def test_fused_lfa_persistent_loop_on_the_emulator(emu, monkeypatch, B, N, d, p_hint, mode):
    """the software pipeline over the point groups of a workgroup (indices two groups ahead, gathered rows one group ahead):
    one workgroup per XCD, so every workgroup walks several groups, the last ones ragged"""
    from oracle import ops_ref
    a = _lfa_case(B, N, d, mode, torch.float32, torch.int64, seed=N + d)
    kw = dict(w2=a["w2"], b2=a["b2"], act2=2) if mode == 2 else {}
    got = ops_pm.lfa_half(mode, a["xyz"], a["nei"], a["f"], a["w1"], a["b1"], 2, a["wfc"], a["wm"], a["bm"], 2, p_hint=p_hint + (1 << 8), **kw)
    want = ops_ref.lfa_half(mode, a["xyz"], a["nei"], a["f"], a["w1"], a["b1"], 2, a["wfc"], a["wm"], a["bm"], 2, **kw)
    assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_fused_lfa_reads_a_level_as_the_prefix_of_a_finer_coordinate_table(emu):
    """cld_xyz{i+1} is the first quarter of cld_xyz{i} (linemod_dataset.py:322-323): the coarser level gathers its coordinates
    from the finer level's table through the frame stride, no copy"""
    a = _lfa_case(3, 40, 32, 1, torch.float32, torch.int64, seed=5)
    big = torch.rand(3, 160, 3)
    big[:, :40] = a["xyz"]
    table = ops_pm.xyz_table(big)
    args = (a["nei"], a["f"], a["w1"], a["b1"], 2, a["wfc"], a["wm"], a["bm"], 2)
    assert torch.equal(ops_pm.lfa_half(1, table[:, :40], *args), ops_pm.lfa_half(1, a["xyz"], *args))


def test_emulated_library_reports_argument_errors_like_the_product(emu):
    x = torch.randn(1, 10, 12)                       # K = 12 is not a multiple of 8
    w = torch.randn(8, 12)
    with pytest.raises(_lib.FFB6DNativeError, match="multiples of 8"):
        ops_pm.mlp(x, w)
    assert ops.ACT_RELU == 1


# ---------------------------------------------------------------------------------------------------------------
# launch paths of the per-thread kernels (csrc/upconv.hip, csrc/posenc.hip): grid arithmetic, XCD-band workgroup order and
# kernel-form selection run here exactly as on the GPU (their bodies alone: tests/test_hostsim_cpu.py)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,cin,cout,h,w", [(2, 16, 8, 6, 8), (1, 8, 64, 3, 10), (3, 8, 12, 5, 7), (1, 8, 8, 1, 1)])
def test_folded_up_block_through_the_real_launchers(emu, B, cin, cout, h, w):
    """forward_pm.up_block in the folded form (z GEMM on the emulated MFMA kernel + upconv_combine through its launcher:
    blocked form where the map is an exact x2 with OW % 4 == 0, one-pixel form elsewhere) against torch's modules"""
    from ffb6d_amd import forward_pm, model
    g = torch.Generator().manual_seed(cin + cout + h)
    ub = model.UpBlock(cin, cout).eval()
    with torch.no_grad():
        for p in ub.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        ub.conv[2].running_mean.copy_(torch.randn(cout, generator=g) * 0.1)
        ub.conv[2].running_var.copy_(torch.rand(cout, generator=g) + 0.5)
        ub.conv[3].weight.fill_(0.3)
    x = torch.randn(B, cin, h, w, generator=g)
    with torch.no_grad():
        want = ub.conv(x)
        got = forward_pm.up_block(ub, x.permute(0, 2, 3, 1).contiguous())
    assert got.shape == (B, 2 * h, 2 * w, cout)
    assert float((got.permute(0, 3, 1, 2) - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("B,C,h,w,dt", [(2, 16, 6, 8, torch.float32), (1, 64, 9, 10, torch.bfloat16), (1, 32, 1, 1, torch.bfloat16),
                                        (1, 48, 5, 17, torch.float32)])
def test_upconv_combine_forms_are_bit_identical_on_the_emulator(emu, B, C, h, w, dt):
    """csrc/upconv.hip: the LDS-staged form (8 x 16 output pixels of a 64-byte channel chunk per workgroup, window staged with LDS-DMA
    loads; round 6) against the per-thread forms -- ragged tiles, one-pixel maps, several chunks, both precisions: equal bits"""
    lib = _lib.load()
    g = torch.Generator().manual_seed(C + h + w)
    z = torch.randn(B, h, w, 9 * C, generator=g).to(dt)
    z[0, h // 2, w // 2, 5] = float("nan")                     # a NaN stays confined to the outputs whose taps blend it
    shift = torch.randn(C, generator=g)
    outs = []
    try:
        for form in (3, 2, 1, 0):
            lib.ffb6d_upconv_set_form(form)
            outs.append(ops_pm.upconv_combine(z, shift, 0.25, (2 * h, 2 * w)))
    finally:
        lib.ffb6d_upconv_set_form(2)
    for o in outs[1:]:
        assert torch.equal(torch.isnan(o), torch.isnan(outs[0]))
        assert torch.equal(torch.nan_to_num(o).view(torch.uint8), torch.nan_to_num(outs[0]).view(torch.uint8))
    assert 0 < int(torch.isnan(outs[0]).sum()) < outs[0].numel() // 2


@pytest.mark.parametrize("B,N,K,cout,dt", [(2, 40, 16, 16, torch.float32), (1, 33, 16, 128, torch.bfloat16), (3, 17, 5, 24, torch.float32)])
def test_posenc_mlp_through_the_real_launcher(emu, B, N, K, cout, dt):
    from oracle import ops_ref
    g = torch.Generator().manual_seed(N + cout)
    xyz = torch.randn(B, N, 3, generator=g)
    idx = torch.randint(0, N, (B, N, K), generator=g)
    w = torch.randn(cout, 10, generator=g) * 0.5
    bias = torch.randn(cout, generator=g)
    want = torch.relu(ops_ref.relative_pos_encoding(xyz, idx).double() @ w.double().t() + bias.double())
    got = ops_pm.posenc_mlp(xyz, idx, w, bias, 1, dtype=dt)
    assert got.shape == (B, N, K, cout) and got.dtype == dt
    assert float((got.double() - want).abs().max()) <= (1e-5 if dt == torch.float32 else 1e-2) * float(want.abs().max())


# ---------------------------------------------------------------------------------------------------------------
# the whole fused inference forward on the emulator
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,n_pts,height,width", [("fp32", 1024, 120, 160), ("bf16", 1024, 120, 160), ("fp32", 1100, 104, 136)])
def test_whole_fused_forward_on_the_emulator_matches_the_plain_torch_restatement(emu, precision, n_pts, height, width):
    """forward_pm.forward (ffb6d.py:203-337 on point-major rows) with EVERY hand-written kernel of the inference path run
    from its product source on the CPU -- GEMM forms, fused attentive pooling, row gathers / max pooling, fused position
    encoding, BatchNorm glue, pyramid pooling, folded up-convolution -- on a 120x160 frame with 1024 points, the index pyramid
    from the CPU oracle, dense 3x3 convolutions on torch-CPU; against oracle/forward_ref.py: all three outputs and both
    embeddings after each of the 7 fusion stages, fp32 at the GPU suite's bar (1e-5 of the range), bf16 at its bf16 bars.
    The third case is ragged everywhere, like the reference's own default of 12800 points (common.py: 480*640//24 -> levels
    12800 / 3200 / 800 / 200 / 50): two frames of 1100 / 275 / 68 / 17 points and 13 x 17 ... 104 x 136 pixel maps -- no level is a multiple of a
    tile, a wave or a 16-row group, so every kernel runs its partial-tile paths (buffer range checks, row masks)."""
    import json
    import numpy as np
    from ffb6d_amd import forward_pm, model, synth
    from oracle import forward_ref
    from oracle import knn as oknn
    from oracle import pyramid as opyr
    from conftest import GOLDEN
    frames = synth.make_batch(9, 2 if n_pts != 1024 else 1, n_points=n_pts, height=height, width=width)      # the ragged case with two frames: per-frame strides
    inputs = {"rgb": torch.from_numpy(frames["rgb"]).float(), "cld_rgb_nrm": torch.from_numpy(frames["cld_rgb_nrm"]),
              "choose": torch.from_numpy(frames["choose"]).long()}
    for k, v in opyr.build_batch(frames, oknn.knn_search).items():
        inputs[k] = torch.from_numpy(v.astype(np.int64) if v.dtype == np.int32 else v)
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as fh:
        sd = synth.synth_state_dict_from_shapes(json.load(fh), seed=0, n_classes=5)
    net = model.FFB6D(n_classes=5, n_pts=n_pts)
    net.load_state_dict(sd)
    net.eval()
    net.precision = precision
    keep = forward_pm.UPCONV_FOLD
    forward_pm.UPCONV_FOLD = None                      # folded up-convolution in both precisions
    taps, ref_taps = {}, {}
    try:
        with torch.no_grad():
            ep = forward_pm.forward(net, inputs, {}, two_streams=False, taps=taps)
            ref = forward_ref.ffb6d_forward(sd, inputs, taps=ref_taps)
    finally:
        forward_pm.UPCONV_FOLD = keep
    assert len(ref_taps) == 14 and sorted(taps) == sorted(ref_taps)
    for k, want in list(ref.items()) + sorted(ref_taps.items()):
        got = ep[k] if k in ep else taps[k]
        scale = float(want.abs().max())
        err = (got - want).abs()
        if precision == "fp32":
            assert float(err.max()) <= 1e-5 * scale, (k, float(err.max()) / scale)
        else:
            assert float(err.max()) <= 5e-2 * scale and float(err.mean()) <= 8e-3 * scale, (k, float(err.max()) / scale)


def test_head_forms_equal_the_dense_last_stage_on_the_emulator(emu):
    """the last colour stage at the picked pixels + the heads' forms against the dense evaluation of the same network
    (tests/test_forward_gpu.py's own check with CPU tensors; one ragged frame: 1100 points, 72 x 88 pixels -- two whole forwards)"""
    import test_forward_gpu as TF
    TF.test_head_forms_equal_the_dense_last_stage(torch.device("cpu"), "fp32", n_pts=1100, height=72, width=88, n_frames=1)


# ---------------------------------------------------------------------------------------------------------------
# the drop-in boundary, executed: the UNMODIFIED reference model with our operators patched in
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.reference
@pytest.mark.parametrize("rows", [False, True])
def test_reference_model_with_our_operators_patched_in_equals_the_reference(emu, rows):
    """patch.patch_reference on the reference's own FFB6D (ffb6d/models/ffb6d.py, RandLANet.py): random_sample,
    nearest_interpolation, gather_neighbour, relative_pos_encoding and Att_pooling.forward become ffb6d_amd.ops (the
    channel-major kernels of csrc/neighbour_ops.hip, run here on the emulator); everything else -- modules, weights, the
    forward's control flow -- stays the reference's.  Same frame, same weights: the patched forward must reproduce the
    unpatched one (gathers / max pooling exactly, the attentive pooling's softmax-sum to fp32 rounding).
    rows=True: the channels-last operators of ffb6d_amd.ops_cl instead (what this package's training step uses) -- the reference's
    own permutes / views around them must keep working on their channels_last-strided outputs."""
    import numpy as np
    from ffb6d_amd import patch, synth
    from oracle import knn as oknn
    from oracle import pyramid as opyr
    from oracle import ref_harness as rh
    m_ffb6d, m_randla, _ = rh.reference_modules()
    frames = synth.make_batch(7, 1, n_points=1024, height=120, width=160)
    inputs = {"rgb": torch.from_numpy(frames["rgb"].astype(np.float32)), "cld_rgb_nrm": torch.from_numpy(frames["cld_rgb_nrm"]),
              "choose": torch.from_numpy(frames["choose"].astype(np.int64))}
    for k, v in opyr.build_batch(frames, oknn.knn_search).items():
        inputs[k] = torch.from_numpy(v.astype(np.int64) if v.dtype == np.int32 else v)
    ref_model = rh.build_reference_model(n_classes=5, n_pts=1024)
    ref_model.load_state_dict(synth.synth_state_dict(ref_model, 0))
    ref_model.eval()
    with torch.no_grad():
        want = {k: v.clone() for k, v in ref_model(dict(inputs)).items()}
    mp = pytest.MonkeyPatch()
    mp.setattr(ops, "_need_gpu", lambda *ts: None)
    mp.setattr(ops, "_stream", lambda t: None)
    calls = {}
    for name in ("ffb6d_random_sample_f32", "ffb6d_nearest_interpolation_f32", "ffb6d_gather_neighbour_f32",
                 "ffb6d_relative_pos_encoding_f32", "ffb6d_att_pool_f32", "ffb6d_random_sample_pm", "ffb6d_gather_rows_pm",
                 "ffb6d_att_pool_rows"):
        def counted(*a, _fn=getattr(emu, name), _name=name):
            calls[_name] = calls.get(_name, 0) + 1
            return _fn(*a)
        mp.setattr(emu, name, counted, raising=False)
    undo = patch.patch_reference(m_ffb6d, m_randla, rows=rows)
    try:
        with torch.no_grad():
            got = ref_model(dict(inputs))
    finally:
        undo()
        mp.undo()
    # the patched forward really went through the five native entry points (ffb6d.py:240-312, RandLANet.py:196-250):
    # 4 + 7 max poolings (sub-sampling, r2p), 4 + 7 interpolations (decoder, p2r), 2 x 4 feature gathers, 4 encodings, 2 x 4 attentive poolings
    if rows:                          # row kernels: every gather (interpolation or neighbour) is ffb6d_gather_rows_pm
        assert calls == {"ffb6d_random_sample_pm": 11, "ffb6d_gather_rows_pm": 19, "ffb6d_relative_pos_encoding_f32": 4,
                         "ffb6d_att_pool_rows": 8}, calls
    else:
        assert calls == {"ffb6d_random_sample_f32": 11, "ffb6d_nearest_interpolation_f32": 11, "ffb6d_gather_neighbour_f32": 8,
                         "ffb6d_relative_pos_encoding_f32": 4, "ffb6d_att_pool_f32": 8}, calls
    assert sorted(got) == sorted(want)
    for k in want:
        scale = float(want[k].abs().max())
        err = float((got[k] - want[k]).abs().max()) / scale
        print(k, "patched reference vs reference: max err / range %.2e" % err)
        assert err <= 1e-5, (k, err)


# ---------------------------------------------------------------------------------------------------------------
# exact KNN (the reference's C ABI knn_.h:4-27) on the emulator: Morton preparation, tile-pruned row search
# (one DPP row per query: 16-lane rendezvous), K = 1 pruned search, LDS scan, distance pick -- bit-exact indices
# ---------------------------------------------------------------------------------------------------------------
def test_knn_through_the_reference_c_abi_on_the_emulator_matches_the_reference_goldens(emu):
    """tests/golden/knn_small.npz = output of the reference's own nanoflann build (knn_.cxx:104-135): every case through
    cpp_knn_batch[_omp] with host pointers (what knn.pyx binds), executed by the emulated kernels"""
    import numpy as np
    from conftest import GOLDEN
    from ffb6d_amd import nearest_neighbors as nn
    from oracle import knn as oknn
    z = np.load(os.path.join(GOLDEN, "knn_small.npz"))
    for name in sorted({k.split("/")[0] for k in z.files}):
        sup, qry, K, want = z[name + "/support"], z[name + "/query"], int(z[name + "/K"]), z[name + "/idx"]
        for omp in (False, True):
            got = nn.knn_batch(sup, qry, K, omp=omp)
            assert got.dtype == np.int64
            np.testing.assert_array_equal(got, want, err_msg=name)         # the golden cases are tie-free
    one = nn.knn(z["self_k16/support"][0], z["self_k16/query"][0][:100], 16)
    np.testing.assert_array_equal(one, z["self_k16/idx"][0][:100])


@pytest.mark.parametrize("B,S,Q,K", [(1, 3000, 500, 16), (2, 2500, 700, 1), (1, 5000, 300, 5), (1, 100, 50, 16), (2, 4000, 4000, 2)])
def test_knn_kernels_on_the_emulator_are_bit_exact_against_the_oracle(emu, B, S, Q, K):
    """pruned searches (S >= 512 (2048 until round 5): row kernel for 2 <= K <= 16, lane-per-query kernel for K = 1) and the LDS scan, incl. a
    cloud with duplicated points and (0,0,0) pixels (the ties of linemod_dataset.py:198, 276-277: lowest index wins)"""
    import numpy as np
    from ffb6d_amd import nearest_neighbors as nn
    from oracle import knn as oknn
    g = np.random.default_rng(S + K)
    sup = g.standard_normal((B, S, 3)).astype(np.float32)
    sup[:, S // 2:S // 2 + 40] = sup[:, :40]                 # duplicates
    sup[:, -25:] = 0.0                                       # invalid-depth pixels
    qry = g.standard_normal((B, Q, 3)).astype(np.float32)
    qry[:, :10] = sup[:, :10]
    np.testing.assert_array_equal(nn.knn_batch(sup, qry, K, omp=True), oknn.knn_batch(sup, qry, K))


def test_knn_on_the_emulator_tie_rule_on_quantised_clouds(emu):
    """Property test (hypothesis, 12 seeded examples): coordinates quantised to a coarse lattice, so that most neighbourhoods
    contain exact distance ties and duplicated points -- the answer is only defined by nanoflann's rule (KNNResultSet::addPoint,
    nanoflann.hpp:115-139: among equal distances the lowest index wins), which the oracle restates.  Small sets take the LDS
    scan, S >= 512 the Morton-pruned kernels; K from 1 to 16 including K == S."""
    import numpy as np
    from hypothesis import example, given, settings, strategies as st
    from ffb6d_amd import nearest_neighbors as nn
    from oracle import knn as oknn

    @settings(max_examples=12, deadline=None, derandomize=True)
    @example(1, 2048, 8, 16, 3, 1)        # the Morton-pruned row kernel on a lattice of 27 distinct points
    @example(2, 2048, 8, 1, 2, 2)         # ... and the K = 1 kernel
    @example(3, 16, 5, 16, 2, 1)          # K == S
    @given(st.integers(0, 2 ** 31 - 1), st.sampled_from([16, 17, 63, 200, 512, 2048]), st.integers(1, 24), st.integers(1, 16),
           st.sampled_from([2, 3, 5, 9]), st.integers(1, 2))
    def check(seed, S, Q, K, levels, B):
        g = np.random.default_rng(seed)
        sup = (g.integers(0, levels, (B, S, 3)) / np.float32(levels)).astype(np.float32)
        qry = (g.integers(0, levels, (B, Q, 3)) / np.float32(levels)).astype(np.float32)
        K = min(K, S)
        np.testing.assert_array_equal(nn.knn_batch(sup, qry, K, omp=True), oknn.knn_batch(sup, qry, K))
    check()


@pytest.mark.reference
def test_knn_on_the_emulator_equals_the_reference_nanoflann(emu):
    """the reference's knn_.cxx + nanoflann compiled in place (oracle/_ref) on the same clouds, tie runs canonicalised"""
    import numpy as np
    from ffb6d_amd import nearest_neighbors as nn
    from oracle import knn as oknn
    from oracle import ref_harness as rh
    g = np.random.default_rng(3)
    sup = g.standard_normal((2, 3500, 3)).astype(np.float32)
    qry = g.standard_normal((2, 900, 3)).astype(np.float32)
    for K in (1, 16):
        got, want = nn.knn_batch(sup, qry, K, omp=True), rh.ref_knn_batch(sup, qry, K, omp=True)
        for b in range(sup.shape[0]):
            np.testing.assert_array_equal(oknn.canonical_ties(got[b], sup[b], qry[b])[0], oknn.canonical_ties(want[b], sup[b], qry[b])[0])


def test_distance_pick_on_the_emulator_matches_the_reference_goldens(emu, monkeypatch):
    """cpp_knn_batch_distance_pick (knn_.h:21-27, knn_.cxx:138-271; std::mt19937 restated in the kernel): the reference's own
    output with the clock pinned (tests/golden/knn_pick_small.npz)"""
    import numpy as np
    from conftest import GOLDEN
    from ffb6d_amd import nearest_neighbors as nn
    z = np.load(os.path.join(GOLDEN, "knn_pick_small.npz"))
    for name in sorted({k.split("/")[0] for k in z.files}):
        pts, K, seed = z[name + "/pts"], int(z[name + "/K"]), int(z[name + "/seed"])
        monkeypatch.setenv("FFB6D_KNN_PICK_SEED", str(seed))
        idx, q = nn.knn_batch_distance_pick(pts, z[name + "/idx"].shape[1], K)
        np.testing.assert_array_equal(idx, z[name + "/idx"], err_msg=name)
        np.testing.assert_array_equal(q, z[name + "/queries"], err_msg=name)


def test_index_pyramid_of_a_frame_on_the_emulator_equals_the_oracle(emu):
    """the 22 searches of a frame (linemod_dataset.py:299-353; 120x160 image, 1024 points) through our helper_tool mirror
    (DataProcessing.knn_search -> cpp_knn_batch_omp) -- all 26 index tensors and 4 xyz levels equal the CPU oracle's"""
    import numpy as np
    from ffb6d_amd import synth
    from ffb6d_amd.helper_tool import DataProcessing
    from oracle import knn as oknn
    from oracle import pyramid as opyr
    frames = synth.make_batch(3, 2, n_points=1024, height=120, width=160)
    got = opyr.build_batch(frames, DataProcessing.knn_search)
    want = opyr.build_batch(frames, oknn.knn_search)
    assert sorted(got) == sorted(want) and len(want) == 30           # 26 index tensors + 4 xyz levels
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


@pytest.mark.parametrize("n_pts,height,width", [(4096, 120, 160), (4100, 104, 136)])
def test_pyramid_builder_with_sets_prepared_together_on_the_emulator(emu, n_pts, height, width):
    """pyramid.PyramidBuilder: the searches planned up front, their Morton-ordered sets prepared together
    (ffb6d_knn_prepare_multi), then level by level -- every tensor equals the CPU oracle's pyramid; the prepared sets are
    byte-identical to sets prepared one by one"""
    import numpy as np
    from ffb6d_amd import nearest_neighbors as nn
    from ffb6d_amd import pyramid, synth
    from oracle import knn as oknn
    from oracle import pyramid as opyr
    # cloud (4096) and stride-2 grid (4800): pruned searches, the rest scans; second case: ragged sets (4100 / 1025 / 256 / 64 points,
    # 52 x 68 ... 13 x 17 grids)
    frames = synth.make_batch(4, 2, n_points=n_pts, height=height, width=width)
    want = opyr.build_batch(frames, oknn.knn_search)
    got = pyramid.build_index_pyramid(torch.from_numpy(frames['cld']), torch.from_numpy(frames['dpt_xyz']), index_dtype=torch.int32)
    assert sorted(got) == sorted(want)
    for k in want:
        np.testing.assert_array_equal(got[k].numpy().astype(want[k].dtype), want[k], err_msg=k)
    sets = [torch.from_numpy(frames['cld']), torch.from_numpy(frames['cld'][:, :2100].copy())]
    for p, m in zip(sets, nn.prepare_many(sets)):
        assert torch.equal(m.blob[:-256], nn.PreparedPoints(p).blob[:-256])      # (the last < 256 bytes are alignment padding)


@pytest.mark.parametrize("B,H,W,C,dt", [(2, 12, 16, 64, torch.float32), (1, 7, 9, 8, torch.float32), (1, 1, 1, 16, torch.float32),
                                        (2, 10, 6, 16, torch.bfloat16)])
def test_fused_stem_pass_on_the_emulator(emu, B, H, W, C, dt):
    """BatchNorm + ReLU + MaxPool2d(3, 2, 1) of the colour stem in one kernel against the torch modules (even and odd maps,
    a NaN pixel: it reaches exactly the windows that contain it)"""
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(B, H, W, C, generator=g)
    if H > 4:
        x[0, 3, 2, 1] = float("nan")
    x = x.to(dt)
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    want = torch.nn.functional.max_pool2d(torch.relu(x.float() * scale + shift).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    got = ops_pm.affine_relu_maxpool(x, scale, shift)
    assert got.shape == want.shape and got.dtype == dt
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    if dt == torch.float32:
        assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))
    else:
        assert float((torch.nan_to_num(got.float()) - torch.nan_to_num(want)).abs().max()) <= 1e-2 * float(torch.nan_to_num(want).abs().max())
