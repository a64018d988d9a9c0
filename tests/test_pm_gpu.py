"""GPU parity of the point-major / pixel-major ("channels last") inference operators (ffb6d_amd/ops_pm.py,
csrc/mlp_pm.hip ...) against float64 / plain-torch references of the same mathematics.  GEMM bar: 1e-5
(north_star: pooled features within 1e-5 fp32); gathers and max pooling: bit exact."""
import pytest
import torch

from ffb6d_amd import ops, ops_pm

pytestmark = pytest.mark.gpu


def _ref(x1, w, bias, act, x2=None, add=None, gather=None):
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
    return y.float()


# (B, P, K1, K2, Cout, act, py [>0 gather, <0 add], tile_hint)
CASES = [
    (2, 12288, 64, 64, 128, 1, 0, 0), (1, 4800, 1024, 0, 1024, 1, 48, 0), (2, 4800, 128, 0, 64, 1, 768, 0),
    (8, 48, 1024, 0, 512, 1, 0, 0), (8, 192, 512, 256, 256, 2, 0, 0), (8, 192, 256, 0, 256, 1, 48, 0),
    (1, 12288, 128, 0, 22, 0, 0, 0), (1, 12288, 128, 0, 3, 0, 0, 0), (3, 301, 24, 8, 37, 1, 13, 0),
    (2, 130, 128, 0, 128, 2, 0, 0), (1, 196608, 16, 0, 16, 2, 0, 0), (2, 3072, 32, 16, 64, 2, 0, 0),
    (1, 4800, 512, 0, 1024, 1, -1, 0), (1, 1, 8, 0, 8, 0, 0, 0), (1, 33, 8, 8, 5, 1, 0, 0),
    (2, 1000, 64, 0, 136, 1, 7, 0),
]
CASES += [(2, 777, 40, 24, 72, 2, 50, h) for h in (1, 2, 3, 4, 5)] + [(1, 70000, 64, 0, 64, 1, 300, 0), (2, 40000, 16, 0, 16, 2, 0, 0)]
CASES += [(2, 777, 40, 24, 72, 2, 50, h) for h in (1, 2, 3, 4, 5)] + [(1, 4100, 256, 0, 200, 1, -1, h) for h in (1, 2, 3, 4, 5)]
# the stream form (hint 6: whole-row loads / stores through LDS images, W resident): every register-set count (K = 24 .. 128),
# channel-tile count, ragged last tile, two sources, gathered / added epilogue rows; the last two are picked by tile_hint 0
CASES += [(2, 777, 40, 24, 72, 2, 50, 6), (1, 100000, 64, 0, 64, 1, 0, 6), (2, 5001, 24, 0, 16, 0, 0, 6), (1, 4100, 128, 0, 128, 1, -1, 6),
          (3, 1300, 32, 96, 100, 2, 70, 6), (1, 130, 64, 0, 32, 1, 0, 6), (1, 40000, 48, 16, 64, 1, 300, 6), (2, 20000, 64, 0, 64, 2, 900, 0),
          (1, 70000, 128, 0, 128, 1, -1, 0)]
# the LDS-tiled form (hint 7: 128 x 128 tiles, 128-byte row segments staged through LDS, rows of whole segments only): one to
# many steps, two sources, ragged rows / channels, gathered / added epilogue rows; gathered operand rows live in
# test_..._operand_gather; the last one is picked by tile_hint 0
CASES += [(1, 4800, 1024, 0, 1024, 1, 48, 7), (2, 777, 32, 32, 72, 2, 50, 7), (1, 4100, 256, 0, 200, 1, -1, 7), (8, 192, 512, 256, 256, 2, 0, 7),
          (3, 301, 96, 0, 37, 1, 13, 7), (1, 130, 32, 0, 8, 0, 0, 7), (1, 70000, 256, 0, 256, 1, 0, 0)]
# the tile-sequence form (hint 8 + 256 * plan; plan bits 0-3 = tiles per workgroup of region A, 4-7 = of region B, 8-15 / 16-22 = tiles / 16
# of regions B / C): gathered and added epilogue rows, two sources, ragged channels and rows, a three-region plan; the last one is
# picked by tile_hint 0
SEQ = lambda ta, tb=1, b16=0, c16=0: 8 + 256 * (ta | tb << 4 | b16 << 8 | c16 << 16)      # noqa: E731
# round 6: balanced contiguous sequences per XCD (they may run from one point tile into the next): rounds per workgroup slot, or an explicit
# number of sequences per XCD
LIN = lambda rounds, per_xcd=0: 8 + 256 * (0xF0 | rounds | per_xcd << 8)                  # noqa: E731
CASES += [(1, 4800, 1024, 0, 1024, 1, 48, SEQ(2)), (2, 777, 160, 0, 392, 2, 50, SEQ(3)), (1, 4100, 256, 0, 200, 1, -1, SEQ(2)),
          (8, 192, 512, 256, 520, 2, 0, SEQ(4)), (2, 2100, 128, 0, 640, 0, 0, SEQ(4, 2, 2, 2)), (1, 9000, 128, 128, 1024, 1, 300, SEQ(8, 2, 4, 8)),
          (1, 140000, 128, 0, 256, 1, 0, 0)]
CASES += [(1, 4800, 1024, 0, 1024, 1, 48, LIN(2, 7)), (2, 3000, 160, 0, 392, 2, 50, LIN(2, 3)), (1, 9000, 256, 0, 200, 1, -1, LIN(2, 5)),
          (8, 1920, 512, 256, 520, 2, 0, LIN(1, 9)), (2, 38400 // 2, 512, 0, 512, 1, 192, LIN(2)), (1, 98304, 64, 64, 384, 1, 0, 0)]


@pytest.mark.parametrize("B,P,K1,K2,Cout,act,py,hint", CASES)
def test_mlp_pm_matches_fp64_reference(device, B, P, K1, K2, Cout, act, py, hint):
    g = torch.Generator().manual_seed(K1 + Cout + P)
    x1 = torch.randn(B, P, K1, generator=g)
    x2 = torch.randn(B, P, K2, generator=g) if K2 else None
    w = torch.randn(Cout, K1 + K2, generator=g) / (K1 + K2) ** 0.5
    bias = torch.randn(Cout, generator=g)
    gather = (torch.randn(B, py, Cout, generator=g), torch.randint(0, py, (B, P), generator=g)) if py > 0 else None
    add = torch.randn(B, P, Cout, generator=g) if py < 0 else None
    want = _ref(x1, w, bias, act, x2, add, gather)
    d = lambda t: None if t is None else t.to(device)
    got = ops_pm.mlp(d(x1), d(w), d(bias), act, x2=d(x2), add=d(add),
                     gather=None if gather is None else (d(gather[0]), d(gather[1])), tile_hint=hint).cpu()
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_tile_sequence_and_big_tile_forms_equal_the_lds_tiled_form_bit_for_bit(device):
    """mlp_pm_seq_kernel (fp32, hint 8 + 256 * plan) and mlp_pm_big_kernel (bf16, hint 9) feed every accumulator the same products in
    the same k order as mlp_pm_lds_kernel (hint 7) and run the same epilogue arithmetic: equal bits on the device -- with gathered and
    added epilogue rows, two sources, no bias / identity activation (negative zeros must survive the additions of "nothing"), ragged
    rows and channels, a three-region plan"""
    g = torch.Generator().manual_seed(5)
    d = lambda t: None if t is None else t.to(device)                                 # noqa: E731
    for dt, hints in ((torch.float32, (SEQ(2), SEQ(3), SEQ(4, 2, 2, 2), LIN(2, 1), LIN(2, 2), LIN(1))), (BF, (9, 9 + 256, 9 + 256 * 2, 9 + 256 * 3))):
        for B, P, K1, K2, C, py in ((2, 1500, 192, 0, 400, 40), (1, 2100, 128, 128, 640, -1), (3, 700, 256, 0, 528, 0)):
            x1 = torch.randn(B, P, K1, generator=g).to(dt)
            x2 = torch.randn(B, P, K2, generator=g).to(dt) if K2 else None
            w = (torch.randn(C, K1 + K2, generator=g) / (K1 + K2) ** 0.5).to(dt)
            bias = torch.randn(C, generator=g)
            kw = {}
            if py > 0:
                kw["gather"] = (d(torch.randn(B, py, C, generator=g).to(dt)), d(torch.randint(0, py, (B, P), generator=g)))
            elif py < 0:
                kw["add"] = d(torch.randn(B, P, C, generator=g).to(dt))
            for b, act in ((bias, 2), (None, 0)):
                want = ops_pm.mlp(d(x1), d(w), d(b), act, x2=d(x2), tile_hint=7, **kw)
                for h in hints:
                    got = ops_pm.mlp(d(x1), d(w), d(b), act, x2=d(x2), tile_hint=h, **kw)
                    assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), (str(dt), K1, K2, C, py, act, h)


def test_mlp_pm_channel_slices_and_int32_indices(device):
    """Operands / outputs that are channel slices of wider row buffers (row stride > row length), int32 gather
    indices, no bias: what the fused forward feeds it (cat buffers are written in place, never copied)."""
    g = torch.Generator().manual_seed(3)
    B, P, py = 2, 500, 60
    wide = torch.randn(B, P, 96, generator=g).to(device)
    x1, x2 = wide[..., :32], wide[..., 64:88]
    w = (torch.randn(48, 56, generator=g) / 7).to(device)
    Y = torch.randn(B, py, 48, generator=g).to(device)
    idx = torch.randint(0, py, (B, P), generator=g).to(device)
    out_wide = torch.full((B, P, 80), float("nan"), device=device)
    got = ops_pm.mlp(x1, w, None, ops.ACT_LEAKY, x2=x2, gather=(Y, idx.int()), out=out_wide[..., 16:64])
    want = _ref(x1.cpu(), w.cpu(), None, 2, x2.cpu(), None, (Y.cpu(), idx.cpu()))
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)
    assert torch.isnan(out_wide[..., :16]).all() and torch.isnan(out_wide[..., 64:]).all()   # neighbours untouched


def test_mlp_pm_rejects_what_it_cannot_do(device):
    from ffb6d_amd import _lib
    x = torch.randn(1, 10, 12, device=device)
    with pytest.raises(_lib.FFB6DNativeError):       # K not a multiple of 8
        ops_pm.mlp(x, torch.randn(4, 12, device=device))
    with pytest.raises(_lib.FFB6DNativeError):       # CPU tensor: no fallback
        ops_pm.mlp(x.cpu(), torch.randn(4, 12))


def test_mlp_pm_operand_gather_and_log_softmax(device):
    """`choose` pick fused into the head GEMM (x1 rows gathered by index, ffb6d.py:309-312) and the colour branch's
    `final` 1x1 conv + LogSoftmax as one launch (pspnet.py:108-112)."""
    g = torch.Generator().manual_seed(5)
    B, M, P = 2, 700, 333
    img = torch.randn(B, M, 64, generator=g)
    pts = torch.randn(B, P, 64, generator=g)
    choose = torch.randint(0, M, (B, P), generator=g)
    w = torch.randn(128, 128, generator=g) / 11
    bias = torch.randn(128, generator=g)
    picked = torch.gather(img, 1, choose.unsqueeze(2).expand(-1, -1, 64))
    want = _ref(picked, w, bias, 1, pts)
    for dt, hint in ((torch.int64, 0), (torch.int32, 0), (torch.int64, 7)):          # 7: the LDS-tiled form's loader gathers too
        got = ops_pm.mlp(img.to(device), w.to(device), bias.to(device), ops.ACT_RELU, x2=pts.to(device),
                         x1_gather=choose.to(device).to(dt), tile_hint=hint).cpu()
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    for C in (64, 32, 16, 40):
        x = torch.randn(3, 50, 7, 64, generator=g)
        wf = torch.randn(C, 64, generator=g) / 8
        bf = torch.randn(C, generator=g)
        want = torch.log_softmax((x.double() @ wf.double().t() + bf.double()), dim=-1).float()
        for hint in (0, 6):             # 1050 rows: tile kernels; hint 6: the stream form's log-softmax epilogue
            got = ops_pm.mlp(x.to(device), wf.to(device), bf.to(device), ops_pm.ACT_LOG_SOFTMAX, tile_hint=hint).cpu()
            torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("rows,cout,acts,sliced", [(517, 22, (1, 1, 0), True), (128, 3, (1, 2, 0), False), (33, 24, (0, 1, 1), True),
                                                    (100000, 22, (1, 1, 0), True)])
def test_mlp_chain3_matches_the_separate_layers(device, rows, cout, acts, sliced):
    """csrc/mlp_chain.hip (forward_pm.HEADS_CHAIN_FUSED): 128 -> 128 -> 128 -> c as one launch, hidden activations in registers,
    against float64 (1e-5 of the range, the GEMM bar) and against three ffb6d_mlp_pm launches (1e-6: same products, same k order
    wherever the separate launches take a tile kernel)."""
    if rows > 1000 and torch.device(device).type == "cpu":
        pytest.skip("the large case is for the device")
    g = torch.Generator().manual_seed(rows + cout)
    wide = torch.randn(rows, 384, generator=g)
    x = (wide[:, 128:256] if sliced else wide[:, :128].contiguous()).to(device)           # a channel slice: row stride 384
    ws = [torch.randn(128, 128, generator=g) / 11, torch.randn(128, 128, generator=g) / 11, torch.randn(cout, 128, generator=g) / 11]
    bs = [torch.randn(128, generator=g), torch.randn(128, generator=g), torch.randn(cout, generator=g)]
    cpad = -(-cout // 4) * 4
    w3p, b3p = torch.zeros(32, 128), torch.zeros(32)
    w3p[:cout], b3p[:cout] = ws[2], bs[2]
    parts = [(ops_pm.k_chunked(w).to(device), b.to(device), a) for w, b, a in zip([ws[0], ws[1], w3p], [bs[0], bs[1], b3p], acts)]
    got = ops_pm.mlp_chain3(x, parts[0], parts[1], parts[2], cpad)
    assert got.shape == (rows, cpad) and not got[:, cout:].any()
    y, y64 = x, x.double().cpu()
    for w, b, a in zip(ws, bs, acts):
        y = ops_pm.mlp(y, w.to(device), b.to(device), a)
        y64 = y64 @ w.double().t() + b.double()
        y64 = torch.relu(y64) if a == 1 else (torch.nn.functional.leaky_relu(y64, 0.2) if a == 2 else y64)
    scale = float(y64.abs().max())
    assert float((got[:, :cout].double().cpu() - y64).abs().max()) <= 1e-5 * scale
    assert float((got[:, :cout] - y).abs().max()) <= 1e-6 * scale


@pytest.mark.parametrize("rows,cout,acts,sliced", [(517, 22, (1, 1, 0), True), (128, 3, (1, 2, 0), False), (33, 24, (0, 1, 1), True),
                                                    (100000, 22, (1, 1, 0), True)])
def test_mlp_chain3_bf16_matches_the_separate_layers(device, rows, cout, acts, sliced):
    """the bfloat16 chain (round 5): layers 2 and 3 take k in the order the accumulators of the previous layer supply it (weights
    k-chunked with perm=True).  Against float64 of the same bf16-rounded operands WITH the hidden activations rounded to bf16 where the
    chain (and the separate launches) round them, and against three ffb6d_mlp_pm launches."""
    if rows > 1000 and torch.device(device).type == "cpu":
        pytest.skip("the large case is for the device")
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(rows + cout + 1)
    wide = torch.randn(rows, 384, generator=g).to(BF)
    x = (wide[:, 128:256] if sliced else wide[:, :128].contiguous()).to(device)
    ws = [(torch.randn(128, 128, generator=g) / 11).to(BF), (torch.randn(128, 128, generator=g) / 11).to(BF), (torch.randn(cout, 128, generator=g) / 11).to(BF)]
    bs = [torch.randn(128, generator=g), torch.randn(128, generator=g), torch.randn(cout, generator=g)]
    cpad = -(-cout // 4) * 4
    w3p, b3p = torch.zeros(32, 128, dtype=BF), torch.zeros(32)
    w3p[:cout], b3p[:cout] = ws[2], bs[2]
    parts = [(ops_pm.k_chunked(w, perm).to(device), b.to(device), a)
             for w, b, a, perm in zip([ws[0], ws[1], w3p], [bs[0], bs[1], b3p], acts, (False, True, True))]
    got = ops_pm.mlp_chain3(x, parts[0], parts[1], parts[2], cpad)
    assert got.dtype == BF and got.shape == (rows, cpad) and not got[:, cout:].any()
    y, y64 = x, x.double().cpu()
    for i, (w, b, a) in enumerate(zip(ws, bs, acts)):
        y = ops_pm.mlp(y, w.to(device), b.to(device), a)
        y64 = y64 @ w.double().t() + b.double()
        y64 = torch.relu(y64) if a == 1 else (torch.nn.functional.leaky_relu(y64, 0.2) if a == 2 else y64)
        if i < 2:
            y64 = y64.to(BF).double()                          # the hidden activations are bf16 in both forms
    # a hidden value that sits on a bf16 rounding boundary may round the other way (fp32 sums in another order): compare on the
    # output's scale with the bf16 bar of the GEMM tests
    want = y64
    tol = 2.0 ** -7 * want.abs() + 2.0 ** -7 * float(want.abs().mean())
    assert not bool(((got[:, :cout].double().cpu() - want).abs() > tol).any()), float((got[:, :cout].double().cpu() - want).abs().max())
    assert not bool(((got[:, :cout].double() - y.double()).abs().cpu() > tol).any())


@pytest.mark.parametrize("B,h,w,C,P,idt", [(2, 6, 8, 8, 40, torch.int64), (1, 5, 7, 16, 70, torch.int32), (3, 1, 1, 8, 4, torch.int64),
                                           (2, 12, 16, 64, 333, torch.int64)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_upsampled_patch_rows_are_the_unfolded_upsampled_map(device, B, h, w, C, P, idt, dt):
    """forward_pm.LAST_STAGE_AT_CHOSEN: the 3x3 patches of the align_corners x2 up-sampling around picked pixels (corners and edges
    among them: zero padding) == unfold(upsample(x)) at those pixels.  Against this package's own up-sampling kernel: equal bits;
    against ATen's: 1e-6 (fp32)."""
    F = torch.nn.functional
    g = torch.Generator().manual_seed(h * w + C + P)
    x = torch.randn(B, h, w, C, generator=g).to(dt)
    OH, OW = 2 * h, 2 * w
    idx = torch.randint(0, OH * OW, (B, P), generator=g)
    idx[:, :4] = torch.tensor([0, OW - 1, (OH - 1) * OW, OH * OW - 1])             # the four corners
    got = ops_pm.upsampled_patch_rows(x.to(device), idx.to(idt).to(device), (OH, OW)).cpu()
    assert got.shape == (B, P, 9 * C) and got.dtype == dt

    def patches(up):                                                                # up [B,OH,OW,C] -> [B,P,9*C], tap-major
        cols = F.unfold(up.permute(0, 3, 1, 2).float(), 3, padding=1).view(B, C, 9, OH * OW)            # [B, C, tap, pixel]
        pick = torch.gather(cols, 3, idx.view(B, 1, 1, P).expand(-1, C, 9, -1))                          # [B, C, 9, P]
        return pick.permute(0, 3, 2, 1).reshape(B, P, 9 * C)
    ours = ops_pm.bilinear_resize(x.to(device), (OH, OW), True).cpu()
    assert torch.equal(got.float(), patches(ours))
    aten = F.interpolate(x.float().permute(0, 3, 1, 2), size=(OH, OW), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    tol = 1e-6 if dt == torch.float32 else 1e-2
    torch.testing.assert_close(got.float(), patches(aten), rtol=tol, atol=tol)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_stacked_head_gemm_equals_the_separate_launches(device, dt):
    """forward_pm.HEADS_SHARE_FIRST: the first layers of the three heads (ffb6d.py:316-318, same gathered rows) as one GEMM over
    stacked weights, the heads continuing on channel slices of its output -- every output channel is the dot product the separate
    launch computes, in the same k order: equal bits, first layer and the layer after it (which reads a slice with a 3x row stride)"""
    g = torch.Generator().manual_seed(11)
    B, M, C = 2, 900, 128
    P = 517 if torch.device(device).type == "cpu" else 33001        # on the GPU: enough rows for the LDS-tiled / stream / 32 x 256 forms
    img = torch.randn(B, M, 64, generator=g).to(dt).to(device)
    pts = torch.randn(B, P, 64, generator=g).to(dt).to(device)
    choose = torch.randint(0, M, (B, P), generator=g).to(device)
    ws = [(torch.randn(C, 128, generator=g) / 11).to(dt).to(device) for _ in range(3)]
    bs = [torch.randn(C, generator=g).to(device) for _ in range(3)]
    w2 = (torch.randn(C, C, generator=g) / 11).to(dt).to(device)
    y0 = ops_pm.mlp(img, torch.cat(ws).contiguous(), torch.cat(bs).contiguous(), ops.ACT_RELU, x2=pts, x1_gather=choose)
    assert y0.shape == (B, P, 3 * C)
    for h in range(3):
        sep = ops_pm.mlp(img, ws[h], bs[h], ops.ACT_RELU, x2=pts, x1_gather=choose)
        part = y0[..., h * C:(h + 1) * C]
        assert torch.equal(part, sep), h
        assert torch.equal(ops_pm.mlp(part, w2, bs[0], ops.ACT_RELU), ops_pm.mlp(sep, w2, bs[0], ops.ACT_RELU)), h
    # forward_pm.HEADS_ALIGN_LAST: 22 / 3 output channels as whole 16-byte rows (zero weight rows appended), the padding sliced away
    mult = 16 // img.element_size()
    for cout in (22, 3):
        cp = -(-cout // mult) * mult
        wl, bl = w2[:cout].contiguous(), bs[1][:cout].contiguous()
        wp, bp = wl.new_zeros(cp, C), bl.new_zeros(cp)
        wp[:cout], bp[:cout] = wl, bl
        got = ops_pm.mlp(y0[..., :C], wp, bp, ops.ACT_NONE)
        ref = ops_pm.mlp(y0[..., :C], wl, bl, ops.ACT_NONE)
        # same products; equal bits whenever both shapes take kernels with the same k order (every form but the K-split tile, which
        # ffb6d_mlp_pm_tile picks for few rows): asserted as 1e-6 of the range so that the kernel choice stays free
        assert got.shape[-1] == cp and float((got[..., :cout].float() - ref.float()).abs().max()) <= 1e-6 * float(ref.float().abs().max()) \
            + (0 if dt == torch.float32 else 8e-3 * float(ref.float().abs().max())), cout
        assert not got[..., cout:].any()


@pytest.mark.parametrize("B,M,C,Np,K,dt", [(2, 3072, 64, 768, 16, torch.int64), (1, 19200, 64, 3072, 16, torch.int32),
                                           (3, 192, 512, 48, 16, torch.int64), (2, 100, 8, 37, 5, torch.int64),
                                           (1, 4800, 1024, 48, 16, torch.int32)])
def test_random_sample_and_gather_rows_pm_are_bit_exact(device, B, M, C, Np, K, dt):
    g = torch.Generator().manual_seed(M + C)
    f = torch.randn(B, M, C, generator=g)
    idx = torch.randint(0, M, (B, Np, K), generator=g)
    want = torch.gather(f, 1, idx.reshape(B, -1, 1).expand(-1, -1, C)).view(B, Np, K, C).max(dim=2).values
    got = ops_pm.random_sample(f.to(device), idx.to(device).to(dt)).cpu()
    assert torch.equal(got, want)
    up = torch.randint(0, M, (B, 4 * Np + 3), generator=g)
    want = torch.gather(f, 1, up.unsqueeze(2).expand(-1, -1, C))
    assert torch.equal(ops_pm.gather_rows(f.to(device), up.to(device).to(dt)).cpu(), want)


def test_random_sample_pm_propagates_nan_like_torch_max(device):
    f = torch.zeros(1, 20, 4)
    f[0, 7, 2] = float("nan")
    idx = torch.arange(16).view(1, 1, 16).repeat(1, 2, 1)
    idx[0, 1] += 4                                    # second point's neighbourhood: 4..19 (also holds row 7)
    got = ops_pm.random_sample(f.to(device), idx.to(device)).cpu()
    assert torch.isnan(got[0, :, 2]).all() and (got[0, :, [0, 1, 3]] == 0).all()


def test_relative_pos_encoding_pm_matches_reference_formula(device):
    from oracle import ops_ref
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(2, 500, 3, generator=g)
    nei = torch.randint(0, 500, (2, 500, 16), generator=g)
    want = ops_ref.relative_pos_encoding(xyz, nei)                       # [B,N,K,10], RandLANet.py:216-223
    got = ops_pm.relative_pos_encoding(xyz.to(device), nei.to(device)).cpu()
    assert got.shape == (2, 500, 16, 16) and (got[..., 10:] == 0).all()
    assert torch.equal(got[..., 1:10], want[..., 1:])
    torch.testing.assert_close(got[..., 0], want[..., 0], rtol=2e-7, atol=0)     # sqrt: <= 1 ulp vs torch CPU


def test_colour_branch_glue_pm_matches_torch(device):
    g = torch.Generator().manual_seed(2)
    F = torch.nn.functional
    x = torch.randn(2, 12, 16, 64, generator=g)                                # [B,H,W,C]
    sc, sh, rs, rb = (torch.randn(64, generator=g) for _ in range(4))
    res = torch.randn(2, 12, 16, 64, generator=g)
    d = lambda t: t.to(device)
    want = F.leaky_relu(x * sc + sh + res * rs + rb, 0.25)
    got = ops_pm.affine_act_(d(x).clone(), d(sc), d(sh), ops.ACT_LEAKY, 0.25, residual=d(res), res_affine=(d(rs), d(rb))).cpu()
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    want = torch.relu(x * sc + sh + res)
    got = ops_pm.affine_act_(d(x).clone(), d(sc), d(sh), ops.ACT_RELU, residual=d(res)).cpu()
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    for slope in (1.5, -0.3):          # a learned PReLU slope may leave [0, 1]: max(v, slope * v) would be wrong there
        want = F.prelu(x * sc + sh, torch.tensor([slope]))
        got = ops_pm.affine_act_(d(x).clone(), d(sc), d(sh), ops.ACT_LEAKY, slope).cpu()
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    for size, ac in (((24, 32), True), ((30, 41), False), ((12, 16), False)):
        want = F.interpolate(x.permute(0, 3, 1, 2), size=size, mode="bilinear", align_corners=ac).permute(0, 2, 3, 1)
        got = ops_pm.bilinear_resize(d(x), size, ac).cpu()
        torch.testing.assert_close(got, want.contiguous(), rtol=1e-5, atol=1e-5)
    sizes = (1, 2, 3, 6)
    xx = torch.randn(2, 60, 80, 32, generator=g)
    want = torch.cat([F.adaptive_avg_pool2d(xx.permute(0, 3, 1, 2), s).flatten(2) for s in sizes], dim=2).transpose(1, 2)
    got = ops_pm.psp_pool(d(xx), sizes).cpu()
    torch.testing.assert_close(got, want.contiguous(), rtol=1e-5, atol=1e-5)
    z = torch.randn(2, 50, 16, generator=g)
    want, off = 0, 0
    for s in sizes:
        lvl = z[:, off:off + s * s].transpose(1, 2).reshape(2, 16, s, s)
        want = want + F.interpolate(lvl, size=(60, 80), mode="bilinear", align_corners=False)
        off += s * s
    got = ops_pm.psp_prior_sum(d(z), sizes, (60, 80)).cpu()
    torch.testing.assert_close(got, want.permute(0, 2, 3, 1).contiguous(), rtol=1e-5, atol=1e-5)
    for M, hw in ((1024, (12, 16)), (512, (9, 7)), (256, (5, 6))):          # rows of whole waves: the per-wave index arithmetic (round 5)
        z = torch.randn(2, 50, M, generator=g)
        want, off = 0, 0
        for s in sizes:
            want = want + F.interpolate(z[:, off:off + s * s].transpose(1, 2).reshape(2, M, s, s), size=hw, mode="bilinear", align_corners=False)
            off += s * s
        got = ops_pm.psp_prior_sum(d(z), sizes, hw).cpu()
        torch.testing.assert_close(got, want.permute(0, 2, 3, 1).contiguous(), rtol=1e-5, atol=1e-5)
        xx = torch.randn(2, hw[0], hw[1], M // 2, generator=g)
        want = torch.cat([F.adaptive_avg_pool2d(xx.permute(0, 3, 1, 2), s).flatten(2) for s in sizes], dim=2).transpose(1, 2)
        torch.testing.assert_close(ops_pm.psp_pool(d(xx), sizes).cpu(), want.contiguous(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,N,C1,C2,dt", [(2, 1000, 16, 16, torch.int64), (1, 777, 32, 32, torch.int32), (2, 192, 64, 64, torch.int64),
                                          (3, 48, 128, 128, torch.int64), (1, 50, 24, 8, torch.int64), (1, 3, 16, 48, torch.int32)])
def test_att_pool_pm_matches_fp64_reference(device, B, N, C1, C2, dt):
    """Attentive pooling with the neighbour gather and the score GEMM fused (RandLANet.py:243-248):
    sum_k S * softmax_k(fc(S)), S = cat(gather(f), g)."""
    g = torch.Generator().manual_seed(N + C1)
    f = torch.randn(B, N, C1, generator=g)
    nei = torch.randint(0, N, (B, N, 16), generator=g)
    pair = torch.randn(B, N, 16, C2, generator=g)
    d_ = C1 + C2
    w = torch.randn(d_, d_, generator=g) / d_ ** 0.5 * 3
    S = torch.cat([torch.gather(f, 1, nei.reshape(B, -1, 1).expand(-1, -1, C1)).view(B, N, 16, C1), pair], dim=3).double()
    scores = torch.softmax(S @ w.double().t(), dim=2)
    want = (S * scores).sum(dim=2).float()
    got = ops_pm.att_pool(f.to(device), nei.to(device).to(dt), pair.to(device), w.to(device)).cpu()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


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


def _lfa_run(a, mode, device, p_hint=0):
    kw = dict(w2=a["w2"].to(device), b2=a["b2"].to(device), act2=2) if mode == 2 else {}
    return ops_pm.lfa_half(mode, a["xyz"].to(device), a["nei"].to(device), a["f"].to(device), a["w1"].to(device), a["b1"].to(device), 2,
                           a["wfc"].to(device), a["wm"].to(device), a["bm"].to(device), 2, p_hint=p_hint, **kw)


# (B, N, d, p_hint): the four widths of the network at (scaled-down) level shapes, ragged tails, groups straddling frames, every
# group size (4 = one wave per workgroup, + 8 = weights resident in LDS); the last two are the real level-0 / level-3 shapes of BASELINE configuration 2
LFA_CASES = [(2, 1000, 32, 1), (1, 777, 32, 2), (2, 771, 64, 0), (3, 193, 128, 1), (2, 190, 128, 2), (3, 47, 256, 1), (1, 50, 256, 2),
             (2, 1001, 32, 3), (2, 1001, 32, 4), (2, 771, 64, 4), (2, 771, 64, 10),
             (8, 12288, 32, 0), (8, 192, 256, 0)]


@pytest.mark.parametrize("B,N,d,p_hint", LFA_CASES)
@pytest.mark.parametrize("mode", [1, 2])
def test_fused_lfa_half_matches_fp64_reference(device, B, N, d, p_hint, mode):
    """Building_block.forward, RandLANet.py:196-214 -- one launch per half (csrc/lfa_pm.hip): neighbour gather + position encoding +
    mlp1 (+ mlp2) + Att_pooling (fc, softmax over the 16 neighbours, weighted sum, mlp), pair rows in LDS only; against the
    float64 restatement oracle/ops_ref.lfa_half, 1e-5 of the output range (the GEMM bar of this suite)"""
    from oracle import ops_ref
    idt = torch.int64 if (N + mode) % 2 else torch.int32
    a = _lfa_case(B, N, d, mode, torch.float32, idt, seed=N + d + mode)
    got = _lfa_run(a, mode, device, p_hint).cpu()
    kw = dict(w2=a["w2"], b2=a["b2"], act2=2) if mode == 2 else {}
    want = ops_ref.lfa_half(mode, a["xyz"], a["nei"], a["f"], a["w1"], a["b1"], 2, a["wfc"], a["wm"], a["bm"], 2, **kw)
    err = float((got.double() - want).abs().max()) / float(want.abs().max())
    assert got.shape == want.shape and err <= 1e-5, f"max err {err:.2e} of range"


@pytest.mark.parametrize("B,N,d", [(2, 1000, 32), (2, 771, 64), (3, 193, 128), (3, 47, 256)])
def test_fused_lfa_equals_the_unfused_chain(device, B, N, d):
    """The fused halves against the round-2 chain of operators they replace (posenc_mlp -> att_pool -> mlp [-> mlp -> att_pool ->
    mlp]), which has its own parity tests above: same arithmetic up to the summation order of the GEMMs"""
    a = _lfa_case(B, N, d, 2, torch.float32, torch.int64, seed=d)
    dev = {k: v.to(device) for k, v in a.items()}
    g = torch.Generator().manual_seed(1)
    wfc1 = (torch.randn(d, d, generator=g) / d ** 0.5 * 2).to(device)
    wm1, bm1 = (torch.randn(d // 2, d, generator=g) / d ** 0.5).to(device), (torch.randn(d // 2, generator=g) / 2).to(device)
    g1 = ops_pm.posenc_mlp(dev["xyz"], dev["nei"], dev["w1"], dev["b1"], 2)
    agg = ops_pm.mlp(ops_pm.att_pool(dev["f"], dev["nei"], g1, wfc1), wm1, bm1, 2)
    want = ops_pm.mlp(ops_pm.att_pool(agg, dev["nei"], ops_pm.mlp(g1, dev["w2"], dev["b2"], 2), dev["wfc"]), dev["wm"], dev["bm"], 2)
    h1 = ops_pm.lfa_half(1, dev["xyz"], dev["nei"], dev["f"], dev["w1"], dev["b1"], 2, wfc1, wm1, bm1, 2)
    got = ops_pm.lfa_half(2, dev["xyz"], dev["nei"], h1, dev["w1"], dev["b1"], 2, dev["wfc"], dev["wm"], dev["bm"], 2,
                          w2=dev["w2"], b2=dev["b2"], act2=2)
    assert float((h1 - agg).abs().max()) <= 1e-5 * float(agg.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("B,N,d", [(2, 1000, 32), (2, 192, 128), (1, 50, 256)])
@pytest.mark.parametrize("mode", [1, 2])
def test_fused_lfa_half_bf16(device, B, N, d, mode):
    """bf16 rows, fp32 accumulation: against float64 arithmetic on the same bf16 operands with the kernel's two store roundings
    (pair rows, pooled rows) restated"""
    from oracle import ops_ref
    a = _lfa_case(B, N, d, mode, torch.bfloat16, torch.int64, seed=N + d + mode)
    got = _lfa_run(a, mode, device).cpu()
    kw = dict(w2=a["w2"], b2=a["b2"], act2=2) if mode == 2 else {}
    want = ops_ref.lfa_half(mode, a["xyz"], a["nei"], a["f"], a["w1"], a["b1"], 2, a["wfc"], a["wm"], a["bm"], 2,
                            store=lambda t: t.to(torch.bfloat16).to(torch.float64), **kw)
    assert got.dtype == torch.bfloat16
    assert float((got.double() - want).abs().max()) <= 2e-2 * float(want.abs().max())


# ---------------------------------------------------------------------------------------------------------------
# bfloat16 rows (BASELINE.json configuration 5: mixed precision).  References are computed in float64 FROM THE SAME
# bf16-rounded operands, so what is checked is the kernel (fp32 accumulation / arithmetic, one rounding at the store):
# |err| <= 2^-8 * |want| (half an ulp of bf16 is 2^-9) + a small absolute term for cancellation.
# ---------------------------------------------------------------------------------------------------------------
BF = torch.bfloat16


def _close_bf16(got, want, what=""):
    got, want = got.double(), want.double()
    tol = 2.0 ** -8 * want.abs() + 2.0 ** -8 * float(want.abs().mean())
    bad = (got - want).abs() > tol
    assert not bool(bad.any()), (what, float((got - want).abs().max()), float(want.abs().max()))


@pytest.mark.parametrize("B,P,K1,K2,Cout,act,py,hint", [
    (2, 12288, 64, 64, 128, 1, 0, 0), (1, 4800, 1024, 0, 1024, 1, 48, 0), (8, 48, 1024, 0, 512, 1, 0, 0),
    (8, 192, 512, 256, 256, 2, 0, 0), (1, 12288, 128, 0, 22, 0, 0, 0), (3, 301, 32, 16, 40, 1, 13, 0),
    (1, 196608, 16, 0, 16, 2, 0, 0), (1, 4800, 512, 0, 1024, 1, -1, 0), (1, 70000, 128, 0, 64, 1, 300, 0)] +
    [(2, 777, 48, 32, 72, 2, 50, h) for h in (1, 2, 3, 4, 5, 6)] + [(2, 777, 64, 64, 72, 2, 50, 7), (1, 4800, 1024, 0, 1024, 1, 48, 7),
                                                                  (1, 4100, 512, 0, 200, 1, -1, 7), (8, 192, 512, 256, 256, 2, 0, 7),
                                                                  (1, 70000, 512, 0, 256, 1, 0, 0)] +
    [(1, 100000, 64, 0, 64, 1, 0, 6), (2, 5001, 48, 0, 16, 0, 0, 6), (1, 4100, 256, 0, 128, 1, -1, 6), (3, 1300, 64, 96, 104, 2, 70, 6),
     (1, 70000, 64, 64, 128, 1, 300, 0), (2, 20000, 256, 0, 64, 2, 0, 0)] +
    # the 256 x 256 tile with LDS-DMA operand loads (hint 9, csrc/mlp_pm_big.hip; round 6): gathered / added epilogue rows, two sources,
    # ragged rows and channels (whole 16-channel groups: other widths go to form 7), sequences of 3 / 4 channel tiles per workgroup with a
    # ragged last group; the last two are picked by tile_hint 0 (>= 512 tiles of 256 x 256)
    [(1, 4800, 1024, 0, 1024, 1, 48, 9), (2, 777, 64, 64, 80, 2, 50, 9), (1, 4100, 512, 0, 208, 1, -1, 9), (8, 192, 512, 256, 256, 2, 0, 9),
     (3, 301, 128, 0, 784, 1, 13, 9 + 256 * 3), (1, 520, 192, 64, 48, 0, 0, 9), (1, 9000, 256, 0, 1040, 2, -1, 9 + 256 * 4),
     (2, 38400, 512, 0, 512, 1, 192, 0), (1, 140000, 256, 0, 256, 2, -1, 0)])
def test_mlp_pm_bf16(device, B, P, K1, K2, Cout, act, py, hint):
    g = torch.Generator().manual_seed(K1 + Cout + P)
    r = lambda *s: torch.randn(*s, generator=g).to(BF)                               # noqa: E731
    x1, x2 = r(B, P, K1), (r(B, P, K2) if K2 else None)
    w = (torch.randn(Cout, K1 + K2, generator=g) / (K1 + K2) ** 0.5).to(BF)
    bias = torch.randn(Cout, generator=g)
    gather = (r(B, py, Cout), torch.randint(0, py, (B, P), generator=g)) if py > 0 else None
    add = r(B, P, Cout) if py < 0 else None
    want = _ref(x1.double(), w.double(), bias, act, None if x2 is None else x2.double(), None if add is None else add.double(),
                None if gather is None else (gather[0].double(), gather[1]))
    d = lambda t: None if t is None else t.to(device)                                 # noqa: E731
    got = ops_pm.mlp(d(x1), d(w), d(bias), act, x2=d(x2), add=d(add),
                     gather=None if gather is None else (d(gather[0]), d(gather[1])), tile_hint=hint).cpu()
    assert got.dtype == BF and got.shape == want.shape
    _close_bf16(got, want, (K1, K2, Cout))


def test_att_pool_and_log_softmax_bf16(device):
    g = torch.Generator().manual_seed(4)
    for B, N, C1, C2 in ((2, 1000, 16, 16), (1, 192, 64, 64), (2, 48, 128, 128)):
        f = torch.randn(B, N, C1, generator=g).to(BF)
        nei = torch.randint(0, N, (B, N, 16), generator=g)
        pair = torch.randn(B, N, 16, C2, generator=g).to(BF)
        d_ = C1 + C2
        w = (torch.randn(d_, d_, generator=g) / d_ ** 0.5 * 3).to(BF)
        S = torch.cat([torch.gather(f, 1, nei.reshape(B, -1, 1).expand(-1, -1, C1)).view(B, N, 16, C1), pair], dim=3).double()
        want = (S * torch.softmax(S @ w.double().t(), dim=2)).sum(dim=2)
        got = ops_pm.att_pool(f.to(device), nei.to(device), pair.to(device), w.to(device)).cpu()
        assert got.dtype == BF
        _close_bf16(got, want, ("att", C1))
    x = torch.randn(3, 350, 64, generator=g).to(BF)
    wf, bf = (torch.randn(64, 64, generator=g) / 8).to(BF), torch.randn(64, generator=g)
    want = torch.log_softmax(x.double() @ wf.double().t() + bf.double(), dim=-1)
    for hint in (0, 6):
        got = ops_pm.mlp(x.to(device), wf.to(device), bf.to(device), ops_pm.ACT_LOG_SOFTMAX, tile_hint=hint).cpu()
        _close_bf16(got, want, ("log_softmax", hint))


def test_row_operators_bf16(device):
    """gathers / max pooling bit exact on bf16 rows; affine, bilinear, pyramid pooling and position encoding equal to the
    float64 evaluation of the same bf16-rounded inputs up to the final rounding."""
    g = torch.Generator().manual_seed(6)
    F = torch.nn.functional
    d = lambda t: t.to(device)                                                       # noqa: E731
    f = torch.randn(2, 500, 64, generator=g).to(BF)
    idx = torch.randint(0, 500, (2, 120, 16), generator=g)
    want = torch.gather(f, 1, idx.reshape(2, -1, 1).expand(-1, -1, 64)).view(2, 120, 16, 64).float().max(dim=2).values.to(BF)
    assert torch.equal(ops_pm.random_sample(d(f), d(idx)).cpu(), want)
    up = torch.randint(0, 500, (2, 333), generator=g)
    assert torch.equal(ops_pm.gather_rows(d(f), d(up)).cpu(), torch.gather(f, 1, up.unsqueeze(2).expand(-1, -1, 64)))
    x = torch.randn(2, 12, 16, 64, generator=g).to(BF)
    sc, sh = torch.randn(64, generator=g), torch.randn(64, generator=g)
    res = torch.randn(2, 12, 16, 64, generator=g).to(BF)
    want = F.relu(x.double() * sc.double() + sh.double() + res.double())
    _close_bf16(ops_pm.affine_act_(d(x).clone(), d(sc), d(sh), ops.ACT_RELU, residual=d(res)).cpu(), want, "affine")
    want = F.interpolate(x.double().permute(0, 3, 1, 2), size=(24, 32), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    _close_bf16(ops_pm.bilinear_resize(d(x), (24, 32), True).cpu(), want, "bilinear")
    sizes = (1, 2, 3, 6)
    xx = torch.randn(2, 60, 80, 32, generator=g).to(BF)
    want = torch.cat([F.adaptive_avg_pool2d(xx.double().permute(0, 3, 1, 2), s).flatten(2) for s in sizes], dim=2).transpose(1, 2)
    got = ops_pm.psp_pool(d(xx), sizes).cpu()
    assert got.dtype == torch.float32
    torch.testing.assert_close(got.double(), want.contiguous(), rtol=1e-5, atol=1e-5)
    z = torch.randn(2, 50, 16, generator=g)
    want, off = 0, 0
    for s in sizes:
        want = want + F.interpolate(z[:, off:off + s * s].double().transpose(1, 2).reshape(2, 16, s, s), size=(60, 80), mode="bilinear",
                                    align_corners=False)
        off += s * s
    _close_bf16(ops_pm.psp_prior_sum(d(z), sizes, (60, 80), dtype=BF).cpu(), want.permute(0, 2, 3, 1), "prior")
    z = torch.randn(2, 50, 1024, generator=g)
    want, off = 0, 0
    for s in sizes:
        want = want + F.interpolate(z[:, off:off + s * s].double().transpose(1, 2).reshape(2, 1024, s, s), size=(12, 16), mode="bilinear",
                                    align_corners=False)
        off += s * s
    _close_bf16(ops_pm.psp_prior_sum(d(z), sizes, (12, 16), dtype=BF).cpu(), want.permute(0, 2, 3, 1), "prior, whole waves per pixel")
    from oracle import ops_ref
    xyz = torch.rand(2, 300, 3, generator=g)
    nei = torch.randint(0, 300, (2, 300, 16), generator=g)
    got = ops_pm.relative_pos_encoding(d(xyz), d(nei), dtype=BF).cpu()
    assert got.shape == (2, 300, 16, 16) and (got[..., 10:] == 0).all()
    assert torch.equal(got[..., :10], ops_ref.relative_pos_encoding(xyz, nei).to(BF))


# ---------------------------------------------------------------------------------------------------------------
# folded up-convolution (csrc/upconv.hip) and fused position encoding + mlp1 (csrc/posenc.hip); the same kernel bodies
# run on the host in tests/test_hostsim_cpu.py
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,cin,cout,h,w", [(2, 64, 64, 24, 32), (1, 256, 64, 30, 40), (2, 1024, 256, 6, 8), (1, 16, 8, 1, 1),
                                            (3, 24, 40, 5, 7)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_folded_upconv_matches_upsample_conv_bn_prelu(device, B, cin, cout, h, w, dt):
    """PSPUpsample (pspnet.py:34-45) through forward_pm.up_block in the folded form against the torch modules in float64."""
    from ffb6d_amd import forward_pm, model
    if dt == torch.bfloat16 and (cin % 16 or cout % 8):
        pytest.skip("bf16 rows need 16-channel inputs and 8-channel units")
    g = torch.Generator().manual_seed(cin + cout)
    ub = model.UpBlock(cin, cout).eval()
    with torch.no_grad():
        ub.conv[1].weight.copy_(torch.randn(ub.conv[1].weight.shape, generator=g) / (9 * cin) ** 0.5)
        ub.conv[1].bias.copy_(torch.randn(cout, generator=g) * 0.1)
        ub.conv[2].weight.copy_(torch.rand(cout, generator=g) + 0.5)
        ub.conv[2].bias.copy_(torch.randn(cout, generator=g) * 0.1)
        ub.conv[2].running_mean.copy_(torch.randn(cout, generator=g) * 0.1)
        ub.conv[2].running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    x = torch.randn(B, cin, h, w, generator=g)
    with torch.no_grad():
        want = ub.double().conv(x.double()).float()
    ub = ub.float().to(device)
    keep = forward_pm.UPCONV_FOLD
    forward_pm.UPCONV_FOLD = None                                   # every block
    try:
        with torch.no_grad():
            got = forward_pm.up_block(ub, x.to(device).permute(0, 2, 3, 1).contiguous().to(dt))
    finally:
        forward_pm.UPCONV_FOLD = keep
    assert got.shape == (B, 2 * h, 2 * w, cout) and got.dtype == dt
    got = got.float().permute(0, 3, 1, 2).cpu()
    scale = float(want.abs().max())
    err = float((got - want).abs().max()) / scale
    print("folded upconv", (B, cin, cout, h, w), dt, "max err / range %.2e" % err)
    assert err <= (1e-5 if dt == torch.float32 else 3e-2)


@pytest.mark.parametrize("B,N,K,cout,idt", [(2, 3072, 16, 32, torch.int64), (1, 12288, 16, 16, torch.int32), (8, 48, 16, 128, torch.int64),
                                            (3, 17, 5, 24, torch.int64), (1, 1, 1, 8, torch.int64)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_fused_posenc_mlp_matches_encoding_then_fp64_matmul(device, B, N, K, cout, idt, dt):
    """mlp1(relative_pos_encoding(.)) (RandLANet.py:196-199, 216-223) in one pass against the float64 product of the
    standalone encoding kernel's (bit-exact, test above) output."""
    if dt == torch.bfloat16 and cout % 8:
        pytest.skip("bf16 rows come in 8-channel units")
    g = torch.Generator().manual_seed(N + cout)
    xyz = torch.randn(B, N, 3, generator=g).to(device)
    idx = torch.randint(0, N, (B, N, K), generator=g).to(idt).to(device)
    w = torch.zeros(cout, 16)
    w[:, :10] = torch.randn(cout, 10, generator=g) * 0.5
    w[:, 10:] = 3.0                                                 # padding columns must be ignored
    bias = torch.randn(cout, generator=g)
    enc = ops_pm.relative_pos_encoding(xyz, idx)[..., :10].double().cpu()
    for act, f in ((0, lambda v: v), (1, torch.relu), (2, lambda v: torch.nn.functional.leaky_relu(v, 0.2))):
        want = f(enc @ w[:, :10].double().t() + bias.double())
        got = ops_pm.posenc_mlp(xyz, idx, w.to(device), bias.to(device), act, dtype=dt)
        assert got.shape == (B, N, K, cout) and got.dtype == dt
        err = float((got.double().cpu() - want).abs().max()) / float(want.abs().max())
        assert err <= (1e-5 if dt == torch.float32 else 1e-2), (act, err)


@pytest.mark.parametrize("B,H,W,C,dt", [(8, 240, 320, 64, torch.float32), (2, 12, 16, 64, torch.float32), (1, 7, 9, 8, torch.float32),
                                        (2, 60, 80, 64, torch.bfloat16)])
def test_fused_stem_pass_matches_bn_relu_maxpool(device, B, H, W, C, dt):
    """BatchNorm + ReLU + MaxPool2d(3, 2, 1) of the colour stem (ffb6d.py:222) in one kernel against torch on the same device"""
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(B, H, W, C, generator=g).to(dt).to(device)
    scale, shift = (torch.rand(C, generator=g) + 0.5).to(device), torch.randn(C, generator=g).to(device)
    want = torch.nn.functional.max_pool2d(torch.relu(x.float() * scale + shift).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    got = ops_pm.affine_relu_maxpool(x, scale, shift)
    assert got.shape == want.shape and got.dtype == dt
    if dt == torch.float32:
        assert torch.equal(got, want)
    else:
        assert float((got.float() - want).abs().max()) <= 1e-2 * float(want.abs().max())
