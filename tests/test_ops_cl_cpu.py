"""Host logic of the channels-last training operators (ffb6d_amd/ops_cl.py) that needs no GPU: the row views (no copy for
channels_last memory, row stride of a channel slice), the inverted index of a gather (GatherPlan.csr) against a plain
scatter-add, and the loud failure on CPU tensors."""
import pytest
import torch

from ffb6d_amd import _lib, ops_cl


def test_row_views_alias_channels_last_memory():
    x = torch.randn(2, 8, 5, 3).contiguous(memory_format=torch.channels_last)
    r = ops_cl.to_rows(x)
    assert r.shape == (2, 15, 8) and r.is_contiguous() and r.data_ptr() == x.data_ptr()
    assert torch.equal(r[1, 4], x[1, :, 1, 1])
    back = ops_cl.from_rows(r, (5, 3))
    assert back.data_ptr() == x.data_ptr() and back.stride() == x.stride() and torch.equal(back, x)
    col = torch.randn(2, 8, 7, 1).contiguous(memory_format=torch.channels_last)        # [B,C,N,1] as a 1x1 convolution writes it
    assert ops_cl.to_rows(col).data_ptr() == col.data_ptr()
    nchw = torch.randn(2, 8, 5, 3)                                                      # other layouts: one copy, same values
    r2 = ops_cl.to_rows(nchw)
    assert r2.data_ptr() != nchw.data_ptr() and torch.equal(r2[0, 7], nchw[0, :, 2, 1])


def test_channel_slices_of_a_concatenation_keep_their_row_stride():
    full = torch.randn(2, 24, 6, 4).contiguous(memory_format=torch.channels_last)
    for lo, hi in ((0, 8), (8, 24)):
        part = full[:, lo:hi]
        rows, ld = ops_cl._rows_ld(part)
        assert ld == 24 and rows.shape == (2, 24, hi - lo) and rows.data_ptr() == part.data_ptr()
        assert torch.equal(rows[1, 5], part[1, :, 1, 1])
    odd = full[:, 2:10]                                     # not 16-byte aligned: copied
    rows, ld = ops_cl._rows_ld(odd)
    assert ld == 8 and rows.is_contiguous() and torch.equal(rows[0, 3], odd[0, :, 0, 3])
    bf = full.to(torch.bfloat16)[:, 4:12]                    # bf16 unit = 8 channels: offset 4 is unaligned
    rows, ld = ops_cl._rows_ld(bf)
    assert ld == 8 and rows.is_contiguous()


@pytest.mark.parametrize("B,M,U", [(1, 7, 40), (3, 50, 400), (2, 5, 0), (2, 300, 10)])
def test_gather_plan_inverse_reproduces_the_scatter_add(B, M, U):
    g = torch.Generator().manual_seed(B * M + U)
    idx = torch.randint(0, M, (B, U), generator=g)
    grad = torch.randn(B, U, 4, generator=g)
    order, start = ops_cl.GatherPlan(idx, M).csr()
    assert start.shape == (B * M + 1,) and int(start[0]) == 0 and int(start[-1]) == B * U
    flat = grad.reshape(B * U, 4)
    got = torch.stack([flat[order[int(start[r]):int(start[r + 1])]].sum(0) for r in range(B * M)]).reshape(B, M, 4)
    want = torch.zeros(B, M, 4).scatter_add_(1, idx.unsqueeze(2).expand(-1, -1, 4), grad)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    # readers of one source row stay in gather order (stable sort): the fp32 sums are reproducible run to run
    for r in range(B * M):
        seg = order[int(start[r]):int(start[r + 1])]
        assert torch.equal(seg, seg.sort()[0])


def test_cpu_tensors_fail_loudly():
    feat = torch.randn(1, 8, 10, 1)
    idx = torch.zeros(1, 10, 16, dtype=torch.int64)
    for call in (lambda: ops_cl.gather_neighbour(feat, idx), lambda: ops_cl.random_sample(feat, idx),
                 lambda: ops_cl.att_pool(torch.randn(1, 8, 10, 16), torch.randn(1, 8, 10, 16)),
                 lambda: ops_cl.nearest_interpolation(feat, idx[:, :, :1])):
        with pytest.raises(_lib.FFB6DNativeError):
            call()
