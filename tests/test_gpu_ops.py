"""GPU parity tests of the raw C-ABI entry points against the CPU oracle.
Run on an MI355X:  python -m pytest tests -m gpu"""
import math

import pytest
import torch

from conftest import load_golden
from oracle import clt_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"
FP32 = dict(rtol=1e-5, atol=1e-6)   # BASELINE.json tolerance, fp32
BF16 = dict(rtol=1e-2, atol=1e-2)   # BASELINE.json tolerance, bf16


@pytest.fixture(scope="module")
def ops():
    from clt_training_amd import hip_ops
    assert torch.cuda.is_available()
    return hip_ops


def canon(vals, idx):
    return O.topk_merge(vals.float().cpu(), idx.cpu(), vals.shape[1])


# --------------------------------------------------------------------------- decode
@pytest.mark.parametrize("name", ["decoder_d48", "decoder_d64", "decoder_c"])
def test_decode_golden(ops, name):
    t, meta = load_golden(name)
    idx, acts, W, G = (t[k].to(DEV) for k in ("idx", "acts", "W", "G"))
    y = ops.decode_fwd(idx, acts, W)
    torch.testing.assert_close(y.cpu(), t["y"], **FP32)
    ga, _, _ = ops.decode_bwd_acts(idx, acts, W, G)
    torch.testing.assert_close(ga.cpu(), t["gacts"], **FP32)
    seg = ops.segments_build(idx, meta["M"])
    order, offsets = O.segments_build(t["idx"], meta["M"])
    assert torch.equal(seg.perm.cpu().long(), order)
    assert torch.equal(seg.offsets.cpu().long(), offsets)
    gW = ops.spt_dense(seg, acts.float(), G)
    torch.testing.assert_close(gW.cpu(), t["gW"], **FP32)
    torch.testing.assert_close(gW.cpu(), t["spT"], **FP32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,d,M,k", [(257, 768, 4096, 32), (64, 2048, 1024, 64), (33, 4096, 512, 192), (19, 104, 300, 70)])
def test_decode_fused_random(ops, dtype, N, d, M, k):
    g = torch.Generator().manual_seed(N + d)
    W = (torch.randn(M, d, generator=g) / math.sqrt(d)).to(dtype)
    v = torch.rand(N, k, generator=g)
    v[:, ::5] = 0.0
    idx = torch.stack([torch.randperm(M, generator=g)[:k] for _ in range(N)])
    pe = 0.1 * torch.randn(M, generator=g)
    ps = 1 + 0.1 * torch.randn(M, generator=g)
    bias = torch.randn(d, generator=g)
    G = torch.randn(N, d, generator=g).to(dtype)
    tol = FP32 if dtype == torch.float32 else BF16
    a = O.post_affine_fwd(v, idx, pe, ps)
    y_ref = O.decode_fwd(idx, a, W.float()) + bias
    y, acts = ops.decode_fwd(idx.to(DEV), v.to(DEV), W.to(DEV), post_enc=pe.to(DEV), post_scale=ps.to(DEV),
                             bias=bias.to(DEV), want_acts=True)
    assert y.dtype == dtype
    torch.testing.assert_close(acts.cpu(), a, **FP32)
    torch.testing.assert_close(y.float().cpu(), y_ref, **tol)
    # accumulate
    y2 = ops.decode_fwd(idx.to(DEV), v.to(DEV), W.to(DEV), post_enc=pe.to(DEV), post_scale=ps.to(DEV),
                        out=y.clone(), accumulate=True)
    torch.testing.assert_close(y2.float().cpu(), 2 * y_ref - bias, **tol)
    # backward wrt acts + affine
    gW_ref, ga_ref = O.decode_bwd(idx, a, W.float(), G.float())
    u_ref, gp_ref, gs_ref = O.post_affine_bwd(v, idx, pe, ps, ga_ref)
    acc0 = torch.randn(N, k, generator=g)
    gacc, e_post, e_scale = ops.decode_bwd_acts(idx.to(DEV), v.to(DEV), W.to(DEV), G.to(DEV), post_enc=pe.to(DEV),
                                                post_scale=ps.to(DEV), grad_acts=acc0.to(DEV), want_e_post=True)
    torch.testing.assert_close(gacc.cpu(), acc0 + u_ref, rtol=1e-4, atol=1e-4)
    seg = ops.segments_build(idx.to(DEV), M)
    gp, gs = ops.segment_sum(seg, e_post, e_scale)
    torch.testing.assert_close(gp.cpu(), gp_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gs.cpu(), gs_ref, rtol=1e-4, atol=1e-4)
    gW, rs = ops.spt_dense(seg, acts, G.to(DEV), want_rowsum=True)
    assert gW.dtype == dtype
    torch.testing.assert_close(gW.float().cpu(), gW_ref, **tol)
    rs_ref = torch.zeros(M).index_add_(0, idx.reshape(-1), a.reshape(-1))
    torch.testing.assert_close(rs.cpu(), rs_ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,d,M,k,ns,shard", [(130, 768, 2048, 32, 5, None), (67, 768, 1024, 32, 12, (256, 512)),
                                              (9, 2048, 512, 64, 3, None), (21, 104, 300, 70, 2, (100, 200)),
                                              (5, 64, 128, 8, 19, (0, 64)), (11, 1536, 400, 37, 3, None),
                                              (7, 2304, 256, 9, 2, None), (10, 3072, 256, 21, 2, None),
                                              (5, 3584, 200, 13, 2, None), (6, 4096, 300, 33, 3, (50, 250)),
                                              (3, 8192, 64, 5, 2, None)])
def test_decode_multi_source(ops, dtype, N, d, M, k, ns, shard):
    """clt_decode_fwd_multi / clt_decode_bwd_acts_multi == the per-source kernels summed (and the oracle); with a
    latent shard, only the entries of rows [lo, hi) count.  ns = 19 exceeds CLT_DECODE_MAX_SOURCES (two launches).
    The widths cover every piece count of the LDS-staged forward (bf16 rows of 1 .. 8 KiB, entry counts that leave a
    short last batch) and one beyond it (d = 8192: the register gather)."""
    if dtype == torch.float32 and d > 4096:
        pytest.skip("fp32 rows wider than 16 KiB are refused (CLT_ERR_UNSUPPORTED)")
    g = torch.Generator().manual_seed(N * ns + d)
    lo, hi = shard or (0, M)
    Ml = hi - lo
    tol = FP32 if dtype == torch.float32 else BF16
    srcs, ref_y, ref = [], torch.zeros(N, d), []
    G = torch.randn(N, d, generator=g).to(dtype)
    for s_i in range(ns):
        W = (torch.randn(M, d, generator=g) / math.sqrt(d)).to(dtype)
        v = torch.rand(N, k, generator=g)
        v[:, ::7] = 0.0
        idx = torch.stack([torch.randperm(M, generator=g)[:k] for _ in range(N)])
        pe = 0.1 * torch.randn(M, generator=g)
        ps = 1 + 0.1 * torch.randn(M, generator=g)
        weight = 1.0 if s_i == ns - 1 else 0.25
        own = (idx >= lo) & (idx < hi)
        a = O.post_affine_fwd(v, idx, pe, ps) * own
        ref_y += weight * O.decode_fwd(idx, a, W.float())
        _, ga_ref = O.decode_bwd(idx, a, W.float(), G.float() * weight)
        u_ref, _, _ = O.post_affine_bwd(v, idx, pe, ps, ga_ref * own)
        acc0 = torch.randn(N, k, generator=g)
        ref.append((a, acc0, u_ref * own))
        srcs.append(dict(idx=idx.to(DEV), v=v.to(DEV), W=W[lo:hi].contiguous().to(DEV), post_enc=pe[lo:hi].contiguous().to(DEV),
                         post_scale=ps[lo:hi].contiguous().to(DEV), lat_begin=lo, weight=weight))
    bias = torch.randn(d, generator=g)
    y, acts = ops.decode_fwd_multi(srcs, bias=bias.to(DEV), want_acts=True)
    assert y.dtype == dtype
    torch.testing.assert_close(y.float().cpu(), ref_y + bias, **tol)
    for (a, _, _), got in zip(ref, acts):
        torch.testing.assert_close(got.cpu(), a, **FP32)
    # per-source kernels, fp32 output, summed the same way
    y1 = torch.zeros(N, d, device=DEV)
    for s_ in srcs:
        y1 += s_["weight"] * ops.decode_fwd(s_["idx"], s_["v"], s_["W"], post_enc=s_["post_enc"], post_scale=s_["post_scale"],
                                            lat_begin=lo, out_dtype=torch.float32)
    yf = ops.decode_fwd_multi(srcs, out_dtype=torch.float32)
    torch.testing.assert_close(yf, y1, rtol=1e-5, atol=1e-5)
    # accumulate into an existing output
    y2 = ops.decode_fwd_multi(srcs, out=yf.clone(), accumulate=True)
    torch.testing.assert_close(y2, 2 * yf, rtol=1e-6, atol=1e-6)
    # backward
    outs = ops.decode_bwd_acts_multi([dict(s_, grad_acts=r[1].to(DEV), want_e_post=True) for s_, r in zip(srcs, ref)], G.to(DEV))
    for s_, (a, acc0, u_ref), (ga, e_post, e_scale) in zip(srcs, ref, outs):
        torch.testing.assert_close(ga.cpu(), acc0 + u_ref, rtol=1e-4, atol=1e-4)
        Gs = (G.float() * s_["weight"]).to(DEV)
        ga1, ep1, es1 = ops.decode_bwd_acts(s_["idx"], s_["v"], s_["W"], Gs, post_enc=s_["post_enc"], post_scale=s_["post_scale"],
                                            grad_acts=acc0.to(DEV), want_e_post=True, lat_begin=lo)
        torch.testing.assert_close(ga, ga1, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(e_post, ep1, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(e_scale, es1, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,d,M,k,shard,scale", [(300, 768, 1024, 32, None, True), (1500, 768, 512, 32, (128, 384), True),
                                                 (64, 2048, 4096, 64, None, False), (19, 104, 300, 70, (100, 200), True),
                                                 (40, 4096, 70000, 16, None, True), (64, 768, 20000, 32, None, True),
                                                 (64, 768, 40000, 32, None, False), (50, 768, 70000, 16, None, True),
                                                 (40, 2048, 70000, 16, None, True)])
def test_decode_bwd_rows_equals_separate_kernels(ops, dtype, N, d, M, k, shard, scale):
    """clt_decode_bwd_rows(_adam) == clt_decode_bwd_acts + clt_segment_sum + clt_spt_dense(_adam), bit for bit
    (same dot-product and summation order), with and without a latent shard; every rows-per-wave variant
    (table sizes 256 .. 70000)."""
    g = torch.Generator().manual_seed(N + M)
    lo, hi = shard or (0, M)
    W = (torch.randn(M, d, generator=g) / math.sqrt(d)).to(dtype)[lo:hi].contiguous().to(DEV)
    v = torch.rand(N, k, generator=g)
    v[:, ::5] = 0.0
    v = v.to(dtype).to(DEV)
    idx = torch.stack([torch.randperm(M, generator=g)[:k] for _ in range(N)]).to(DEV)
    if M == 512:
        idx[:, 0] = 130  # one latent on every token: a long segment
    pe = (0.1 * torch.randn(M, generator=g))[lo:hi].contiguous().to(DEV)
    ps = (1 + 0.1 * torch.randn(M, generator=g))[lo:hi].contiguous().to(DEV) if scale else None
    G = torch.randn(N, d, generator=g).to(dtype).to(DEV)
    acc0 = torch.randn(N, k, generator=g).to(DEV)
    _, acts = ops.decode_fwd(idx, v, W, post_enc=pe, post_scale=ps, want_acts=True, lat_begin=lo)
    seg = ops.segments_build(idx, hi - lo, lat_begin=lo)
    # separate kernels
    ga1, e_post, e_scale = ops.decode_bwd_acts(idx, v, W, G, post_enc=pe, post_scale=ps, grad_acts=acc0.clone(),
                                               want_e_post=True, lat_begin=lo)
    gp1, gs1 = ops.segment_sum(seg, e_post.view(-1), e_scale.view(-1) if scale else None)
    gW1 = ops.spt_dense(seg, acts.view(-1), G, out_dtype=dtype)
    # one row pass
    ga2 = acc0.clone()
    gW2, gp2, gs2 = ops.decode_bwd_rows(seg, acts.view(-1), v, W, G, ga2, post_enc=pe, post_scale=ps, want_g_post=True)
    off = seg.offsets.long()
    short = (off[1:] - off[:-1]) <= 512  # longer rows are pre-reduced in blocks: another fp32 summation order
    assert torch.equal(gW2, gW1)
    assert torch.equal(ga2, ga1)
    assert torch.equal(gp2[short], gp1[short])
    torch.testing.assert_close(gp2, gp1, rtol=1e-4, atol=1e-4)
    assert (gs2 is None and gs1 is None) or torch.equal(gs2[short], gs1[short])
    if gs2 is not None:
        torch.testing.assert_close(gs2, gs1, rtol=1e-4, atol=1e-4)
    # against the oracle as well
    own = ((idx >= lo) & (idx < hi)).cpu()
    ga_ref = (W.float().cpu()[(idx.cpu() - lo).clamp(0, hi - lo - 1)] * G.float().cpu().unsqueeze(1)).sum(-1) * own
    u_ref = ga_ref * ps.cpu()[(idx.cpu() - lo).clamp(0, hi - lo - 1)] if scale else ga_ref
    torch.testing.assert_close(ga2.cpu(), acc0.cpu() + u_ref, rtol=1e-4, atol=1e-4)
    # Adam variant: parameter, moments and activation gradients as grad -> clt_adam_step / clt_spt_dense_adam
    m1, v1, W1 = torch.zeros_like(W), torch.zeros_like(W), W.clone()
    m2, v2, W2 = torch.zeros_like(W), torch.zeros_like(W), W.clone()
    hp = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    from clt_training_amd.adam import Adam
    p3 = torch.nn.Parameter(W.clone())
    opt3 = Adam([p3], **hp)
    for step in (1, 2):
        ops.spt_dense_adam(seg, acts.view(-1), G, W1, m1, v1, step=step, **hp)
        p3.grad = gW1.clone()
        opt3.step()
        assert torch.equal(W1[short], p3.data[short])  # clt_spt_dense -> .grad -> clt_adam_step
        torch.testing.assert_close(W1.float(), p3.data.float(), rtol=1e-2, atol=2e-3)
        ga3 = acc0.clone()
        _, gp3, gs3 = ops.decode_bwd_rows(seg, acts.view(-1), v, W2, G, ga3, post_enc=pe, post_scale=ps, want_g_post=True,
                                          adam=dict(m=m2, v=v2, step=step, **hp))
        if step == 1:
            assert torch.equal(ga3, ga1) and torch.equal(gp3[short], gp1[short])
            torch.testing.assert_close(gp3, gp1, rtol=1e-4, atol=1e-4)
        assert torch.equal(W2, W1) and torch.equal(m2, m1) and torch.equal(v2, v1)


def test_long_rows_prereduced(ops):
    """Segments of more than 512 entries (latents active on most tokens) go through the block pre-reduction of
    clt_spt_dense_adam / clt_decode_bwd_rows_adam: results against clt_spt_dense -> clt_adam_step and
    clt_decode_bwd_acts -> clt_segment_sum (fp32: only the summation order of the long rows differs)."""
    from clt_training_amd.adam import Adam
    N, k, M, d = 4000, 8, 300, 768
    g = torch.Generator().manual_seed(11)
    rest = torch.stack([12 + torch.randperm(M - 12, generator=g)[: k - 3] for _ in range(N)])
    third = torch.where(torch.rand(N, generator=g) < 0.3, torch.tensor(9), torch.tensor(10))
    idx = torch.cat([torch.full((N, 1), 7), third[:, None], torch.full((N, 1), 11), rest], 1).to(DEV)  # rows 7, 11: 4000
    v = torch.rand(N, k, generator=g)
    v[:, ::3] = 0.0
    v = v.to(DEV)
    W = (torch.randn(M, d, generator=g) / math.sqrt(d)).to(DEV)
    G = torch.randn(N, d, generator=g).to(DEV)
    pe = (0.1 * torch.randn(M, generator=g)).to(DEV)
    ps = (1 + 0.1 * torch.randn(M, generator=g)).to(DEV)
    _, acts = ops.decode_fwd(idx, v, W, post_enc=pe, post_scale=ps, want_acts=True)
    seg = ops.segments_build(idx, M)
    off = seg.offsets.long()
    assert int((off[1:] - off[:-1]).max()) == N and int(((off[1:] - off[:-1]) > 512).sum()) >= 4
    ga1, e_post, e_scale = ops.decode_bwd_acts(idx, v, W, G, post_enc=pe, post_scale=ps, want_e_post=True)
    gp1, gs1 = ops.segment_sum(seg, e_post.view(-1), e_scale.view(-1))
    gW1, rs1 = ops.spt_dense(seg, acts.view(-1), G, want_rowsum=True)
    hp = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    p3 = torch.nn.Parameter(W.clone())
    opt3 = Adam([p3], **hp)
    W1, m1, v1 = W.clone(), torch.zeros_like(W), torch.zeros_like(W)
    W2, m2, v2 = W.clone(), torch.zeros_like(W), torch.zeros_like(W)
    for step in (1, 2):
        p3.grad = gW1.clone()
        opt3.step()
        rs2 = ops.spt_dense_adam(seg, acts.view(-1), G, W1, m1, v1, step=step, want_rowsum=True, **hp)
        ga2 = torch.zeros_like(ga1)
        _, gp2, gs2 = ops.decode_bwd_rows(seg, acts.view(-1), v, W2, G, ga2, post_enc=pe, post_scale=ps, want_g_post=True,
                                          adam=dict(m=m2, v=v2, step=step, **hp))
        assert torch.equal(W2, W1) and torch.equal(m2, m1) and torch.equal(v2, v1)
        torch.testing.assert_close(m1, opt3.state[p3]["exp_avg"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(W1, p3.data, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(rs2, rs1, rtol=1e-4, atol=1e-3)
        if step == 1:
            assert torch.equal(ga2, ga1)  # per-entry dot products: the same order everywhere
            torch.testing.assert_close(gp2, gp1, rtol=1e-4, atol=1e-3)
            torch.testing.assert_close(gs2, gs1, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,d,M,k,ns,lo,hi", [(1, 64, 33, 1, 1, 0, 33), (3, 104, 301, 70, 2, 0, 301), (5, 768, 127, 3, 4, 64, 65),
                                              (7, 64, 50, 8, 3, 50, 50 + 0), (2, 2048, 9, 9, 2, 0, 9)])
def test_decode_edges(ops, dtype, N, d, M, k, ns, lo, hi):
    """Ragged and degenerate shapes through the fan-out and row-pass kernels: one token, k = 1, k > 64 (several
    64-entry chunks), odd row counts, a one-row shard, an EMPTY shard (every entry belongs to another rank: outputs
    must be the bias / zeros and every table row must still get its (zero-gradient) Adam update)."""
    g = torch.Generator().manual_seed(N * 7 + k)
    if hi == lo:  # empty shard: a table of 4 rows that no index names
        lo, hi, M_tab = M, M + 4, M + 4
    else:
        M_tab = M
    Ml = hi - lo
    G = torch.randn(N, d, generator=g).to(dtype).to(DEV)
    bias = torch.randn(d, generator=g).to(DEV)
    srcs, ref_y = [], torch.zeros(N, d)
    for _ in range(ns):
        Wf = (torch.randn(M_tab, d, generator=g) / math.sqrt(d)).to(dtype)
        v = torch.rand(N, k, generator=g)
        idx = torch.stack([torch.randperm(M, generator=g)[:k] for _ in range(N)])
        own = (idx >= lo) & (idx < hi)
        ref_y += O.decode_fwd(idx.clamp(max=M_tab - 1), v * own, Wf.float())
        srcs.append(dict(idx=idx.to(DEV), v=v.to(DEV), W=Wf[lo:hi].contiguous().to(DEV), lat_begin=lo, own=own))
    tol = FP32 if dtype == torch.float32 else BF16
    y = ops.decode_fwd_multi([{a: b for a, b in s_.items() if a != "own"} for s_ in srcs], bias=bias)
    torch.testing.assert_close(y.float().cpu(), ref_y + bias.cpu(), **tol)
    hp = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    for s_ in srcs:
        seg = ops.segments_build(s_["idx"], Ml, lat_begin=lo)
        assert int(seg.offsets[-1]) == int(s_["own"].sum())
        acts = (s_["v"] * s_["own"].to(DEV)).contiguous()
        W0 = s_["W"].clone()
        # plain gradient, then the fused Adam form: both against the oracle's dense gradient
        ga = torch.zeros(N, k, device=DEV)
        gW, _, _ = ops.decode_bwd_rows(seg, acts.view(-1), s_["v"], s_["W"], G, ga)
        loc = (s_["idx"].cpu() - lo).clamp(0, Ml - 1)
        gW_ref = torch.zeros(Ml, d).index_add_(0, loc.reshape(-1), ((s_["v"].cpu() * s_["own"]).reshape(-1, 1)
                                                                      * G.float().cpu().repeat_interleave(k, 0)))
        torch.testing.assert_close(gW.float().cpu(), gW_ref, **tol)
        ga_ref = (W0.float().cpu()[loc] * G.float().cpu().unsqueeze(1)).sum(-1) * s_["own"]
        torch.testing.assert_close(ga.cpu(), ga_ref, rtol=1e-3, atol=1e-3)
        m_, v_ = torch.zeros_like(W0), torch.zeros_like(W0)
        ga2 = torch.zeros(N, k, device=DEV)
        ops.decode_bwd_rows(seg, acts.view(-1), s_["v"], s_["W"], G, ga2, adam=dict(m=m_, v=v_, step=1, **hp))
        assert torch.equal(ga2, ga)
        g_r = gW.float()
        torch.testing.assert_close(m_.float(), 0.1 * g_r, rtol=1e-2, atol=1e-6)
        torch.testing.assert_close(v_.float(), 0.001 * g_r * g_r, rtol=2e-2, atol=1e-9)
        moved = (s_["W"].float() - W0.float()).abs()
        assert bool(((moved > 0) == (g_r != 0)).all()) or dtype == torch.bfloat16


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,d", [(257, 768), (64, 4096), (33, 104)])
def test_sae_row_extras(ops, dtype, M, d):
    """clt_rows_unit_norm / clt_rows_project_out against the reference's torch expressions (sparse_coder.py:687-711), and
    the projection fused into the row pass (project_out) bit for bit against plain gradient -> clt_rows_project_out ->
    clt_adam_step."""
    from clt_training_amd.adam import Adam
    g = torch.Generator().manual_seed(M + d)
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else BF16
    W = torch.randn(M, d, generator=g).to(dtype)
    ref = W.float() / (W.float().norm(dim=1, keepdim=True) + torch.finfo(dtype).eps)
    Wn = ops.rows_unit_norm(W.clone().to(DEV))
    torch.testing.assert_close(Wn.float().cpu(), ref, **tol)
    torch.testing.assert_close(Wn.float().norm(dim=1).cpu(), torch.ones(M), rtol=1e-2, atol=1e-2)
    Gr = torch.randn(M, d, generator=g).to(dtype)
    refp = Gr.float() - (Gr.float() * Wn.float().cpu()).sum(1, keepdim=True) * Wn.float().cpu()
    Gp = ops.rows_project_out(Gr.clone().to(DEV), Wn)
    torch.testing.assert_close(Gp.float().cpu(), refp, **tol)
    assert float((Gp.float() * Wn.float()).sum(1).abs().max()) < (1e-4 if dtype == torch.float32 else 0.25)
    # fused into the decoder row pass
    N, k = 96, 8
    idx = torch.stack([torch.randperm(M, generator=g)[:k] for _ in range(N)]).to(DEV)
    v = torch.rand(N, k, generator=g).to(dtype).to(DEV)
    G = torch.randn(N, d, generator=g).to(dtype).to(DEV)
    _, acts = ops.decode_fwd(idx, v, Wn, want_acts=True)
    seg = ops.segments_build(idx, M)
    hp = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    p3 = torch.nn.Parameter(Wn.clone())
    opt3 = Adam([p3], **hp)
    W2, m2, v2 = Wn.clone(), torch.zeros_like(Wn), torch.zeros_like(Wn)
    for step in (1, 2):
        ga = torch.zeros(N, k, device=DEV)
        gW, _, _ = ops.decode_bwd_rows(seg, acts.view(-1), v, p3.data, G, ga)
        p3.grad = ops.rows_project_out(gW, p3.data)
        opt3.step()
        ga2 = torch.zeros(N, k, device=DEV)
        ops.decode_bwd_rows(seg, acts.view(-1), v, W2, G, ga2, adam=dict(m=m2, v=v2, step=step, project=True, **hp))
        assert torch.equal(ga2, ga) and torch.equal(W2, p3.data)
        assert torch.equal(m2, opt3.state[p3]["exp_avg"]) and torch.equal(v2, opt3.state[p3]["exp_avg_sq"])


def test_row_pass_and_fan_out_at_full_c2_size(ops):
    """BASELINE configs[1] layer size (N=8192, M=98304, d=768, k=32, bf16) through size-independent properties:
    scaling the bag weights / the loss gradient by 2 scales the outputs by exactly 2 (power-of-two scaling commutes with
    every rounding), the per-latent sums of grad_acts equal those of the per-token kernel, two shards' partial
    decodes add up to the unsharded one, and the segments are a permutation sorted by latent."""
    N, M, d, k = 8192, 98304, 768, 32
    g = torch.Generator(device=DEV).manual_seed(9)
    W = (torch.randn(M, d, generator=g, device=DEV) / math.sqrt(d)).bfloat16()
    idx = torch.stack([torch.randperm(M, generator=g, device=DEV)[:k] for _ in range(256)]).repeat(N // 256, 1)
    idx = (idx + torch.arange(N, device=DEV)[:, None] * 7919) % M  # distinct within a row, different per token
    v = torch.rand(N, k, generator=g, device=DEV).bfloat16()
    G = torch.randn(N, d, generator=g, device=DEV).bfloat16()
    src = dict(idx=idx, v=v, W=W)
    y1, (a1,) = ops.decode_fwd_multi([src], want_acts=True, out_dtype=torch.float32)
    y2 = ops.decode_fwd_multi([dict(src, v=(v.float() * 2).bfloat16())], out_dtype=torch.float32)
    assert torch.equal(y2, 2 * y1)
    half = M // 2
    parts = [ops.decode_fwd_multi([dict(idx=idx, v=v, W=W[lo:lo + half], lat_begin=lo)], out_dtype=torch.float32)
             for lo in (0, half)]
    torch.testing.assert_close(parts[0] + parts[1], y1, rtol=1e-5, atol=1e-5)
    seg = ops.segments_build(idx, M)
    keys = idx.reshape(-1)[seg.perm.long()]
    assert bool((keys[1:] >= keys[:-1]).all()) and int(seg.offsets[-1]) == N * k
    assert torch.equal(torch.sort(seg.perm.long()).values, torch.arange(N * k, device=DEV))
    ga1, ga2 = torch.zeros(N, k, device=DEV), torch.zeros(N, k, device=DEV)
    gW1, _, _ = ops.decode_bwd_rows(seg, a1.view(-1), v, W, G, ga1)
    gW2, _, _ = ops.decode_bwd_rows(seg, (2 * a1).view(-1), v, W, (G.float() * 2).bfloat16(), ga2)
    assert torch.equal(gW2.float(), 4 * gW1.float()) and torch.equal(ga2, 2 * ga1)
    ga_tok, _, _ = ops.decode_bwd_acts(idx, v, W, G)
    assert torch.equal(ga_tok, ga1)


def test_segments_skewed(ops):
    """One latent active on every token (long segment) + many empty latents."""
    N, k, M, d = 700, 8, 512, 64
    g = torch.Generator().manual_seed(5)
    idx = torch.stack([torch.cat([torch.tensor([7]), 8 + torch.randperm(M - 8, generator=g)[: k - 1]]) for _ in range(N)])
    vals = torch.randn(N, k, generator=g)
    dense = torch.randn(N, d, generator=g)
    seg = ops.segments_build(idx.to(DEV), M)
    order, offsets = O.segments_build(idx, M)
    assert torch.equal(seg.perm.cpu().long(), order) and torch.equal(seg.offsets.cpu().long(), offsets)
    out = ops.spt_dense(seg, vals.to(DEV), dense.to(DEV))
    torch.testing.assert_close(out.cpu(), O.sparse_transpose_dense_matmul(idx, vals, dense, M), rtol=1e-4, atol=1e-4)
    gate = torch.randn(N, k, generator=g)
    out2, rs = ops.spt_dense(seg, vals.to(DEV), dense.to(DEV), gate=gate.to(DEV), want_rowsum=True)
    vg = vals * (gate > 0)
    torch.testing.assert_close(out2.cpu(), O.sparse_transpose_dense_matmul(idx, vg, dense, M), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(rs.cpu(), torch.zeros(M).index_add_(0, idx.reshape(-1), vg.reshape(-1)), rtol=1e-4, atol=1e-4)


# --------------------------------------------------------------------------- top-k
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,L,k", [(37, 512, 32), (5, 98304, 32), (9, 24576, 64), (6, 16384, 192), (11, 8200, 100),
                                   (16, 130, 7), (3, 4096, 256)])
def test_topk_rows_bit_exact(ops, dtype, N, L, k):
    g = torch.Generator().manual_seed(L + k)
    pre = torch.relu(torch.randn(N, L, generator=g)).to(dtype)  # bf16 rounding creates many ties
    if dtype == torch.float32:
        pre = (pre * 8).round() / 8  # force ties in fp32 too
    v_ref, i_ref = O.topk_lowest_index(pre.float(), k)
    v, i = ops.topk_rows(pre.to(DEV), k)
    assert torch.equal(i.cpu(), i_ref)          # bit-exact, already in canonical order
    assert torch.equal(v.float().cpu(), v_ref)


def test_topk_short_rows_fill_rule(ops):
    """Rows with fewer than k positives are completed with the lowest unselected columns (value 0)."""
    N, L, k = 8, 4096, 32
    pre = torch.zeros(N, L)
    pre[0, [5, 0, 4000]] = torch.tensor([3.0, 1.0, 2.0])
    pre[1, :] = 0
    pre[2, 10:30] = 1.0
    pre[3, 100:200] = torch.arange(100, 0, -1).float()
    pre[4, L - 1] = 9.0
    pre[5, 0:31] = 2.0
    pre[6, 1:40:2] = 5.0
    v_ref, i_ref = O.topk_lowest_index(pre, k)
    for dt in (torch.float32, torch.bfloat16):
        v, i = ops.topk_rows(pre.to(dt).to(DEV), k)
        assert torch.equal(i.cpu(), i_ref)
        assert torch.equal(v.float().cpu(), v_ref)


def test_topk_merge_explicit(ops):
    """TP merge (fused_encoder.py:359-360): candidates [N, k*tp] with global indices, zeros allowed."""
    N, k, tp = 50, 32, 4
    g = torch.Generator().manual_seed(3)
    vals = torch.relu(torch.randn(N, k * tp, generator=g)).to(torch.bfloat16)
    idx = torch.stack([torch.randperm(100000, generator=g)[: k * tp] for _ in range(N)])
    v_ref, i_ref = O.topk_merge(vals.float(), idx, k)
    v, i = ops.topk_rows(vals.to(DEV), k, cand_idx=idx.to(DEV))
    assert torch.equal(i.cpu(), i_ref) and torch.equal(v.float().cpu(), v_ref)
    counts = torch.randint(k, k * tp + 1, (N,), generator=g)
    v_ref2, i_ref2 = [], []
    for n in range(N):
        a, b = O.topk_merge(vals[n : n + 1, : counts[n]].float(), idx[n : n + 1, : counts[n]], k)
        v_ref2.append(a); i_ref2.append(b)
    v, i = ops.topk_rows(vals.to(DEV), k, cand_idx=idx.to(DEV), counts=counts.to(DEV))
    assert torch.equal(i.cpu(), torch.cat(i_ref2)) and torch.equal(v.float().cpu(), torch.cat(v_ref2))


# --------------------------------------------------------------------------- encoder
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,M,d", [(128, 128, 64), (300, 520, 768), (77, 1000, 48), (1024, 4096, 256),
                                   (2600, 2048, 128)])  # d = 128: four k-steps per tile, the 16x16x32 kernel's minimum
def test_encoder_gemm_exact_integers(ops, dtype, N, M, d):
    """Small-integer operands: every product and partial sum is exact in fp32, so the MFMA
    result must equal the oracle bit for bit whatever the accumulation order."""
    g = torch.Generator().manual_seed(N + M + d)
    x = torch.randint(-3, 4, (N, d), generator=g).float()
    W = torch.randint(-3, 4, (M, d), generator=g).float()
    b = torch.randint(-5, 6, (M,), generator=g).float()
    ref = O.encoder_preacts(x, W, b, None)
    pre = ops.encoder_gemm_relu(x.to(dtype).to(DEV), W.to(dtype).to(DEV), b.to(DEV), out_dtype=torch.float32)
    assert torch.equal(pre.cpu(), ref)
    if dtype == torch.bfloat16:
        pre16 = ops.encoder_gemm_relu(x.to(dtype).to(DEV), W.to(dtype).to(DEV), b.to(dtype).to(DEV))
        assert torch.equal(pre16.cpu(), ref.to(torch.bfloat16))
        v, i = ops.encode_topk(x.to(dtype).to(DEV), W.to(dtype).to(DEV), b.to(DEV), 32)
        v_ref, i_ref = O.topk_lowest_index(ref.to(torch.bfloat16).float(), 32)
        assert torch.equal(i.cpu(), i_ref) and torch.equal(v.float().cpu(), v_ref)


@pytest.mark.parametrize("name", ["encoder_a", "encoder_b"])
def test_encoder_golden_fp32(ops, name):
    t, meta = load_golden(name)
    v, i = ops.encode_topk(t["x"].to(DEV), t["W"].to(DEV), t["b"].to(DEV), meta["k"])
    assert torch.equal(i.cpu(), t["idx"])
    torch.testing.assert_close(v.cpu(), t["vals"], **FP32)


def test_encoder_gemm_random(ops):
    g = torch.Generator().manual_seed(0)
    N, M, d = 512, 2048, 768
    x = torch.randn(N, d, generator=g)
    W = torch.randn(M, d, generator=g) / math.sqrt(d)
    b = 0.1 * torch.randn(M, generator=g)
    ref = O.encoder_preacts(x, W, b, None)
    pre = ops.encoder_gemm_relu(x.to(DEV), W.to(DEV), b.to(DEV))
    torch.testing.assert_close(pre.cpu(), ref, rtol=1e-5, atol=1e-5)  # atol of tests/test_encode.py (assert_close default)
    xb, Wb = x.to(torch.bfloat16), W.to(torch.bfloat16)
    ref16 = O.encoder_preacts(xb.float(), Wb.float(), b, torch.bfloat16)
    pre16 = ops.encoder_gemm_relu(xb.to(DEV), Wb.to(DEV), b.to(DEV))
    torch.testing.assert_close(pre16.float().cpu(), ref16, **BF16)


def _plain_encode_topk(ops, x, W, b, k):
    return ops.topk_rows(ops.encoder_gemm_relu(x, W, b), k)


@pytest.mark.parametrize("N,M,d,k,bias", [
    (8192, 98304, 768, 32, "f32"),      # BASELINE configs[1] layer shape: S = 16
    (1000, 20488, 128, 32, "bf16"),     # ragged token and latent edges (M % 256 = 8), S = 4
    (513, 12288, 768, 32, "f32"),       # a tensor-parallel shard of the above: S = 2
    (5000, 16384, 128, 32, "bf16"),     # d = 128 (the 16x16x32 kernel's shortest k loop), several tiles per workgroup
    (300, 65536, 64, 64, None),         # k = 64, no bias
    (260, 90112, 96, 130, "f32"),       # k > 128: the widest selection state
])
def test_fused_encode_topk_equals_plain(ops, N, M, d, k, bias):
    """clt_encode_topk_fwd's fused path (sample tiles -> thresholds -> filtered GEMM -> merge) must give exactly
    what GEMM + row top-k give: same accumulators, same key order, so bit-identical values and indices."""
    g = torch.Generator().manual_seed(N + M + k)
    x = (torch.randn(N, d, generator=g) + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16).to(DEV)
    W = (torch.randn(M, d, generator=g) / math.sqrt(d)).to(torch.bfloat16).to(DEV)
    b = None
    if bias == "f32":
        b = (0.1 * torch.randn(M, generator=g)).to(DEV)
    elif bias == "bf16":
        b = (0.1 * torch.randn(M, generator=g)).to(torch.bfloat16).to(DEV)
    v_ref, i_ref = _plain_encode_topk(ops, x, W, b, k)
    v, i = ops.encode_topk(x, W, b, k)
    assert torch.equal(i, i_ref)
    assert torch.equal(v, v_ref)
    if N * M <= 1000 * 20488:  # and the plain pair against the oracle, where that is cheap
        ref = O.encoder_preacts(x.float().cpu(), W.float().cpu(), None if b is None else b.float().cpu(), torch.bfloat16)
        sl = O.topk_selection_slack(ref, i.cpu()).max().item()
        assert sl <= 2.0 ** -5, sl  # near-ties only: accumulation-order differences of one bf16 ulp


def test_fused_encode_topk_few_positives_and_overflow(ops):
    """Rows with fewer than k positive pre-activations (fill rule), and a degenerate batch whose candidate lists
    overflow (thousands of exact ties at the threshold), which must come out right through the gated fallback."""
    g = torch.Generator().manual_seed(5)
    N, M, d, k = 300, 32768, 64, 32
    x = torch.randn(N, d, generator=g).to(torch.bfloat16).to(DEV)
    W = (torch.randn(M, d, generator=g) / 8).to(torch.bfloat16).to(DEV)
    b = torch.full((M,), -9.0)            # almost nothing survives the ReLU
    b[torch.randperm(M, generator=g)[:20]] = 3.0
    b = b.to(DEV)
    v_ref, i_ref = _plain_encode_topk(ops, x, W, b, k)
    assert (v_ref == 0).any() and (v_ref > 0).any()
    v, i = ops.encode_topk(x, W, b, k)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref)
    # every latent row identical: all M pre-activations of a token tie
    W2 = W[:1].expand(M, d).contiguous()
    v_ref, i_ref = _plain_encode_topk(ops, x, W2, None, k)
    v, i = ops.encode_topk(x, W2, None, k)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref)
    pos = v_ref[:, 0] > 0
    assert torch.equal(i_ref[pos], torch.arange(k, device=DEV).expand(int(pos.sum()), k))  # lowest-index tie rule


def test_fused_encode_topk_fallback_is_per_token_tile(ops):
    """Only the 256-token tiles that hold a failing token are redone the plain way.  Tokens 300..340 see thousands of
    exact ties at their threshold (their candidate lists overflow); every other token is ordinary.  The result is that
    of GEMM + row top-k everywhere, and exactly one of the 5 token tiles is flagged."""
    g = torch.Generator().manual_seed(9)
    N, M, d, k = 1200, 32768, 64, 32
    x = torch.randn(N, d, generator=g)
    x[300:341] = 0.0  # pre-activation = bias for these tokens
    x = x.to(torch.bfloat16).to(DEV)
    W = (torch.randn(M, d, generator=g) / 8).to(torch.bfloat16).to(DEV)
    b = (0.01 * torch.randn(M, generator=g)).to(DEV)
    b[::3] = 0.5  # a third of the latents tie for the best value of the zero tokens
    v_ref, i_ref = _plain_encode_topk(ops, x, W, b, k)
    v, i = ops.encode_topk(x, W, b, k)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref)
    assert torch.equal(i[300, :4].cpu(), torch.tensor([0, 3, 6, 9]))  # lowest-index tie rule through the fallback
    flagged, tiles = ops.encode_topk_fallback_tiles(x, M)
    assert (flagged, tiles) == (1, 5)
    # an ordinary batch flags nothing
    x2 = torch.randn(N, d, generator=g).to(torch.bfloat16).to(DEV)
    v, i = ops.encode_topk(x2, W, b, k)
    v_ref, i_ref = _plain_encode_topk(ops, x2, W, b, k)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref)
    assert ops.encode_topk_fallback_tiles(x2, M)[0] == 0


@pytest.mark.parametrize("N,M,d,k", [(2048, 12288, 768, 32),    # C2 at tp8: the shard of 98304 latents
                                     (1024, 16384, 2048, 64),   # configs[2] (Llama-3.2-1B CLT) at tp8
                                     (2048, 98304, 768, 32),    # C2 unsharded
                                     (16384, 12288, 768, 32),   # the tp8 shard on the gathered batch of 8 ranks: ~95
                                                                # queue entries per wave and tile (flush at row ends)
                                     (512, 131072, 256, 192)])  # configs[3]'s latents and k (d cut to 256): ~500 candidates
                                                                # per token in > 256 non-empty list groups for the merge
def test_fused_encode_topk_stays_on_the_fused_path(ops, N, M, d, k):
    """A guard on BEHAVIOUR, not results (those are exact either way): on an ordinary dictionary the fused path must not
    send token tiles through the plain-GEMM fallback -- at latent-shard widths the candidate density per lane is ~5x the
    unsharded one, and a hit queue sized for the unsharded shape overflowed on 4 of 5 token tiles there."""
    g = torch.Generator(device=DEV).manual_seed(N + M)
    x = torch.randn(N, d, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(M, d, device=DEV, generator=g) / d**0.5).to(torch.bfloat16)
    b = 0.1 * torch.randn(M, device=DEV, generator=g)
    v, i = ops.encode_topk(x, W, b, k)
    flagged, tiles = ops.encode_topk_fallback_tiles(x, M)
    assert tiles == (N + 255) // 256 and flagged == 0, (flagged, tiles)
    v_ref, i_ref = _plain_encode_topk(ops, x, W, b, k)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref)


@pytest.mark.parametrize("rho", [0.25, 0.5])
def test_fused_encode_topk_holds_on_a_tile_correlated_dictionary(ops, rho):
    """The other behaviour guard: a 'trained-like' dictionary (tools/bench_fused.py: Zipf firing frequencies, a few
    latents active on more than half of the tokens, the latents of a 256-row tile sharing a direction with strength rho).
    With the sample drawn as every S-th TILE a token aligned with a tile's direction never showed up in the sample and
    31 of 32 token tiles went through the fallback at rho = 0.25; sampled as every S-th 16-row GROUP inside every tile
    (grp_actual, csrc/gemm_fast_dev.h) the threshold sees every tile, and no candidate list can overflow any more."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from tools.bench_fused import dictionary

    N, M, d, k = 2048, 98304, 768, 32
    gx = torch.Generator(device=DEV).manual_seed(12345)
    x = (torch.randn(N, d, device=DEV, generator=gx) + 0.1 * torch.randn(d, device=DEV, generator=gx)).bfloat16()
    W, b = dictionary("hostile", M, d, rho)
    v, i = ops.encode_topk(x, W, b, k)
    flagged, tiles = ops.encode_topk_fallback_tiles(x, M)
    assert tiles == N // 256 and flagged <= 1, (flagged, tiles)
    v_ref, i_ref = _plain_encode_topk(ops, x, W, b, k)
    assert torch.equal(i, i_ref) and torch.equal(v, v_ref)


def test_decode_bwd_rows_takes_two_byte_aligned_activations(ops):
    """The row pass fetches an entry's bf16 activation as the aligned 32-bit word that holds it (one load instruction for
    either dtype, so that all of an entry's scalars are in flight together): an activation array that starts at an odd
    element -- 2-byte aligned, which the ABI allows -- must give the same bits as a 4-byte aligned copy."""
    N, d, M, k = 700, 256, 2048, 16
    g = torch.Generator().manual_seed(5)
    W = (torch.randn(M, d, generator=g) / math.sqrt(d)).to(torch.bfloat16).to(DEV)
    buf = torch.rand(N * k + 1, generator=g).to(torch.bfloat16).to(DEV)
    v_odd = buf[1:].view(N, k)  # storage offset 1: contiguous, 2-byte aligned
    v_even = v_odd.clone()
    assert v_odd.data_ptr() % 4 == 2 and v_even.data_ptr() % 4 == 0 and v_odd.is_contiguous()
    idx = torch.stack([torch.randperm(M, generator=g)[:k] for _ in range(N)]).to(DEV)
    pe = (0.1 * torch.randn(M, generator=g)).to(DEV)
    G = torch.randn(N, d, generator=g).to(torch.bfloat16).to(DEV)
    seg = ops.segments_build(idx, M)
    outs = []
    for v in (v_even, v_odd):
        _, acts = ops.decode_fwd(idx, v, W, post_enc=pe, want_acts=True)
        ga = torch.zeros(N, k, device=DEV)
        gW, gp, _ = ops.decode_bwd_rows(seg, acts.view(-1), v, W, G, ga, post_enc=pe, want_g_post=True)
        outs.append((gW, gp, ga))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("seed", [0, 1])
def test_fused_encode_topk_random_shapes(ops, seed):
    """tools/fuzz_fused.py: 30 random (N, M, d, k, bias dtype, data kind) cases per seed -- ragged N, M at and around the
    fused path's limit (M = 80 k), widths with and without the 16x16x32 kernels, integer-valued operands with exact ties,
    rows with fewer than k positives -- each bit-identical to GEMM + row top-k."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from tools.fuzz_fused import run

    assert run(30, seed, verbose=False) == 0


# --------------------------------------------------------------------------- FVU
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("masked", [False, True])
def test_fvu(ops, dtype, masked):
    g = torch.Generator().manual_seed(1)
    N, d = 1000, 768
    y = (torch.randn(N, d, generator=g) + 0.3).to(dtype)
    yhat = (y.float() + 0.5 * torch.randn(N, d, generator=g)).to(dtype)
    mask = None
    if masked:
        mask = torch.ones(N, dtype=torch.bool)
        mask[::128] = False
    fvu, e, tv = O.fvu_fwd(y.float(), yhat.float(), mask)
    G_ref = O.fvu_bwd(e, tv, mask, 0.5).float()
    sc, G = ops.fvu_fwd_bwd(y.to(DEV), yhat.to(DEV), None if mask is None else mask.to(DEV), gscale=0.5)
    torch.testing.assert_close(sc[0].cpu(), fvu.float(), rtol=1e-4, atol=1e-6)
    tol = dict(rtol=1e-4, atol=1e-9) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-8)
    torch.testing.assert_close(G.float().cpu(), G_ref, **tol)


# --------------------------------------------------------------------------- Adam
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adam(dtype):
    from clt_training_amd.adam import Adam
    g = torch.Generator().manual_seed(2)
    shapes = [(1000, 96), (77,), (3, 5), (70000,)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(dtype).to(DEV)) for s in shapes]
    ref = [(p.detach().float().cpu().double(), torch.zeros(s).double(), torch.zeros(s).double()) for p, s in zip(ps, shapes)]
    opt = Adam([dict(params=ps[:2], lr=torch.tensor(1e-2)), dict(params=ps[2:], lr=3e-3)], betas=(0.9, 0.999))
    lrs = [1e-2, 1e-2, 3e-3, 3e-3]
    for step in range(1, 4):
        for i, p in enumerate(ps):
            gr = torch.randn(shapes[i], generator=g).to(dtype)
            p.grad = gr.to(DEV)
            pp, m, v = ref[i]
            ref[i] = O.adam_step(pp, gr.float(), m, v, step, lrs[i])
            if dtype == torch.bfloat16:  # states are stored in bf16 between steps
                ref[i] = tuple(O.round_to(t.float(), torch.bfloat16).double() for t in ref[i])
        opt.step()
        for i, p in enumerate(ps):
            tol = FP32 if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
            torch.testing.assert_close(p.detach().float().cpu(), ref[i][0].float(), **tol)


# ----------------------------------------------------------------------------- groupmax (fused_encoder.py:373-413)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,M,k,pad", [(37, 4096, 32, 0), (5, 120, 8, 0), (3, 12, 3, 0), (1, 64, 64, 0), (9, 1024, 1, 0),
                                       (64, 98304, 32, 0), (17, 2048, 16, 64)])
def test_groupmax_rows_vs_oracle(ops, dtype, N, M, k, pad):
    """Bit-exact values and indices against the oracle (itself pinned on the reference's goldens): vector and scalar
    group widths, a strided matrix, exact ties (the lowest column wins) and all-zero groups (first column, value 0)."""
    g = torch.Generator().manual_seed(N * 7 + M + k)
    pre = torch.relu(torch.randn(N, M + pad, generator=g)).to(dtype)
    gs = M // k
    pre[0, : gs] = 0                                  # an all-zero group
    if gs >= 4:
        pre[-1, gs - 1] = pre[-1, gs - 3] = 100.0     # a tie inside the first group of the last row
        pre[-1, M - 1] = pre[-1, M - gs] = 50.0       # first and last column of the last group
    dev = pre.to(DEV)[:, :M]
    v, i = ops.groupmax_rows(dev, k)
    rv, ri = O.groupmax(pre[:, :M].float(), k)
    assert i.dtype == torch.int64 and torch.equal(i.cpu(), ri)
    assert torch.equal(v.float().cpu(), rv)
    assert int(i[0, 0]) == 0 and float(v[0, 0]) == 0.0
    if gs >= 4 and k > 1:
        assert int(i[-1, 0]) == gs - 3 and int(i[-1, -1]) == M - gs


def test_encode_groupmax_vs_reference_golden(ops):
    """clt_encode_groupmax_fwd and the autograd seam ``fused_encoder(..., "groupmax")`` against the golden produced by
    the reference's own fused_encoder (fp32, rtol 1e-5; indices bit-exact)."""
    from clt_training_amd.fused_encoder import fused_encoder

    t, meta = load_golden("encoder_groupmax")
    k = meta["k"]
    v, i = ops.encode_topk(t["x"].to(DEV), t["W"].to(DEV), t["b"].to(DEV), k, activation="groupmax")
    assert torch.equal(i.cpu(), t["idx"])
    torch.testing.assert_close(v.cpu(), t["vals"], **FP32)
    for tag in ("sum", "up"):
        x, W, b = (t[n].to(DEV).requires_grad_() for n in ("x", "W", "b"))
        out = fused_encoder(x, W, b, k, "groupmax")
        assert torch.equal(out.top_indices.cpu(), t["idx"])
        up = torch.ones_like(out.top_acts) if tag == "sum" else t["up"].to(DEV)
        (out.top_acts * up).sum().backward()
        torch.testing.assert_close(W.grad.cpu(), t[f"gW_{tag}"], **FP32)
        torch.testing.assert_close(b.grad.cpu(), t[f"gb_{tag}"], **FP32)
        torch.testing.assert_close(x.grad.cpu(), t[f"gx_{tag}"], **FP32)


def test_encode_groupmax_bf16_layer_shape(ops):
    """BASELINE configs[1] layer shape in bf16: the group maxima of the kernel's own bf16 pre-activations, bit for bit
    (the GEMM is the one the top-k path's plain form uses; its parity is tested above)."""
    g = torch.Generator().manual_seed(5)
    N, M, d, k = 2048, 98304, 768, 32
    x = (torch.randn(N, d, generator=g) + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16).to(DEV)
    W = (torch.randn(M, d, generator=g) / math.sqrt(d)).to(torch.bfloat16).to(DEV)
    b = (0.1 * torch.randn(M, generator=g)).to(DEV)
    v, i = ops.encode_topk(x, W, b, k, activation="groupmax")
    pre = ops.encoder_gemm_relu(x, W, b)
    rv, ri = pre.view(N, k, M // k).max(dim=-1)
    assert torch.equal(v, rv)
    gathered = pre.gather(1, i)
    assert torch.equal(gathered, rv)                       # the reported column holds the maximum
    assert bool((i // (M // k) == torch.arange(k, device=DEV)).all())   # one column per group, in group order
    first = torch.where(pre.view(N, k, -1) == rv.unsqueeze(-1), torch.arange(M // k, device=DEV), M).min(-1).values
    assert torch.equal(i, first + torch.arange(0, M, M // k, device=DEV))  # and it is the FIRST maximum


# ----------------------------------------------------------------------------- SignSGD (sign_sgd.py:19-23)
@pytest.mark.parametrize("pdt", [torch.float32, torch.bfloat16])
def test_signsgd_step_equals_torch(pdt):
    """clt_signsgd_step over a table of tensors (vector body, scalar tail, several chunks, a tensor without a gradient,
    zero gradients) against ``p.add_(p.grad.sign(), alpha=-lr)``: bit for bit."""
    from clt_training_amd.adam import SignSGD

    g = torch.Generator().manual_seed(0)
    shapes = [(50_000,), (7,), (129, 33), (1,), (4096, 8)]
    ps = [torch.randn(s, generator=g).to(pdt).to(DEV).requires_grad_() for s in shapes]
    lrs = [3e-3, 1e-2]
    opt = SignSGD([dict(params=ps[:3], lr=lrs[0]), dict(params=ps[3:], lr=lrs[1])])
    for step in range(3):
        want = []
        for i, p in enumerate(ps):
            if i == 1 and step == 1:
                p.grad = None
                want.append(p.detach().clone())
                continue
            gr = torch.randn(p.shape, generator=g).to(pdt)  # (torch ties a .grad to the parameter's dtype)
            gr[gr.abs() < 0.3] = 0  # sign(0) = 0: those parameters keep their bits
            p.grad = gr.to(DEV)
            want.append(p.detach().clone().add_(p.grad.sign(), alpha=-lrs[0 if i < 3 else 1]))
        opt.step()
        for p, w in zip(ps, want):
            assert torch.equal(p.detach(), w)


def test_encode_batchtopk_vs_reference_golden(ops):
    """activation="batchtopk" (fused_encoder.py:204-240, 315-318) through the autograd seam against the golden from the
    reference's own fused_encoder: [N, 4k] outputs, k * N positive survivors, indices bit-exact where the activation is
    positive (the reference leaves the index of a zeroed slot to torch.topk's order among equal zeros), grads rtol 1e-5."""
    from clt_training_amd.fused_encoder import fused_encoder

    t, meta = load_golden("encoder_batchtopk")
    k = meta["k"]
    for tag in ("sum", "up"):
        x, W, b = (t[n].to(DEV).requires_grad_() for n in ("x", "W", "b"))
        out = fused_encoder(x, W, b, k, "batchtopk")
        v, i = out.top_acts.detach().cpu(), out.top_indices.cpu()
        assert v.shape == (meta["N"], 4 * k) and int((v > 0).sum()) == k * meta["N"]
        live = t["vals"] > 0
        torch.testing.assert_close(v, t["vals"], **FP32)
        assert torch.equal(v > 0, live) and torch.equal(i[live], t["idx"][live])
        up = torch.ones_like(out.top_acts) if tag == "sum" else t["up"].to(DEV)
        (out.top_acts * up).sum().backward()
        torch.testing.assert_close(W.grad.cpu(), t[f"gW_{tag}"], **FP32)
        torch.testing.assert_close(b.grad.cpu(), t[f"gb_{tag}"], **FP32)
        torch.testing.assert_close(x.grad.cpu(), t[f"gx_{tag}"], **FP32)


@pytest.mark.parametrize("N,k,M,lo,hi", [
    (1, 1, 1, 0, 1),                    # one entry, one latent, one 1-bit pass
    (3, 5, 64, 0, 64),                  # fewer entries than a round
    (300, 7, 300_001, 0, 300_001),      # 19-bit keys: three passes
    (5000, 16, 70_000, 30_000, 50_000), # a latent shard: foreign entries are dropped by the first pass
    (257, 33, (1 << 20) + 3, 1 << 19, (1 << 20) + 3),  # 20-bit local keys, shard at the top of the range
    (2048, 8, 40, 0, 40),               # every segment long (410 entries per latent on average), 6-bit keys
    (700, 4, 1000, 990, 1000),          # almost nothing owned
])
def test_segments_radix_sort_shapes(ops, N, k, M, lo, hi):
    """clt_segments_build's own radix sort (csrc/segments.hip) against a stable torch sort: pass counts 1-3, ragged
    last blocks, shards, empty and long segments.  Bit-exact: perm (owned prefix), offsets, owned count."""
    g = torch.Generator().manual_seed(N * 31 + k)
    idx = torch.randint(0, M, (N, k), generator=g)
    seg = ops.segments_build(idx.to(DEV), hi - lo, lat_begin=lo)
    flat = idx.flatten()
    owned = (flat >= lo) & (flat < hi)
    ids = torch.nonzero(owned).flatten()
    keys = flat[ids] - lo
    order = torch.sort(keys, stable=True).indices
    want_perm = ids[order]
    want_off = torch.zeros(hi - lo + 1, dtype=torch.long)
    want_off[1:] = torch.cumsum(torch.bincount(keys, minlength=hi - lo), 0)
    assert torch.equal(seg.offsets.cpu().long(), want_off)
    assert torch.equal(seg.perm.cpu().long()[: ids.numel()], want_perm)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_candidates_pack_unpack(ops, dtype):
    """The exchange format of the TP top-k merge (clt_candidates_pack / _unpack) against its definition in torch ops,
    including the [rank][token][k] -> [token][rank * k] regrouping of what an all-to-all delivers."""
    g = torch.Generator().manual_seed(3)
    tp, n, k = 4, 37, 5
    v = torch.rand(tp * n, k, generator=g).to(dtype)
    v[0, 0] = 0.0
    idx = torch.randint(0, 1 << 20, (tp * n, k), generator=g)
    off = 3 << 20
    want = (v.float().view(torch.int32).to(torch.int64) << 32) | (idx + off)
    got = ops.candidates_pack(v.to(DEV), idx.to(DEV), off)
    assert torch.equal(got.cpu(), want)
    cand = want.view(tp, n, k).permute(1, 0, 2).reshape(n, tp * k)
    uv, ui = ops.candidates_unpack(got, tp, k, dtype)
    assert uv.dtype == dtype and torch.equal(uv.cpu(), (cand >> 32).to(torch.int32).view(torch.float32).to(dtype))
    assert torch.equal(ui.cpu(), cand & 0xFFFFFFFF)
    pv, pi = ops.candidates_unpack(got, 1, k, dtype)
    assert torch.equal(pv.cpu(), v) and torch.equal(pi.cpu(), idx + off)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,d,M,k,n_jobs", [(600, 768, 2048, 16, 3), (300, 2048, 512, 8, 2), (257, 64, 300, 5, 5)])
def test_decode_bwd_rows_adam_multi_equals_one_call_per_job(ops, dtype, N, d, M, k, n_jobs):
    """clt_decode_bwd_rows_adam_multi (every live source of a target in one launch) is bitwise n_jobs separate
    clt_decode_bwd_rows_adam calls: tables, both Adam moments, grad_acts, post-encoder sums -- with a long segment (one
    latent on every token) in the first job, which goes through the batched long-row helper."""
    g = torch.Generator().manual_seed(N + d)
    G = torch.randn(N, d, generator=g).to(dtype).to(DEV)

    def make():
        gg = torch.Generator().manual_seed(7)
        jobs = []
        for j in range(n_jobs):
            idx = torch.stack([torch.randperm(M, generator=gg)[:k] for _ in range(N)])
            if j == 0:
                idx[:, 0] = 11  # long segment (N > 512 in the first case)
                idx[:, 1:] = 12 + idx[:, 1:] % (M - 12)
                idx = torch.stack([torch.cat([r[:1], torch.unique(r[1:])[: k - 1]]) if torch.unique(r[1:]).numel() >= k - 1
                                   else r for r in idx])
            idx = idx.to(DEV)
            v = torch.rand(N, k, generator=gg).to(dtype).to(DEV)
            W = (0.1 * torch.randn(M, d, generator=gg)).to(dtype).to(DEV)
            pe = (0.1 * torch.randn(M, generator=gg)).to(DEV)
            ps = (1 + 0.1 * torch.randn(M, generator=gg)).to(DEV) if j % 2 == 0 else None
            _, acts = ops.decode_fwd(idx, v, W, post_enc=pe, post_scale=ps, want_acts=True)
            jobs.append(dict(seg=ops.segments_build(idx, M), acts=acts.view(-1), v=v, W=W,
                             grad_acts=torch.randn(N, k, generator=gg).to(DEV), post_enc=pe, post_scale=ps, want_g_post=True,
                             adam=dict(m=(0.01 * torch.randn(M, d, generator=gg)).to(dtype).to(DEV),
                                       v=(0.001 * torch.rand(M, d, generator=gg)).to(dtype).to(DEV), lr=1e-3,
                                       betas=(0.9, 0.999), eps=1e-8, step=3)))
        return jobs

    a, b = make(), make()
    outs_a = ops.decode_bwd_rows_adam_multi(a, G)
    for ja, jb, (gp_a, gs_a) in zip(a, b, outs_a):
        _, gp_b, gs_b = ops.decode_bwd_rows(jb["seg"], jb["acts"], jb["v"], jb["W"], G, jb["grad_acts"],
                                            post_enc=jb["post_enc"], post_scale=jb["post_scale"], want_g_post=True,
                                            adam=jb["adam"])
        assert torch.equal(ja["W"], jb["W"]) and torch.equal(ja["adam"]["m"], jb["adam"]["m"])
        assert torch.equal(ja["adam"]["v"], jb["adam"]["v"]) and torch.equal(ja["grad_acts"], jb["grad_acts"])
        assert torch.equal(gp_a, gp_b) and (gs_a is None) == (gs_b is None) and (gs_a is None or torch.equal(gs_a, gs_b))


@pytest.mark.parametrize("L,N,M,d,k,bias", [(3, 2048, 12288, 768, 32, True), (2, 2048, 16384, 256, 64, False)])
def test_encode_topk_layers_equals_layer_by_layer(ops, L, N, M, d, k, bias):
    """clt_encode_topk_fwd_layers (all layers' encoders stacked into one set of launches, inputs read in place from the
    side-by-side [N, L * d] block) is bit for bit L separate clt_encode_topk_fwd calls; shapes the stacked path does not
    take return None."""
    g = torch.Generator(device=DEV).manual_seed(L * N + M)
    xbuf = torch.randn(N, L * d, device=DEV, generator=g).to(torch.bfloat16)
    Ws = [(torch.randn(M, d, device=DEV, generator=g) / d**0.5).to(torch.bfloat16) for _ in range(L)]
    bs = [0.1 * torch.randn(M, device=DEV, generator=g) for _ in range(L)] if bias else None
    out = ops.encode_topk_layers(xbuf, Ws, bs, k)
    assert out is not None
    v, i = out
    for l in range(L):
        vr, ir = ops.encode_topk(xbuf[:, l * d:(l + 1) * d].contiguous(), Ws[l], bs[l] if bias else None, k)
        assert torch.equal(i[l], ir) and torch.equal(v[l], vr), l
    assert ops.encode_topk_layers(xbuf[:300].contiguous(), Ws, bs, k) is None  # N_l not a multiple of 2048


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_spt_dense_adam_multi_equals_one_call_per_job(ops, dtype):
    """clt_spt_dense_adam_multi (the encoder backward of several layers in one launch) is bitwise one
    clt_spt_dense_adam call per job: tables, both moments, row sums; different dense rows and gates per job, a long
    segment in the first."""
    N, d, M, k, n_jobs = 600, 256, 1024, 8, 3

    def make():
        g = torch.Generator().manual_seed(21)
        jobs = []
        for j in range(n_jobs):
            idx = torch.stack([torch.randperm(M - 1, generator=g)[:k] + 1 for _ in range(N)])
            if j == 0:
                idx[:, 0] = 0  # one latent on every token: a long segment (N > 512)
            jobs.append(dict(seg=ops.segments_build(idx.to(DEV), M), vals=torch.randn(N * k, generator=g).to(DEV),
                             dense=torch.randn(N, d, generator=g).to(dtype).to(DEV),
                             gate=(torch.randn(N * k, generator=g)).to(dtype).to(DEV), want_rowsum=True,
                             p=(0.1 * torch.randn(M, d, generator=g)).to(dtype).to(DEV),
                             m=(0.01 * torch.randn(M, d, generator=g)).to(dtype).to(DEV),
                             v=(0.001 * torch.rand(M, d, generator=g)).to(dtype).to(DEV)))
        return jobs

    a, b = make(), make()
    sums = ops.spt_dense_adam_multi(a, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, step=2)
    for ja, jb, rs_a in zip(a, b, sums):
        rs_b = ops.spt_dense_adam(jb["seg"], jb["vals"], jb["dense"], jb["p"], jb["m"], jb["v"], lr=1e-3, betas=(0.9, 0.999),
                                  eps=1e-8, step=2, gate=jb["gate"], want_rowsum=True)
        assert torch.equal(ja["p"], jb["p"]) and torch.equal(ja["m"], jb["m"]) and torch.equal(ja["v"], jb["v"])
        assert torch.equal(rs_a, rs_b)
