"""Oracle-backed double of ``clt_training_amd.hip_ops`` (same call signatures, torch-CPU arithmetic).
TEST INFRASTRUCTURE ONLY: lets the CPU tests drive the sharding / collective logic of
``clt_training_amd.engine`` under gloo.  Never imported by the product."""
from __future__ import annotations


import torch

from . import clt_oracle as O


class Segments:
    def __init__(self, perm, offsets, M, k):
        self.perm, self.offsets, self.M, self.k = perm, offsets, M, k


def encode_topk(x, W, b, k, out_dtype=None, activation="topk"):
    out_dtype = out_dtype or x.dtype
    pre = O.encoder_preacts(x.float(), W.float(), None if b is None else b.float(), out_dtype)
    v, i = O.topk_lowest_index(pre, k) if activation == "topk" else O.groupmax(pre, k)
    return v.to(out_dtype), i


def batch_threshold(v, k):
    return O.batch_threshold(v.float(), k).to(v.dtype)


def groupmax_rows(vals, k, out_dtype=None):
    v, i = O.groupmax(vals.float(), k)
    return v.to(out_dtype or vals.dtype), i


def topk_rows(vals, k, cand_idx=None, counts=None, out_dtype=None):
    if cand_idx is None:
        v, i = O.topk_lowest_index(vals.float(), k)
    else:
        v, i = O.topk_merge(vals.float(), cand_idx, k)
    return v.to(out_dtype or vals.dtype), i


def _local(idx, M, lat_begin):
    r = idx - lat_begin
    valid = (r >= 0) & (r < M)
    return torch.where(valid, r, torch.zeros_like(r)), valid


def decode_fwd(idx, v, W, *, post_enc=None, post_scale=None, bias=None, out=None, out_dtype=None,
               accumulate=False, round_acts=False, want_acts=False, lat_begin=0):
    r, valid = _local(idx, W.shape[0], lat_begin)
    a = v.float()
    if post_enc is not None:
        a = a + post_enc[r] * (a > 0)
    if post_scale is not None:
        a = a * post_scale[r]
    if round_acts:
        a = a.to(torch.bfloat16).float()
    a = a * valid
    y = O.decode_fwd(r, a, W.float())
    if bias is not None:
        y = y + bias
    out_dtype = out_dtype or W.dtype
    if accumulate:
        out.copy_((out.float() + y).to(out.dtype))
        y = out
    else:
        y = y.to(out_dtype)
    return (y, a) if want_acts else y


def decode_bwd_acts(idx, v, W, G, *, post_enc=None, post_scale=None, grad_acts=None, want_e_post=False, lat_begin=0):
    r, valid = _local(idx, W.shape[0], lat_begin)
    ga = (W.float()[r] * G.float().unsqueeze(1)).sum(-1) * valid
    vf = v.float()
    pos = (vf > 0).float()
    u = ga * post_scale[r] if post_scale is not None else ga
    e_scale = ga * (vf + (post_enc[r] * pos if post_enc is not None else 0)) if post_scale is not None else None
    e_post = u * pos if want_e_post else None
    if grad_acts is None:
        grad_acts = u
    else:
        grad_acts += u
    return grad_acts, e_post, e_scale


def decode_fwd_multi(sources, *, bias=None, out=None, out_dtype=None, accumulate=False, round_acts=False,
                     want_acts=False):
    """fp32 sum over all sources, one rounding (clt_decode_fwd_multi)."""
    W0 = sources[0]["W"]
    total = torch.zeros(sources[0]["idx"].shape[0], W0.shape[1])
    acts = []
    for s in sources:
        y, a = decode_fwd(s["idx"], s["v"], s["W"], post_enc=s.get("post_enc"), post_scale=s.get("post_scale"),
                          out_dtype=torch.float32, round_acts=round_acts, want_acts=True, lat_begin=s.get("lat_begin", 0))
        total = total + y.float() * s.get("weight", 1.0)
        acts.append(a)
    if bias is not None:
        total = total + bias
    if accumulate:
        out.copy_((out.float() + total).to(out.dtype))
    else:
        out = total.to(out_dtype or W0.dtype)
    return (out, acts) if want_acts else out


def decode_bwd_acts_multi(sources, G):
    return [decode_bwd_acts(s["idx"], s["v"], s["W"], G.float() * s.get("weight", 1.0), post_enc=s.get("post_enc"),
                            post_scale=s.get("post_scale"), grad_acts=s.get("grad_acts"),
                            want_e_post=s.get("want_e_post", False), lat_begin=s.get("lat_begin", 0)) for s in sources]


def segments_build(idx, M, lat_begin=0):
    r = idx.reshape(-1) - lat_begin
    key = torch.where((r >= 0) & (r < M), r, torch.full_like(r, M))
    order = torch.argsort(key, stable=True)
    counts = torch.bincount(key, minlength=M + 1)[:M]
    offsets = torch.zeros(M + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(counts, 0)
    return Segments(order, offsets, M, idx.shape[-1])


def _scatter(seg, vals, n_cols=None):
    n_real = int(seg.offsets[-1])
    e = seg.perm[:n_real]
    rows = torch.repeat_interleave(torch.arange(seg.M), seg.offsets[1:] - seg.offsets[:-1])
    return e, rows


def spt_dense(seg, vals, dense, *, gate=None, out_dtype=None, want_rowsum=False, out=None):
    e, rows = _scatter(seg, vals)
    w = vals.float()[e]
    if gate is not None:
        w = w * (gate.float()[e] > 0)
    res = torch.zeros(seg.M, dense.shape[1], dtype=torch.float64)
    res.index_add_(0, rows, dense.double()[e // seg.k] * w.double().unsqueeze(1))
    res = res.to(out_dtype or dense.dtype)
    if want_rowsum:
        rs = torch.zeros(seg.M).index_add_(0, rows, w)
        return res, rs
    return res


def decode_bwd_rows(seg, acts, v, W, G, grad_acts, *, post_enc=None, post_scale=None, want_g_post=False, adam=None):
    """The three separate ops in sequence (clt_decode_bwd_rows); the CPU double has no fused optimizer."""
    assert adam is None
    N = G.shape[0]
    k = seg.k
    real = seg.perm[: int(seg.offsets[-1])]
    rows = torch.repeat_interleave(torch.arange(seg.M), seg.offsets[1:] - seg.offsets[:-1])
    r_of = torch.zeros(N * k, dtype=torch.long)
    r_of[real] = rows
    own = torch.zeros(N * k, dtype=torch.bool)
    own[real] = True
    ga = (W.float()[r_of] * G.float()[torch.arange(N * k) // k]).sum(-1) * own
    vf = v.float().reshape(-1)
    pos = (vf > 0).float()
    u = ga * post_scale[r_of] if post_scale is not None else ga
    grad_acts.view(-1).add_(u * own)
    g_post = torch.zeros(seg.M).index_add_(0, r_of[real], (u * pos)[real]) if want_g_post else None
    g_scale = None
    if post_scale is not None:
        es = ga * (vf + (post_enc[r_of] * pos if post_enc is not None else 0))
        g_scale = torch.zeros(seg.M).index_add_(0, r_of[real], es[real])
    gW = spt_dense(seg, acts.reshape(-1), G, out_dtype=W.dtype)
    return gW, g_post, g_scale


def segment_sum(seg, a, b=None):
    e, rows = _scatter(seg, a)
    oa = torch.zeros(seg.M).index_add_(0, rows, a.float()[e])
    ob = None if b is None else torch.zeros(seg.M).index_add_(0, rows, b.float()[e])
    return oa, ob


def fvu_colsum(y, mask):
    m = torch.ones(y.shape[0]) if mask is None else mask.float()
    return torch.cat([(y.float() * m[:, None]).sum(0), m.sum().reshape(1)])


def fvu_partial(y, yhat, mask, colsum_count):
    m = torch.ones(y.shape[0]) if mask is None else mask.float()
    mean = colsum_count[:-1] / colsum_count[-1]
    e = (y.float() - yhat.float()) * m[:, None]
    tv = ((y.float() - mean) ** 2 * m[:, None]).sum()
    return torch.stack([e.pow(2).sum(), tv])


def fvu_grad(y, yhat, mask, l2tv, gscale=1.0, g_dtype=None):
    m = torch.ones(y.shape[0]) if mask is None else mask.float()
    e = (y.float() - yhat.float()) * m[:, None]
    G = (-2.0 * gscale / l2tv[1]) * e
    sc = torch.stack([l2tv[0] / l2tv[1], l2tv[0], l2tv[1]])
    return sc, G.to(g_dtype or yhat.dtype)


class Adam(torch.optim.Adam):
    """torch.optim.Adam stands in for the HIP multi-tensor Adam in CPU tests."""


class SignSGD(torch.optim.Optimizer):
    """sparsify/sign_sgd.py:5-23 restated (``p -= lr * sign(p.grad)``): the CPU double of clt_signsgd_step."""

    def __init__(self, params, lr: float = 1e-3):
        if lr <= 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        super().__init__(params, {"lr": lr})

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None, "Closure is not supported."
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    p.add_(p.grad.sign(), alpha=-group["lr"])
