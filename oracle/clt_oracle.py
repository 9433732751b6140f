"""CPU oracle for the SAE/CLT train-step hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in explicit torch-CPU arithmetic (no autograd in the
backward formulas), the algorithm of the reference's train step.  It is the
*checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
``clt_training_amd`` never imports it and has no CPU fallback.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
below against golden vectors produced by importing the reference itself in the
build container (``tests/golden/make_golden.py``; fixtures under
``tests/golden/*.safetensors``).  Two things are pinned only by our own
restatement ("parity unpinned"): the tie order of ``torch.topk``
(the reference leaves it unspecified; we define lowest-index-wins) and the
step-0 data dependent init when it is not covered by the trainer golden.

All citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import torch
from torch import Tensor

# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------


def round_to(x: Tensor, dtype: Optional[torch.dtype]) -> Tensor:
    """Emulate a store in ``dtype`` followed by a read back as fp32/fp64."""
    if dtype is None or dtype == x.dtype:
        return x
    return x.to(dtype).to(x.dtype)


# --------------------------------------------------------------------------
# encoder  (sparsify/fused_encoder.py:293-322, 371-372; nanogpt.py:321)
# --------------------------------------------------------------------------


def encoder_preacts(x: Tensor, W: Tensor, b: Optional[Tensor], out_dtype=None) -> Tensor:
    """``relu(x @ W.T + b)`` -- fused_encoder.py:311-312.

    ``out_dtype`` emulates autocast: the GEMM result is rounded once to that
    dtype (bf16 on GPU, sparse_coder.py:672-676) before relu/top-k.
    """
    pre = x.double() @ W.double().T
    if b is not None:
        pre = pre + b.double()
    pre = pre.to(x.dtype if x.dtype in (torch.float32, torch.float64) else torch.float32)
    pre = round_to(pre, out_dtype)
    return torch.relu(pre)


def topk_lowest_index(pre: Tensor, k: int):
    """Row top-k with the documented tie rule: among equal values the lowest
    column index wins (fused_encoder.py:30 leaves the order to torch.topk).

    Returns values/indices in descending-value, ascending-index order."""
    vals, idx = torch.sort(pre, dim=1, descending=True, stable=True)
    return vals[:, :k].contiguous(), idx[:, :k].contiguous()


def groupmax(pre: Tensor, k: int):
    """fused_encoder.py:408-413: k consecutive groups of M / k columns; value and column of each group's maximum
    (ties: the lowest column -- what ``Tensor.max(dim)`` returns on CPU)."""
    n, m = pre.shape
    assert m % k == 0
    gs = m // k
    grp = pre.reshape(n, k, gs)
    vals = grp.max(dim=-1).values
    cols = torch.arange(gs).expand_as(grp)
    first = torch.where(grp == vals.unsqueeze(-1), cols, torch.full_like(cols, gs)).min(dim=-1).values
    return vals.contiguous(), (first + torch.arange(0, m, gs)).contiguous()


def batch_threshold(vals: Tensor, k: int) -> Tensor:
    """batch_topk (fused_encoder.py:204-240, single device) on the per-row top-4k values ``vals [N, 4k]``: the k * N
    largest of the whole batch survive, everything below the smallest of them becomes 0."""
    n = vals.shape[0]
    thr = torch.topk(vals.flatten(), k * n, sorted=False).values.min()
    return torch.where(vals < thr, torch.zeros_like(vals), vals)


def encoder_forward(x, W, b, k, out_dtype=None, activation="topk"):
    """fused_encoder(..., activation) forward -- fused_encoder.py:310-322,372 ("topk"), :373-413 ("groupmax"),
    :315-318 + 204-240 ("batchtopk": [N, 4k] outputs; the zeroed slots keep the index they had in the row's top-4k --
    the reference leaves those to torch.topk's order among equal zeros, and nothing downstream reads them: their
    activation is 0 and FusedEncoder.backward masks their gradient, fused_encoder.py:97)."""
    pre = encoder_preacts(x, W, b, out_dtype)
    if activation == "batchtopk":
        v, i = topk_lowest_index(pre, 4 * k)
        return batch_threshold(v, k), i
    return topk_lowest_index(pre, k) if activation == "topk" else groupmax(pre, k)


def topk_selection_slack(pre: Tensor, idx: Tensor) -> Tensor:
    """How far a selection is from an exact top-k of ``pre``: per row,
    max(unselected) - min(selected), clamped at 0 (0 for an exact top-k)."""
    sel = torch.zeros_like(pre, dtype=torch.bool).scatter_(1, idx, True)
    lo = torch.where(sel, pre, torch.full_like(pre, float("inf"))).min(1).values
    hi = torch.where(sel, torch.full_like(pre, float("-inf")), pre).max(1).values
    return (hi - lo).clamp_min(0)


def topk_merge(cand_vals: Tensor, cand_idx: Tensor, k: int):
    """TP candidate merge -- fused_encoder.py:359-360: second top-k over the
    all-gathered ``[N, k*tp]`` candidates, then gather of the global indices.
    Tie rule: higher value first, then lowest *global latent index*."""
    n, c = cand_vals.shape
    # sort by (value desc, index asc): stable sort by index, then stable by value
    o1 = torch.argsort(cand_idx, dim=1, stable=True)
    v1 = torch.gather(cand_vals, 1, o1)
    i1 = torch.gather(cand_idx, 1, o1)
    o2 = torch.argsort(v1, dim=1, descending=True, stable=True)
    v2 = torch.gather(v1, 1, o2)[:, :k].contiguous()
    i2 = torch.gather(i1, 1, o2)[:, :k].contiguous()
    return v2, i2


def encoder_backward(x, W, vals, idx, grad_vals, need_grad_x=False):
    """FusedEncoder.backward -- fused_encoder.py:92-201 (single device).

    gv = g * (vals > 0)                                  (:97)
    grad_b[idx] += gv                                    (:140-143)
    grad_W[idx, :] += gv * x   (fp32)                    (:150-155, kernels.py:14-54,81)
    grad_x = sum_j gv * W[idx]                           (:100-105)
    """
    M, d = W.shape
    n, k = idx.shape
    gv = grad_vals.float() * (vals > 0).float()
    grad_b = torch.zeros(M, dtype=torch.float32)
    grad_b.index_add_(0, idx.reshape(-1), gv.reshape(-1))
    grad_W = sparse_transpose_dense_matmul(idx, gv, x.float(), M)
    grad_x = None
    if need_grad_x:
        grad_x = decode_fwd(idx, gv, W.float())
    return grad_W, grad_b, grad_x, gv


def sparse_transpose_dense_matmul(idx: Tensor, vals: Tensor, dense: Tensor, n_out: int) -> Tensor:
    """``sparse.T @ dense`` -- kernels.py:14-54 (+ COO kernel :113-216).
    out[r, :] = sum_{(a, j): idx[a, j] == r} vals[a, j] * dense[a, :]; fp32, zeros elsewhere (:81)."""
    a, k = idx.shape
    out = torch.zeros(n_out, dense.shape[1], dtype=torch.float64)
    rows = dense.double().repeat_interleave(k, dim=0) * vals.double().reshape(-1, 1)
    out.index_add_(0, idx.reshape(-1), rows)
    return out.float()


# --------------------------------------------------------------------------
# post-encoder affine  (sparsify/sparse_coder.py:103-141)
# --------------------------------------------------------------------------


def post_affine_fwd(v: Tensor, idx: Tensor, post_enc: Tensor, post_scale: Optional[Tensor]):
    """a' = (v + p[idx] * (v > 0)) * s[idx] -- sparse_coder.py:136-140."""
    a = v + post_enc[idx] * (v > 0).to(v.dtype)
    if post_scale is not None:
        a = a * post_scale[idx]
    return a


def post_affine_bwd(v, idx, post_enc, post_scale, grad_a):
    """Backward of post_affine_fwd (autograd of sparse_coder.py:136-140).

    ds[r] += ga * (v + p[r] [v>0]);  u = ga * s[idx];  dp[r] += u [v>0];  dv = u.
    """
    M = post_enc.shape[0]
    pos = (v > 0).to(grad_a.dtype)
    if post_scale is not None:
        pre_scale = v + post_enc[idx] * pos
        g_scale = torch.zeros(M, dtype=grad_a.dtype)
        g_scale.index_add_(0, idx.reshape(-1), (grad_a * pre_scale).reshape(-1))
        u = grad_a * post_scale[idx]
    else:
        g_scale = None
        u = grad_a
    g_post = torch.zeros(M, dtype=grad_a.dtype)
    g_post.index_add_(0, idx.reshape(-1), (u * pos).reshape(-1))
    return u, g_post, g_scale


# --------------------------------------------------------------------------
# sparse decoder (sparsify/utils.py:442-457; xformers.py:15-45, 125-176)
# --------------------------------------------------------------------------


def decode_fwd(idx: Tensor, acts: Tensor, W: Tensor) -> Tensor:
    """y[n, :] = sum_j acts[n, j] * W[idx[n, j], :]  (fp32+ accumulate) -- utils.py:442-445."""
    rows = W[idx]  # [N, k, d]
    return (rows.double() * acts.double().unsqueeze(-1)).sum(1).to(torch.float32)


def decode_bwd(idx: Tensor, acts: Tensor, W: Tensor, G: Tensor):
    """Backward of the embedding bag -- xformers.py:125-176.
    gW[e, :] = sum acts * G[n, :] (dense [M, d]);  g_acts[n, j] = G[n, :] . W[e, :]."""
    gW = sparse_transpose_dense_matmul(idx, acts, G, W.shape[0])
    g_acts = (W[idx].double() * G.double().unsqueeze(1)).sum(-1).float()
    return gW, g_acts


def segments_build(idx: Tensor, M: int):
    """Sorted-by-latent permutation + offsets (the COO sort of kernels.py:42,
    made stable so that every segment lists its (token, slot) entries in
    ascending entry id)."""
    flat = idx.reshape(-1)
    order = torch.argsort(flat, stable=True)
    counts = torch.bincount(flat, minlength=M)
    offsets = torch.zeros(M + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(counts, 0)
    return order, offsets


# --------------------------------------------------------------------------
# FVU (sparsify/sparse_coder.py:207-221)
# --------------------------------------------------------------------------


def fvu_fwd(y: Tensor, yhat: Tensor, mask: Optional[Tensor]):
    """e = (y - yhat) m;  tv = sum m (y - ybar_m)^2;  fvu = sum e^2 / tv."""
    y = y.double()
    yhat = yhat.double()
    e = y - yhat
    if mask is None:
        tv = (y - y.mean(0)).pow(2).sum()
    else:
        lm = mask.double().unsqueeze(-1)
        e = e * lm
        ybar = (y * lm).sum(0) / lm.sum(0)
        tv = ((y - ybar).pow(2) * lm).sum()
    l2 = e.pow(2).sum()
    return (l2 / tv), e, tv


def fvu_bwd(e: Tensor, tv: Tensor, mask: Optional[Tensor], gscale: float = 1.0):
    """d fvu / d yhat = -2 e m / tv  (e already carries one factor m; m^2 = m)."""
    return (-2.0 * gscale / tv) * e


# --------------------------------------------------------------------------
# Adam + schedule  (sparsify/trainer.py:45-58, 152-193, 797-802)
# --------------------------------------------------------------------------


def warmup_linear_factor(step: int, warmup_steps: int, num_training_steps: int) -> float:
    """WarmupLinearSchedule.lr_lambda -- trainer.py:51-58."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - warmup_steps)))


def auto_lr(num_latents: int, lr: Optional[float] = None) -> float:
    """trainer.py:169-171."""
    return lr or 2e-4 / (num_latents / (2**14)) ** 0.5


def adam_step(p, g, m, v, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (no weight decay, no amsgrad), one step.
    ``step`` is the 1-based step count.  Returns new (p, m, v) in fp64."""
    p, g, m, v = p.double(), g.double(), m.double(), v.double()
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1**step
    bc2 = 1 - b2**step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


# --------------------------------------------------------------------------
# SAE-only extras (sparse_coder.py:687-711)
# --------------------------------------------------------------------------


def unit_norm_decoder(W_dec: Tensor, eps: Optional[float] = None) -> Tensor:
    eps = torch.finfo(W_dec.dtype).eps if eps is None else eps
    return W_dec / (W_dec.norm(dim=1, keepdim=True) + eps)


def remove_parallel_grad(W_dec: Tensor, g: Tensor) -> Tensor:
    par = (g * W_dec).sum(1, keepdim=True)
    return g - par * W_dec


# --------------------------------------------------------------------------
# parameter container + step-0 init (sparse_coder.py:233-445; trainer.py:547-592)
# --------------------------------------------------------------------------


@dataclass
class LayerParams:
    """One SparseCoder's parameters (sparse_coder.py:233-445), plain tensors.

    ``W_decs``/``b_decs``/``W_skips`` have one entry per target slot in
    ``coalesce_topk="none"`` CLTs and exactly one entry otherwise
    (sparse_coder.py:324-338, 377-387)."""

    W_enc: Tensor
    b_enc: Tensor
    W_decs: list
    b_decs: list
    W_skips: list  # entries may be None (no skip connection)
    post_encs: list
    post_enc_scales: Optional[list]
    n_targets: int
    in_norm: Optional[Tensor] = None
    out_norm: Optional[Tensor] = None
    transcode: bool = True

    def clone(self):
        c = lambda t: None if t is None else t.clone()
        return LayerParams(
            c(self.W_enc), c(self.b_enc), [c(t) for t in self.W_decs], [c(t) for t in self.b_decs],
            [c(t) for t in self.W_skips], [c(t) for t in self.post_encs],
            None if self.post_enc_scales is None else [c(t) for t in self.post_enc_scales],
            self.n_targets, c(self.in_norm), c(self.out_norm), self.transcode,
        )


def n_targets_rule(position: int, n_layers: int, cross_layer: int):
    """trainer.py:125-134 (n_targets only)."""
    if cross_layer > 0:
        return min(cross_layer, n_layers - position)
    return 0


def step0_init(lp: LayerParams, x: Tensor, y: Tensor, normalize_io: bool):
    """Data dependent init on the first batch -- trainer.py:547-592."""
    d = x.shape[-1]
    if lp.transcode:
        mean = -x.mean(0)
        lp.b_enc[:] = lp.W_enc.float() @ mean.float()  # F.linear(mean, W, 0*b) :565
    lp.b_decs[0][:] = y.mean(0)  # :572-578
    if normalize_io:
        in_norm = x.norm(dim=-1).mean()
        out_norm = y.norm(dim=-1).mean()
        lp.in_norm[:] = in_norm
        lp.out_norm[:] = out_norm
        for b in lp.b_decs:
            b[:] = b * (d**0.5 / out_norm)
        lp.b_enc[:] = lp.b_enc * (d**0.5 / in_norm)


# --------------------------------------------------------------------------
# whole CLT train step (runner.py:25-173, sparse_coder.py:143-229, trainer.py:598-682)
# --------------------------------------------------------------------------


@dataclass
class StepResult:
    fvus: list
    top_acts: list
    top_idx: list
    grads: list  # list (per layer) of dict name -> tensor
    sae_outs: list = field(default_factory=list)


def _normalize_in(lp: LayerParams, x, normalize_io):
    if normalize_io:
        return x * (x.shape[-1] ** 0.5 / lp.in_norm)  # sparse_coder.py:622-625
    return x


def clt_train_step(
    layers: list,
    xs: list,
    ys: list,
    mask: Optional[Tensor],
    k: int,
    *,
    coalesce: str = "none",
    normalize_io: bool = False,
    divide_cross_layer: bool = False,
    acc_steps: int = 1,
    act_dtype=None,
    forced_idx: Optional[list] = None,
    activation: str = "topk",
    topk_coalesced: bool = False,
    secondary_target_tied: bool = False,
) -> StepResult:
    """Explicit forward+backward of one FVU train step over all layers, in the
    order the reference's hook runs them (trainer.py:598-682): for each layer j
    encode, decode every live source into target j, FVU, backward, then the
    deferred encoder backward of the sources that just wrote their last target
    (runner.py:56-59, 193-197).

    ``coalesce`` in {"none", "concat", "per-layer"} (runner.py:60-73 vs :93-127 vs :130-158).
    Returns per-layer FVU and the gradient of every parameter, computed with the
    closed forms of SURVEY.md section 9 -- no autograd.
    """
    L = len(layers)
    dt = torch.float64
    grads = []
    for lp in layers:
        g = {"W_enc": torch.zeros_like(lp.W_enc, dtype=dt), "b_enc": torch.zeros_like(lp.b_enc, dtype=dt)}
        for t in range(len(lp.W_decs)):
            g[f"W_decs.{t}"] = torch.zeros_like(lp.W_decs[t], dtype=dt)
        for t in range(len(lp.b_decs)):  # (per_source_tied: one W_dec, but a bias / skip matrix per target)
            g[f"b_decs.{t}"] = torch.zeros_like(lp.b_decs[t], dtype=dt)
            if lp.W_skips[t] is not None:
                g[f"W_skips.{t}"] = torch.zeros_like(lp.W_skips[t], dtype=dt)
        for t in range(len(lp.post_encs)):
            g[f"post_encs.{t}"] = torch.zeros_like(lp.post_encs[t], dtype=dt)
            if lp.post_enc_scales is not None:
                g[f"post_enc_scales.{t}"] = torch.zeros_like(lp.post_enc_scales[t], dtype=dt)
        grads.append(g)

    live = []  # insertion ordered list of source layer ids (runner.outputs)
    state = {}  # per live source: dict(xn, x_enc, v, idx, slot, gv_acc)
    fvus, accts, idxs, sae_outs = [], [], [], []

    k_arg = k  # (k is re-bound per layer below)
    for j in range(L):
        k_cfg = k_arg[j] if isinstance(k_arg, (list, tuple)) else k_arg  # TrainConfig.per_layer_k (trainer.py:364-368)
        lp = layers[j]
        d = xs[j].shape[-1]
        xn = _normalize_in(lp, xs[j].to(dt), normalize_io)
        x_enc = xn if lp.transcode else xn - lp.b_decs[0].to(dt)  # sparse_coder.py:637-638
        if forced_idx is None:
            v, idx = encoder_forward(x_enc, lp.W_enc.to(dt), lp.b_enc.to(dt), k_cfg, None, activation)
        else:
            # reduced-precision parity: evaluate the step on a GIVEN selection (the near-tie order of a
            # bf16 GEMM is not reproducible across accumulation orders); validity of the selection is
            # checked separately with topk_selection_slack().
            idx = forced_idx[j]
            v = encoder_preacts(x_enc, lp.W_enc.to(dt), lp.b_enc.to(dt), None).gather(1, idx)
        if act_dtype is not None:
            v = round_to(v, act_dtype)
        accts.append(v.clone())
        idxs.append(idx.clone())
        k = v.shape[1]  # batchtopk: 4k slots per row
        live.append(j)
        state[j] = dict(xn=xn, x_enc=x_enc, v=v, idx=idx, slot=0, gv=torch.zeros_like(v, dtype=dt))

        n_live = len(live)
        divide_by = max(1, n_live - 1) if divide_cross_layer else 1  # runner.py:46-49
        y = ys[j].to(dt)

        # ---------------- forward ----------------
        per_src = {}
        if coalesce == "none":
            addition = torch.zeros_like(y)
            for i in live:
                st = state[i]
                li = layers[i]
                t = st["slot"]
                scale = None if li.post_enc_scales is None else li.post_enc_scales[t].to(dt)
                a = post_affine_fwd(st["v"], st["idx"], li.post_encs[t].to(dt), scale)
                tw = t if t < len(li.W_decs) else 0  # per_source_tied: one decoder for all targets (sparse_coder.py:324-338)
                out = (li.W_decs[tw].to(dt)[st["idx"]] * a.unsqueeze(-1)).sum(1)  # sparse_coder.py:660
                out = out + li.b_decs[t].to(dt)  # sparse_coder.py:661
                if li.W_skips[t] is not None:
                    out = out + st["xn"] @ li.W_skips[t].to(dt).T  # :190
                per_src[i] = dict(a=a, t=t)
                if i != j:
                    addition = addition + out  # runner.py:70-71
                else:
                    sae_out = out + addition / divide_by  # runner.py:63, sparse_coder.py:191
            if secondary_target_tied:
                # runner.py:75-91: a second decode of ALL candidates (post-affine values, indices mod M) through the
                # current layer's slot-0 decoder, with its bias and skip once more, on top of the sum above
                cat_v2 = torch.cat([per_src[i]["a"] for i in live], 1)
                cat_i2 = torch.cat([state[i]["idx"] for i in live], 1)
                sae_out = sae_out + (lp.W_decs[0].to(dt)[cat_i2] * cat_v2.unsqueeze(-1)).sum(1) + lp.b_decs[0].to(dt)
                if lp.W_skips[0] is not None:
                    sae_out = sae_out + xn @ lp.W_skips[0].to(dt).T
        elif coalesce in ("concat", "per-layer"):
            cv, ci = [], []
            for pos, i in enumerate(live):
                st = state[i]
                li = layers[i]
                t = st["slot"]
                scale = None if li.post_enc_scales is None else li.post_enc_scales[t].to(dt)
                a = post_affine_fwd(st["v"], st["idx"], li.post_encs[t].to(dt), scale)
                per_src[i] = dict(a=a, t=t)
                cv.append(a)
                # runner.py:52-54: idx + pos * M; "concat" takes it off again (% M, :122), "per-layer" decodes with
                # it (the target's W_dec has one block of M rows per live source, sparse_coder.py:306-307)
                ci.append(st["idx"] + pos * li.W_enc.shape[0] if coalesce == "per-layer" else st["idx"])
            cat_v = torch.cat(cv, 1)
            cat_i = torch.cat(ci, 1)
            sel = None
            if topk_coalesced:  # runner.py:96-117: keep the k best of the concatenated candidates
                full_v = cat_v
                sel = torch.sort(cat_v, dim=1, descending=True, stable=True).indices[:, :k]
                cat_v, cat_i = cat_v.gather(1, sel), cat_i.gather(1, sel)
            sae_out = (lp.W_decs[0].to(dt)[cat_i] * cat_v.unsqueeze(-1)).sum(1) + lp.b_decs[0].to(dt)
            if lp.W_skips[0] is not None:
                sae_out = sae_out + xn @ lp.W_skips[0].to(dt).T
        else:
            raise ValueError(coalesce)

        denorm = (lp.out_norm.to(dt) / d**0.5) if normalize_io else 1.0  # sparse_coder.py:627-630
        sae_out = sae_out * denorm
        sae_outs.append(sae_out.float())
        fvu, e, tv = fvu_fwd(y, sae_out, mask)
        fvus.append(fvu.float())

        # ---------------- backward of fvu / acc_steps ----------------
        G = fvu_bwd(e, tv, mask, 1.0 / acc_steps) * denorm  # grad wrt pre-denormalised sum
        if coalesce == "none":
            ga2 = None
            if secondary_target_tied:
                grads[j]["b_decs.0"] += G.sum(0)
                if lp.W_skips[0] is not None:
                    grads[j]["W_skips.0"] += G.T @ xn
                kk2 = cat_i2.shape[1]
                grads[j]["W_decs.0"] += torch.zeros_like(grads[j]["W_decs.0"]).index_add_(
                    0, cat_i2.reshape(-1), G.repeat_interleave(kk2, 0) * cat_v2.reshape(-1, 1))
                ga2 = (lp.W_decs[0].to(dt)[cat_i2] * G.unsqueeze(1)).sum(-1)
            for pos, i in enumerate(live):
                st = state[i]
                li = layers[i]
                t = per_src[i]["t"]
                a = per_src[i]["a"]
                Gi = G if i == j else G / divide_by
                grads[i][f"b_decs.{t}"] += Gi.sum(0)
                if li.W_skips[t] is not None:
                    grads[i][f"W_skips.{t}"] += Gi.T @ st["xn"]
                tw = t if t < len(li.W_decs) else 0
                gW = torch.zeros_like(grads[i][f"W_decs.{tw}"]).index_add_(
                    0, st["idx"].reshape(-1), Gi.repeat_interleave(st["idx"].shape[1], 0) * a.reshape(-1, 1))
                ga = (li.W_decs[tw].to(dt)[st["idx"]] * Gi.unsqueeze(1)).sum(-1)
                if ga2 is not None:
                    ga = ga + ga2[:, pos * k:(pos + 1) * k]
                grads[i][f"W_decs.{tw}"] += gW
                scale = None if li.post_enc_scales is None else li.post_enc_scales[t].to(dt)
                u, gp, gs = post_affine_bwd(st["v"], st["idx"], li.post_encs[t].to(dt), scale, ga)
                grads[i][f"post_encs.{t}"] += gp
                if gs is not None:
                    grads[i][f"post_enc_scales.{t}"] += gs
                st["gv"] += u
        else:
            grads[j]["b_decs.0"] += G.sum(0)
            if lp.W_skips[0] is not None:
                grads[j]["W_skips.0"] += G.T @ xn
            kk = cat_i.shape[1]
            gW = torch.zeros_like(grads[j]["W_decs.0"]).index_add_(
                0, cat_i.reshape(-1), G.repeat_interleave(kk, 0) * cat_v.reshape(-1, 1))
            grads[j]["W_decs.0"] += gW
            ga_cat = (lp.W_decs[0].to(dt)[cat_i] * G.unsqueeze(1)).sum(-1)
            if sel is not None:  # torch.topk backward: the gradient goes to the selected positions only
                ga_cat = torch.zeros_like(full_v).scatter_(1, sel, ga_cat)
            off = 0
            for i in live:
                st = state[i]
                li = layers[i]
                t = per_src[i]["t"]
                ks = st["idx"].shape[1]  # (per_layer_k: the sources' bags differ in width)
                ga = ga_cat[:, off : off + ks]
                off += ks
                scale = None if li.post_enc_scales is None else li.post_enc_scales[t].to(dt)
                u, gp, gs = post_affine_bwd(st["v"], st["idx"], li.post_encs[t].to(dt), scale, ga)
                grads[i][f"post_encs.{t}"] += gp
                if gs is not None:
                    grads[i][f"post_enc_scales.{t}"] += gs
                st["gv"] += u

        # ---------------- advance slots; deferred encoder backward ----------------
        for i in list(live):
            st = state[i]
            li = layers[i]
            will_be_last = st["slot"] + 1 >= li.n_targets  # sparse_coder.py:91-93
            st["slot"] += 1
            if will_be_last:
                # FusedEncoder.backward (fused_encoder.py:92-155) in fp64
                gvm = st["gv"] * (st["v"] > 0).to(dt)
                gW64 = torch.zeros_like(grads[i]["W_enc"]).index_add_(
                    0, st["idx"].reshape(-1), st["x_enc"].repeat_interleave(st["idx"].shape[1], 0) * gvm.reshape(-1, 1))
                gb64 = torch.zeros_like(grads[i]["b_enc"]).index_add_(0, st["idx"].reshape(-1), gvm.reshape(-1))
                grads[i]["W_enc"] += gW64
                grads[i]["b_enc"] += gb64
                if not li.transcode:
                    # SAE: encoder input is xn - b_dec, so b_dec also receives -grad_x (sparse_coder.py:637-638)
                    gx = (li.W_enc.to(dt)[st["idx"]] * gvm.unsqueeze(-1)).sum(1)
                    grads[i]["b_decs.0"] += -gx.sum(0)
                live.remove(i)
                del state[i]

    return StepResult(fvus, accts, idxs, grads, sae_outs)
