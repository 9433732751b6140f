"""CPU baseline leg of bench.py.  TEST/BENCH INFRASTRUCTURE ONLY (never imported by the product).

A torch-CPU restatement of the reference's *eager* train step, op for op
(F.linear -> relu -> torch.topk(sorted=False) -> post-encoder affine -> F.embedding_bag
(utils.py:442-445) -> skip GEMM -> FVU -> autograd, encoder grad-W / grad-b by index_add
(fused_encoder.py:140-155) -> torch.optim.Adam), fp32, all host cores.  The CLT wiring is the
explicit formula of SURVEY.md section 3.3 (coalesce_topk="none").  Its arithmetic is pinned
by tests/test_cpu_baseline.py against oracle.clt_oracle (itself pinned on reference goldens)."""
from __future__ import annotations

import time

import torch
import torch.nn.functional as F


class _Encoder(torch.autograd.Function):
    """fused_encoder.py:71-201 restated with index_add for grad-W (kernels.py:14-54 semantics)."""

    @staticmethod
    def forward(ctx, x, W, b, k):
        pre = F.relu(F.linear(x, W, b))
        vals, idx = torch.topk(pre, k, dim=1, sorted=False)
        ctx.save_for_backward(x, vals, idx)
        ctx.M = W.shape[0]
        ctx.mark_non_differentiable(idx)
        return vals, idx

    @staticmethod
    def backward(ctx, g, _):
        x, vals, idx = ctx.saved_tensors
        gv = g * (vals > 0)
        k = idx.shape[1]
        gb = torch.zeros(ctx.M).index_add_(0, idx.flatten(), gv.flatten())
        gW = torch.zeros(ctx.M, x.shape[1]).index_add_(0, idx.flatten(), x.repeat_interleave(k, 0) * gv.flatten()[:, None])
        return None, gW, gb, None


def make_layers(L, d, M, cross, seed=0):
    layers = []
    for i in range(L):
        torch.manual_seed(seed)
        T = min(cross, L - i)
        enc = torch.nn.Linear(d, M)
        p = dict(W_enc=enc.weight, b_enc=enc.bias)
        for t in range(T):
            p[f"W_dec{t}"] = torch.nn.Parameter(0.02 * torch.randn(M, d))
            p[f"b_dec{t}"] = torch.nn.Parameter(torch.zeros(d))
            p[f"W_skip{t}"] = torch.nn.Parameter(torch.zeros(d, d))
            p[f"post_enc{t}"] = torch.nn.Parameter(torch.zeros(M))
        p["T"] = T
        layers.append(p)
    return layers


def train_step(layers, xs, ys, mask, k, opt):
    live = []
    fvus = []
    for j, lp in enumerate(layers):
        v, idx = _Encoder.apply(xs[j], lp["W_enc"], lp["b_enc"], k)
        live.append(dict(i=j, v=v, idx=idx, slot=0, x=xs[j]))
        out = 0
        for s in live:
            li, t = layers[s["i"]], s["slot"]
            a = s["v"] + li[f"post_enc{t}"][s["idx"]] * (s["v"] > 0)
            o = F.embedding_bag(s["idx"], li[f"W_dec{t}"], per_sample_weights=a, mode="sum") + li[f"b_dec{t}"]
            o = o + s["x"] @ li[f"W_skip{t}"].T
            out = out + o
            s["slot"] += 1
        lm = mask[:, None].float()
        e = (ys[j] - out) * lm
        ybar = (ys[j] * lm).sum(0) / lm.sum(0)
        fvu = e.pow(2).sum() / ((ys[j] - ybar).pow(2) * lm).sum()
        fvus.append(fvu)
        live = [s for s in live if s["slot"] < layers[s["i"]]["T"]]
    # gradients are identical to the reference's per-layer backward + deferred encoder backward
    sum(fvus).backward()
    opt.step()
    opt.zero_grad()
    return [float(f) for f in fvus]


def _time_steps(L, d, M, k, N, min_seconds, max_steps):
    torch.manual_seed(0)
    layers = make_layers(L, d, M, cross=L)
    params = [p for lp in layers for n, p in lp.items() if n != "T"]
    opt = torch.optim.Adam(params, lr=1e-4)
    xs = [torch.randn(N, d) for _ in range(L)]
    ys = [x * 0.5 + torch.randn(N, d) for x in xs]
    mask = torch.ones(N, dtype=torch.bool)
    train_step(layers, xs, ys, mask, k, opt)  # warm-up
    t0 = time.perf_counter()
    steps = 0
    while steps < max_steps and (time.perf_counter() - t0 < min_seconds or steps < 2):
        train_step(layers, xs, ys, mask, k, opt)
        steps += 1
    return steps, time.perf_counter() - t0


def run_c1(min_seconds=6.0):
    """BASELINE.json configs[0] / BASELINE.md section 3, UN-EXTRAPOLATED: the single-layer transcoder (d=768, expansion 32,
    k=32, skip connection, fp32) as whole fwd + bwd + Adam steps on the host cores, at N = 1024 and N = 8192."""
    out = {}
    for N in (1024, 8192):
        steps, dt = _time_steps(1, 768, 24576, 32, N, min_seconds / 2, 20)
        out[f"N{N}"] = dict(tokens_per_s=round(N * steps / dt, 1), steps=steps, seconds=round(dt, 2))
    return dict(config="configs[0]: single-layer transcoder d=768, latents=24576, k=32, fp32, whole steps (not extrapolated)",
                cores=torch.get_num_threads(), **out)


def run(L=2, d=768, M=98304, k=32, N=1024, min_seconds=10.0, max_steps=50, full_L=12):
    steps, dt = _time_steps(L, d, M, k, N, min_seconds, max_steps)
    pairs = L * (L + 1) // 2
    full_pairs = full_L * (full_L + 1) // 2
    return dict(value=N * steps / dt, unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                sample=(f"{steps} fwd+bwd+Adam steps of a {L}-layer CLT slice (of {full_L}; {pairs} of {full_pairs} "
                        f"decoder pairs), d={d}, M={M}, k={k}, N={N}, fp32, torch-CPU eager restatement; {dt:.1f} s; "
                        f"whole-model estimate ~{N * steps / dt * pairs / full_pairs:.0f} tokens/s"))
