"""GPU tests of the latent-shard kernel paths and of the explicit engine (tp = 1) against the oracle."""
import math

import pytest
import torch

from oracle import clt_oracle as O
from oracle import ops_cpu
from clt_training_amd.engine import ROUTES

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from clt_training_amd import hip_ops
    return hip_ops


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tp", [2, 4, 8])
def test_sharded_decode_sums_to_full(ops, dtype, tp):
    """Partial decodes over latent shards (compacted, not masked) add up to the unsharded decode; every
    (token, slot) gradient entry is produced by exactly one shard."""
    g = torch.Generator().manual_seed(tp)
    N, d, M, k = 130, 256, 1024, 70
    W = (torch.randn(M, d, generator=g) / math.sqrt(d)).to(dtype)
    v = torch.rand(N, k, generator=g)
    v[:, ::6] = 0
    idx = torch.stack([torch.randperm(M, generator=g)[:k] for _ in range(N)])
    pe, ps = 0.1 * torch.randn(M, generator=g), 1 + 0.1 * torch.randn(M, generator=g)
    G = torch.randn(N, d, generator=g).to(dtype)
    full, acts_full = ops.decode_fwd(idx.to(DEV), v.to(DEV), W.to(DEV), post_enc=pe.to(DEV), post_scale=ps.to(DEV), want_acts=True,
                                     out_dtype=torch.float32)
    ga_full, ep_full, es_full = ops.decode_bwd_acts(idx.to(DEV), v.to(DEV), W.to(DEV), G.to(DEV), post_enc=pe.to(DEV),
                                                    post_scale=ps.to(DEV), want_e_post=True)
    Ml = M // tp
    acc = torch.zeros(N, d, device=DEV)
    acts = torch.zeros(N, k, device=DEV)
    ga = torch.zeros(N, k, device=DEV)
    ep = torch.zeros(N, k, device=DEV)
    owners = torch.zeros(N, k, device=DEV)
    for r in range(tp):
        lo = r * Ml
        Wl, pel, psl = W[lo:lo + Ml].to(DEV), pe[lo:lo + Ml].to(DEV), ps[lo:lo + Ml].to(DEV)
        _, a = ops.decode_fwd(idx.to(DEV), v.to(DEV), Wl, post_enc=pel, post_scale=psl, want_acts=True, lat_begin=lo,
                              out=acc, accumulate=True)
        acts += a
        owners += (a != 0)
        _, e_post, e_scale = ops.decode_bwd_acts(idx.to(DEV), v.to(DEV), Wl, G.to(DEV), post_enc=pel, post_scale=psl,
                                                 grad_acts=ga, want_e_post=True, lat_begin=lo)
        ep += e_post
        seg = ops.segments_build(idx.to(DEV), Ml, lat_begin=lo)
        ref = ops_cpu.segments_build(idx, Ml, lat_begin=lo)
        n_real = int(ref.offsets[-1])
        assert torch.equal(seg.offsets.cpu().long(), ref.offsets)
        assert torch.equal(seg.perm.cpu().long()[:n_real], ref.perm[:n_real])
        gW = ops.spt_dense(seg, a.view(-1), G.to(DEV), out_dtype=torch.float32)
        gW_ref = ops_cpu.spt_dense(ref, a.cpu().view(-1), G.float(), out_dtype=torch.float32)
        torch.testing.assert_close(gW.cpu(), gW_ref, rtol=1e-4, atol=1e-4)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(acc, full, **tol)
    torch.testing.assert_close(acts, acts_full, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(ga, ga_full, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ep, ep_full, rtol=1e-4, atol=1e-4)
    assert int(owners.max()) <= 1


@pytest.mark.parametrize("masked", [False, True])
def test_fvu_stages_equal_fused(ops, masked):
    g = torch.Generator().manual_seed(3)
    N, d = 500, 128
    y = torch.randn(N, d, generator=g).to(DEV)
    yh = (y.cpu() + 0.3 * torch.randn(N, d, generator=g)).to(DEV)
    m = None
    if masked:
        m = torch.ones(N, dtype=torch.bool)
        m[::9] = False
        m = m.to(DEV)
    sc, G = ops.fvu_fwd_bwd(y, yh, m, gscale=0.5)
    # two token shards
    halves = [(y[:200], yh[:200], None if m is None else m[:200]), (y[200:], yh[200:], None if m is None else m[200:])]
    cs = sum(ops.fvu_colsum(a, mm) for a, _, mm in halves)
    l2tv = sum(ops.fvu_partial(a, b, mm, cs) for a, b, mm in halves)
    outs = [ops.fvu_grad(a, b, mm, l2tv, gscale=0.5) for a, b, mm in halves]
    torch.testing.assert_close(outs[0][0], sc, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(torch.cat([outs[0][1], outs[1][1]]), G, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("case", ["clt", "sae", "concat", "per_layer_k"])
def test_engine_tp1_vs_oracle(case):
    """The explicit engine on the HIP kernels (tp = 1), 3 train steps incl. step-0 init, Adam and schedule.
    ``per_layer_k`` (trainer.py:364-368, in force from the first optimizer step on): the live sources of a target carry
    different k, so the one-launch fan-out does not apply and the step takes the per-source kernels."""
    import sys
    sys.path.insert(0, "tests")
    from test_tp_gloo import CASES, _make_problem, _oracle_layers
    from clt_training_amd import SparseCoderConfig, TrainConfig
    from clt_training_amd.engine import ShardedCoder
    from clt_training_amd.tp import TPTrainer

    sae_kw, L, cross = CASES[dict(clt="clt_skip_scale_norm_div", sae="sae", concat="clt_concat_scale_norm",
                                  per_layer_k="clt_skip_scale_norm_div")[case]]
    pk = [3, 5, 4][:L] if case == "per_layer_k" else []
    d, N = 16, 24
    saes, steps = _make_problem(sae_kw, L, cross, d, N)
    hooks = [f"h.{i}" for i in range(L)]
    cfg = TrainConfig(sae=SparseCoderConfig(**sae_kw), hookpoints=hooks, cross_layer=cross, optimizer="adam", lr=1e-2,
                      lr_warmup_steps=2, per_layer_k=list(pk))
    coders = [ShardedCoder.from_full(s, 0, 1, DEV) for s in saes]
    tr = TPTrainer(cfg, [None] * 4, device=DEV, input_widths={h: d for h in hooks}, coders=coders)
    layers = _oracle_layers(saes)
    state = [dict() for _ in range(L)]
    for s, (xs, ys, m) in enumerate(steps):
        if not sae_kw["transcode"]:
            xs = ys
        if s == 0:
            for l in range(L):
                O.step0_init(layers[l], xs[l], ys[l], sae_kw.get("normalize_io", False))
        batch = {h: (xs[l].to(DEV), ys[l].to(DEV)) for l, h in enumerate(hooks)}
        batch["loss_mask"] = m.to(DEV)
        got = tr.step(batch)
        r = O.clt_train_step(layers, xs, ys, m, pk if pk and s > 0 else sae_kw["k"],
                             normalize_io=sae_kw.get("normalize_io", False),
                             divide_cross_layer=sae_kw.get("divide_cross_layer", False),
                             coalesce=sae_kw.get("coalesce_topk", "none"))
        torch.testing.assert_close(torch.tensor([float(got[h]) for h in hooks]), torch.stack(r.fvus), rtol=2e-4, atol=1e-6)
        lr = 1e-2 * O.warmup_linear_factor(s, 2, 4)
        for l, lp in enumerate(layers):
            tensors = {"W_enc": lp.W_enc, "b_enc": lp.b_enc}
            for i in range(len(lp.W_decs)):
                tensors[f"W_decs.{i}"] = lp.W_decs[i]
                tensors[f"b_decs.{i}"] = lp.b_decs[i]
                if lp.W_skips[i] is not None:
                    tensors[f"W_skips.{i}"] = lp.W_skips[i]
            for i in range(len(lp.post_encs)):
                tensors[f"post_encs.{i}"] = lp.post_encs[i]
                if lp.post_enc_scales is not None:
                    tensors[f"post_enc_scales.{i}"] = lp.post_enc_scales[i]
            for n, p in tensors.items():
                mm, vv = state[l].get(n, (torch.zeros_like(p).double(), torch.zeros_like(p).double()))
                pn, mm, vv = O.adam_step(p, r.grads[l][n], mm, vv, s + 1, lr)
                p.copy_(pn.float())
                state[l][n] = (mm, vv)
    # three Adam steps at lr 1e-2: g / (sqrt(v) + eps) turns fp32 accumulation-order noise on a near-zero gradient element
    # into a visible fraction of lr on that element, hence the absolute floor of 5e-3 * lr
    tol = dict(rtol=1e-4, atol=5e-5)
    for l, lp in enumerate(layers):
        c = coders[l]
        torch.testing.assert_close(c.W_enc.detach().cpu(), lp.W_enc, **tol)
        torch.testing.assert_close(c.b_enc.detach().cpu(), lp.b_enc, **tol)
        for i in range(len(lp.W_decs)):
            torch.testing.assert_close(c.W_decs[i].detach().cpu(), lp.W_decs[i], **tol)
            torch.testing.assert_close(c.b_decs[i].detach().cpu(), lp.b_decs[i], **tol)
        for i in range(len(lp.post_encs)):
            torch.testing.assert_close(c.post_encs[i].detach().cpu(), lp.post_encs[i], **tol)


def test_fused_row_adam_equals_grad_then_adam(monkeypatch):
    """clt_spt_dense_adam (row gradients consumed by the optimizer update in the same kernel) against the two-kernel
    route clt_spt_dense -> .grad -> clt_adam_step: every parameter bit-identical after 3 bf16 train steps."""
    import sys
    sys.path.insert(0, "tests")
    from clt_training_amd import SparseCoderConfig, TrainConfig
    from clt_training_amd.tp import TPTrainer

    L, d, N, k = 3, 64, 512, 8
    hooks = [f"h.{i}" for i in range(L)]
    g = torch.Generator().manual_seed(3)
    batches = []
    for s in range(3):
        b = {}
        for h in hooks:
            x = torch.randn(N, d, generator=g) + 0.1 * torch.randn(d, generator=g)
            b[h] = (x.bfloat16().to(DEV), (0.5 * x + 0.5 * torch.randn(N, d, generator=g)).bfloat16().to(DEV))
        b["loss_mask"] = (torch.rand(N, generator=g) > 0.1).to(DEV)
        batches.append(b)

    def run(fused):
        monkeypatch.setattr(ROUTES, "fused_adam", fused)
        torch.manual_seed(0)
        cfg = TrainConfig(sae=SparseCoderConfig(dtype="bfloat16", expansion_factor=32, k=k, transcode=True,
                                                skip_connection=True, post_encoder_scale=True),
                          hookpoints=list(hooks), cross_layer=L, optimizer="adam", lr=1e-3, lr_warmup_steps=1,
                          log_to_wandb=False)
        tr = TPTrainer(cfg, batches, device=DEV, num_batches=10)
        assert (tr.engine.fused_opt is not None) == fused
        with torch.no_grad():
            gg = torch.Generator(device=DEV).manual_seed(1)
            for sae in tr.saes.values():
                for n, p in sae.named_parameters():
                    if n.startswith("W_dec"):
                        p.copy_((0.05 * torch.randn(p.shape, generator=gg, device=DEV)).to(p.dtype))
        fv = [tr.step(b) for b in batches]
        params = {f"{h}.{n}": p.detach().clone() for h, sae in tr.saes.items() for n, p in sae.named_parameters()}
        return fv, params

    fv_a, pa = run(True)
    fv_b, pb = run(False)
    for n in pa:
        assert torch.equal(pa[n], pb[n]), n
    for a, b in zip(fv_a, fv_b):
        for h in hooks:
            assert float(a[h]) == float(b[h])
    moved = [n for n in pa if "W_decs" in n or n.endswith("W_enc")]
    assert moved


@pytest.mark.parametrize("shape", ["c3_d2048_k64_clt", "c4_d4096_k192_sae"])
def test_engine_fast_path_equals_separate_kernels_at_config_widths(monkeypatch, shape):
    """BASELINE configs[2] / configs[3] widths (Llama-3.2-1B CLT: d=2048, k=64; Llama-3-8B SAE: d=4096, k=192, unit-norm
    decoder) through the engine's fast path (one-launch fan-out, row-pass decoder backward, fused Adam where allowed)
    against the same steps on the single-pair kernels (engine.ROUTES: multi / rows_pass / fused_adam off): bf16
    tolerance of BASELINE.json (the one-launch fan-out rounds the sum over sources once instead of once per source)."""
    from clt_training_amd import SparseCoderConfig, TrainConfig
    from clt_training_amd.tp import TPTrainer

    if shape.startswith("c3"):
        L, d, N, k, M, kw = 3, 2048, 192, 64, 8192, dict(transcode=True, skip_connection=True, post_encoder_scale=True)
    else:
        L, d, N, k, M, kw = 1, 4096, 128, 192, 8192, dict(transcode=False, normalize_decoder=True)
    hooks = [f"h.{i}" for i in range(L)]
    g = torch.Generator().manual_seed(5)
    batches = []
    for s in range(2):
        b = {}
        for h in hooks:
            x = torch.randn(N, d, generator=g) + 0.1 * torch.randn(d, generator=g)
            b[h] = (x.bfloat16().to(DEV), (0.5 * x + 0.5 * torch.randn(N, d, generator=g)).bfloat16().to(DEV))
        b["loss_mask"] = (torch.rand(N, generator=g) > 0.1).to(DEV)
        batches.append(b)

    def run(fast):
        for name in ("multi", "rows_pass", "fused_adam"):
            monkeypatch.setattr(ROUTES, name, fast)
        torch.manual_seed(0)
        cfg = TrainConfig(sae=SparseCoderConfig(dtype="bfloat16", num_latents=M, k=k, **kw), hookpoints=list(hooks),
                          cross_layer=L if kw["transcode"] else 0, optimizer="adam", lr=1e-3, lr_warmup_steps=1,
                          log_to_wandb=False)
        tr = TPTrainer(cfg, batches, device=DEV, num_batches=10)
        with torch.no_grad():
            gg = torch.Generator(device=DEV).manual_seed(1)
            for sae in tr.saes.values():
                for n, p in sae.named_parameters():
                    if n.startswith("W_dec") and kw["transcode"]:
                        p.copy_((0.02 * torch.randn(p.shape, generator=gg, device=DEV)).to(p.dtype))
        p0 = {f"{h}.{n}": p.detach().float().clone() for h, sae in tr.saes.items() for n, p in sae.named_parameters()}
        fv = [{h: float(v) for h, v in tr.step(b).items()} for b in batches]
        return fv, {f"{h}.{n}": p.detach().float().clone() for h, sae in tr.saes.items() for n, p in sae.named_parameters()}, p0

    fv_a, pa, p0 = run(True)
    fv_b, pb, _ = run(False)
    for a, b in zip(fv_a, fv_b):
        for h in hooks:
            assert abs(a[h] - b[h]) <= 1e-2 * abs(b[h]), (h, a[h], b[h])
    for n in pa:
        err = float((pa[n] - pb[n]).norm() / (pb[n].norm() + 1e-12))
        # A parameter that starts at zero IS its Adam update after these two steps: g / (sqrt(v) + eps) turns the two
        # routes' bf16-level gradient differences into percent-level differences on near-zero gradient elements
        # (tests/test_gpu_tp_shapes.py bounds the same quantity against the oracle at 0.15).
        assert err < (1e-2 if float(p0[n].norm()) > 0 else 5e-2), (n, err)


@pytest.mark.parametrize("name", ["clt_none_L3", "clt_none_div_scale_norm_L4", "clt_concat_L3"])
def test_engine_vs_reference_golden(name):
    """The explicit engine (the path bench.py times; plain-gradient route, no optimizer attached) directly against the
    golden vectors the REFERENCE produced for the hook sequence of trainer.py:598-682: FVU per layer, reconstruction,
    merged indices and the gradient of every parameter, fp32, rtol 1e-5.  clt_concat_L3 is BASELINE configs[4]'s mode
    (coalesce_topk="concat", post_encoder_scale, normalize_io)."""
    import sys
    sys.path.insert(0, "tests")
    from conftest import golden_grad_name, load_golden
    from test_gpu_clt import build_sae
    from clt_training_amd import TrainConfig, hip_ops
    from clt_training_amd.engine import CLTEngine, ShardedCoder

    t, meta = load_golden(name)
    L = meta["L"]
    saes = [build_sae(t, meta, l, False) for l in range(L)]
    coders = [ShardedCoder.from_full(s, 0, 1, DEV) for s in saes]
    eng = CLTEngine(TrainConfig(sae=saes[0].cfg, hookpoints=[f"h.{i}" for i in range(L)], cross_layer=meta["cross_layer"]),
                    coders, hip_ops)
    eng.keep_outputs = True
    eng.reset()
    eng.begin_step(packed=True)
    mask = t["mask"].to(DEV)
    for l in range(L):
        fvu = eng.layer_step(l, t[f"x.{l}"].to(DEV), t[f"y.{l}"].to(DEV), mask)
        torch.testing.assert_close(fvu.cpu(), t[f"fvu.{l}"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(eng.last_out[l].float().cpu(), t[f"sae_out.{l}"], rtol=1e-5, atol=1e-5)
        _, gi = O.topk_merge(torch.zeros(eng.last_idx[l].shape), eng.last_idx[l].cpu(), eng.last_idx[l].shape[1])
        assert torch.equal(gi.sort(1).values, t[f"idx.{l}"].sort(1).values)
    assert not eng.live
    for l, c in enumerate(coders):
        want = {golden_grad_name(k[len(f"g.{l}."):]): v for k, v in t.items() if k.startswith(f"g.{l}.")}
        got = dict(c.named_parameters())
        assert set(want) == set(got), (sorted(want), sorted(got))
        for n, ref in want.items():
            assert got[n].grad is not None, (l, n)
            # rtol 1e-5 with an absolute floor of 1e-5 x the tensor's largest magnitude: an element that is a sum with
            # cancellation (the library GEMMs behind the skip-matrix gradients accumulate in an order that depends on
            # the kernel the library picks on the box) carries an error relative to its terms
            atol = max(2e-7, 1e-5 * float(ref.abs().max()))
            torch.testing.assert_close(got[n].grad.float().cpu(), ref, rtol=1e-5, atol=atol, msg=lambda m: f"{l}.{n}: {m}")


def test_engine_stacked_encoders_equal_layer_by_layer(monkeypatch):
    """engine.ROUTES.enc_layers (all layers' fused encode + top-k stacked into one set of launches from step 1 on:
    clt_encode_topk_fwd_layers) leaves every step bit for bit what the layer-by-layer encoders give: fvu and every
    parameter after 3 steps of a 3-layer bf16 CLT with skip connections at N = 2048 tokens."""
    from clt_training_amd import SparseCoderConfig, TrainConfig
    from clt_training_amd.tp import TPTrainer

    L, d, N, k, M = 3, 256, 2048, 16, 4096
    hooks = [f"h.{i}" for i in range(L)]
    g = torch.Generator().manual_seed(11)
    batches = []
    for s in range(3):
        b = {}
        for h in hooks:
            x = torch.randn(N, d, generator=g) + 0.1 * torch.randn(d, generator=g)
            b[h] = (x.bfloat16().to(DEV), (0.5 * x + 0.5 * torch.randn(N, d, generator=g)).bfloat16().to(DEV))
        b["loss_mask"] = (torch.rand(N, generator=g) > 0.1).to(DEV)
        batches.append(b)
    calls = []

    def run(stacked):
        monkeypatch.setattr(ROUTES, "enc_layers", stacked)
        torch.manual_seed(0)
        cfg = TrainConfig(sae=SparseCoderConfig(dtype="bfloat16", num_latents=M, k=k, transcode=True, skip_connection=True),
                          hookpoints=list(hooks), cross_layer=L, optimizer="adam", lr=1e-3, lr_warmup_steps=1,
                          log_to_wandb=False)
        tr = TPTrainer(cfg, batches, device=DEV, num_batches=10)
        real = tr.engine.encode_all

        def spy(xs):
            out = real(xs)
            calls.append((stacked, out is not None))
            return out

        tr.engine.encode_all = spy
        with torch.no_grad():
            gg = torch.Generator(device=DEV).manual_seed(1)
            for sae in tr.saes.values():
                for n, p in sae.named_parameters():
                    if n.startswith("W_dec"):
                        p.copy_((0.02 * torch.randn(p.shape, generator=gg, device=DEV)).to(p.dtype))
        fv = [{h: float(v) for h, v in tr.step(b).items()} for b in batches]
        return fv, {f"{h}.{n}": p.detach().clone() for h, sae in tr.saes.items() for n, p in sae.named_parameters()}

    fv_a, pa = run(True)
    fv_b, pb = run(False)
    assert calls.count((True, True)) == 2 and (False, True) not in calls  # steps 1 and 2 took the stacked path
    assert fv_a == fv_b
    for n in pa:
        assert torch.equal(pa[n], pb[n]), n
