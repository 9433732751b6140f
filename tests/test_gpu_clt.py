"""GPU parity of the host mirror (SparseCoder / MidDecoder / CrossLayerRunner / Trainer) against
(a) golden vectors produced by the reference itself, fp32, rtol 1e-5, index sets bit-exact;
(b) the CPU oracle in the bf16 mode, rtol 1e-2 (BASELINE.json)."""
import math

import pytest
import torch

from conftest import golden_grad_name, golden_layers, load_golden
from clt_training_amd.engine import ROUTES
from oracle import clt_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build_sae(t, meta, l, single, dtype=torch.float32, cfg_over=None):
    from clt_training_amd import SparseCoder, SparseCoderConfig

    cfgd = dict(meta["cfg"])
    if not single:
        cfgd["n_targets"] = min(meta["cross_layer"], meta["L"] - l)
        # trainer.py:125-131: sources still listed at position l (p + n_targets(p) >= l, as the reference counts them)
        cfgd["n_sources"] = sum(1 for p in range(l + 1) if p + min(meta["cross_layer"], meta["L"] - p) >= l)
    cfgd.update(cfg_over or {})
    cfg = SparseCoderConfig(**cfgd)
    sae = SparseCoder(meta["d"], cfg, DEV, dtype=torch.float32)
    pre = "p." if single else f"p.{l}."
    sd = sae.state_dict()
    with torch.no_grad():
        for k, v in t.items():
            if k.startswith(pre):
                name = k[len(pre):]
                sd[name].copy_(v.to(DEV).to(sd[name].dtype))
    return sae


def run_reference_sequence(saes, xs, ys, mask):
    """The hook sequence of trainer.py:598-682."""
    from clt_training_amd import CrossLayerRunner

    runner = CrossLayerRunner()
    outs, encs = [], []
    for l, sae in enumerate(saes):
        enc = runner.encode(xs[l], sae)
        out = runner.decode(enc, ys[l], module_name=f"layers.{l}", detach_grad=True, loss_mask=mask)
        encs.append((enc.latent_indices, (enc.original_activations if hasattr(enc, "original_activations")
                                          else enc.latent_acts).detach()))
        outs.append(out)
        out.fvu.backward()
        runner.restore()
    assert len(runner.outputs) == 0
    return outs, encs


@pytest.fixture()
def fp32_exact():
    import clt_training_amd.sparse_coder as sc

    old = sc.AUTOCAST_ENABLED
    sc.AUTOCAST_ENABLED = False
    torch.backends.cuda.matmul.allow_tf32 = False
    yield
    sc.AUTOCAST_ENABLED = old


def check_against_golden(t, saes, outs, encs, single):
    for l, sae in enumerate(saes):
        sfx = "" if single else f".{l}"
        v, i = O.topk_merge(encs[l][1].float().cpu(), encs[l][0].cpu(), encs[l][0].shape[1])
        live = t[f"vals{sfx}"] > 0  # (batchtopk: the index of a zeroed slot is not defined by the reference)
        assert torch.equal(v > 0, live) and torch.equal(i[live], t[f"idx{sfx}"][live])  # bit-exact index sets
        torch.testing.assert_close(v, t[f"vals{sfx}"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(outs[l].sae_out.float().cpu(), t[f"sae_out{sfx}"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(outs[l].fvu.float().cpu(), t[f"fvu{sfx}"], rtol=1e-5, atol=1e-6)
        gpre = "g." if single else f"g.{l}."
        seen = 0
        for key, ref in t.items():
            if key.startswith(gpre):
                p = dict(sae.named_parameters())[key[len(gpre):]]
                assert p.grad is not None, key
                torch.testing.assert_close(p.grad.float().cpu(), ref, rtol=1e-5, atol=2e-7, msg=lambda m: f"{key}: {m}")
                seen += 1
        assert seen >= 4


@pytest.mark.parametrize("name", ["single_transcoder_skip", "single_transcoder_scale_norm", "single_sae",
                                  "single_transcoder_groupmax", "single_transcoder_batchtopk"])
def test_single_layer_vs_reference_golden(name, fp32_exact):
    t, meta = load_golden(name)
    sae = build_sae(t, meta, 0, True)
    mask = t["mask"].to(DEV) if "mask" in t else None
    outs, encs = run_reference_sequence([sae], [t["x"].to(DEV)], [t["y"].to(DEV)], mask)
    check_against_golden(t, [sae], outs, encs, True)
    if name == "single_sae":
        sae.remove_gradient_parallel_to_decoder_directions()
        torch.testing.assert_close(sae.W_dec.grad.cpu(), t["g_proj.W_dec"], rtol=1e-5, atol=2e-7)


@pytest.mark.parametrize("name", ["clt_none_L3", "clt_none_div_scale_norm_L4", "clt_concat_L3", "clt_concat_topkcoal_L3",
                                  "clt_perlayer_L3", "clt_none_per_source_tied_L3",
                                  "clt_none_secondary_tied_L3"])
def test_clt_vs_reference_golden(name, fp32_exact):
    t, meta = load_golden(name)
    L = meta["L"]
    saes = [build_sae(t, meta, l, False) for l in range(L)]
    xs = [t[f"x.{l}"].to(DEV) for l in range(L)]
    ys = [t[f"y.{l}"].to(DEV) for l in range(L)]
    outs, encs = run_reference_sequence(saes, xs, ys, t["mask"].to(DEV))
    check_against_golden(t, saes, outs, encs, False)


def rel_err(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("name", ["clt_none_div_scale_norm_L4", "clt_concat_L3"])
def test_clt_bf16_mode_vs_oracle(name):
    """Mode B (bf16 encoder / decoder weights, bf16 autocast): rtol 1e-2 against the oracle
    evaluated on the same bf16-rounded weights."""
    t, meta = load_golden(name)
    L = meta["L"]
    saes = [build_sae(t, meta, l, False, cfg_over=dict(dtype="bfloat16")) for l in range(L)]
    assert saes[0].encoder.weight.dtype == torch.bfloat16 and saes[0].W_dec.dtype == torch.bfloat16
    xs = [t[f"x.{l}"].to(torch.bfloat16) for l in range(L)]
    ys = [t[f"y.{l}"].to(torch.bfloat16) for l in range(L)]
    outs, encs = run_reference_sequence(saes, [x.to(DEV) for x in xs], [y.to(DEV) for y in ys], t["mask"].to(DEV))
    layers = golden_layers(t, meta, L)
    for lp, sae in zip(layers, saes):  # the oracle sees exactly the rounded parameters
        lp.W_enc = sae.encoder.weight.detach().float().cpu()
        lp.b_enc = sae.encoder.bias.detach().float().cpu()
        lp.W_decs = [(sae.W_decs[i] if hasattr(sae, "W_decs") else sae.W_dec).detach().float().cpu()
                     for i in range(len(lp.W_decs))]
    cfg = meta["cfg"]
    forced = [e[0].cpu() for e in encs]
    res = O.clt_train_step(layers, [x.float() for x in xs], [y.float() for y in ys], t["mask"], cfg["k"],
                           coalesce=cfg.get("coalesce_topk", "none"), normalize_io=cfg.get("normalize_io", False),
                           divide_cross_layer=cfg.get("divide_cross_layer", False), act_dtype=torch.bfloat16,
                           forced_idx=forced)
    d = meta["d"]
    for l, sae in enumerate(saes):
        # the selection must be an exact top-k of the fp64 pre-activations up to bf16 resolution
        lp = layers[l]
        xn = xs[l].double() * (d**0.5 / lp.in_norm.double()) if cfg.get("normalize_io") else xs[l].double()
        pre = O.encoder_preacts(xn, lp.W_enc.double(), lp.b_enc.double(), None)
        slack = O.topk_selection_slack(pre, forced[l])
        assert float((slack / pre.max(1).values).max()) < 2e-2
        assert all(len(set(r.tolist())) == cfg["k"] for r in forced[l])
        torch.testing.assert_close(encs[l][1].float().cpu(), res.top_acts[l].float(), rtol=1e-2, atol=1e-2)
        assert rel_err(outs[l].sae_out.float().cpu(), res.sae_outs[l]) < 1e-2
        assert abs(float(outs[l].fvu) - float(res.fvus[l])) < 1e-2 * float(res.fvus[l])
        for n, p in sae.named_parameters():
            ref = res.grads[l][golden_grad_name(n)]
            if ref.norm() > 0:
                assert rel_err(p.grad.detach().float().cpu(), ref) < 2e-2, (l, n, rel_err(p.grad.detach().float().cpu(), ref))


def set_rows(idx):
    return [frozenset(r.tolist()) for r in idx]


def test_trainer_three_steps_vs_oracle(fp32_exact):
    """Trainer (step-0 init, runner order, Adam, WarmupLinearSchedule) against the oracle restatement."""
    from clt_training_amd import SparseCoderConfig, TrainConfig, Trainer

    L, d, N, k, cross = 3, 32, 64, 4, 2
    hooks = [f"h.{i}.mlp" for i in range(L)]
    g = torch.Generator().manual_seed(0)
    steps = 4
    data = []
    for s in range(steps):
        b = {}
        for h in hooks:
            x = torch.randn(N, d, generator=g) + 0.2 * torch.randn(d, generator=g)
            y = 0.5 * x + 0.5 * torch.randn(N, d, generator=g)
            b[h] = (x, y)
        m = torch.ones(N, dtype=torch.bool)
        m[::16] = False
        b["loss_mask"] = m
        data.append(b)
    sae_cfg = SparseCoderConfig(transcode=True, expansion_factor=4, k=k, skip_connection=True, normalize_io=True,
                                post_encoder_scale=True)
    cfg = TrainConfig(sae=sae_cfg, hookpoints=hooks, cross_layer=cross, optimizer="adam", lr=1e-2, lr_warmup_steps=2)
    gpu_data = [{h: tuple(v.to(DEV) for v in val) if isinstance(val, tuple) else val.to(DEV) for h, val in b.items()}
                for b in data]
    tr = Trainer(cfg, gpu_data, device=DEV)
    # decoders start at zero in a transcoder: give them (and post_enc) non-trivial values
    gi = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for sae in tr.saes.values():
            for n, p in sae.named_parameters():
                if n.startswith("W_dec"):
                    p.copy_((torch.randn(p.shape, generator=gi) / math.sqrt(d)).to(DEV))
                if n.startswith("post_encs"):
                    p.copy_((0.05 * torch.randn(p.shape, generator=gi)).to(DEV))

    # ---- oracle trainer
    layers = []
    for l, h in enumerate(hooks):
        sae = tr.saes[h]
        sd = {n: p.detach().float().cpu().clone() for n, p in sae.state_dict().items()}
        T = sae.cfg.n_targets
        def lst(stem, plural):
            return [sd[stem if i == 0 else f"{plural}.{i}"] for i in range(T)]
        layers.append(O.LayerParams(sd["encoder.weight"], sd["encoder.bias"], lst("W_dec", "W_decs"),
                                    lst("b_dec", "b_decs"), lst("W_skip", "W_skips"),
                                    [sd[f"post_encs.{i}"] for i in range(T)],
                                    [sd[f"post_enc_scales.{i}"] for i in range(T)], T, sd["in_norm"], sd["out_norm"], True))
    state = [dict() for _ in range(L)]
    fvu_hist = []
    for s in range(steps):
        if s == 0:
            for l, h in enumerate(hooks):
                O.step0_init(layers[l], data[0][h][0], data[0][h][1], True)
        res = O.clt_train_step(layers, [data[s][h][0] for h in hooks], [data[s][h][1] for h in hooks],
                               data[s]["loss_mask"], k, normalize_io=True)
        fvu_hist.append([float(f) for f in res.fvus])
        lr = 1e-2 * O.warmup_linear_factor(s, 2, steps)
        for l in range(L):
            lp = layers[l]
            tensors = {"W_enc": lp.W_enc, "b_enc": lp.b_enc}
            for i in range(lp.n_targets):
                tensors[f"W_decs.{i}"] = lp.W_decs[i]; tensors[f"b_decs.{i}"] = lp.b_decs[i]
                tensors[f"W_skips.{i}"] = lp.W_skips[i]; tensors[f"post_encs.{i}"] = lp.post_encs[i]
                tensors[f"post_enc_scales.{i}"] = lp.post_enc_scales[i]
            for n, p in tensors.items():
                m, v = state[l].get(n, (torch.zeros_like(p).double(), torch.zeros_like(p).double()))
                pn, m, v = O.adam_step(p, res.grads[l][n], m, v, s + 1, lr)
                p.copy_(pn.float())
                state[l][n] = (m, v)
    # ---- HIP trainer
    got_fvu = []
    for b in gpu_data:
        f = tr.step(b)
        got_fvu.append([float(f[h]) for h in hooks])
    torch.testing.assert_close(torch.tensor(got_fvu), torch.tensor(fvu_hist), rtol=2e-4, atol=1e-6)
    for l, h in enumerate(hooks):
        sae = tr.saes[h]
        torch.testing.assert_close(sae.encoder.weight.detach().cpu(), layers[l].W_enc, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(sae.encoder.bias.detach().cpu(), layers[l].b_enc, rtol=1e-4, atol=1e-5)
        for i in range(sae.cfg.n_targets):
            torch.testing.assert_close(sae.W_decs[i].detach().cpu(), layers[l].W_decs[i], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(sae.b_decs[i].detach().cpu(), layers[l].b_decs[i], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(sae.W_skips[i].detach().cpu(), layers[l].W_skips[i], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(sae.post_enc_scales[i].detach().cpu(), layers[l].post_enc_scales[i], rtol=1e-4, atol=1e-5)


def test_trainer_fused_adam_equals_grad_then_adam(monkeypatch):
    """The drop-in Trainer (SparseCoder / CrossLayerRunner / autograd) with the row gradients consumed by the Adam update
    inside the gradient kernels (fused_state) against the same steps through .grad -> Adam.step: every parameter bit
    for bit after 3 bf16 steps, for a CLT and for a unit-norm SAE (projection inside the row pass)."""
    from clt_training_amd import SparseCoderConfig, TrainConfig, Trainer

    for kw, L, cross in ((dict(transcode=True, skip_connection=True, post_encoder_scale=True), 3, 3),
                         (dict(transcode=False, normalize_decoder=True), 1, 0)):
        d, N, k = 64, 512, 8
        hooks = [f"h.{i}" for i in range(L)]
        g = torch.Generator().manual_seed(3)
        batches = []
        for s_ in range(3):
            b = {}
            for h in hooks:
                x = torch.randn(N, d, generator=g) + 0.1 * torch.randn(d, generator=g)
                b[h] = (x.bfloat16().to(DEV), (0.5 * x + 0.5 * torch.randn(N, d, generator=g)).bfloat16().to(DEV))
            b["loss_mask"] = (torch.rand(N, generator=g) > 0.1).to(DEV)
            batches.append(b)

        def run(fused):
            monkeypatch.setattr(ROUTES, "fused_adam", fused)
            torch.manual_seed(0)
            cfg = TrainConfig(sae=SparseCoderConfig(dtype="bfloat16", expansion_factor=32, k=k, **kw), hookpoints=list(hooks),
                              cross_layer=cross, optimizer="adam", lr=1e-3, lr_warmup_steps=1, log_to_wandb=False)
            tr = Trainer(cfg, batches, device=DEV, num_batches=10)
            assert tr._fused == fused
            with torch.no_grad():
                gg = torch.Generator(device=DEV).manual_seed(1)
                for sae in tr.saes.values():
                    for n, p in sae.named_parameters():
                        if n.startswith("W_dec") and kw["transcode"]:
                            p.copy_((0.05 * torch.randn(p.shape, generator=gg, device=DEV)).to(p.dtype))
            fv = [{h: float(v) for h, v in tr.step(b).items()} for b in batches]
            return fv, {f"{h}.{n}": p.detach().clone() for h, sae in tr.saes.items() for n, p in sae.named_parameters()}

        fv_a, pa = run(True)
        fv_b, pb = run(False)
        for n in pa:
            assert torch.equal(pa[n], pb[n]), n
        assert fv_a == fv_b


@pytest.mark.parametrize("kind", ["clt", "sae"])
@pytest.mark.parametrize("which", ["autograd", "engine"])
def test_model_driven_fit_and_resume(which, kind, tmp_path):
    """SURVEY.md section 8f rank 2 (trainer.py:73-111, 472-545, 684-692, 818-861, 275-354): a frozen random-init GPT-2
    drives ``fit`` through forward hooks; the parameters equal those of the same trainer fed the cached activations, and
    a run saved after two steps and resumed in a fresh trainer (``load_state`` + the dataset offset of :410-417) ends
    with the same parameters as the uninterrupted run."""
    transformers = pytest.importorskip("transformers")
    from clt_training_amd import SparseCoderConfig, TrainConfig, Trainer
    from clt_training_amd.activations import ModelActivations
    from clt_training_amd.tp import TPTrainer

    cls = Trainer if which == "autograd" else TPTrainer
    torch.manual_seed(0)
    mcfg = transformers.GPT2Config(n_layer=3, n_embd=32, n_head=2, vocab_size=50, n_positions=32, bos_token_id=1,
                                   eos_token_id=1, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    model = transformers.GPT2LMHeadModel(mcfg).eval().to(DEV)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(2, 50, (16, 32), generator=g)
    ids[:, 0] = 1

    class Rows(list):  # a list of rows with the `select` the resume path asks of a dataset (trainer.py:411-417)
        def select(self, rng):
            return Rows(self[i] for i in rng)

    rows = Rows({"input_ids": r} for r in ids)

    # "sae": plain unit-norm autoencoders on the residual stream (block outputs are tuples: trainer.py:500-504), BOS
    # filtered; the decoder's gradient projection and Adam update run inside the row pass
    wild, names = (["h.*.mlp"], ["h.0.mlp", "h.1.mlp", "h.2.mlp"]) if kind == "clt" else (["h.[0-9]"], ["h.0", "h.1", "h.2"])

    def make(run, hooks):
        torch.manual_seed(0)
        if kind == "clt":
            sae = SparseCoderConfig(transcode=True, expansion_factor=8, k=8, skip_connection=True)
            kw = dict(cross_layer=3, remove_first_token=True)
        else:
            sae = SparseCoderConfig(transcode=False, expansion_factor=8, k=8, normalize_decoder=True)
            kw = dict(cross_layer=0, filter_bos=True)
        return TrainConfig(sae=sae, hookpoints=hooks, optimizer="adam", lr=1e-2, lr_warmup_steps=2, batch_size=4,
                           log_to_wandb=False, save_dir=str(tmp_path), run_name=run, save_every=2, **kw)

    def params(tr):
        return {f"{h}.{n}": p.detach().clone() for h, s in tr.saes.items() for n, p in s.named_parameters()}

    def warm(tr):  # transcoder decoders start at zero
        gg = torch.Generator(device=DEV).manual_seed(1)
        with torch.no_grad():
            for s in tr.saes.values():
                for n, p in s.named_parameters():
                    if n.startswith("W_dec") and kind == "clt":
                        p.copy_((0.1 * torch.randn(p.shape, generator=gg, device=DEV)).to(p.dtype))

    # ---- model-driven, 4 steps; checkpoints land after steps 1 and 3 (save_every=2: (step + 1) % 2 == 0 with step = 1, 3)
    tr = cls(make("driven", list(wild)), rows, model)
    warm(tr)
    tr.fit()
    assert tr.global_step == 4 and tr.cfg.hookpoints == names
    want = params(tr)

    # ---- cached activations, the same 4 steps
    cfg2 = make("cached", list(names))
    cached = list(ModelActivations(model, rows, cfg2))
    tr2 = cls(cfg2, cached, device=DEV, num_batches=4)
    warm(tr2)
    for b in cached:
        tr2.step(b)
    got = params(tr2)
    for n in want:
        assert torch.equal(want[n], got[n]), n

    # ---- resume: one step, save, fresh trainer, load_state, fit the rest
    tr3 = cls(make("resume", list(wild)), rows, model)
    warm(tr3)
    tr3.step(next(iter(tr3.dataset)))
    tr3.save()
    tr4 = cls(make("resume", list(wild)), rows, model)
    tr4.load_state(f"{tmp_path}/resume")
    assert tr4.global_step == 1
    tr4.fit()
    assert tr4.global_step == 4
    got = params(tr4)
    for n in want:
        torch.testing.assert_close(got[n], want[n], rtol=0, atol=0, msg=n)


def test_trainer_signum_hip_equals_torch_signsgd(monkeypatch):
    """optimizer="signum" with force_lr_warmup (trainer.py:224-250): three bf16 CLT steps with the HIP SignSGD against the
    same trainer stepping with the torch restatement of sign_sgd.py on the same (HIP-computed) gradients: bit for bit.
    Without force_lr_warmup the reference wraps it in ScheduleFree (third-party, absent): that must raise."""
    from clt_training_amd import SparseCoderConfig, TrainConfig, Trainer
    from clt_training_amd import adam as adam_mod
    from oracle import ops_cpu

    d, N, k, L = 64, 512, 8, 3
    hooks = [f"h.{i}" for i in range(L)]
    g = torch.Generator().manual_seed(3)
    batches = []
    for s_ in range(3):
        b = {}
        for h in hooks:
            x = torch.randn(N, d, generator=g) + 0.1 * torch.randn(d, generator=g)
            b[h] = (x.bfloat16().to(DEV), (0.5 * x + 0.5 * torch.randn(N, d, generator=g)).bfloat16().to(DEV))
        batches.append(b)

    def make_cfg(**kw):
        return TrainConfig(sae=SparseCoderConfig(dtype="bfloat16", expansion_factor=32, k=k, transcode=True,
                                                 skip_connection=True), hookpoints=list(hooks), cross_layer=L,
                           optimizer="signum", lr_warmup_steps=2, log_to_wandb=False, **kw)

    with pytest.raises(NotImplementedError, match="force_lr_warmup"):
        Trainer(make_cfg(), batches, device=DEV, num_batches=10)

    def run(torch_double):
        if torch_double:
            monkeypatch.setattr(adam_mod, "SignSGD", ops_cpu.SignSGD)
        torch.manual_seed(0)
        tr = Trainer(make_cfg(force_lr_warmup=True), batches, device=DEV, num_batches=10)
        assert not tr._fused and type(tr.optimizers[0]).__name__ == "SignSGD"
        # trainer.py:225-231: 5e-3 / sqrt(M / 2^14) per group, a float
        assert abs(tr.optimizers[0].param_groups[0]["initial_lr"] - 5e-3 / (d * 32 / 2**14) ** 0.5) < 1e-12
        with torch.no_grad():
            gg = torch.Generator(device=DEV).manual_seed(1)
            for sae in tr.saes.values():
                for n, p in sae.named_parameters():
                    if n.startswith("W_dec"):
                        p.copy_((0.05 * torch.randn(p.shape, generator=gg, device=DEV)).to(p.dtype))
        fv = [{h: float(v) for h, v in tr.step(b).items()} for b in batches]
        return fv, {f"{h}.{n}": p.detach().clone() for h, sae in tr.saes.items() for n, p in sae.named_parameters()}

    fv_a, pa = run(False)
    fv_b, pb = run(True)
    for n in pa:
        assert torch.equal(pa[n], pb[n]), n
    assert fv_a == fv_b
    assert fv_a[2]["h.2"] < fv_a[0]["h.2"] or fv_a[2]["h.0"] < fv_a[0]["h.0"]  # and it trains
