"""SURVEY.md rows a18 (parameter set + init), a19 (step-0 init), a20 (Adam + LambdaLR order), a22 (fired-latent
bookkeeping) and f1 (checkpoint format) pinned on RUNS OF THE REFERENCE TRAINER: tests/golden/make_golden.py drove
``sparsify.trainer.Trainer.fit`` (mesh=None, optimizer="adam", FVU loss) on a random-init GPT-2 for three batches and
recorded the coders' parameters as constructed, after the data-dependent init, after every optimizer step, each step's
gradients and learning rate, ``num_tokens_since_fired`` and the activations the hooks saw; the checkpoint that run wrote
is kept under tests/golden/*_ckpt/.

CPU: the oracle restatement against those records.  GPU (-m gpu): ``Trainer`` (autograd mirror) and ``TPTrainer`` (the
explicit engine bench.py times), fed the recorded activations, against the same records -- fp32, rtol 1e-5."""
from pathlib import Path

import pytest
import torch

from conftest import GOLDEN, golden_grad_name, golden_layers, load_golden
from oracle import clt_oracle as O

CASES = ["trainer_clt_b1_09", "trainer_clt_b1_00", "trainer_sae_b1_09"]
FP32 = dict(rtol=1e-5, atol=1e-6)


def close(got, ref, what, rtol=1e-5, floor=2e-7):
    """rtol 1e-5 in fp32 (BASELINE.json), element-wise, with an absolute floor of rtol x the tensor's largest magnitude:
    the reference accumulates its GEMMs and column sums in fp32, so an element that is a sum with cancellation carries
    an error relative to the terms, not to the result."""
    atol = max(floor, rtol * float(ref.abs().max()))
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")


def oracle_tensors(lp):
    t = {"W_enc": lp.W_enc, "b_enc": lp.b_enc}
    for i in range(len(lp.W_decs)):
        t[f"W_decs.{i}"] = lp.W_decs[i]
        t[f"b_decs.{i}"] = lp.b_decs[i]
        if lp.W_skips[i] is not None:
            t[f"W_skips.{i}"] = lp.W_skips[i]
        t[f"post_encs.{i}"] = lp.post_encs[i]
        if lp.post_enc_scales is not None:
            t[f"post_enc_scales.{i}"] = lp.post_enc_scales[i]
    return t


def snapshot(t, tag, l):
    """{oracle key: tensor} of a recorded state dict (aliases like W_dec / W_decs.0 collapse onto one key)."""
    pre = f"{tag}.{l}."
    return {golden_grad_name(k[len(pre):]): v for k, v in t.items() if k.startswith(pre)}


def make_cfg(meta, hooks):
    from clt_training_amd import SparseCoderConfig, TrainConfig

    return TrainConfig(sae=SparseCoderConfig(**meta["cfg"]), batch_size=2, optimizer="adam", lr=meta["lr"],
                       lr_warmup_steps=meta["warmup"], b1=meta["b1"], hookpoints=list(hooks),
                       cross_layer=meta["cross_layer"], log_to_wandb=False, dead_feature_threshold=meta["dead_feature_threshold"])


def batches(t, meta, device="cpu"):
    out = []
    for s in range(meta["steps"]):
        b = {h: (t[f"x{s}.{l}"].to(device), t[f"y{s}.{l}"].to(device)) for l, h in enumerate(meta["hooks"])}
        b["loss_mask"] = t["mask"].to(device)
        out.append(b)
    return out


@pytest.mark.parametrize("name", CASES)
def test_constructor_reproduces_the_reference_parameters(name):
    """a18: same names, shapes, dtypes and VALUES as the reference's SparseCoder under the Trainer's seeding rule
    (trainer.py:117-146: torch.manual_seed(seed) before every coder), bit for bit."""
    from clt_training_amd import Trainer

    t, meta = load_golden(name)
    d = meta["d"]
    tr = Trainer(make_cfg(meta, meta["hooks"]), batches(t, meta), device="cpu", input_widths={h: d for h in meta["hooks"]})
    for l, h in enumerate(meta["hooks"]):
        sd = tr.saes[h].state_dict()
        want = {k[len(f"p0.{l}."):]: v for k, v in t.items() if k.startswith(f"p0.{l}.")}
        assert list(sd) == [k for k in sd if k in want] and set(sd) == set(want), (sorted(sd), sorted(want))
        for k, v in want.items():
            assert sd[k].dtype == v.dtype and torch.equal(sd[k], v), (l, k)


@pytest.mark.parametrize("name", CASES)
def test_oracle_vs_reference_trainer(name):
    t, meta = load_golden(name)
    L, cfg, steps = meta["L"], meta["cfg"], meta["steps"]
    transcode = cfg.get("transcode", False)
    sae_unit = not transcode and cfg.get("normalize_decoder", True)
    nio = cfg.get("normalize_io", False)
    layers = golden_layers(t, meta, L, prefix="p0")
    since = [torch.zeros(lp.W_enc.shape[0], dtype=torch.long) for lp in layers]
    state = [dict() for _ in range(L)]
    for s in range(steps):
        ys = [t[f"y{s}.{l}"] for l in range(L)]
        xs = [t[f"x{s}.{l}"] for l in range(L)] if transcode else ys  # trainer.py:524
        if s == 0:
            for l in range(L):
                O.step0_init(layers[l], xs[l], ys[l], nio)  # trainer.py:547-592
        if sae_unit:  # trainer.py:595-596
            for lp in layers:
                lp.W_decs[0] = O.unit_norm_decoder(lp.W_decs[0].double()).float()
        if s == 0:
            for l, lp in enumerate(layers):
                want = snapshot(t, "init", l)
                for n, p in oracle_tensors(lp).items():
                    close(p, want[n], f"init {l}.{n}")
                if nio:
                    torch.testing.assert_close(lp.in_norm, want["in_norm"], **FP32)
                    torch.testing.assert_close(lp.out_norm, want["out_norm"], **FP32)
        r = O.clt_train_step(layers, xs, ys, t["mask"], cfg["k"], normalize_io=nio,
                             divide_cross_layer=cfg.get("divide_cross_layer", False))
        if sae_unit:  # trainer.py:793-795
            for l, lp in enumerate(layers):
                r.grads[l]["W_decs.0"] = O.remove_parallel_grad(lp.W_decs[0].double(), r.grads[l]["W_decs.0"])
        lr = meta["lr"] * O.warmup_linear_factor(s, meta["warmup"], meta["num_batches"])
        torch.testing.assert_close(t[f"lr{s}"], torch.full((L,), lr), rtol=1e-6, atol=0)  # LambdaLR order, :797-802
        for l, lp in enumerate(layers):
            gwant = snapshot(t, f"grad{s}", l)
            for n, p in oracle_tensors(lp).items():
                close(r.grads[l][n].float(), gwant[n], f"grad step {s} {l}.{n}")
                mm, vv = state[l].get(n, (torch.zeros_like(p).double(), torch.zeros_like(p).double()))
                pn, mm, vv = O.adam_step(p, r.grads[l][n], mm, vv, s + 1, lr, b1=meta["b1"])
                p.copy_(pn.float())
                state[l][n] = (mm, vv)
            want = snapshot(t, f"after{s}", l)
            for n, p in oracle_tensors(lp).items():
                # Adam's g / (sqrt(v) + eps) turns fp32-level noise on a near-zero gradient element into a visible
                # fraction of lr on that element: absolute floor 1e-3 * lr
                close(p, want[n], f"after step {s} {l}.{n}", floor=1e-3 * meta["lr"])
            # trainer.py:622-628, 806-816
            since[l] += meta["tokens_per_step"]
            since[l][r.top_idx[l].flatten()] = 0
    for l in range(L):
        assert torch.equal(since[l], t[f"since_fired.{l}"]), l


def test_load_a_checkpoint_the_reference_wrote(tmp_path):
    """f1: ``sae.safetensors`` + ``cfg.json`` written by the reference's save_to_disk (sparse_coder.py:593-612) and
    ``optimizer_0.pt`` / ``lr_scheduler_0.pt`` / ``state.pt`` / ``rank_0_state.pt`` written by its Trainer
    (trainer.py:910-964: flattened tuple-keyed optimizer state) load here, and what save() writes back has the same keys."""
    from clt_training_amd import SparseCoder, Trainer, ckpt

    name = "trainer_sae_b1_09"
    t, meta = load_golden(name)
    path = GOLDEN / f"{name}_ckpt"
    for l, h in enumerate(meta["hooks"]):
        sae = SparseCoder.load_from_disk(path / h)
        for k, v in sae.state_dict().items():
            assert torch.equal(v, t[f"after2.{l}.{k}"]), (h, k)
    tr = Trainer(make_cfg(meta, meta["hooks"]), batches(t, meta), device="cpu",
                 input_widths={h: meta["d"] for h in meta["hooks"]})
    tr.load_state(path)
    assert tr.global_step == 3 and tr.optimizers[0]._step == 3
    assert tr.lr_schedulers[0].last_epoch == 3
    ref_flat = torch.load(path / "optimizer_0.pt", weights_only=True)
    ref = ckpt.unflatten_state(ref_flat)
    pos = 0
    for l, h in enumerate(meta["hooks"]):
        assert torch.equal(tr.num_tokens_since_fired[h], t[f"since_fired.{l}"])
        for n, p in tr.saes[h].named_parameters():
            st = tr.optimizers[0].state[p]
            assert torch.equal(st["exp_avg"], ref["state"][pos]["exp_avg"]), (h, n)
            assert torch.equal(st["exp_avg_sq"], ref["state"][pos]["exp_avg_sq"]), (h, n)
            assert torch.equal(p.detach(), t[f"after2.{l}.{n}"])
            pos += 1
    # and back: the file save() writes has the reference's keys (so the reference's unflatten_dict +
    # torch.optim.Adam.load_state_dict accept it)
    tr.save(tmp_path / "again")
    mine = torch.load(tmp_path / "again" / "optimizer_0.pt", weights_only=True)
    assert set(ref_flat) <= set(mine) and set(mine) - set(ref_flat) == {("clt_step",)}
    for k, v in ref_flat.items():
        if torch.is_tensor(v):
            assert torch.equal(mine[k].float().cpu(), v.float()), k
        elif k[-1] != "initial_lr":
            assert mine[k] == v, (k, mine[k], v)


def test_flatten_layout_round_trip():
    from clt_training_amd import ckpt

    nested = {"state": {0: {"step": torch.tensor(2.0), "exp_avg": torch.ones(2)}, 1: {"exp_avg": torch.zeros(1)}},
              "param_groups": [{"lr": 0.1, "betas": (0.9, 0.99), "params": [0, 1]}, {"lr": 0.2, "params": []}]}
    flat = ckpt.flatten_state(nested)
    assert flat[("param_groups", ("list", 0), "params", ("list", 1))] == 1  # utils.py:193-205
    assert flat[("param_groups", ("list", 0), "betas")] == (0.9, 0.99)
    assert ("param_groups", ("list", 1), "params") not in flat  # an empty list has no leaves (as in the reference)
    back = ckpt.unflatten_state(flat)
    assert back["param_groups"][0]["params"] == [0, 1] and back["state"][1]["exp_avg"].shape == (1,)


# ----------------------------------------------------------------------------------------------- GPU
def _gpu_check(tr, t, meta, coder_tensors):
    """Step the trainer over the recorded activations; compare with the reference after every optimizer step."""
    L = meta["L"]
    for s, b in enumerate(batches(t, meta, "cuda")):
        tr.step(b)
        for l, h in enumerate(meta["hooks"]):
            want = snapshot(t, f"after{s}", l)
            for n, p in coder_tensors(tr.saes[h]).items():
                close(p.detach().float().cpu(), want[n], f"after step {s} {l}.{n}", floor=1e-3 * meta["lr"])
    for l, h in enumerate(meta["hooks"]):
        assert torch.equal(tr.num_tokens_since_fired[h].cpu(), t[f"since_fired.{l}"]), l


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_trainer_vs_reference_trainer(name):
    """The autograd mirror (SparseCoder / CrossLayerRunner / Trainer, fused Adam inside the gradient kernels) on the GPU
    against the reference Trainer's own run: step-0 init, three optimizer steps, fired-latent counters."""
    import clt_training_amd.sparse_coder as sc
    from clt_training_amd import Trainer

    t, meta = load_golden(name)
    old, sc.AUTOCAST_ENABLED = sc.AUTOCAST_ENABLED, False
    try:
        tr = Trainer(make_cfg(meta, meta["hooks"]), batches(t, meta, "cuda"), device="cuda",
                     input_widths={h: meta["d"] for h in meta["hooks"]})
        for l, h in enumerate(meta["hooks"]):  # device RNG differs from the CPU's: start from the reference's parameters
            tr.saes[h].load_state_dict({k[len(f"p0.{l}."):]: v for k, v in t.items() if k.startswith(f"p0.{l}.")})

        def tensors(sae):
            return {golden_grad_name(n): p for n, p in sae.named_parameters()}

        _gpu_check(tr, t, meta, tensors)
    finally:
        sc.AUTOCAST_ENABLED = old


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_engine_vs_reference_trainer(name):
    """The explicit engine (tp.TPTrainer at world 1: the path bench.py times) against the reference Trainer's own run."""
    from clt_training_amd import SparseCoder
    from clt_training_amd.engine import ShardedCoder
    from clt_training_amd.tp import TPTrainer
    from dataclasses import replace

    t, meta = load_golden(name)
    cfg = make_cfg(meta, meta["hooks"])
    L, d = meta["L"], meta["d"]
    coders = []
    for l, h in enumerate(meta["hooks"]):
        n_targets = min(meta["cross_layer"], L - l) if meta["cross_layer"] > 0 else 0
        full = SparseCoder(d, replace(cfg.sae, n_targets=n_targets), "cpu", dtype=torch.float32)
        full.load_state_dict({k[len(f"p0.{l}."):]: v for k, v in t.items() if k.startswith(f"p0.{l}.")})
        coders.append(ShardedCoder.from_full(full, 0, 1, "cuda"))
    tr = TPTrainer(cfg, batches(t, meta, "cuda"), device="cuda", input_widths={h: d for h in meta["hooks"]}, coders=coders)

    def tensors(c):
        return {n.replace("W_enc", "W_enc").replace("b_enc", "b_enc"): p for n, p in c.named_parameters()}

    _gpu_check(tr, t, meta, tensors)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["autograd", "engine"])
@pytest.mark.parametrize("name", CASES)
def test_gpu_model_driven_fit_vs_oracle_and_reference(name, which, tmp_path):
    """the SAME random-init GPT-2 and token ids the reference run used (rebuilt from the
    recorded seed) drive ``fit`` on the GPU through the forward hooks.  (i) the activations the hooks hand to the step
    equal the ones the reference's hooks saw (GPU vs CPU forward: rtol 1e-4); (ii) the trained parameters and the
    fired-latent counters equal the ORACLE's run over exactly those recorded activations (fp32 tolerance, Adam floor);
    (iii) they also track the reference Trainer's own final parameters (looser: the activations differ in the last bits
    and Adam's first steps amplify that on near-zero gradient elements)."""
    transformers = pytest.importorskip("transformers")
    import clt_training_amd.sparse_coder as sc
    from clt_training_amd import Trainer
    from clt_training_amd.tp import TPTrainer
    from dataclasses import replace

    t, meta = load_golden(name)
    L, d, steps, seed, cfgd = meta["L"], meta["d"], meta["steps"], meta["seed"], meta["cfg"]
    T, B = 16, 2
    assert meta["tokens_per_step"] == B * T
    torch.manual_seed(seed)  # as make_golden.golden_trainer
    gcfg = transformers.GPT2Config(n_layer=L, n_embd=d, n_head=2, vocab_size=96, n_positions=T, bos_token_id=0,
                                   eos_token_id=0)
    model = transformers.GPT2LMHeadModel(gcfg).eval().to("cuda")
    g = torch.Generator().manual_seed(seed + 1)
    ids = torch.randint(1, 96, (B * steps, T), generator=g)
    rows = [{"input_ids": r} for r in ids]
    cfg = replace(make_cfg(meta, ["h.*.mlp"]), remove_first_token=True, filter_bos=True, batch_size=B,
                  save_dir=str(tmp_path), run_name="run", save_every=10**9)
    old, sc.AUTOCAST_ENABLED = sc.AUTOCAST_ENABLED, False
    try:
        tr = (Trainer if which == "autograd" else TPTrainer)(cfg, rows, model)
        assert list(tr.saes) == meta["hooks"]
        with torch.no_grad():  # device RNG differs from the CPU's: start from the reference's parameters
            for l, h in enumerate(meta["hooks"]):
                own = {golden_grad_name(n): p for n, p in tr.saes[h].named_parameters()}
                for n, v in snapshot(t, "p0", l).items():  # (reference names -> oracle names: encoder.weight -> W_enc)
                    if n in own:  # (buffers such as in_norm / out_norm are set by the step-0 init)
                        own[n].copy_(v)
        seen = []
        step = tr.step

        def recording_step(batch):
            seen.append({k: (tuple(x.detach().clone() for x in v) if isinstance(v, tuple) else v.detach().clone())
                         for k, v in batch.items()})
            return step(batch)

        tr.step = recording_step
        tr.fit()
    finally:
        sc.AUTOCAST_ENABLED = old
    assert tr.global_step == steps and len(seen) == steps
    transcode = cfgd.get("transcode", False)
    # (i) what the hooks recorded
    for s, b in enumerate(seen):
        assert torch.equal(b["loss_mask"].cpu().flatten(), t["mask"])
        for l, h in enumerate(meta["hooks"]):
            x, y = b[h]
            torch.testing.assert_close(y.float().cpu(), t[f"y{s}.{l}"], rtol=1e-4, atol=1e-5)
            if transcode:
                torch.testing.assert_close(x.float().cpu(), t[f"x{s}.{l}"], rtol=1e-4, atol=1e-5)
    # (ii) the oracle over exactly those activations
    nio = cfgd.get("normalize_io", False)
    sae_unit = not transcode and cfgd.get("normalize_decoder", True)
    layers = golden_layers(t, meta, L, prefix="p0")
    since = [torch.zeros(lp.W_enc.shape[0], dtype=torch.long) for lp in layers]
    state = [dict() for _ in range(L)]
    for s, b in enumerate(seen):
        ys = [b[h][1].float().cpu() for h in meta["hooks"]]
        xs = [b[h][0].float().cpu() for h in meta["hooks"]] if transcode else ys
        if s == 0:
            for l in range(L):
                O.step0_init(layers[l], xs[l], ys[l], nio)
        if sae_unit:
            for lp in layers:
                lp.W_decs[0] = O.unit_norm_decoder(lp.W_decs[0].double()).float()
        r = O.clt_train_step(layers, xs, ys, t["mask"], cfgd["k"], normalize_io=nio,
                             divide_cross_layer=cfgd.get("divide_cross_layer", False))
        if sae_unit:
            for l, lp in enumerate(layers):
                r.grads[l]["W_decs.0"] = O.remove_parallel_grad(lp.W_decs[0].double(), r.grads[l]["W_decs.0"])
        lr = meta["lr"] * O.warmup_linear_factor(s, meta["warmup"], meta["num_batches"])
        for l, lp in enumerate(layers):
            for n, p in oracle_tensors(lp).items():
                mm, vv = state[l].get(n, (torch.zeros_like(p).double(), torch.zeros_like(p).double()))
                pn, mm, vv = O.adam_step(p, r.grads[l][n], mm, vv, s + 1, lr, b1=meta["b1"])
                p.copy_(pn.float())
                state[l][n] = (mm, vv)
            since[l] += meta["tokens_per_step"]
            since[l][r.top_idx[l].flatten()] = 0
    for l, h in enumerate(meta["hooks"]):
        got = {golden_grad_name(n): p.detach().float().cpu() for n, p in tr.saes[h].named_parameters()}
        for n, p in oracle_tensors(layers[l]).items():
            close(got[n], p, f"oracle, final {l}.{n}", floor=1e-3 * meta["lr"])
        assert torch.equal(tr.num_tokens_since_fired[h].cpu(), since[l]), l
        # (iii) and the reference Trainer's own end state
        want = snapshot(t, f"after{steps - 1}", l)
        for n, p in got.items():
            close(p, want[n], f"reference, final {l}.{n}", rtol=1e-3, floor=0.25 * meta["lr"])
