"""Real-activation driver (sparsify/trainer.py:73-111, 472-545, 684-692, 804-861) on CPU:
hookpoint resolution, recorded activations against the model's own hidden states, the BOS / first-token mask,
``hookpoints_in``, and a model-driven ``fit`` of the sharded trainer (oracle-backed ops double) against the same
trainer fed the cached activations.  The model is a random-init GPT-2 built from a config (no downloads)."""
import json
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

transformers = pytest.importorskip("transformers")


def tiny_gpt2(n_layer=3, d=16, seed=0):
    torch.manual_seed(seed)
    cfg = transformers.GPT2Config(n_layer=n_layer, n_embd=d, n_head=2, vocab_size=50, n_positions=16, bos_token_id=1,
                                  eos_token_id=1, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    return transformers.GPT2LMHeadModel(cfg).eval()


def token_rows(n=8, T=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    rows = torch.randint(2, 50, (n, T), generator=g)
    rows[:, 0] = 1   # BOS in front
    rows[1, 4] = 1   # and one in the middle of a sequence
    return [{"input_ids": r} for r in rows]


def test_natural_order_and_hookpoint_resolution():
    from clt_training_amd import TrainConfig
    from clt_training_amd.activations import natural_key, resolve_hookpoints, resolve_widths

    assert sorted(["h.10.mlp", "h.9.mlp", "h.1.mlp"], key=natural_key) == ["h.1.mlp", "h.9.mlp", "h.10.mlp"]
    model = tiny_gpt2(n_layer=12)
    cfg = TrainConfig(hookpoints=["h.*.mlp"])
    assert resolve_hookpoints(cfg, model) == [f"h.{i}.mlp" for i in range(12)]       # trainer.py:76-83
    cfg = TrainConfig(layers=[0, 2])
    assert resolve_hookpoints(cfg, model) == ["h.0", "h.2"]                            # trainer.py:90-92
    cfg = TrainConfig()
    assert resolve_hookpoints(cfg, model) == [f"h.{i}" for i in range(12)]            # trainer.py:86-88
    assert resolve_widths(model, ["h.0.mlp", "h.3"]) == {"h.0.mlp": 16, "h.3": 16}    # utils.py:110-142


def test_recorded_activations_match_hidden_states():
    from clt_training_amd import SparseCoderConfig, TrainConfig
    from clt_training_amd.activations import ModelActivations

    model = tiny_gpt2()
    rows = token_rows()
    cfg = TrainConfig(sae=SparseCoderConfig(transcode=True), hookpoints=["h.0", "h.1", "h.2"], batch_size=4,
                      filter_bos=True, remove_first_token=False)
    acts = ModelActivations(model, rows, cfg)
    assert len(acts) == 2
    steps = list(acts)
    assert len(steps) == 2
    ids = torch.stack([r["input_ids"] for r in rows])
    for s, step in enumerate(steps):
        x = ids[4 * s:4 * s + 4]
        with torch.no_grad():
            hs = model(x, output_hidden_states=True).hidden_states
        for i in range(3):
            inp, out = step[f"h.{i}"]
            assert inp.shape == (32, 16)                                               # flatten(0, 1), trainer.py:522-524
            torch.testing.assert_close(inp, hs[i].flatten(0, 1))                       # a block's input: hidden state i
            if i < 2:                                                                  # (the last one has ln_f applied)
                torch.testing.assert_close(out, hs[i + 1].flatten(0, 1))
        assert torch.equal(step["loss_mask"], ~(x == 1).flatten())                     # trainer.py:686, 609
    # an autoencoder reconstructs the module OUTPUT (trainer.py:524)
    cfg_sae = TrainConfig(sae=SparseCoderConfig(transcode=False), hookpoints=["h.1"], batch_size=4)
    st = next(iter(ModelActivations(model, rows, cfg_sae)))
    assert torch.equal(st["h.1"][0], st["h.1"][1])
    assert bool(st["loss_mask"].all())                                                 # filter_bos off (trainer.py:687-688)
    cfg_first = TrainConfig(sae=SparseCoderConfig(transcode=True), hookpoints=["h.1"], batch_size=4, remove_first_token=True)
    m = next(iter(ModelActivations(model, rows, cfg_first)))["loss_mask"].view(4, 8)
    assert not bool(m[:, 0].any()) and bool(m[:, 1:].all())                            # trainer.py:689-690


def test_hookpoints_in_override_the_input():
    from clt_training_amd import SparseCoderConfig, TrainConfig
    from clt_training_amd.activations import ModelActivations

    model = tiny_gpt2()
    rows = token_rows()
    cfg = TrainConfig(sae=SparseCoderConfig(transcode=True), hookpoints=["h.0.mlp", "h.1.mlp"],
                      hookpoints_in=["h.0.ln_2"], batch_size=8)
    step = next(iter(ModelActivations(model, rows, cfg)))
    seen = {}
    hs = [model.base_model.get_submodule(n).register_forward_hook(
        lambda mod, i, o, n=n: seen.__setitem__(n, (i[0].flatten(0, 1), o.flatten(0, 1)))) for n in ("h.0.ln_2", "h.0.mlp", "h.1.mlp")]
    with torch.no_grad():
        model(torch.stack([r["input_ids"] for r in rows]))
    for h in hs:
        h.remove()
    torch.testing.assert_close(step["h.0.mlp"][0], seen["h.0.ln_2"][0])  # the residual stream INTO ln_2 (:482-485, 509-511)
    torch.testing.assert_close(step["h.0.mlp"][1], seen["h.0.mlp"][1])
    torch.testing.assert_close(step["h.1.mlp"][0], seen["h.1.mlp"][0])  # no override for the second hookpoint
    torch.testing.assert_close(step["h.1.mlp"][1], seen["h.1.mlp"][1])


def test_model_driven_fit_equals_cached_activations(tmp_path):
    from clt_training_amd import SparseCoderConfig, TrainConfig
    from clt_training_amd.activations import ModelActivations
    from clt_training_amd.tp import TPTrainer
    from oracle import ops_cpu as ops  # oracle-backed double of hip_ops: test infrastructure

    model = tiny_gpt2()
    rows = token_rows(n=12)

    def make_cfg(hooks, **kw):
        return TrainConfig(sae=SparseCoderConfig(transcode=True, expansion_factor=4, k=4, skip_connection=True),
                           hookpoints=hooks, cross_layer=3, optimizer="adam", lr=1e-2, lr_warmup_steps=2, batch_size=4,
                           remove_first_token=True, log_to_wandb=False, save_dir=str(tmp_path), save_every=2, **kw)

    # ---- model-driven: wildcard hookpoints, widths from the model, batches from forward hooks
    cfg = make_cfg(["h.*.mlp"], run_name="driven", k_decay_steps=4, k_anneal_mul=2)
    tr = TPTrainer(cfg, rows, model, device="cpu", ops=ops, optimizer_cls=ops.Adam)
    assert cfg.hookpoints == ["h.0.mlp", "h.1.mlp", "h.2.mlp"]
    assert [s.cfg.n_targets for s in tr.saes.values()] == [3, 2, 1]
    logged, seen_k = [], []
    tr.wandb_log = lambda info, step: logged.append((step, info))
    tr.fit(log=lambda step, fv: seen_k.append(next(iter(tr.saes.values())).cfg.k))
    assert tr.global_step == 3
    # k schedule (trainer.py:356-374, 804): step 0 runs at cfg.sae.k, then the anneal from k * mul is read at the
    # pre-increment step count: after steps 0, 1, 2 -> progress 0, 1/4, 2/4 of 8 -> 4
    assert seen_k == [8, 7, 6]
    assert [s for s, _ in logged] == [1, 2, 3]                                           # divmod(global_step + 1, ...), :791
    assert set(logged[0][1]) == {"k"} | {f"{p}/h.{i}.mlp" for p in ("fvu", "dead_pct") for i in range(3)}  # :829-851
    for f in ("h.0.mlp/sae.safetensors", "h.0.mlp/cfg.json", "optimizer_0.pt", "lr_scheduler_0.pt", "state.pt",
              "rank_0_state.pt", "config.json"):
        assert (tmp_path / "driven" / f).exists(), f
    assert json.loads((tmp_path / "driven" / "config.json").read_text())["hookpoints"] == cfg.hookpoints

    # ---- the same steps from cached activations
    cfg2 = make_cfg(["h.0.mlp", "h.1.mlp", "h.2.mlp"], run_name="cached", k_decay_steps=4, k_anneal_mul=2)
    cached = list(ModelActivations(model, rows, cfg2))
    tr2 = TPTrainer(cfg2, cached, device="cpu", ops=ops, optimizer_cls=ops.Adam, num_batches=3)
    fv2 = [dict(tr2.step(b)) for b in cached]
    for (n, p), (n2, q) in zip(((n, p) for s in tr.saes.values() for n, p in s.named_parameters()),
                               ((n, p) for s in tr2.saes.values() for n, p in s.named_parameters())):
        assert n == n2 and torch.equal(p, q), n
    assert abs(logged[-1][1]["fvu/h.2.mlp"] - float(fv2[-1]["h.2.mlp"])) < 1e-6
