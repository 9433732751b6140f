"""Checkpoint format: ``sae.safetensors`` + ``cfg.json`` per hookpoint with the reference's
parameter names, the imputation rules of sparse_coder.py:518-592, and the gather of latent shards into the unsharded file."""
import json

import torch

from clt_training_amd import SparseCoder, SparseCoderConfig
from clt_training_amd.engine import ShardedCoder


def _randomise(sae, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in sae.parameters():
            p.copy_(torch.randn(p.shape, generator=g))
    return sae


def test_round_trip_keeps_reference_names(tmp_path):
    cfg = SparseCoderConfig(transcode=True, expansion_factor=4, k=4, skip_connection=True, post_encoder_scale=True,
                            normalize_io=True, n_targets=3, n_sources=1)
    sae = _randomise(SparseCoder(16, cfg, "cpu", dtype=torch.float32), 0)
    sae.save_to_disk(tmp_path / "a")
    with open(tmp_path / "a" / "cfg.json") as f:
        meta = json.load(f)
    assert meta["d_in"] == 16 and meta["k"] == 4 and meta["n_targets"] == 3
    back = SparseCoder.load_from_disk(tmp_path / "a")
    want = {"encoder.weight", "encoder.bias", "W_dec", "b_dec", "W_skip", "in_norm", "out_norm"}
    want |= {f"{s}.{i}" for s in ("W_decs", "b_decs", "W_skips", "post_encs", "post_enc_scales") for i in range(3)}
    assert set(back.state_dict()) == want
    for (n, a), (m, b) in zip(sae.state_dict().items(), back.state_dict().items()):
        assert n == m and torch.equal(a, b), n


def test_single_target_checkpoint_is_imputed_into_per_target_module(tmp_path):
    one = _randomise(SparseCoder(16, SparseCoderConfig(transcode=True, expansion_factor=4, k=4, skip_connection=True), "cpu",
                                 dtype=torch.float32), 1)
    one.save_to_disk(tmp_path / "one")
    cfg3 = SparseCoderConfig(transcode=True, expansion_factor=4, k=4, skip_connection=True, post_encoder_scale=True,
                             n_targets=3, n_sources=1)
    three = SparseCoder(16, cfg3, "cpu", dtype=torch.float32)
    three.load_state(tmp_path / "one")
    assert torch.equal(three.W_decs[0], one.W_dec) and three.W_dec is three.W_decs[0]
    assert torch.equal(three.b_decs[0], one.b_dec) and torch.equal(three.W_skips[0], one.W_skip)
    assert float(three.W_decs[1].detach().abs().sum()) == 0 and float(three.b_decs[2].detach().abs().sum()) == 0
    # post-encoder vectors the checkpoint lacks keep their initial values (ones for slot 0 of the scale, zeros after)
    assert bool((three.post_enc_scales[0] == 1).all()) and bool((three.post_enc_scales[1] == 0).all())
    assert torch.equal(three.encoder.weight, one.encoder.weight)


def test_sharded_coder_gathers_into_the_unsharded_file(tmp_path):
    cfg = SparseCoderConfig(transcode=True, expansion_factor=4, k=4, skip_connection=True, post_encoder_scale=True,
                            n_targets=2, n_sources=1)
    sae = _randomise(SparseCoder(16, cfg, "cpu", dtype=torch.float32), 2)
    ShardedCoder.from_full(sae, 0, 1, "cpu").save_to_disk(tmp_path / "s")
    back = SparseCoder.load_from_disk(tmp_path / "s")
    for (n, a), (m, b) in zip(sae.state_dict().items(), back.state_dict().items()):
        assert n == m and torch.equal(a, b), n
    half = ShardedCoder.load_from_disk(tmp_path / "s", "cpu", rank=1, tp=2)
    assert torch.equal(half.W_enc, sae.encoder.weight[32:]) and torch.equal(half.W_decs[1], sae.W_decs[1][32:])
    assert torch.equal(half.b_decs[1], sae.b_decs[1]) and half.lat_begin == 32


def test_trainer_does_not_fuse_adam_into_a_tied_decoder():
    """per_source_tied (sparse_coder.py:324-338): one W_dec serves every target of a source, and the hook-by-hook local
    backward (trainer.py:674-682) would update it between two of its forward uses if the Adam step were applied inside
    the gradient kernel.  The trainer must leave such runs to .grad -> Adam.step."""
    import torch

    from clt_training_amd import SparseCoderConfig, TrainConfig, Trainer

    hooks = ["h.0", "h.1"]
    data = [{h: (torch.zeros(4, 16), torch.zeros(4, 16)) for h in hooks}]
    for tied in (False, True):
        cfg = TrainConfig(sae=SparseCoderConfig(transcode=True, expansion_factor=2, k=2, per_source_tied=tied),
                          hookpoints=list(hooks), cross_layer=2, optimizer="adam", log_to_wandb=False)
        tr = Trainer(cfg, data, device="cpu")
        assert tr._fused == (not tied)
        assert hasattr(tr.saes["h.0"], "W_decs") == (not tied)


def test_load_many_local(tmp_path):
    """sparse_coder.py:448-479: every hookpoint directory of a run, in natural order; ``layers`` selects."""
    import torch

    from clt_training_amd import SparseCoder, SparseCoderConfig

    names = ["h.10.mlp", "h.2.mlp", "h.9.mlp"]
    for i, n in enumerate(names):
        torch.manual_seed(i)
        SparseCoder(8, SparseCoderConfig(transcode=True, expansion_factor=2, k=2), "cpu", dtype=torch.float32).save_to_disk(tmp_path / n)
    (tmp_path / "config.json").write_text("{}")  # a file beside the directories is not a hookpoint
    got = SparseCoder.load_many(str(tmp_path), local=True)
    assert list(got) == ["h.2.mlp", "h.9.mlp", "h.10.mlp"]
    torch.manual_seed(1)
    ref = SparseCoder(8, SparseCoderConfig(transcode=True, expansion_factor=2, k=2), "cpu", dtype=torch.float32)
    assert torch.equal(got["h.2.mlp"].encoder.weight, ref.encoder.weight)
    assert list(SparseCoder.load_many(str(tmp_path), local=True, layers=["h.9.mlp", "h.10.mlp"])) == ["h.9.mlp", "h.10.mlp"]


def test_empty_optimizer_state_round_trips_through_the_file_layout():
    """SignSGD keeps no state and Adam has none before its first step: the flattened file then holds no ``state`` leaf
    and the loader must hand ``Optimizer.load_state_dict`` an empty one."""
    from clt_training_amd import ckpt

    p = torch.nn.Parameter(torch.zeros(4))
    for opt in (torch.optim.SGD([p], lr=0.5), torch.optim.Adam([p], lr=0.5)):
        flat = ckpt.to_reference_layout(opt.state_dict(), 0)
        assert not any(k[0] == "state" for k in flat)
        nested, step = ckpt.from_file_layout(flat)
        assert nested["state"] == {} and step == 0
        opt.load_state_dict(nested)  # raised KeyError('state') before
        assert opt.param_groups[0]["lr"] == 0.5


def test_fit_logs_the_end_to_end_keys_and_ranks_best_by_the_trained_loss(tmp_path):
    """trainer.py:829-851, 972-989: ``ce_loss`` / ``kl_loss`` + ``acc_top1`` are logged, ``fvu/...`` only when "fvu" is in
    the loss name, and save_best compares ONE scalar for the end-to-end losses."""
    from types import SimpleNamespace

    from clt_training_amd.fit_loop import FitLoop

    class Stub(FitLoop):
        def __init__(self, loss_fn, losses):
            self.cfg = SimpleNamespace(loss_fn=loss_fn, grad_acc_steps=1, micro_acc_steps=1, wandb_log_frequency=1,
                                       save_every=2, save_best=True, log_to_wandb=False, restart_epoch=False,
                                       save_dir=str(tmp_path), run_name="r", hookpoints=["h"], dead_feature_threshold=10,
                                       k_decay_steps=0, k_anneal_mul=1, sae=SimpleNamespace(k=4), per_layer_k=[])
            self.saes = {"h": SimpleNamespace(num_latents=8)}
            self.num_tokens_since_fired = {"h": torch.zeros(8)}
            self.dataset, self.losses, self.global_step = list(range(len(losses))), losses, 0
            self.saved, self.logged = [], []
            self.init_k_schedule()
            self.wandb_log = lambda info, step: self.logged.append((step, info))

        def step(self, batch):
            v = torch.tensor(self.losses[batch])
            self.last_e2e = {"kl": v, "acc_top1": torch.tensor(0.5)} if "kl" in self.cfg.loss_fn else {"ce": v}
            self.global_step += 1
            return {"h": torch.tensor(0.25)}

        def save(self, path=None, only=None):
            self.saved.append((path, only))

    t = Stub("kl", [3.0, 2.0, 5.0, 4.0]).fit()
    assert [set(i) for _, i in t.logged] == [{"kl_loss", "acc_top1", "dead_pct/h", "k"}] * 4
    assert [i["kl_loss"] for _, i in t.logged] == [3.0, 2.0, 5.0, 4.0]
    best = [p for p, _ in t.saved if p is not None]
    # (step + 1) % save_every with step already advanced: the reference saves after batches 1 and 3 (3.0, then 5.0)
    assert best == [f"{tmp_path}/r/best"] and t.best_loss == 3.0
    t = Stub("kl-fvu", [1.0]).fit()
    assert set(t.logged[0][1]) == {"kl_loss", "acc_top1", "dead_pct/h", "fvu/h", "k"}
    t = Stub("ce", [1.0, 0.5]).fit()
    assert set(t.logged[0][1]) == {"ce_loss", "dead_pct/h", "k"} and t.best_loss == 0.5
    t = Stub("fvu", [1.0, 0.5]).fit()
    assert set(t.logged[0][1]) == {"dead_pct/h", "fvu/h", "k"} and t.best_loss == {"h": 0.25}
    assert (f"{tmp_path}/r/best/h", ["h"]) in t.saved
