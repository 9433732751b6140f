"""world_size-2 gloo test (CPU) of the latent-sharded engine: sharding, candidate merge, partial decode +
reduce-scatter, token-sharded FVU, gradient ownership and the replicated-gradient all-reduce must reproduce
the single-process oracle step exactly (fp32).  The arithmetic backend is the oracle-backed ops double."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_problem(sae_kw, L, cross, d, N, seed=0):
    from clt_training_amd import SparseCoder, SparseCoderConfig

    g = torch.Generator().manual_seed(seed)
    saes = []
    for pos in range(L):
        torch.manual_seed(pos)
        n_targets = min(cross, L - pos) if cross > 0 else 0
        cfg = SparseCoderConfig(**sae_kw, n_targets=n_targets)
        sae = SparseCoder(d, cfg, "cpu", dtype=torch.float32)
        with torch.no_grad():
            for n, p in sae.named_parameters():
                if n.startswith("W_dec") and cfg.transcode:
                    p.copy_(torch.randn(p.shape, generator=g) / d**0.5)
                elif "post_enc_scale" in n:
                    p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
                elif "post_enc" in n:
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
                elif n.startswith("W_skip"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g) / d**0.5)
        saes.append(sae)
    steps = []
    for s in range(3):
        xs = [torch.randn(N, d, generator=g) + 0.2 * torch.randn(d, generator=g) for _ in range(L)]
        ys = [0.5 * x + 0.5 * torch.randn(N, d, generator=g) for x in xs]
        m = torch.ones(N, dtype=torch.bool)
        m[::7] = False
        steps.append((xs, ys, m))
    return saes, steps


def _oracle_layers(saes):
    from oracle import clt_oracle as O

    layers = []
    for sae in saes:
        T = max(1, sae.cfg.n_targets if sae.cfg.transcode else 1)
        Td = 1 if sae.cfg.coalesce_topk == "concat" else T  # concat: one decoder / bias / skip per layer
        sd = {n: p.detach().clone() for n, p in sae.state_dict().items()}

        def lst(stem, plural):
            n = T if stem.startswith("post_enc") else Td
            return [sd[stem if (i == 0 and stem in sd) else f"{plural}.{i}"] for i in range(n)]

        scales = lst("post_enc_scale", "post_enc_scales") if sae.cfg.post_encoder_scale else None
        skips = lst("W_skip", "W_skips") if sae.cfg.skip_connection else [None] * Td
        layers.append(O.LayerParams(sd["encoder.weight"], sd["encoder.bias"], lst("W_dec", "W_decs"), lst("b_dec", "b_decs"),
                                    skips, lst("post_enc", "post_encs"), scales, sae.cfg.n_targets,
                                    sd.get("in_norm"), sd.get("out_norm"), sae.cfg.transcode))
    return layers


def _worker(rank, world, port, sae_kw, L, cross, d, N, out_dir, device="cpu", resume=False, train_kw=None):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from clt_training_amd import TrainConfig
    from clt_training_amd.engine import ShardedCoder
    from clt_training_amd.tp import TPTrainer
    if device == "cpu":
        from oracle import ops_cpu as ops  # oracle-backed double of hip_ops: test infrastructure
        adam_cls = ops.Adam
    else:  # tests/test_gpu_tp2.py: the HIP kernels, both ranks on the one GPU, collectives staged through gloo
        from clt_training_amd import hip_ops as ops
        adam_cls = None

    saes, steps = _make_problem(sae_kw, L, cross, d, N)
    hooks = [f"h.{i}" for i in range(L)]
    from clt_training_amd import SparseCoderConfig
    cfg = TrainConfig(**{**dict(sae=SparseCoderConfig(**sae_kw), hookpoints=hooks, cross_layer=cross, optimizer="adam",
                              lr=1e-2, lr_warmup_steps=2, finetune="skip-step0-init"), **(train_kw or {})})
    coders = [ShardedCoder.from_full(s, rank, world, device) for s in saes]
    tr = TPTrainer(cfg, [None] * 4, device=device, input_widths={h: d for h in hooks}, ops=ops,
                   optimizer_cls=adam_cls, coders=coders)
    Nr = N // world
    fv = []
    for s_i, (xs, ys, m) in enumerate(steps):
        if resume and s_i == 2:
            # checkpoint after two steps (shards gathered into the reference's unsharded sae.safetensors), then carry on
            # in a FRESH trainer that starts from the untrained parameters and loads the checkpoint
            tr.save(f"{out_dir}/ckpt")
            coders = [ShardedCoder.from_full(s, rank, world, device) for s in _make_problem(sae_kw, L, cross, d, N)[0]]
            tr = TPTrainer(cfg, [None] * 4, device=device, input_widths={h: d for h in hooks}, ops=ops,
                           optimizer_cls=adam_cls, coders=coders)
            tr.load_state(f"{out_dir}/ckpt")
            assert tr.global_step == 2
        batch = {h: (xs[l][rank * Nr:(rank + 1) * Nr].to(device), ys[l][rank * Nr:(rank + 1) * Nr].to(device))
                 for l, h in enumerate(hooks)}
        batch["loss_mask"] = m[rank * Nr:(rank + 1) * Nr].to(device)
        f = tr.step(batch)
        fv.append([float(f[h]) for h in hooks])
        # fired-latent bookkeeping (trainer.py:622-628, 806-816): this rank's latents that fired were reset to 0,
        # the others aged by the step's global token count
        for l, h in enumerate(hooks):
            loc = tr.engine.last_idx[l].flatten() - coders[l].lat_begin
            loc = loc[(loc >= 0) & (loc < coders[l].m_local)]
            since = tr.num_tokens_since_fired[h]
            assert bool((since[loc] == 0).all()) and not bool(tr.did_fire[h].any())
            quiet = torch.ones_like(since, dtype=torch.bool)
            quiet[loc] = False
            assert bool((since[quiet] >= N).all())
    state = {f"{l}.{n}": p.detach().cpu().clone() for l, c in enumerate(coders) for n, p in c.named_parameters()}
    torch.save(dict(fvu=fv, state=state), f"{out_dir}/rank{rank}.pt")
    dist.destroy_process_group()


CASES = {
    "clt_skip_scale_norm_div": (dict(transcode=True, expansion_factor=4, k=4, skip_connection=True, post_encoder_scale=True,
                                     normalize_io=True, divide_cross_layer=True), 3, 2),
    "sae": (dict(transcode=False, expansion_factor=4, k=4, normalize_decoder=False), 1, 0),
    # fused_encoder.py:373-413: each shard owns k / tp whole groups; the exchange is a concatenation
    "clt_groupmax": (dict(transcode=True, expansion_factor=4, k=4, skip_connection=True, activation="groupmax"), 3, 3),
    # fused_encoder.py:204-240, 315-318: row top-4k, merge, batch-wide threshold on the merged rows
    "clt_batchtopk": (dict(transcode=True, expansion_factor=4, k=2, skip_connection=True, activation="batchtopk"), 3, 2),
    # BASELINE configs[4]: per-target tied decoder, one bag of k * n_live candidates per token (runner.py:93-127)
    "clt_concat_scale_norm": (dict(transcode=True, expansion_factor=4, k=4, skip_connection=True, post_encoder_scale=True,
                                   normalize_io=True, coalesce_topk="concat"), 3, 3),
    # optimizer="signum" with force_lr_warmup: SignSGD (sign_sgd.py:19-23) + the linear warm-up (trainer.py:224-250)
    "clt_signum": (dict(transcode=True, expansion_factor=4, k=4, skip_connection=True), 3, 2,
                   dict(optimizer="signum", force_lr_warmup=True, lr=1e-3)),
}


def run_case(case, tmp_path, device="cpu", resume=False):
    from oracle import clt_oracle as O

    sae_kw, L, cross, *rest = CASES[case]
    train_kw = rest[0] if rest else {}
    d, N, world = 16, 24, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, sae_kw, L, cross, d, N, str(tmp_path), device, resume, train_kw), nprocs=world,
             join=True)
    res = [torch.load(f"{tmp_path}/rank{r}.pt", weights_only=True) for r in range(world)]
    # ---- single-process oracle: 3 steps of clt_train_step + Adam with the warm-up schedule
    saes, steps = _make_problem(sae_kw, L, cross, d, N)
    layers = _oracle_layers(saes)
    state = [dict() for _ in range(L)]
    want_fvu = []
    for s, (xs, ys, m) in enumerate(steps):
        if not sae_kw["transcode"]:
            xs = ys  # an SAE reconstructs its own input (trainer.py:524)
        r = O.clt_train_step(layers, xs, ys, m, sae_kw["k"], normalize_io=sae_kw.get("normalize_io", False),
                             divide_cross_layer=sae_kw.get("divide_cross_layer", False),
                             activation=sae_kw.get("activation", "topk"), coalesce=sae_kw.get("coalesce_topk", "none"))
        want_fvu.append([float(f) for f in r.fvus])
        lr = train_kw.get("lr", 1e-2) * O.warmup_linear_factor(s, 2, 4)
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
                if train_kw.get("optimizer") == "signum":
                    p.sub_(lr * torch.sign(r.grads[l][n].float()))  # sign_sgd.py:23
                    continue
                mm, vv = state[l].get(n, (torch.zeros_like(p).double(), torch.zeros_like(p).double()))
                pn, mm, vv = O.adam_step(p, r.grads[l][n], mm, vv, s + 1, lr)
                p.copy_(pn.float())
                state[l][n] = (mm, vv)
    for r in range(world):
        torch.testing.assert_close(torch.tensor(res[r]["fvu"]), torch.tensor(want_fvu), rtol=1e-4, atol=1e-6)
    Ml = (d * 4) // world
    for l, lp in enumerate(layers):
        for r in range(world):
            st = res[r]["state"]
            lo, hi = r * Ml, (r + 1) * Ml
            torch.testing.assert_close(st[f"{l}.W_enc"], lp.W_enc[lo:hi], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(st[f"{l}.b_enc"], lp.b_enc[lo:hi], rtol=1e-4, atol=1e-5)
            for i in range(len(lp.W_decs)):
                torch.testing.assert_close(st[f"{l}.W_decs.{i}"], lp.W_decs[i][lo:hi], rtol=1e-4, atol=1e-5)
                torch.testing.assert_close(st[f"{l}.b_decs.{i}"], lp.b_decs[i], rtol=1e-4, atol=1e-5)
                if lp.W_skips[i] is not None:
                    torch.testing.assert_close(st[f"{l}.W_skips.{i}"], lp.W_skips[i], rtol=1e-4, atol=1e-5)
            for i in range(len(lp.post_encs)):
                torch.testing.assert_close(st[f"{l}.post_encs.{i}"], lp.post_encs[i][lo:hi], rtol=1e-4, atol=1e-5)
                if lp.post_enc_scales is not None:
                    torch.testing.assert_close(st[f"{l}.post_enc_scales.{i}"], lp.post_enc_scales[i][lo:hi], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", list(CASES))
def test_tp2_matches_single_process_oracle(case, tmp_path):
    run_case(case, tmp_path)


def test_tp2_checkpoint_resume(tmp_path):
    """trainer.py:910-964 / 258-348 on the sharded trainer: save after two steps (latent shards gathered into the
    reference's unsharded ``sae.safetensors`` + ``cfg.json``; per-rank optimizer state), resume in a fresh trainer,
    and the third step still matches the uninterrupted single-process oracle."""
    run_case("clt_skip_scale_norm_div", tmp_path, resume=True)
    from safetensors.torch import load_file

    sd = load_file(f"{tmp_path}/ckpt/h.0/sae.safetensors")
    assert sd["encoder.weight"].shape == (64, 16) and "W_decs.1" in sd and "post_enc_scales.0" in sd
