"""optimizer="muon" (sparsify/muon.py:12-203, trainer.py:194-222) against what the
REFERENCE's own Muon and Trainer produced (tests/golden/make_golden.py muon: muon_steps, trainer_clt_muon).  The
Newton-Schulz iteration runs in bf16 in the reference too, so a different GEMM accumulation order moves the result by
bf16 roundings: tolerances are relative Frobenius errors of the UPDATES, 2-3e-2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_grad_name, load_golden


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_newton_schulz_vs_reference(device):
    from clt_training_amd.muon import newton_schulz

    t, meta = load_golden("muon_steps")
    for tag in ("", "_wide"):
        out = newton_schulz(t[f"ns_in{tag}"].to(device), meta["ns_steps"])
        assert out.dtype == torch.bfloat16 and out.shape == t[f"ns_in{tag}"].shape
        assert rel(out.float().cpu(), t[f"ns_out{tag}"]) < 2e-2, tag


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_muon_steps_vs_reference(device):
    """Three steps on a tall, a square and a wide matrix: weight decay, momentum, Nesterov, Newton-Schulz, the
    0.2 sqrt(max(shape)) scale (muon.py:131-203)."""
    from clt_training_amd.muon import Muon

    t, meta = load_golden("muon_steps")
    n = meta["n_params"]
    ps = [torch.nn.Parameter(t[f"p0.{i}"].clone().to(device)) for i in range(n)]
    opt = Muon(ps, lr=meta["lr"], momentum=meta["momentum"], nesterov=meta["nesterov"], weight_decay=meta["weight_decay"],
               ns_steps=meta["ns_steps"])
    for s in range(meta["steps"]):
        for i, p in enumerate(ps):
            p.grad = t[f"g{s}.{i}"].clone().to(device)
        opt.step()
        for i, p in enumerate(ps):
            upd, want = p.detach().cpu() - t[f"p{s}.{i}"], t[f"p{s + 1}.{i}"] - t[f"p{s}.{i}"]
            assert rel(upd, want) < 3e-2, (s, i, rel(upd, want))
            # (the next step starts from the reference's parameters: errors do not compound into the comparison)
            with torch.no_grad():
                p.copy_(t[f"p{s + 1}.{i}"].to(device))


def _sharded_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clt_training_amd.engine import Comm
    from clt_training_amd.muon import newton_schulz

    g = torch.Generator().manual_seed(5)
    G = torch.randn(128, 24, generator=g)
    rows = 128 // world
    mine = newton_schulz(G[rank * rows:(rank + 1) * rows].contiguous(), 5, Comm(None), rows_sharded=True, full_rows=128)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    if rank == 0:
        torch.save(dict(sharded=torch.cat(parts).float(), full=newton_schulz(G, 5).float()), out)
    dist.destroy_process_group()


def test_row_sharded_newton_schulz_equals_unsharded(tmp_path):
    """Latent shards: the Gram matrix is the all-reduced sum of the shards' X_r X_r^T; the result is the unsharded
    iteration's row block (bf16 tolerance: the partial sums are rounded in a different order)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "ns.pt")
    mp.spawn(_sharded_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=True)
    assert rel(r["sharded"], r["full"]) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["autograd", "engine"])
def test_trainer_muon_vs_reference_trainer(which):
    """Both trainers with optimizer="muon" (Muon for the matrices with a 0-step warm-up schedule, Adam for the vectors
    with the configured one) fed the activations the REFERENCE Trainer's hooks saw, against that run: every step starts
    from the reference's parameters of the step before, and the step's UPDATE is compared (matrices: bf16 Newton-Schulz
    tolerance; vectors: Adam, fp32 tolerance with the lr floor)."""
    import clt_training_amd.sparse_coder as sc
    from dataclasses import replace
    from test_trainer_golden import batches, close, make_cfg, snapshot
    from clt_training_amd import SparseCoder, Trainer
    from clt_training_amd.engine import ShardedCoder
    from clt_training_amd.tp import TPTrainer

    t, meta = load_golden("trainer_clt_muon")
    assert meta["optimizer"] == "muon"
    cfg = replace(make_cfg(meta, meta["hooks"]), optimizer="muon")
    L, d = meta["L"], meta["d"]
    old, sc.AUTOCAST_ENABLED = sc.AUTOCAST_ENABLED, False
    try:
        if which == "autograd":
            tr = Trainer(cfg, batches(t, meta, "cuda"), device="cuda", input_widths={h: d for h in meta["hooks"]})
            for l, h in enumerate(meta["hooks"]):
                tr.saes[h].load_state_dict({k[len(f"p0.{l}."):]: v for k, v in t.items() if k.startswith(f"p0.{l}.")})
        else:
            coders = []
            for l, h in enumerate(meta["hooks"]):
                n_targets = min(meta["cross_layer"], L - l) if meta["cross_layer"] > 0 else 0
                full = SparseCoder(d, replace(cfg.sae, n_targets=n_targets), "cpu", dtype=torch.float32)
                full.load_state_dict({k[len(f"p0.{l}."):]: v for k, v in t.items() if k.startswith(f"p0.{l}.")})
                coders.append(ShardedCoder.from_full(full, 0, 1, "cuda"))
            tr = TPTrainer(cfg, batches(t, meta, "cuda"), device="cuda", input_widths={h: d for h in meta["hooks"]},
                           coders=coders)
        assert len(tr.optimizers) == 2 and type(tr.optimizers[0]).__name__ == "Muon"
        for s, b in enumerate(batches(t, meta, "cuda")):
            tr.step(b)
            for l, h in enumerate(meta["hooks"]):
                before = snapshot(t, "init" if s == 0 else f"after{s - 1}", l)
                want = snapshot(t, f"after{s}", l)
                for n, p in tr.saes[h].named_parameters():
                    key = golden_grad_name(n)
                    got = p.detach().float().cpu()
                    if p.ndim >= 2:
                        upd, ref = got - before[key], want[key] - before[key]
                        assert rel(upd, ref) < 5e-2, (s, l, n, rel(upd, ref))
                    else:
                        close(got, want[key], f"step {s} {l}.{n}", floor=1e-3 * meta["lr"])
                    with torch.no_grad():  # the next step starts from the reference's state
                        p.copy_(want[key].to(p.device))
    finally:
        sc.AUTOCAST_ENABLED = old


def test_load_the_muon_checkpoint_the_reference_wrote(tmp_path):
    """optimizer_0.pt (Muon: exp_avg per matrix) and optimizer_1.pt (Adam for the vectors), lr_scheduler_{0,1}.pt as the
    reference Trainer wrote them (trainer.py:923-962) load into ``Trainer``; what save() writes back has the same keys."""
    from dataclasses import replace
    from conftest import GOLDEN
    from test_trainer_golden import batches, make_cfg
    from clt_training_amd import Trainer, ckpt

    t, meta = load_golden("trainer_clt_muon")
    path = GOLDEN / "trainer_clt_muon_ckpt"
    cfg = replace(make_cfg(meta, meta["hooks"]), optimizer="muon")
    tr = Trainer(cfg, batches(t, meta), device="cpu", input_widths={h: meta["d"] for h in meta["hooks"]})
    tr.load_state(path)
    assert tr.global_step == 3 and tr.lr_schedulers[0].last_epoch == 3 and tr.lr_schedulers[1].last_epoch == 3
    ref0 = ckpt.unflatten_state(torch.load(path / "optimizer_0.pt", weights_only=True))
    mats = [p for sae in tr.saes.values() for p in sae.parameters() if p.ndim >= 2]
    assert len(ref0["state"]) == len(mats)
    # the reference orders its parameters by NAME within a coder (trainer.py:195-199); so does Trainer for muon: position
    # i of the file is matrix i here
    names = [n for sae in tr.saes.values() for n, p in sorted(sae.named_parameters()) if p.ndim >= 2]
    assert names[:3] == ["W_dec", "W_decs.1", "W_skip"], names[:5]  # (slot 0 goes by its alias, as in the reference)
    mats = [p for sae in tr.saes.values() for _, p in sorted(sae.named_parameters()) if p.ndim >= 2]
    for i, p in enumerate(mats):
        assert torch.equal(tr.optimizers[0].state[p]["exp_avg"], ref0["state"][i]["exp_avg"]), (i, names[i])
    tr.save(tmp_path / "again")
    for i in (0, 1):
        a = torch.load(tmp_path / "again" / f"optimizer_{i}.pt", weights_only=True)
        b = torch.load(path / f"optimizer_{i}.pt", weights_only=True)
        assert {k for k in b if k[0] == "state"} <= set(a), i
