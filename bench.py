"""bench.py -- train tokens/s of the GPT-2 CLT (12 layers, cross_layer=12, expansion 128, k=32,
skip connection, bf16) fwd + bwd + Adam step on MI355X, with synthetic cached activations.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU over RCCL.  Launched by torch.distributed.run, every rank reads RANK / LOCAL_RANK /
     WORLD_SIZE / MASTER_* from the environment; started plainly (`python bench.py --gpus N`, WORLD_SIZE unset) it
     starts its N ranks itself, as a torch.distributed.run child, BEFORE this process touches the GPU.)

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job tokens/s, plus
`roofline` for the dominant kernel (HIP events on the launch stream, over the timed region),
`kernels` (every C-ABI kernel: launches/step, mean ms, achieved vs peak) and `cpu_baseline`
(the torch-CPU restatement of the reference's eager step timed on this box's host cores)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16 = 2.5e15   # MI355X dense bf16 MFMA, FLOP/s (MI355X_MICROARCH.md)
PEAK_HBM = 8.0e12    # HBM3E, B/s


def synth_batches(hooks, N, d, n_batches, dtype, device, seed=0):
    """x = randn + 0.1*randn(d); y = x @ R_l + 0.5*randn; mask drops token 0 of every 128."""
    out = []
    for s in range(n_batches):
        b = {}
        for l, h in enumerate(hooks):
            g = torch.Generator(device=device).manual_seed(1000 * s + l + seed)
            gr = torch.Generator(device=device).manual_seed(7 + l)
            x = torch.randn(N, d, generator=g, device=device) + 0.1 * torch.randn(d, generator=g, device=device)
            R = torch.randn(d, d, generator=gr, device=device) / d**0.5
            y = x @ R + 0.5 * torch.randn(N, d, generator=g, device=device)
            b[h] = (x.to(dtype), y.to(dtype))
        m = torch.ones(N, dtype=torch.bool, device=device)
        m[::128] = False
        b["loss_mask"] = m
        out.append(b)
    return out


class SimComm:
    """Per-rank COMPUTE rehearsal of a tp-way run on one GPU (--simulate-tp W): this process does what rank 0 of a
    W-rank group does, with every collective replaced by a local stand-in of the same shapes (all-gather = rolled
    copies, so tokens and candidates differ per pretend rank; reduce-scatter = a slice; all-reduce = identity).
    The arithmetic results are meaningless and no RCCL time is included: it sizes kernels at their tp shapes, it is
    not a measurement of the metric."""

    def __init__(self, world, m_local):
        self.enabled, self.group, self.world, self.rank, self.backend, self.m_local = True, None, world, 0, "sim", m_local

    def all_gather_rows(self, t, tag=None):
        return torch.cat([t.roll(37 * r, 0) for r in range(self.world)], 0)

    def all_gather_cat1(self, t, tag=None):
        parts = [t.roll(37 * r, 0) for r in range(self.world)]
        if t.dtype == torch.int64:  # global candidate indices: pretend rank r owns [r * m_local, (r + 1) * m_local)
            parts = [p + r * self.m_local for r, p in enumerate(parts)]
        return torch.cat(parts, 1).contiguous()

    def all_to_all_rows(self, t, tag=None):
        n = t.shape[0] // self.world
        parts = [t[:n].roll(37 * r, 0) for r in range(self.world)]
        if t.dtype == torch.int64:  # packed (value bits << 32 | global index) candidates of pretend rank r
            parts = [p + r * self.m_local for r, p in enumerate(parts)]
        return torch.cat(parts, 0)

    def reduce_scatter_rows(self, t, tag=None):
        return t[: t.shape[0] // self.world].clone()

    def all_reduce_(self, t, tag=None):
        return t


def algorithmic_work(name, N, d, M, k, es, P_total, world=1, pairs_per_target=1.0, adam_bytes=None):
    """Algorithmic flops / bytes per launch -> (amount, 'flop'|'byte').
    N = tokens of the whole group, M = latents of this rank, k / world = entries a rank owns per token and source;
    pairs_per_target = mean number of live sources per target launch ((L + 1) / 2 with cross_layer = L)."""
    if name in ("clt_decode_fwd_multi", "clt_decode_bwd_acts_multi"):  # the pairs of one target: gathered rows + one row
        return es * N * (pairs_per_target * k / world + 1) * d, "byte"
    if name in ("clt_decode_bwd_rows_adam", "clt_decode_bwd_rows_adam_multi"):
        # per (source, target) pair: parameter / exp_avg / exp_avg_sq rows read + written, G rows gathered; the batched
        # entry point does `jobs` pairs per launch (the caller multiplies)
        return es * (6 * M * d + N * k / world * d), "byte"
    if name == "clt_decode_bwd_rows":
        return es * (2 * M * d + N * k / world * d), "byte"
    if name in ("clt_encoder_gemm_relu", "clt_encode_topk_fwd", "clt_encode_groupmax_fwd"):  # the fused call is priced as its GEMM alone
        return 2.0 * N * d * M, "flop"
    if name == "clt_topk_rows":
        return N * M * es, "byte"
    if name in ("clt_decode_fwd", "clt_decode_bwd_acts"):
        return es * N * (k + 1) * d, "byte"
    if name == "clt_spt_dense":
        return es * (M * d + N * k / world * d), "byte"
    if name in ("clt_spt_dense_adam", "clt_spt_dense_adam_multi"):  # gathered rows + read/write of the parameter, exp_avg and exp_avg_sq rows (per job)
        return es * (6 * M * d + N * k / world * d), "byte"
    if name == "clt_adam_step":  # p, g, exp_avg, exp_avg_sq read + p, exp_avg, exp_avg_sq written, each in the parameter's dtype
        return 7.0 * (adam_bytes if adam_bytes is not None else es * P_total), "byte"
    if name == "clt_fvu_fwd_bwd":
        return N * d * es * 6, "byte"
    return None, None


def host_launch_cost(args, device, steps=10):
    """Pure host cost of one step's launch sequence, ms: the SAME sequence (layers, pairs, routes) on a twin whose
    tables are 16x smaller and whose batch is 256 tokens, so the GPU finishes each step long before the host has
    issued the next one and no launch call ever blocks on a full queue."""
    from clt_training_amd import SparseCoderConfig, TrainConfig
    from clt_training_amd.tp import TPTrainer

    L, d = args.layers, args.d_model
    hooks = [f"h.{i}.mlp" for i in range(L)]
    sae_cfg = SparseCoderConfig(dtype=args.dtype if args.dtype == "bfloat16" else "none", expansion_factor=8, k=args.k,
                                transcode=not args.sae, skip_connection=not args.sae, activation=args.activation,
                                coalesce_topk=args.coalesce, post_encoder_scale=args.coalesce == "concat")
    cfg = TrainConfig(sae=sae_cfg, hookpoints=hooks, cross_layer=0 if args.sae else L, optimizer=args.optimizer, lr=2e-4,
                      lr_warmup_steps=50, log_to_wandb=False, force_lr_warmup=args.optimizer == "signum")
    act_dtype = torch.bfloat16 if args.dtype == "bfloat16" else torch.float32
    batches = synth_batches(hooks, 256, d, 2, act_dtype, device)
    twin = TPTrainer(cfg, batches, device=device, num_batches=10_000)
    for i in range(3):
        twin.step(batches[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        twin.step(batches[i % 2])
    t = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t / steps * 1e3


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run the same command line under torch.distributed.run
    (one rank per GPU, rendezvous on 127.0.0.1) as a CHILD process and return its exit code.  Nothing in this
    process has initialised the GPU (torch.cuda.device_count() does not), and it never does."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < n and not os.environ.get("CLT_BENCH_REHEARSAL"):
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    print("bench.py: starting", n, "ranks:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tokens", type=int, default=8192, help="tokens per step per GPU")
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--d-model", type=int, default=768)
    ap.add_argument("--expansion", type=int, default=128)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float32"])
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the side legs after the timed region (CPU baseline, host-launch twin): profiling runs")
    ap.add_argument("--path", default="engine", choices=["engine", "autograd"],
                    help="engine: explicit CLTEngine (also the multi-GPU path); autograd: SparseCoder/CrossLayerRunner")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--sae", action="store_true",
                    help="plain SAE per hookpoint (no transcoding, no skip, unit-norm decoder): BASELINE configs[3] shape "
                         "with --layers 1 --d-model 4096 --expansion 32 --k 192; a secondary data point, not the metric")
    ap.add_argument("--activation", default="topk", choices=["topk", "groupmax", "batchtopk"],
                    help="groupmax / batchtopk (fused_encoder.py:373-413 / 204-240) are secondary data points, not the metric")
    ap.add_argument("--optimizer", default="adam", choices=["adam", "signum"],
                    help="signum = SignSGD + linear warm-up (force_lr_warmup); a secondary data point, not the metric")
    ap.add_argument("--simulate-tp", type=int, default=0,
                    help="rehearse rank 0's compute of a W-rank run on one GPU (collectives are local stand-ins; "
                         "NOT a measurement of the metric: the line is marked 'simulated')")
    ap.add_argument("--coalesce", default="none", choices=["none", "concat"],
                    help="concat = BASELINE configs[4]: per-target tied decoder (one W_dec per layer), post_encoder_scale; "
                         "a secondary data point, not the metric")
    ap.add_argument("--routes-off", default="", help="A/B timing only: comma-separated engine.ROUTES switches to turn off")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="HIP-event timing of the C-ABI calls on every n-th timed step (each event pair leaves ~10 us of "
                         "idle GPU at a call boundary: ~1.4 ms/step when every call of every step is bracketed)")
    ap.add_argument("--routes-on", default="", help="A/B timing only: engine.ROUTES switches that are off by default")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        if os.environ.get("CLT_BENCH_REHEARSAL"):  # every rank on GPU 0, gloo on host copies: a functional check of
            local_rank = 0                          # the multi-rank path on a one-GPU box, not a measurement
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        print(f"bench.py: rank {dist.get_rank()} of {dist.get_world_size()} ranks, backend {dist.get_backend()}, "
              f"device cuda:{local_rank}", file=sys.stderr, flush=True)
        if args.gpus != world:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is used",
                  file=sys.stderr, flush=True)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from clt_training_amd import SparseCoderConfig, TrainConfig, Trainer
    from clt_training_amd import hip_ops
    from clt_training_amd.engine import ROUTES

    for name in filter(None, args.routes_off.split(",")):
        assert hasattr(ROUTES, name), name
        setattr(ROUTES, name, False)
    for name in filter(None, args.routes_on.split(",")):
        assert hasattr(ROUTES, name), name
        setattr(ROUTES, name, True)

    L, d, k = args.layers, args.d_model, args.k
    M = d * args.expansion
    N = args.tokens
    hooks = [f"h.{i}.mlp" for i in range(L)]
    sae_cfg = SparseCoderConfig(dtype=args.dtype if args.dtype == "bfloat16" else "none", expansion_factor=args.expansion,
                                k=k, transcode=not args.sae, skip_connection=not args.sae, activation=args.activation,
                                coalesce_topk=args.coalesce, post_encoder_scale=args.coalesce == "concat")
    cfg = TrainConfig(sae=sae_cfg, hookpoints=hooks, cross_layer=0 if args.sae else L, optimizer=args.optimizer, lr=2e-4,
                      lr_warmup_steps=50, log_to_wandb=False, force_lr_warmup=args.optimizer == "signum")
    act_dtype = torch.bfloat16 if args.dtype == "bfloat16" else torch.float32
    es = 2 if args.dtype == "bfloat16" else 4

    if world > 1 or args.path == "engine":
        from clt_training_amd.tp import TPTrainer

        batches = synth_batches(hooks, N, d, 2, act_dtype, device, seed=rank * 7919)
        sim = SimComm(args.simulate_tp, M // args.simulate_tp) if args.simulate_tp > 1 and world == 1 else None
        trainer = TPTrainer(cfg, batches, device=device, num_batches=10_000, comm=sim)
        parallelism = f"tp{world}" if world > 1 else "single"
        if world > 1 and os.environ.get("CLT_BENCH_REHEARSAL"):
            parallelism = f"REHEARSAL of tp{world}: every rank on GPU 0, gloo on host copies, no RCCL"
        if sim is not None:
            world, parallelism = args.simulate_tp, f"SIMULATED rank 0 of tp{args.simulate_tp}, no RCCL"
    else:
        batches = synth_batches(hooks, N, d, 2, act_dtype, device)
        trainer = Trainer(cfg, batches, device=device, num_batches=10_000)
        parallelism = "single"
    # transcoder decoders are zero-initialised (sparse_coder.py:304-338): use the "warm" variant so
    # the decode / backward arithmetic is not trivially zero
    with torch.no_grad():
        for sae in trainer.saes.values():
            g = torch.Generator(device=device).manual_seed(1)
            for n, p in sae.named_parameters():
                if n.startswith("W_dec"):
                    p.copy_((0.02 * torch.randn(p.shape, generator=g, device=device)).to(p.dtype))
                elif n.startswith("post_encs"):
                    p.copy_(0.01 * torch.randn(p.shape, generator=g, device=device))
    P_total = sum(p.numel() for sae in trainer.saes.values() for p in sae.parameters())
    # parameters that still go through clt_adam_step: everything but the [M, d] matrices when their update is fused
    fused = getattr(getattr(trainer, "engine", None), "fused_opt", None) is not None or getattr(trainer, "_fused", False)
    small = [p for sae in trainer.saes.values() for p in sae.parameters() if not (fused and p.dim() == 2 and p.shape[0] > d)]
    P_adam = sum(p.numel() for p in small)
    B_adam = sum(p.numel() * p.element_size() for p in small)  # (biases, skips and post-encoder vectors are fp32)

    real_world = int(os.environ.get("WORLD_SIZE", "1"))

    def barrier():
        if real_world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        trainer.step(batches[i % len(batches)])
    barrier()
    prof = hip_ops.KernelProfile()
    hip_ops.PROFILE = prof
    t0 = time.perf_counter()
    stride = max(1, args.event_stride)
    sampled = len(range(0, args.steps, stride))
    comm = getattr(trainer, "comm", None)
    coll = {} if real_world > 1 and hasattr(comm, "timing") else None  # collectives: events on the sampled steps only
    for i in range(args.steps):
        prof.active = i % stride == 0
        if coll is not None:
            comm.timing = coll if prof.active else None
        fv = trainer.step(batches[i % len(batches)])
    if coll is not None:
        comm.timing = None
    t_enqueue = time.perf_counter() - t0  # host time to issue the steps: close to dt means launch-bound
    barrier()
    dt = time.perf_counter() - t0
    hip_ops.PROFILE = None
    if real_world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    tokens_per_step = N * world
    value = tokens_per_step * args.steps / dt

    if rank == 0:
        # ---- per-kernel table and the dominant kernel's roofline
        Ml = M // world
        kernels = {}
        dominant, dom_ms = None, -1.0
        for name, (cnt, ms) in prof.summary().items():
            amount, kind = algorithmic_work(name, tokens_per_step, d, Ml, k, es, P_adam, world, 1.0 if args.sae else (L + 1) / 2, B_adam)
            if amount is not None and name in prof.units:  # batched entry point: pairs per launch, counted live
                amount *= prof.units[name] / cnt
            ent = {"launches_per_step": cnt / sampled, "mean_ms": round(ms, 4),
                   "ms_per_step": round(ms * cnt / sampled, 3)}
            if name in prof.units:
                ent["jobs_per_launch"] = round(prof.units[name] / cnt, 3)
            if amount is not None:
                rate = amount / (ms * 1e-3)
                peak = PEAK_BF16 if kind == "flop" else PEAK_HBM
                ent.update({"achieved": round(rate / (1e12 if kind == "flop" else 1e9), 1),
                            "unit": "TFLOP/s" if kind == "flop" else "GB/s", "frac": round(rate / peak, 4),
                            "bound": "mfma" if kind == "flop" else "hbm"})
                if ent["ms_per_step"] > dom_ms:
                    dominant, dom_ms = name, ent["ms_per_step"]
            kernels[name] = ent
        roofline = None
        if dominant:
            e = kernels[dominant]
            # HBM bytes per launch from the PMC passes kept under profiles/ (separate rocprofv3 --pmc runs at these
            # layer shapes; counters cannot be read from inside this process)
            traffic = None
            def _ver(path):  # rNN_pmc_traffic[_vK].json -> (NN, K): the newest pass (tools/pmc_traffic.py)
                import re
                m = re.match(r"r(\d+)_pmc_traffic(?:_v(\d+))?\.json$", path.name)
                return (int(m.group(1)), int(m.group(2) or 0)) if m else (-1, 0)
            pmcs = sorted((ROOT / "profiles").glob("r*_pmc_traffic*.json"), key=_ver)
            if pmcs and world == 1 and (N, d, M, k, es) == (8192, 768, 98304, 32, 2):
                pk = json.loads(pmcs[-1].read_text())["kernels"]
                traffic = pk.get(dominant, {}).get("hbm_bytes")
                if dominant in prof.units:  # batched launch: PMC bytes per (source, target) pair x the pairs per launch
                    per_job = pk.get(dominant, {}).get("hbm_bytes_per_job") or pk.get(dominant[:-len("_multi")], {}).get("hbm_bytes")
                    traffic = None if per_job is None else int(per_job * prof.units[dominant] / prof.summary()[dominant][0])
            roofline = {"kernel": dominant, "bound": e["bound"], "achieved": e["achieved"],
                        "peak": 2500.0 if e["bound"] == "mfma" else 8000.0, "unit": e["unit"], "frac": e["frac"],
                        "traffic": traffic}
        # Whole-step roofline of SURVEY.md section 8d (the reference's own cost model, scripts/benchmarks/distributed.py:
        # 30-38, with this part's peaks), per rank: encoder flops + skip GEMM flops at the MFMA peak, decode forward
        # (B_dec per pair), decode backward (2 B_dec per pair), encoder sparse backward (B_dec per layer) and dense
        # Adam (7 * s * P) at the HBM peak.  frac = that time / the measured step.
        pairs = L if args.sae else L * (L + 1) // 2
        Ntot = tokens_per_step
        B_dec = es * Ntot * (k / world + 1) * d
        flops = L * 2.0 * Ntot * d * Ml + pairs * 3 * 2.0 * N * d * d
        hbm = pairs * 3 * B_dec + L * B_dec + 7.0 * es * P_total
        roof_s = flops / PEAK_BF16 + hbm / PEAK_HBM
        step_roofline = {"ms": round(roof_s * 1e3, 3), "frac": round(roof_s / (dt / args.steps), 4),
                         "tokens_per_s_at_roofline": round(tokens_per_step / roof_s, 1)}
        host_ms = None
        # (not in profiling runs: --no-cpu-baseline leaves out both side legs, so that a rocprofv3 trace of the command
        # holds the timed workload's kernels only -- the twin launches the same kernels at tiny shapes)
        if real_world == 1 and args.simulate_tp <= 1 and args.path == "engine" and not args.no_cpu_baseline:
            host_ms = round(host_launch_cost(args, device), 3)
        cpu = None
        if not args.no_cpu_baseline and real_world == 1:  # rank 0 at N = 1 only
            sys.path.insert(0, str(ROOT))
            from oracle import cpu_baseline

            cpu = cpu_baseline.run(L=2, d=d, M=M, k=k, N=1024, min_seconds=args.cpu_seconds, full_L=L)
            cpu["c1"] = cpu_baseline.run_c1(min_seconds=6.0)  # BASELINE configs[0], whole steps, not extrapolated
        line = {
            "metric": "train tokens/sec, GPT-2 CLT (exp128,k32) fwd+bwd+Adam", "value": round(value, 1),
            "unit": "tokens/s", "n_gpus": real_world,
            # simulated: not a measurement of the metric -- one GPU doing rank 0's compute (--simulate-tp), or several
            # ranks sharing GPU 0 over gloo (CLT_BENCH_REHEARSAL)
            "simulated": (args.simulate_tp > 1 and real_world == 1) or (real_world > 1 and bool(os.environ.get("CLT_BENCH_REHEARSAL"))),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            # wall time the host spent enqueueing the timed steps.  It INCLUDES back-pressure: the HIP queue is bounded, so
            # once the host is a few hundred launches ahead of a GPU-bound step its launch calls block.  The pure host
            # cost of this launch sequence is host_launch_ms_per_step (host_launch_cost: same sequence, tiny kernels).
            "host_enqueue_wall_ms_per_step": round(t_enqueue / args.steps * 1e3, 3), "host_launch_ms_per_step": host_ms,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.dtype == "bfloat16" else "f32", "data": "synthetic",
            "config": {"workload": f"GPT-2 small CLT: {L} layers, cross_layer={L}, d={d}, latents={M}, k={k}, "
                                   f"skip_connection, {args.dtype} weights, {'Adam' if args.optimizer == 'adam' else 'SignSGD'}{'' if args.activation == 'topk' else ', ' + args.activation}; {N} tokens/step/GPU "
                                   f"({'BASELINE.json configs[4]: coalesce_topk=concat, post_encoder_scale' if args.coalesce == 'concat' else 'BASELINE.json configs[1]'})",
                       "tokens_per_step": tokens_per_step, "params_per_gpu": P_total, "parallelism": parallelism},
            "fvu_last": round(float(sum(float(v) for v in fv.values()) / len(fv)), 5),
            "roofline": roofline, "step_roofline": step_roofline,
            "kernel_timing": f"HIP events around every C-ABI call on {sampled} of the {args.steps} timed steps (every "
                             f"{stride}{'st' if stride == 1 else 'th'}), on the launch stream",
            "kernels": kernels, "cpu_baseline": cpu,
        }
        if real_world > 1:
            # what the scaling number is made of (the method of scripts/benchmarks/distributed.py:11-57, per collective):
            # each class of exchange timed with HIP events on the stream that issued it, rank 0, sampled steps; with
            # overlap on, the encoder half's collectives (input all-gather, candidate exchange) run on the side stream
            # beside the decoder / backward kernels of the previous layer, so their time is not additive to the step
            from clt_training_amd.engine import ROUTES as _R

            line["rccl"] = {"world": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend()}
            line["collectives"] = comm.timing_summary(sampled, coll) if coll is not None else None
            line["collectives_timing"] = "HIP events around every collective on the issuing stream, rank 0, sampled steps"
            line["overlap"] = {"two_streams": bool(_R.overlap), "encoder_half_on_side_stream": bool(_R.overlap)}
        print(json.dumps(line), flush=True)
    if real_world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
