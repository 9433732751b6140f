"""Latent-sharded trainer: ``TPTrainer`` drives ``CLTEngine`` over one ``torch.distributed`` group
(one process per GPU; backend "nccl" is RCCL over xGMI on ROCm).  It keeps the reference Trainer's
construction rule, Adam groups + LR rule, warm-up schedule, step-0 init and step order
(sparsify/trainer.py:113-193, 547-592, 791-816); the mesh convention is (dp=1, tp=world)
(sparsify/__main__.py:173-177)."""
from __future__ import annotations


from dataclasses import replace
from typing import Optional

import torch
import torch.distributed as dist

from . import ckpt
from .config import TrainConfig
from .engine import ROUTES, CLTEngine, Comm, ShardedCoder
from .fit_loop import FitLoop


class TPTrainer(FitLoop):
    def __init__(self, cfg: TrainConfig, dataset, model=None, mesh=None, *, device="cuda",
                 input_widths: Optional[dict] = None, num_batches: Optional[int] = None, ops=None, group=None,
                 optimizer_cls=None, coders: Optional[list] = None, comm: Optional[Comm] = None):
        self.check_supported(cfg)
        if model is not None:  # frozen HF model + token rows: every rank feeds its own shard of the sequences
            dataset, input_widths, num_batches, device = self.attach_model(cfg, dataset, model)
        assert cfg.hookpoints, "cached-activation training needs explicit cfg.hookpoints"
        cfg.hookpoints = cfg.hookpoints[:: cfg.layer_stride]
        self.cfg, self.dataset, self.model = cfg, dataset, model
        self.device = torch.device(device)
        self.comm = comm or Comm(group)
        if ops is None:
            from . import hip_ops as ops  # the product path: HIP kernels, no CPU fallback
        self.ops = ops
        if input_widths is None:
            first = next(iter(dataset))
            input_widths = {h: first[h][0].shape[-1] for h in cfg.hookpoints}
        self.saes = {}
        if coders is None:
            sources = []
            for position, hook in enumerate(cfg.hookpoints):  # trainer.py:113-146
                torch.manual_seed(cfg.init_seeds[0] + 1000003 * self.comm.rank)  # every shard draws its own rows
                if cfg.cross_layer > 0:
                    n_targets = min(cfg.cross_layer, len(cfg.hookpoints) - position)
                    sources.append(position + n_targets)
                    while sources[0] < position:
                        sources.pop(0)
                    n_sources = len(sources)
                else:
                    n_targets = n_sources = 0
                sae_cfg = replace(cfg.sae, n_targets=n_targets, n_sources=n_sources)
                self.saes[hook] = ShardedCoder(input_widths[hook], sae_cfg, self.device, self.comm.rank, self.comm.world)
        else:
            self.saes = dict(zip(cfg.hookpoints, coders))
        self.engine = CLTEngine(cfg, [self.saes[h] for h in cfg.hookpoints], ops, self.comm)
        if num_batches is None:
            num_batches = len(dataset)
        # trainer.py:166-173: the FULL latent count sets the LR
        muon_kw = None
        if cfg.optimizer == "muon":  # the [M_local, d] tables are row blocks of the full matrices (muon.newton_schulz)
            tables = [p for s in self.saes.values() for p in [s.W_enc, *s.W_decs] if p.requires_grad]
            muon_kw = dict(comm=self.comm, sharded={id(p) for p in tables},
                           full_rows={id(p): p.shape[0] * self.comm.world for p in tables})
        self.optimizers, self.lr_schedulers = self.build_optimizers(
            cfg, [[p for p in sae.parameters() if p.requires_grad] for sae in self.saes.values()],
            [sae.num_latents for sae in self.saes.values()], num_batches, adam_cls=optimizer_cls,
            signsgd_cls=getattr(ops, "SignSGD", None), muon_kw=muon_kw)
        opt = self.optimizers[0]
        # Decoder / encoder row gradients go straight into the optimizer update (no .grad round trip) when nothing
        # else consumes them: one micro-step per optimizer step, and an optimizer that has the fused form (Adam).
        # (a unit-norm SAE decoder's gradient projection happens inside the same row pass: engine.layer_step)
        fuse = (cfg.optimizer in ("adam", "adam8") and hasattr(opt, "fused_rows")
                and cfg.grad_acc_steps * cfg.micro_acc_steps == 1 and ROUTES.fused_adam)
        self.engine.fused_opt = opt if fuse else None
        self.global_step = 0
        self.num_tokens_since_fired = {h: torch.zeros(s.m_local, device=self.device, dtype=torch.long)
                                       for h, s in self.saes.items()}
        # one spare slot at the end takes the indices owned by other ranks (no boolean-mask indexing: no host sync)
        self._did_fire = {h: torch.zeros(s.m_local + 1, device=self.device, dtype=torch.bool) for h, s in self.saes.items()}
        self.did_fire = {h: f[:-1] for h, f in self._did_fire.items()}
        self.num_tokens_in_step = 0
        self.last_fvu = {}
        self._enc_stream = None
        self.init_k_schedule()

    @torch.no_grad()
    def step(self, batch: dict):
        cfg, eng = self.cfg, self.engine
        acc_steps = cfg.grad_acc_steps * cfg.micro_acc_steps
        eng.reset()
        eng.begin_step(packed=self.global_step > 0 or bool(cfg.finetune))
        mask = batch.get("loss_mask")
        n_tok = 0
        # Two streams under tp > 1: the encoder half of layer j + 1 (input all-gather, GEMM + top-k, candidate
        # exchange) is enqueued on a side stream before the decoder / backward half of layer j on the main stream, so
        # the collectives of either half run behind the kernels of the other.  Step 0 (data-dependent init) is serial.
        names = cfg.hookpoints
        overlap = ((self.comm.world > 1 or ROUTES.overlap_single) and self.device.type == "cuda"
                   and self.global_step > 0 and ROUTES.overlap)
        pending = {}
        pre = None
        if not overlap and (self.global_step > 0 or bool(cfg.finetune)):  # (step 0: data-dependent init, layer by layer)
            pre = eng.encode_all([batch[name][0] for name in names])
        if overlap:
            main = torch.cuda.current_stream(self.device)
            if self._enc_stream is None:
                self._enc_stream = torch.cuda.Stream(self.device)
            self._enc_stream.wait_stream(main)  # last step's parameter updates; this step's batch

            def launch_encode(jj):
                xx, yy = batch[names[jj]]
                with torch.cuda.stream(self._enc_stream):
                    e = eng.encode_part(jj, xx, yy)
                    ev = torch.cuda.Event()
                    ev.record(self._enc_stream)
                pending[jj] = (e, ev)

            launch_encode(0)
        for j, name in enumerate(names):
            x, y = batch[name]
            sae = self.saes[name]
            n_tok = x.shape[0] * self.comm.world
            if self.global_step == 0 and not cfg.finetune:
                eng.first_step_init(j, x if cfg.sae.transcode else y, y)
            if sae.cfg.normalize_decoder and not cfg.sae.transcode:  # trainer.py:595-596 (row-wise: shard-local)
                W = sae.W_decs[0]
                if hasattr(self.ops, "rows_unit_norm"):
                    self.ops.rows_unit_norm(W.data)
                else:
                    W.data /= W.data.norm(dim=1, keepdim=True) + torch.finfo(W.dtype).eps
            enc = pre[j] if pre is not None else None
            if overlap:
                if j + 1 < len(names):
                    launch_encode(j + 1)
                enc, ev = pending.pop(j)
                main.wait_event(ev)
                for t in enc.values():
                    if t is not None:
                        t.record_stream(main)  # allocated on the side stream, consumed (and released) on this one
            self.last_fvu[name] = eng.layer_step(j, x, y, mask, acc_steps, enc=enc)
            # fired-latent bookkeeping on this rank's latent range (trainer.py:622-628)
            loc = eng.last_idx[j].flatten() - sae.lat_begin
            if self.comm.world > 1:
                loc = torch.where((loc >= 0) & (loc < sae.m_local), loc, sae.m_local)
            self._did_fire[name].index_fill_(0, loc, True)
        self.num_tokens_in_step += n_tok
        step, substep = divmod(self.global_step + 1, cfg.grad_acc_steps)
        if substep == 0:
            eng.sync_replicated_grads()
            if cfg.sae.normalize_decoder and not cfg.sae.transcode:  # sparse_coder.py:696-711 (row-wise)
                for sae in self.saes.values():
                    W = sae.W_decs[0]
                    if W.grad is None:
                        continue  # projected and consumed inside the row pass (clt_decode_bwd_rows_adam, project_out)
                    if hasattr(self.ops, "rows_project_out") and W.grad.dtype == W.dtype:
                        self.ops.rows_project_out(W.grad, W.data)
                    else:
                        W.grad -= (W.grad * W.data).sum(1, keepdim=True) * W.data
            for opt in self.optimizers:
                opt.step()
                opt.zero_grad()
            for sch in self.lr_schedulers:
                sch.step()
            self.set_correct_k()  # trainer.py:804
            with torch.no_grad():  # trainer.py:806-816
                for name, counts in self.num_tokens_since_fired.items():
                    counts += self.num_tokens_in_step
                    counts.masked_fill_(self.did_fire[name], 0)  # no host sync, unlike counts[mask] = 0
                self.num_tokens_in_step = 0
                for fired in self._did_fire.values():
                    fired.zero_()
        self.global_step += 1
        return self.last_fvu

    # ------------------------------------------------------------------ checkpoint (trainer.py:910-964, 258-348)
    def save(self, path=None, only=None):
        """``{path}/{hookpoint}/sae.safetensors + cfg.json`` in the reference's unsharded format (the latent shards are
        gathered to rank 0), ``lr_scheduler_0.pt``, ``state.pt`` (global_step) and ``rank_{r}_state.pt``
        (num_tokens_since_fired, best_loss) as the reference writes them, and ``optimizer_0.pt``: the shards of every
        state tensor gathered to rank 0 and flattened exactly as the reference does (utils.py:193-277, 280-360), so the
        checkpoint resumes on any tp degree, here or in the reference.  EXCEPTION: ``optimizer="muon"`` writes per-rank
        ``optimizer_{i}.rank{r}.pt`` files (shard-local momentum) and resumes on the same tp degree only -- neither a
        reference-written nor a ``Trainer``-written Muon checkpoint loads here (load_state says so)."""
        from pathlib import Path

        path = Path(path or self.default_path())
        rank = self.comm.rank
        if rank == 0:
            path.mkdir(parents=True, exist_ok=True)
            self.save_config(path)
        if self.comm.world > 1:
            dist.barrier(group=self.comm.group)
        for name, sae in self.saes.items():
            if only is None or name in only:
                sae.save_to_disk(path / name, self.comm)
        if self.cfg.optimizer == "muon":
            # Muon + Adam: each rank keeps its shard's state (resumes on the same tp degree only; the gathered,
            # reference-layout file below is written for the single-optimizer configurations)
            for i, opt in enumerate(self.optimizers):
                torch.save(opt.state_dict(), path / f"optimizer_{i}.rank{rank}.pt")
        else:
            flat = self._gathered_optimizer_state()  # collective: every rank takes part, rank 0 gets the dict
            if rank == 0:
                torch.save(flat, path / "optimizer_0.pt")
        torch.save({"num_tokens_since_fired": {h: t.cpu() for h, t in self.num_tokens_since_fired.items()},
                    "best_loss": self.best_loss}, path / f"rank_{rank}_state.pt")
        if rank == 0:
            for i, sch in enumerate(self.lr_schedulers):
                torch.save(sch.state_dict(), path / f"lr_scheduler_{i}.pt")
            torch.save({"global_step": self.global_step}, path / "state.pt")
        if self.comm.world > 1:
            dist.barrier(group=self.comm.group)

    def load_state(self, path):
        """Resume from ``save(path)`` -- or from a checkpoint the reference wrote -- on any tp degree (Muon: from this
        class's own per-rank files at the same tp degree only)."""
        from pathlib import Path

        path = Path(path)
        rank = self.comm.rank
        with torch.no_grad():
            for name, sae in self.saes.items():
                loaded = ShardedCoder.load_from_disk(path / name, self.device, rank, self.comm.world)
                for p, q in zip(sae.parameters(), loaded.parameters()):
                    p.copy_(q.to(p.dtype))
                for (_, b), (_, c) in zip(sae.named_buffers(), loaded.named_buffers()):
                    b.copy_(c)
        if self.cfg.optimizer == "muon":
            if not (path / f"optimizer_0.rank{rank}.pt").exists():
                raise FileNotFoundError(
                    f"{path}: no optimizer_0.rank{rank}.pt -- a Muon checkpoint resumes under TPTrainer only from "
                    "TPTrainer.save() at the same tp degree (the reference / Trainer layout optimizer_{0,1}.pt is not "
                    "re-sharded for Muon)")
            for i, opt in enumerate(self.optimizers):
                opt.load_state_dict(torch.load(path / f"optimizer_{i}.rank{rank}.pt", map_location=self.device,
                                               weights_only=True))
        else:
            self._scatter_optimizer_state(torch.load(path / "optimizer_0.pt", map_location="cpu", weights_only=True))
        for i, sch in enumerate(self.lr_schedulers):
            sch.load_state_dict(torch.load(path / f"lr_scheduler_{i}.pt", weights_only=True))
        self.global_step = torch.load(path / "state.pt", weights_only=True)["global_step"]
        fired = torch.load(path / f"rank_{rank}_state.pt", map_location=self.device, weights_only=True)
        for h, t in fired["num_tokens_since_fired"].items():
            self.num_tokens_since_fired[h].copy_(t)
        saved_best = fired.get("best_loss") or {}
        if isinstance(saved_best, dict) and isinstance(self.best_loss, dict):
            for h, v in saved_best.items():
                if h in self.best_loss:
                    self.best_loss[h] = v
        return self

    # The reference gathers the tp shards of every optimizer state tensor to rank 0 and writes ONE flattened
    # ``optimizer_0.pt`` whose parameter order is ``sae.parameters()`` of the unsharded module (utils.py:193-277,
    # 280-360; trainer.py:923-937): the file does not depend on the tp degree.  Same here.
    @staticmethod
    def _reference_order(sae: ShardedCoder):
        """[(reference parameter name, this shard's parameter, latent-sharded?)] in the order of the unsharded
        ``SparseCoder.parameters()`` (the names and their order do not depend on the widths: a tiny skeleton gives them)."""
        from .sparse_coder import SparseCoder

        skel = SparseCoder(8, replace(sae.cfg, num_latents=8, k=1), "cpu", dtype=torch.float32)
        lists = dict(W_dec=sae.W_decs, b_dec=sae.b_decs, W_skip=sae.W_skips, post_enc=sae.post_encs,
                     post_enc_scale=sae.post_enc_scales)
        out = []
        for name, _ in skel.named_parameters():
            if name == "encoder.weight":
                p, sharded = sae.W_enc, True
            elif name == "encoder.bias":
                p, sharded = sae.b_enc, True
            else:
                stem, _, slot = name.partition(".")
                stem = stem[:-1] if slot else stem  # "W_decs.1" -> ("W_dec", 1); "W_dec" -> ("W_dec", 0)
                p = lists[stem][int(slot) if slot else 0]
                sharded = stem in ("W_dec", "post_enc", "post_enc_scale")
            out.append((name, p, sharded))
        return out

    def _gathered_optimizer_state(self):
        opt = self.optimizers[0]
        state, groups, pos = {}, [], 0
        for sae, g in zip(self.saes.values(), opt.param_groups):
            ids = []
            for _, p, sharded in self._reference_order(sae):
                if not p.requires_grad:
                    continue
                st = opt.state.get(p, {})
                ent = {}
                for key in ("exp_avg", "exp_avg_sq"):
                    if key in st:
                        t = st[key].detach()
                        ent[key] = (self.comm.all_gather_rows(t) if sharded else t).cpu()
                if ent:
                    if "step" in st:  # (torch.optim.Adam in the CPU tests keeps one per parameter)
                        ent["step"] = st["step"].detach().cpu()
                    state[pos] = ent
                ids.append(pos)
                pos += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        if self.comm.rank != 0:
            return None
        return ckpt.to_reference_layout({"state": state, "param_groups": groups}, getattr(opt, "_step", 0))

    def _scatter_optimizer_state(self, blob):
        nested, step = ckpt.from_file_layout(blob)
        opt = self.optimizers[0]
        pos = 0
        for sae, g, saved in zip(self.saes.values(), opt.param_groups, nested["param_groups"]):
            for _, p, sharded in self._reference_order(sae):
                if not p.requires_grad:
                    continue
                st = nested["state"].get(pos)
                pos += 1
                if st is None:
                    continue
                mine = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()
                        if k not in ("exp_avg", "exp_avg_sq")}  # (torch.optim.Adam's per-parameter step)
                for key in ("exp_avg", "exp_avg_sq"):
                    if key not in st:  # (an optimizer without moments, or a file saved before the first step)
                        continue
                    t = st[key]
                    if sharded:
                        t = t[sae.lat_begin:sae.lat_begin + sae.m_local]
                    assert t.shape == p.shape, (key, tuple(t.shape), tuple(p.shape))
                    mine[key] = t.to(device=p.device, dtype=p.dtype).contiguous().clone()
                opt.state[p] = mine
            for key in ("lr", "initial_lr"):
                if key in saved:
                    g[key] = saved[key]
        if step is not None:
            opt._step = step
        opt._tables = {}
