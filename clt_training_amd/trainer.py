"""Trainer over cached activations: the train-step hot path of sparsify/trainer.py.

The reference's ``Trainer`` (trainer.py:61-997) hooks a frozen HF model; everything
between the hook and the optimizer step is reproduced here -- SAE construction rule
(:113-146), Adam parameter groups + LR rule (:166-193), ``WarmupLinearSchedule`` (:45-58),
step-0 data-dependent init (:547-592), the encode/decode/backward/restore order of the hook
(:598-682), optimizer/scheduler order (:791-802) and fired-latent bookkeeping
(:622-628, 806-816) -- with the model forward replaced by an iterator of cached activations
``{hookpoint: (inputs[N, d], outputs[N, d])}`` (BASELINE.json north_star)."""
from __future__ import annotations

from dataclasses import replace
from typing import Optional

import torch
from torch import Tensor

from . import ckpt, fused_state
from .adam import Adam
from .config import TrainConfig
from .engine import ROUTES
from .e2e import EndToEnd, TokenBatches
from .fit_loop import FitLoop, WarmupLinearSchedule  # noqa: F401 (re-exported)
from .runner import CrossLayerRunner
from .sparse_coder import SparseCoder


class Trainer(EndToEnd, FitLoop):
    """``Trainer(cfg, dataset, model=None, mesh=None)``.

    With ``model=None``: ``cfg.hookpoints`` names the layers in network order; ``dataset`` is a sized iterable of
    steps, each a dict ``hookpoint -> (inputs, outputs)`` of cached activations already on
    the GPU (and optionally ``"loss_mask" -> bool[N]``); ``input_widths`` maps hookpoints to
    d_model (inferred from the first batch when omitted).
    With a frozen HF ``model``: ``dataset`` yields ``{"input_ids": ...}`` rows as in the reference and the step dicts
    come from forward hooks on the model (``activations.ModelActivations``)."""

    def __init__(self, cfg: TrainConfig, dataset, model=None, mesh=None, *, device="cuda",
                 input_widths: Optional[dict] = None, num_batches: Optional[int] = None):
        if mesh is not None:
            raise NotImplementedError("use clt_training_amd.tp for latent sharding over RCCL")
        # loss_fn != "fvu" (trainer.py:739-785): the coders run inside the model's forward (e2e.EndToEnd); the dataset
        # then stays a stream of token batches
        e2e = cfg.loss_fn != "fvu"
        if e2e and model is None:
            raise ValueError("the end-to-end losses need the model: Trainer(cfg, token_dataset, model)")
        if model is not None:
            tokens = dataset
            dataset, input_widths, num_batches, device = self.attach_model(cfg, dataset, model)
            if e2e:
                dataset = TokenBatches(tokens, cfg)
        assert cfg.hookpoints, "cached-activation training needs explicit cfg.hookpoints"
        self.check_supported(cfg, e2e_ok=True)
        cfg.hookpoints = cfg.hookpoints[:: cfg.layer_stride]
        self.cfg = cfg
        self.dataset = dataset
        self.model = model
        self.mesh = None
        self.device = torch.device(device)
        if input_widths is None:
            first = next(iter(dataset))
            input_widths = {h: first[h][0].shape[-1] for h in cfg.hookpoints}

        # ---- SAE construction (trainer.py:113-146)
        self.saes = {}
        sources = []
        for position, hook in enumerate(cfg.hookpoints):
            for seed in cfg.init_seeds:
                torch.manual_seed(seed)
                name = f"{hook}/seed{seed}" if len(cfg.init_seeds) > 1 else hook
                if cfg.cross_layer > 0:
                    n_targets = min(cfg.cross_layer, len(cfg.hookpoints) - position)
                    sources.append(position + n_targets)
                    while sources[0] < position:
                        sources.pop(0)
                    n_sources = len(sources)
                else:
                    n_targets = n_sources = 0
                sae_cfg = replace(cfg.sae, n_targets=n_targets, n_sources=n_sources)
                self.saes[name] = SparseCoder(input_widths[hook], sae_cfg, self.device, dtype=torch.float32)

        if num_batches is None:
            num_batches = len(dataset)
        # ---- optimizer (trainer.py:151-254)
        # (optimizer="muon": the reference orders every coder's parameters by NAME, trainer.py:195-199 -- the order the
        # positions in its optimizer_{0,1}.pt refer to)
        by_name = cfg.optimizer == "muon"
        self.optimizers, self.lr_schedulers = self.build_optimizers(
            cfg, [[p for _, p in sorted(sae.named_parameters())] if by_name else list(sae.parameters())
                  for sae in self.saes.values()],
            [sae.num_latents for sae in self.saes.values()], num_batches)
        # Row gradients of the [M, d] matrices go straight into the Adam update inside the gradient kernels when each
        # optimizer step sees one micro-step (fused_state): no dense .grad, no second pass over the tables.
        self._fused = (cfg.grad_acc_steps * cfg.micro_acc_steps == 1 and ROUTES.fused_adam and not e2e
                       and type(self.optimizers[0]) is Adam
                       # a decoder shared by several targets is read again after its first local backward
                       and not cfg.sae.per_source_tied)
        if self._fused:
            for sae in self.saes.values():
                if sae.cfg.normalize_decoder and not cfg.sae.transcode:
                    fused_state.set_project(sae.W_dec)

        self.global_step = 0
        self.num_tokens_since_fired = {
            name: torch.zeros(sae.num_latents, device=self.device, dtype=torch.long) for name, sae in self.saes.items()}
        self.did_fire = {name: torch.zeros(sae.num_latents, device=self.device, dtype=torch.bool)
                         for name, sae in self.saes.items()}
        self.runner = CrossLayerRunner()
        self.last_fvu = {}
        self.num_tokens_in_step = 0
        self.init_k_schedule()

    # ------------------------------------------------------------------ step-0 init (trainer.py:547-592)
    @torch.no_grad()
    def _first_step_init(self, raw: SparseCoder, inputs: Tensor, outputs: Tensor):
        if self.cfg.sae.transcode:
            mean = -inputs.float().mean(0).to(raw.dtype)
            raw.encoder.bias.data[:] = torch.nn.functional.linear(mean, raw.encoder.weight.data)
        mean = outputs.float().mean(0)
        if not hasattr(raw, "b_decs"):
            raw.b_dec.data[:] = mean.to(raw.b_dec.dtype)
        else:
            raw.b_decs[0].data[:] = mean.to(raw.b_decs[0].dtype)
        if raw.cfg.normalize_io:
            in_norm = inputs.float().norm(dim=-1).mean()
            out_norm = outputs.float().norm(dim=-1).mean()
            raw.in_norm.data[:] = in_norm
            raw.out_norm.data[:] = out_norm
            for b_dec in raw.b_decs if hasattr(raw, "b_decs") else [raw.b_dec]:
                b_dec.data[:] = b_dec.data * ((b_dec.shape[-1] ** 0.5) / out_norm)
            raw.encoder.bias.data[:] = raw.encoder.bias.data * ((raw.encoder.weight.shape[-1] ** 0.5) / in_norm)

    # ------------------------------------------------------------------ the hook body (trainer.py:492-682)
    def hook(self, name: str, inputs: Tensor, outputs: Tensor, loss_mask: Optional[Tensor], acc_steps: int):
        raw = self.saes[name]
        if not self.cfg.sae.transcode:
            inputs = outputs
        if self.global_step == 0 and not self.cfg.finetune:
            self._first_step_init(raw, inputs, outputs)
        if raw.cfg.normalize_decoder and not self.cfg.sae.transcode:
            raw.set_decoder_norm_to_unit_norm()
        encoding = self.runner.encode(
            inputs, sparse_coder=raw,
            dead_mask=self.num_tokens_since_fired[name] > self.cfg.dead_feature_threshold)
        out = self.runner.decode(encoding, outputs, module_name=name, detach_grad=True, loss_mask=loss_mask)
        self.did_fire[name].index_fill_(0, encoding.latent_indices.flatten(), True)  # (indexed assignment syncs on HIP)
        self.last_fvu[name] = out.fvu.detach()
        loss = out.fvu / acc_steps
        loss.backward()
        del loss
        self.runner.restore()

    def step(self, batch: dict):
        """One pass over all hookpoints in network order + optimizer step (trainer.py:684-816)."""
        acc_steps = self.cfg.grad_acc_steps * self.cfg.micro_acc_steps
        fused_state.ACTIVE = self.optimizers[0] if self._fused else None
        fused_state.begin_step()
        try:
            return self._step(batch, acc_steps)
        finally:
            fused_state.end_step()

    def _step(self, batch: dict, acc_steps: int):
        if "input_ids" in batch:  # end-to-end losses: the hooks run inside the model's forward (e2e.EndToEnd)
            return self._step_e2e(batch["input_ids"])
        self.runner.reset()
        loss_mask = batch.get("loss_mask")
        n_tok = 0
        for name in self.cfg.hookpoints:
            x, y = batch[name]
            n_tok = x.shape[0]
            self.hook(name, x, y, loss_mask, acc_steps)
        return self._finish_step(n_tok)

    def _finish_step(self, n_tok: int):
        """Optimizer / scheduler / bookkeeping tail of a step (trainer.py:787-816, 858)."""
        self.num_tokens_in_step += n_tok
        step, substep = divmod(self.global_step + 1, self.cfg.grad_acc_steps)
        if substep == 0:
            if self.cfg.sae.normalize_decoder and not self.cfg.sae.transcode:
                for sae in self.saes.values():
                    sae.remove_gradient_parallel_to_decoder_directions()
            for optimizer in self.optimizers:
                optimizer.step()
                optimizer.zero_grad()
            for scheduler in self.lr_schedulers:
                scheduler.step()
            self.set_correct_k()  # trainer.py:804 (before global_step advances, :858)
            with torch.no_grad():
                for name, counts in self.num_tokens_since_fired.items():
                    counts += self.num_tokens_in_step
                    counts.masked_fill_(self.did_fire[name], 0)  # no host sync, unlike counts[mask] = 0
                self.num_tokens_in_step = 0
                for mask in self.did_fire.values():
                    mask.zero_()
        self.global_step += 1
        return self.last_fvu

    # ------------------------------------------------------------------ checkpoint (trainer.py:910-970, 275-354)
    def save(self, path=None, only=None):
        """``{path}/{hookpoint}/sae.safetensors + cfg.json``, ``optimizer_0.pt`` (flattened tuple keys, utils.py:193-277),
        ``lr_scheduler_0.pt``, ``state.pt``, ``config.json`` and ``rank_0_state.pt`` as the reference lays them out: a run
        saved here resumes in the reference and a run saved by the reference resumes here (tests/test_trainer_golden.py
        loads a checkpoint the reference wrote)."""
        from pathlib import Path

        path = Path(path or self.default_path())
        path.mkdir(parents=True, exist_ok=True)
        for name, sae in self.saes.items():
            if only is None or name in only:
                sae.save_to_disk(f"{path}/{name}")
        # optimizer_0.pt in the reference's flattened layout (utils.py:193-277; trainer.py:931-937)
        for i, opt in enumerate(self.optimizers):  # (optimizer="muon": Muon for the matrices, Adam for the vectors)
            if hasattr(opt, "_step"):
                flat = ckpt.to_reference_layout(opt.state_dict(), opt._step)
            else:
                flat = ckpt.flatten_state(opt.state_dict())
            torch.save(flat, path / f"optimizer_{i}.pt")
        for i, sch in enumerate(self.lr_schedulers):
            torch.save(sch.state_dict(), path / f"lr_scheduler_{i}.pt")
        torch.save({"global_step": self.global_step}, path / "state.pt")
        self.save_config(path)
        torch.save({"num_tokens_since_fired": {h: t.cpu() for h, t in self.num_tokens_since_fired.items()},
                    "best_loss": self.best_loss}, path / "rank_0_state.pt")

    def load_state(self, path):
        """trainer.py:275-354: resume from ``save(path)``."""
        from pathlib import Path

        path = Path(path)
        self.global_step = torch.load(path / "state.pt", weights_only=True)["global_step"]
        rank_state = torch.load(path / "rank_0_state.pt", map_location=self.device, weights_only=True)
        for h, t in rank_state["num_tokens_since_fired"].items():
            if h in self.num_tokens_since_fired:
                self.num_tokens_since_fired[h].copy_(t)
        saved_best = rank_state.get("best_loss", {})
        if not isinstance(saved_best, dict):  # one scalar for the end-to-end losses (trainer.py:294-295)
            self.best_loss = saved_best
        elif isinstance(self.best_loss, dict):
            for h, v in saved_best.items():
                if h in self.best_loss:
                    self.best_loss[h] = v
        for i, sch in enumerate(self.lr_schedulers):
            sch.load_state_dict(torch.load(path / f"lr_scheduler_{i}.pt", weights_only=True))
        for i, opt in enumerate(self.optimizers):
            nested, step = ckpt.from_file_layout(torch.load(path / f"optimizer_{i}.pt", map_location=self.device,
                                                            weights_only=True))
            opt.load_state_dict(nested)  # (written by the reference or by save(): the same layout)
            if step is not None and hasattr(opt, "_step"):
                opt._step = step
        for name, sae in self.saes.items():
            if (path / name / "sae.safetensors").exists():
                sae.load_state(f"{path}/{name}")
            else:
                print(f"No checkpoint found for {name} at {path}, skipping")
        return self


SaeTrainer = Trainer
