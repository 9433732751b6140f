"""The parts of the reference ``Trainer`` around the step that both trainers here share (sparsify/trainer.py):
attaching a frozen HF model (:73-111), the k schedule (:263-265, 356-374, 804), and ``fit`` with its logging keys,
checkpoint cadence and resume offset (:376-443, 818-861)."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Callable, Optional

import torch
import torch.distributed as dist


class WarmupLinearSchedule(torch.optim.lr_scheduler.LambdaLR):
    """trainer.py:45-58."""

    def __init__(self, optimizer, warmup_steps, num_training_steps):
        self.warmup_steps = warmup_steps
        self.num_training_steps = num_training_steps
        super().__init__(optimizer, self.lr_lambda, last_epoch=-1)

    def lr_lambda(self, current_step):
        if current_step < self.warmup_steps:
            return float(current_step) / float(max(1, self.warmup_steps))
        return max(0.0, float(self.num_training_steps - current_step)
                   / float(max(1, self.num_training_steps - self.warmup_steps)))


class FitLoop:
    # set by the trainers: cfg, dataset, model, saes, num_tokens_since_fired, global_step, step(), save()
    wandb_log: Optional[Callable[[dict, int], None]] = None  # ``wandb.log``-shaped sink for the reference's keys

    # ------------------------------------------------------------------ model attachment (trainer.py:73-111, 149)
    @staticmethod
    def attach_model(cfg, dataset, model):
        """-> (dataset of step dicts, input_widths, num_batches, device) for a frozen HF ``model``."""
        from .activations import ModelActivations, resolve_hookpoints, resolve_widths

        resolve_hookpoints(cfg, model)
        cfg.hookpoints = cfg.hookpoints[:: cfg.layer_stride]
        cfg.layer_stride = 1  # applied
        widths = resolve_widths(model, cfg.hookpoints)
        acts = ModelActivations(model, dataset, cfg)
        return acts, widths, len(acts), model.device

    @staticmethod
    def check_supported(cfg, e2e_ok: bool = False):
        """Configurations the hot path does not implement fail loudly instead of training something else."""
        if cfg.loss_fn != "fvu" and not e2e_ok:  # trainer.py:739-780: the coders run inside the model's forward
            raise NotImplementedError("loss_fn != 'fvu' (end-to-end losses) runs on trainer.Trainer with a model, "
                                      "not on the latent-sharded engine")
        if cfg.loss_fn not in ("fvu", "ce", "kl", "kl-fvu"):
            raise ValueError(f"Unknown loss function '{cfg.loss_fn}'")
        if cfg.dead_latent_penalty > 0.0:  # trainer.py:653-676 (DeadLatentLoss)
            raise NotImplementedError("dead_latent_penalty > 0 (DeadLatentLoss) is not built")
        if len(cfg.init_seeds) != 1:  # (the reference's hook looks SAEs up by module name only, trainer.py:513-515, 544)
            raise NotImplementedError("one init seed per run")

    # ------------------------------------------------------------------ optimizers (trainer.py:151-254)
    @staticmethod
    def build_optimizers(cfg, param_lists, num_latents, num_batches, adam_cls=None, signsgd_cls=None, muon_kw=None):
        """One parameter group per sparse coder with the reference's LR rule; -> (optimizers, lr_schedulers)."""
        if cfg.optimizer == "muon":  # trainer.py:194-222: Muon for the matrices, Adam for the vectors, two schedules
            from .muon import Muon

            if adam_cls is None:
                from .adam import Adam as adam_cls
            params = [p for ps in param_lists for p in ps]
            mats = [p for p in params if p.ndim >= 2]
            lr = cfg.lr or 2e-3  # "Muon LR is independent of the number of latents"
            opts = [Muon(mats, lr=lr, **(muon_kw or {})),
                    adam_cls([dict(params=[p for p in params if p.ndim < 2], lr=torch.tensor(lr))], betas=(cfg.b1, cfg.b2))]
            return opts, [WarmupLinearSchedule(opts[0], 0, num_batches),
                          WarmupLinearSchedule(opts[1], cfg.lr_warmup_steps, num_batches)]
        if cfg.optimizer in ("adam", "adam8"):
            if cfg.optimizer == "adam8":  # trainer.py:155-164: torchao's AdamW8bit when importable, else plain Adam
                print("torchao 8-bit Adam not available, using Adam")
            if adam_cls is None:
                from .adam import Adam as adam_cls
            pgs = [dict(params=ps, lr=torch.tensor(cfg.lr or 2e-4 / (m / (2**14)) ** 0.5))  # trainer.py:166-173
                   for ps, m in zip(param_lists, num_latents)]
            opt = adam_cls(pgs, betas=(cfg.b1, cfg.b2))
        elif cfg.optimizer == "signum":
            if not cfg.force_lr_warmup:
                # trainer.py:234-242 wraps SignSGD in schedulefree.ScheduleFreeWrapper(momentum=0.95); that package is
                # a third-party dependency absent from the reference tree and from this image: nothing to pin it on.
                raise NotImplementedError("optimizer='signum' needs force_lr_warmup=True here (plain SignSGD + linear "
                                          "warm-up, trainer.py:239-250); the ScheduleFree wrapper is not built")
            if signsgd_cls is None:
                from .adam import SignSGD as signsgd_cls
            pgs = [dict(params=ps, lr=cfg.lr or 5e-3 / (m / (2**14)) ** 0.5)  # trainer.py:225-231
                   for ps, m in zip(param_lists, num_latents)]
            opt = signsgd_cls(pgs)
        else:
            raise NotImplementedError(f"optimizer {cfg.optimizer!r} is not on the MI355X hot path")
        # get_linear_schedule_with_warmup (trainer.py:246-250) is the same lambda as WarmupLinearSchedule (:45-58)
        return [opt], [WarmupLinearSchedule(opt, cfg.lr_warmup_steps, num_batches)]

    # ------------------------------------------------------------------ k schedule
    def init_k_schedule(self):
        num_latents = next(iter(self.saes.values())).num_latents
        self.initial_k = min(num_latents, self.cfg.sae.k * self.cfg.k_anneal_mul)  # trainer.py:263-265
        self.final_k = self.cfg.sae.k
        # trainer.py:267-271: one running best per hookpoint for the local loss, ONE scalar for the end-to-end losses
        self.best_loss = ({name: float("inf") for name in self.cfg.hookpoints} if self.cfg.loss_fn == "fvu"
                          else float("inf"))

    def get_current_k(self) -> int:
        """trainer.py:356-362: linear decay from ``initial_k`` to ``final_k`` over ``k_decay_steps``."""
        if self.global_step >= self.cfg.k_decay_steps:
            return self.final_k
        progress = self.global_step / self.cfg.k_decay_steps
        return round(self.initial_k * (1 - progress) + self.final_k * progress)

    def set_correct_k(self):
        """trainer.py:364-374 (called after each optimizer step, :804)."""
        if self.cfg.per_layer_k:
            assert len(self.cfg.per_layer_k) == len(self.saes), "Must specify a k for each layer"
            for i, sae in enumerate(self.saes.values()):
                sae.cfg.k = self.cfg.per_layer_k[i]
        else:
            k = self.get_current_k()
            for sae in self.saes.values():
                sae.cfg.k = k

    # ------------------------------------------------------------------ fit (trainer.py:376-443, 684, 818-861)
    def default_path(self) -> str:
        return f'{self.cfg.save_dir}/{self.cfg.run_name or "unnamed"}'  # trainer.py:968

    def _dead_fraction(self, name: str) -> float:
        mask = self.num_tokens_since_fired[name] > self.cfg.dead_feature_threshold
        frac = mask.mean(dtype=torch.float32)
        comm = getattr(self, "comm", None)
        if comm is not None and comm.world > 1:  # equal latent shards: the mean of the shard fractions
            frac = comm.all_reduce_(frac.clone()) / comm.world
        return frac.item()

    def fit(self, log: Optional[Callable[[int, dict], None]] = None):
        torch.set_float32_matmul_precision("high")  # trainer.py:378
        cfg = self.cfg
        rank_zero = not dist.is_initialized() or dist.get_rank() == 0
        if cfg.log_to_wandb and self.wandb_log is None:  # trainer.py:386-400
            import importlib.util

            if importlib.util.find_spec("wandb") is None:  # the same answer on every rank
                if rank_zero:
                    print("Weights & Biases not available, skipping logging.")
                cfg.log_to_wandb = False
            elif rank_zero:
                import wandb

                wandb.init(name=cfg.run_name, project="sparsify", config=cfg.to_dict(), save_code=True)
                self.wandb_log = lambda info, step: wandb.log(info, step=step)
        # Under tp the dead-latent fraction is a collective: every rank must take the logging branch together, so a
        # hand-set ``wandb_log`` sink goes on every rank or on none.
        do_log = cfg.log_to_wandb or self.wandb_log is not None
        dataset = self.dataset
        if self.global_step > 0 and not cfg.restart_epoch:  # trainer.py:410-417 (the reference asserts ``select``)
            if hasattr(dataset, "skip"):
                dataset = dataset.skip(self.global_step)
            elif hasattr(dataset, "__getitem__") and hasattr(dataset, "__len__"):
                src, first = dataset, self.global_step  # lazy, like the reference's select(range(...)): nothing is
                dataset = (src[i] for i in range(first, len(src)))  # loaded before its step
            else:
                import itertools

                dataset = itertools.islice(iter(dataset), self.global_step, None)  # plain iterables of step dicts
        denom = cfg.grad_acc_steps * cfg.micro_acc_steps * cfg.wandb_log_frequency
        avg_fvu = {}
        avg_e2e = {}  # "ce" | "kl" + "acc_top1": the end-to-end losses' running means (trainer.py:448-450, 744, 777-778)
        e2e_key = {"ce": "ce", "kl": "kl", "kl-fvu": "kl"}.get(cfg.loss_fn)

        def running_loss():  # what save_best ranks by (trainer.py:451-455, 746, 779, 783)
            if e2e_key is None:
                return {n: float(v) for n, v in avg_fvu.items()}
            return float(avg_e2e[e2e_key]) if e2e_key in avg_e2e else None

        avg_losses = None
        for batch in dataset:
            fvus = self.step(batch)
            for name, f in fvus.items():  # stays on the device: no host sync per hook (the reference's float(), :648)
                avg_fvu[name] = avg_fvu.get(name, 0.0) + f.detach() / denom
            if e2e_key is not None:
                for key, v in (getattr(self, "last_e2e", None) or {}).items():
                    avg_e2e[key] = avg_e2e.get(key, 0.0) + v.detach() / denom
            step, substep = divmod(self.global_step, cfg.grad_acc_steps)  # step() has advanced global_step (:791, 858)
            if substep == 0:
                saving = (step + 1) % cfg.save_every == 0
                if cfg.save_best and saving:
                    avg_losses = running_loss()  # (a host sync: only on the steps that use it)
                if saving:  # trainer.py:818-823
                    self.save()
                    if cfg.save_best and avg_losses is not None:
                        self.save_best(avg_losses)
                if do_log and (step + 1) % cfg.wandb_log_frequency == 0:
                    info = {}  # trainer.py:829-851
                    if cfg.loss_fn == "ce":
                        info["ce_loss"] = float(avg_e2e.get("ce", 0.0))
                    elif e2e_key == "kl":
                        info["kl_loss"] = float(avg_e2e.get("kl", 0.0))
                        info["acc_top1"] = float(avg_e2e.get("acc_top1", 0.0))
                    for name in self.saes:
                        info[f"dead_pct/{name}"] = self._dead_fraction(name)  # collective under tp: every rank calls
                        if "fvu" in cfg.loss_fn:  # (:844-845: "fvu" and "kl-fvu")
                            info[f"fvu/{name}"] = float(avg_fvu[name])
                    info["k"] = self.get_current_k()
                    if self.wandb_log is not None:
                        self.wandb_log(info, step)
                if cfg.save_best and not saving:
                    avg_losses = running_loss()  # the reference's avg_losses keeps the last batch's running value
                avg_fvu.clear()
                avg_e2e.clear()
            if log is not None:
                log(self.global_step, fvus)
        self.save()  # trainer.py:861
        if cfg.save_best:  # trainer.py:862-863
            if avg_fvu or avg_e2e:
                avg_losses = running_loss()
            if avg_losses is not None and avg_losses != {}:
                self.save_best(avg_losses)
        return self

    def save_best(self, avg_loss: dict):
        """trainer.py:972-993: one checkpoint per hookpoint whose running loss is the lowest so far."""
        base = f"{self.default_path()}/best"
        if not isinstance(avg_loss, dict):  # the end-to-end losses: one scalar, all coders under best/ (:984-989)
            if avg_loss < self.best_loss:
                self.best_loss = avg_loss
                self.save(base)
            return
        for name in self.saes:
            if avg_loss.get(name, float("inf")) < self.best_loss[name]:
                self.best_loss[name] = avg_loss[name]
                self.save(f"{base}/{name}", only=[name])

    def save_config(self, path):
        Path(path).mkdir(parents=True, exist_ok=True)
        with open(Path(path) / "config.json", "w") as f:  # trainer.py:949
            json.dump(self.cfg.to_dict(), f)
