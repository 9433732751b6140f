"""End-to-end losses (sparsify/trainer.py:492-682, 684-785): ``loss_fn`` in
``"ce" | "kl" | "kl-fvu"``.  The sparse coders run INSIDE the forward hooks of the frozen model and their output
REPLACES the module output, so the loss -- cross entropy of the patched model, or the KL divergence from the clean
model's next-token distribution -- back-propagates through the later layers of the model into every coder.

Same kernels as the local-FVU path (fused encode + top-k, sparse decode, sparse encoder / decoder gradients through the
C ABI); what differs is host logic: nothing is detached (``detach_grad=False``, trainer.py:609), the FVU of each hook is
computed without a loss mask (:610) and only logged, the gradient reaches a coder through the model's own autograd graph,
and the optimizer update cannot be applied inside the gradient kernels (a table's gradient is complete only when the
model's backward has passed its hook), so the plain ``.grad`` -> ``Adam.step`` route is used.

``"kl-fvu"`` follows trainer.py:712-730, 755-778 -- a first, clean pass whose hooks train on the local FVU, then the
patched pass whose KL term is weighted by ``kl_coeff`` (0: by fvu / kl) -- but is PARITY UNPINNED: the reference wraps
``kl`` in ``DTensor.from_local(kl, self.mesh, ...)`` unconditionally (:764-769), which raises without a device mesh, so
no run of the reference could be recorded in the build container (tests/golden/make_golden.py, ``e2e``)."""
from __future__ import annotations

from functools import partial

import torch
from torch.utils.data import DataLoader


def _first(t):
    return t[0] if isinstance(t, tuple) else t


class TokenBatches:
    """Sized iterable of ``{"input_ids": LongTensor[B, T]}`` (trainer.py:422-428: unshuffled, ``cfg.batch_size`` rows)."""

    def __init__(self, dataset, cfg):
        self.dataset, self.cfg = dataset, cfg

    def __len__(self):
        return len(self.dataset) // self.cfg.batch_size  # trainer.py:149

    def skip(self, n_batches: int):
        n = n_batches * self.cfg.batch_size
        assert hasattr(self.dataset, "select"), "Dataset must implement `select`"  # trainer.py:410-417
        self.dataset = self.dataset.select(range(n, len(self.dataset)))
        return self

    def __iter__(self):
        for batch in DataLoader(self.dataset, batch_size=self.cfg.batch_size, shuffle=False):
            yield {"input_ids": batch["input_ids"]}


class EndToEnd:
    """Mixin of ``trainer.Trainer``: ``_step_e2e(input_ids)``.  Uses the trainer's ``saes``, ``runner``, ``model``,
    ``_first_step_init``, ``did_fire``, ``last_fvu`` and ``_finish_step``."""

    def _e2e_modules(self):
        cfg = self.cfg
        if getattr(self, "_e2e_mods", None) is None:
            self._e2e_mods = {name: self.model.base_model.get_submodule(name)
                              for name in list(cfg.hookpoints) + list(cfg.hookpoints_in)}
        return self._e2e_mods

    def _step_e2e(self, x: torch.Tensor):
        cfg, model = self.cfg, self.model
        acc_steps = cfg.grad_acc_steps * cfg.micro_acc_steps
        model.requires_grad_(False)  # trainer.py:380-382
        model.eval()
        x = x.to(model.device)
        bos_mask = x == model.config.bos_token_id  # trainer.py:686-690
        if not cfg.filter_bos:
            bos_mask = bos_mask & False
        if cfg.remove_first_token:
            bos_mask[:, 0] = True
        self.runner.reset()
        name_to_module = self._e2e_modules()
        module_to_name = {m: n for n, m in name_to_module.items()}
        hook_set = set(cfg.hookpoints)
        cached_inputs, fvu_losses = {}, {}

        def record_inputs(module, inputs, outputs):  # trainer.py:482-485
            cached_inputs[module_to_name[module]] = _first(inputs)

        def hook(module, inputs, outputs, force_loss_fn=None):
            loss_fn = force_loss_fn or cfg.loss_fn
            aux_out = None
            inputs = _first(inputs)
            if isinstance(outputs, tuple):
                outputs, *aux_out = outputs
            name = module_to_name[module]
            layer_idx = cfg.hookpoints.index(name)
            if layer_idx < len(cfg.hookpoints_in):  # trainer.py:509-511
                inputs = cached_inputs.pop(cfg.hookpoints_in[layer_idx])
            out_shape, outputs_original = outputs.shape, outputs
            outputs = outputs.flatten(0, 1)
            inputs = inputs.flatten(0, 1) if cfg.sae.transcode else outputs
            raw = self.saes[name]
            if self.global_step == 0 and not cfg.finetune:
                self._first_step_init(raw, inputs.detach(), outputs.detach())
            if raw.cfg.normalize_decoder and not cfg.sae.transcode:
                raw.set_decoder_norm_to_unit_norm()
            local = loss_fn == "fvu"
            encoding = self.runner.encode(
                inputs, sparse_coder=raw,
                dead_mask=self.num_tokens_since_fired[name] > cfg.dead_feature_threshold)
            out = self.runner.decode(encoding, outputs, module_name=name, detach_grad=local,
                                     loss_mask=(~bos_mask).flatten(0, 1) if local else None)
            if cfg.loss_fn == "kl-fvu":
                fvu_losses[name] = float(out.fvu.detach())
            self.did_fire[name].index_fill_(0, encoding.latent_indices.flatten(), True)
            self.last_fvu[name] = out.fvu.detach()
            if not local:  # the coder's output takes the module output's place (trainer.py:630-646)
                output = out.sae_out.reshape(out_shape).type_as(outputs)
                if cfg.filter_bos:
                    output = torch.where(bos_mask[..., None], outputs_original, output)
                return (output, *aux_out) if aux_out is not None else output
            (out.fvu / acc_steps).backward()  # the clean pass of "kl-fvu": a local step on the FVU (:676-682)
            self.runner.restore()
            return None

        def attach(force):
            return [mod.register_forward_hook(partial(hook, force_loss_fn=force) if name in hook_set else record_inputs)
                    for name, mod in name_to_module.items()]

        info = {}
        clean_logits = clean_probs = None
        if cfg.loss_fn in ("kl", "kl-fvu"):  # trainer.py:712-737: the clean model's distribution
            handles = attach("fvu") if cfg.loss_fn == "kl-fvu" else []
            try:
                clean_logits = model(x).logits
            finally:
                for h in handles:
                    h.remove()
            clean_probs = clean_logits.softmax(dim=-1)
        handles = attach(None if cfg.loss_fn != "kl-fvu" else "kl")
        try:
            if cfg.loss_fn == "ce":  # trainer.py:741-747
                ce = model(x, labels=x).loss
                (ce / acc_steps).backward()
                info["ce"] = ce.detach()
            else:  # trainer.py:748-778
                dirty_lps = model(x).logits.log_softmax(dim=-1)
                kl = torch.sum(clean_probs * (clean_logits.log_softmax(dim=-1) - dirty_lps), dim=-1).mean()
                info["acc_top1"] = (clean_logits.argmax(dim=-1) == dirty_lps.argmax(dim=-1)).float().mean()
                loss = kl
                if cfg.loss_fn == "kl-fvu":
                    fvu_loss = sum(fvu_losses.values()) / len(fvu_losses)
                    kl_coeff = cfg.kl_coeff
                    if kl_coeff == 0.0:
                        kl_coeff = (fvu_loss / kl).detach()
                    loss = kl * kl_coeff + fvu_loss
                (loss / acc_steps).backward()
                info["kl"] = kl.detach()
        finally:
            for h in handles:
                h.remove()
        self.last_e2e = info
        self._finish_step(x.numel())
        return self.last_fvu
