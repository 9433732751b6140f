"""Real-activation driver: a frozen HF model + a token dataset become the step dicts
``{hookpoint: (inputs[N, d], outputs[N, d]), "loss_mask": bool[N]}`` that ``Trainer`` / ``TPTrainer`` consume.

The reference runs the SAE work INSIDE the forward hooks (sparsify/trainer.py:492-682) because its end-to-end losses
replace module outputs on the fly.  For the local FVU loss -- the only loss on this hot path -- the hook's result never
feeds back into the model, so here the hooks only record (one model forward per step, no graph), and the train step then
runs over the recorded layers in network order: the same arithmetic, and the step the benchmark times.

What is kept from the reference: hookpoint resolution with wildcards / ``layers`` / ``layer_stride``
(trainer.py:73-94, utils.py:97-107), ``resolve_widths`` on the model's dummy inputs (utils.py:110-142), tuple
unpacking of module inputs / outputs (:500-504), ``hookpoints_in`` overriding the input of the i-th hookpoint
(:482-485, 509-511), flattening of batch and sequence (:522-524), the BOS / first-token mask (:685-690) handed on as
``loss_mask = ~bos_mask`` (:609), and an unshuffled ``DataLoader`` of ``cfg.batch_size`` sequences (:422-428)."""
from __future__ import annotations

import re
from fnmatch import fnmatchcase
from typing import Optional

import torch
from torch import nn
from torch.utils.data import DataLoader

from .config import TrainConfig


def natural_key(s: str):
    """Sort key with digit runs compared as integers ('h.10' after 'h.9'): what ``natsorted`` gives for module paths."""
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]


def get_layer_list(model) -> tuple:
    """utils.py:97-107: the one ModuleList whose length is ``config.num_hidden_layers``."""
    N = int(model.config.num_hidden_layers)
    candidates = [(name, mod) for name, mod in model.base_model.named_modules()
                  if isinstance(mod, nn.ModuleList) and len(mod) == N]
    assert len(candidates) == 1, "Could not find the list of layers."
    return candidates[0]


def resolve_hookpoints(cfg: TrainConfig, model) -> list:
    """trainer.py:73-92 (``layer_stride`` is applied by the trainer, :94)."""
    if cfg.hookpoints:
        assert not cfg.layers, "Cannot specify both `hookpoints` and `layers`."
        raw = [name for name, _ in model.base_model.named_modules()
               if any(fnmatchcase(name, pat) for pat in cfg.hookpoints)]
        cfg.hookpoints = sorted(raw, key=natural_key)
    else:
        if not cfg.layers:
            cfg.layers = list(range(int(model.config.num_hidden_layers)))
        layers_name, _ = get_layer_list(model)
        cfg.hookpoints = [f"{layers_name}.{i}" for i in cfg.layers]
    return cfg.hookpoints


def _first(t):
    return t[0] if isinstance(t, tuple) else t


@torch.no_grad()
def resolve_widths(model, module_names: list, dim: int = -1) -> dict:
    """utils.py:110-142: output width of each named module, from one forward over ``model.dummy_inputs``."""
    module_to_name = {model.base_model.get_submodule(name): name for name in module_names}
    shapes = {}

    def hook(module, _, output):
        shapes[module_to_name[module]] = _first(output).shape[dim]

    handles = [mod.register_forward_hook(hook) for mod in module_to_name]
    try:
        model(**{k: v.to(model.device) for k, v in model.dummy_inputs.items()})
    finally:
        for h in handles:
            h.remove()
    return shapes


class ModelActivations:
    """Sized iterable of step dicts.  ``dataset`` yields ``{"input_ids": LongTensor[T]}`` rows (an HF dataset in torch
    format, a ``MemmapDataset``, or a list of such dicts); ``dtype`` optionally casts the recorded activations (the
    trainers take them as they come: fp32 activations go through the bf16 autocast rule of sparse_coder.py:663-685)."""

    def __init__(self, model, dataset, cfg: TrainConfig, *, dtype: Optional[torch.dtype] = None):
        self.model, self.dataset, self.cfg, self.dtype = model, dataset, cfg, dtype
        assert len(cfg.hookpoints_in) <= len(cfg.hookpoints)
        self.name_to_module = {name: model.base_model.get_submodule(name)
                               for name in list(cfg.hookpoints) + list(cfg.hookpoints_in)}

    def __len__(self):
        return len(self.dataset) // self.cfg.batch_size  # trainer.py:149 (the loader still yields a last short batch)

    def skip(self, n_batches: int):
        """Resume: drop the sequences the first ``n_batches`` steps consumed (trainer.py:410-417)."""
        n = n_batches * self.cfg.batch_size
        assert hasattr(self.dataset, "select"), "Dataset must implement `select`"
        self.dataset = self.dataset.select(range(n, len(self.dataset)))
        return self

    @torch.no_grad()
    def __iter__(self):
        cfg, model = self.cfg, self.model
        model.requires_grad_(False)  # trainer.py:380-382
        model.eval()
        module_to_name = {m: n for n, m in self.name_to_module.items()}
        hook_set, in_set = set(cfg.hookpoints), set(cfg.hookpoints_in)
        for batch in DataLoader(self.dataset, batch_size=cfg.batch_size, shuffle=False):
            x = batch["input_ids"].to(model.device)
            bos_mask = x == model.config.bos_token_id  # trainer.py:686-690
            if not cfg.filter_bos:
                bos_mask &= False
            if cfg.remove_first_token:
                bos_mask[:, 0] = True
            rec_in, rec_out = {}, {}

            def record(module, inputs, outputs):
                name = module_to_name[module]
                if name in in_set:  # record_inputs (trainer.py:482-485)
                    rec_in[name] = _first(inputs)
                if name in hook_set:
                    rec_out[name] = (_first(inputs), _first(outputs))

            handles = [mod.register_forward_hook(record) for mod in self.name_to_module.values()]
            try:
                model(x)  # trainer.py:781-782
            finally:
                for h in handles:
                    h.remove()
            step = {}
            for i, name in enumerate(cfg.hookpoints):
                inputs, outputs = rec_out[name]
                if i < len(cfg.hookpoints_in):  # trainer.py:509-511
                    inputs = rec_in[cfg.hookpoints_in[i]]
                outputs = outputs.flatten(0, 1)
                inputs = inputs.flatten(0, 1) if cfg.sae.transcode else outputs
                if self.dtype is not None:
                    inputs, outputs = inputs.to(self.dtype), outputs.to(self.dtype)
                step[name] = (inputs.contiguous(), outputs.contiguous())
            step["loss_mask"] = ~bos_mask.flatten(0, 1)
            yield step
