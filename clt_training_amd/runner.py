"""CrossLayerRunner: mirror of sparsify/runner.py:11-201 for the coalescing modes on the
hot path -- ``coalesce_topk="none"`` (every live source layer decodes into the current
target with its own W_decs[j - i], runner.py:60-71) and ``"concat"`` (per-target tied decoder over
the concatenated candidates, or with ``topk_coalesced`` over their k best, runner.py:93-127) and
``"per-layer"`` (the target's decoder holds one block of rows per live source, runner.py:130-158).  Same bookkeeping: insertion-ordered live sources, deferred encoder
backward (``restore``) for the sources that just wrote their last target."""
from __future__ import annotations

from dataclasses import replace

import torch
from torch import Tensor

from .sparse_coder import MidDecoder, SparseCoder


class CrossLayerRunner(object):
    def __init__(self):
        self.outputs = {}
        self.to_restore = {}

    def encode(self, x: Tensor, sparse_coder: SparseCoder, **kwargs):
        return sparse_coder(x=x, y=None, return_mid_decoder=True, **kwargs)

    def decode(self, mid_out: MidDecoder, y: Tensor, module_name: str, detach_grad: bool = False,
               advance: bool = True, **kwargs):
        cfg = mid_out.sparse_coder.cfg
        if cfg.coalesce_topk == "group" or (cfg.secondary_target_tied and cfg.do_coalesce_topk):
            raise NotImplementedError("coalesce_topk in {'none', 'concat', 'per-layer'} is built")
        self.outputs[module_name] = mid_out

        candidate_indices, candidate_values, layer_mids = [], [], []
        output = 0
        to_delete = set()
        out, hookpoint = None, None
        for i, (hookpoint, layer_mid) in enumerate(self.outputs.items()):
            if detach_grad:
                layer_mid.detach()
            divide_by = max(1, len(self.outputs) - 1) if layer_mid.sparse_coder.cfg.divide_cross_layer else 1
            layer_mids.append(layer_mid)
            if cfg.do_coalesce_topk or cfg.secondary_target_tied:
                # runner.py:52-54: candidates carry their position among the live sources.  "concat" takes it off
                # again ((idx + i*M) % M == idx, :122); "per-layer" decodes with it: the target's W_dec holds one
                # block of M rows per live source (sparse_coder.py:306-307)
                off = i * layer_mid.sparse_coder.num_latents if cfg.coalesce_topk == "per-layer" else 0
                candidate_indices.append(layer_mid.latent_indices + off if off else layer_mid.latent_indices)
                candidate_values.append(layer_mid.current_latent_acts)
            if detach_grad and advance:
                self.to_restore[hookpoint] = (layer_mid, layer_mid.will_be_last)
            if layer_mid.will_be_last:
                to_delete.add(hookpoint)
            if not cfg.do_coalesce_topk:
                out = layer_mid(
                    y,
                    addition=(0 if hookpoint != module_name else (output / divide_by)),
                    no_extras=(hookpoint != module_name) or cfg.secondary_target_tied,
                    denormalize=(hookpoint == module_name) and not cfg.secondary_target_tied,
                    **kwargs,
                )
                if hookpoint != module_name:
                    output = output + out.sae_out
            else:
                layer_mid.next()

        if cfg.secondary_target_tied:  # runner.py:75-91: the current layer's slot-0 decoder also reads every candidate
            new_mid_out = mid_out.copy(indices=torch.cat(candidate_indices, dim=1),
                                       activations=torch.cat(candidate_values, dim=1))
            out = new_mid_out(y, index=0, add_post_enc=False, no_extras=False, denormalize=True, addition=out.sae_out,
                              **kwargs)

        if cfg.do_coalesce_topk:  # "concat" (runner.py:93-127)
            best_indices = torch.cat(candidate_indices, dim=1)
            best_values = torch.cat(candidate_values, dim=1)
            if cfg.topk_coalesced:  # runner.py:96-117: the k best of the [N, k * n_live] candidates (a small torch op;
                best_values, pos = torch.topk(best_values, k=cfg.k, dim=1)  # autograd routes the gradient back)
                best_indices = torch.gather(best_indices, 1, pos)
            new_mid_out = mid_out.copy(indices=best_indices, activations=best_values)
            out = new_mid_out(y, index=0, add_post_enc=False, **kwargs)
            if cfg.coalesce_topk == "concat":
                if advance:
                    del mid_out.x
            else:  # "per-layer" (runner.py:130-158): report the current layer's own latents, 0 elsewhere
                M = mid_out.sparse_coder.num_latents
                out = replace(out, latent_indices=(out.latent_indices % M) * (out.latent_indices // M == i))

        assert hookpoint == module_name  # last output guaranteed to be the current layer
        if not advance:
            for layer_mid in layer_mids:
                layer_mid.prev()
        if advance:
            for hookpoint in to_delete:
                del self.outputs[hookpoint]
        return out

    def __call__(self, x: Tensor, y: Tensor, sparse_coder: SparseCoder, module_name: str, detach_grad: bool = False,
                 dead_mask=None, loss_mask=None, *, encoder_kwargs: dict = {}, decoder_kwargs: dict = {}):
        mid_out = self.encode(x, sparse_coder, dead_mask=dead_mask, **encoder_kwargs)
        return self.decode(mid_out, y, module_name, detach_grad, loss_mask=loss_mask, **decoder_kwargs)

    def restore(self):
        for restorable, was_last in self.to_restore.values():
            if was_last:
                restorable.restore(True)
        self.to_restore.clear()

    def reset(self):
        self.outputs.clear()
        self.to_restore.clear()
