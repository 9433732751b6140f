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


class CrossLayerRunner:
    """Keeps the live source layers (oldest first) and turns one hook call into the prediction of the current target.

    The public surface is the reference's (``outputs``, ``to_restore``, ``encode``, ``decode``, ``__call__``,
    ``restore``, ``reset``; runner.py:11-201); ``decode`` is organised as: bookkeeping over the live sources, then one of
    three combiners -- ``_sum_of_sources`` ("none"), ``_second_read`` (``secondary_target_tied``), ``_one_bag``
    ("concat" / "per-layer")."""

    def __init__(self):
        self.outputs = {}     # hookpoint -> MidDecoder of every source layer that still has targets to write
        self.to_restore = {}  # hookpoint -> (MidDecoder, wrote its last target): deferred encoder backward

    def encode(self, x: Tensor, sparse_coder: SparseCoder, **kwargs):
        return sparse_coder(x=x, y=None, return_mid_decoder=True, **kwargs)

    # ------------------------------------------------------------------ the three ways to combine the live sources
    @staticmethod
    def _sum_of_sources(cfg, mids, y, kwargs):
        """runner.py:60-71: every earlier source decodes with its own W_decs[j - i] (no bias / skip extras of its own
        target slot beyond the decoder's, no loss); the current layer adds their sum, scaled by 1 / divide_by."""
        *earlier, current = mids
        carried = 0
        for mid in earlier:
            carried = carried + mid(y, addition=0, no_extras=True, denormalize=False, **kwargs).sae_out
        share = max(1, len(mids) - 1) if current.sparse_coder.cfg.divide_cross_layer else 1
        return current(y, addition=carried / share, no_extras=cfg.secondary_target_tied,
                       denormalize=not cfg.secondary_target_tied, **kwargs)

    @staticmethod
    def _second_read(current, bag, first, y, kwargs):
        """runner.py:75-91: the current layer's slot-0 decoder reads every live candidate once more, on top of ``first``."""
        again = current.copy(indices=bag[0], activations=bag[1])
        return again(y, index=0, add_post_enc=False, no_extras=False, denormalize=True, addition=first.sae_out, **kwargs)

    @staticmethod
    def _one_bag(cfg, current, bag, n_live, y, kwargs):
        """runner.py:93-158: ONE decode of the concatenated candidates through the target's single (tied) decoder."""
        idx, val = bag
        if cfg.topk_coalesced:  # :96-117: the k best of the [N, k * n_live] candidates (autograd routes the gradient back)
            val, keep = torch.topk(val, k=cfg.k, dim=1)
            idx = torch.gather(idx, 1, keep)
        out = current.copy(indices=idx, activations=val)(y, index=0, add_post_enc=False, **kwargs)
        if cfg.coalesce_topk == "per-layer":  # :130-158: report the current layer's own latents, 0 elsewhere
            M = current.sparse_coder.num_latents
            own_block = n_live - 1
            out = replace(out, latent_indices=(out.latent_indices % M) * (out.latent_indices // M == own_block))
        return out

    def decode(self, mid_out: MidDecoder, y: Tensor, module_name: str, detach_grad: bool = False,
               advance: bool = True, **kwargs):
        cfg = mid_out.sparse_coder.cfg
        if cfg.coalesce_topk == "group" or (cfg.secondary_target_tied and cfg.do_coalesce_topk):
            raise NotImplementedError("coalesce_topk in {'none', 'concat', 'per-layer'} is built")
        self.outputs[module_name] = mid_out  # the newest entry: dicts keep insertion order
        names, mids = list(self.outputs), list(self.outputs.values())
        assert names[-1] == module_name

        # ---- bookkeeping, before any source moves on to its next target slot
        if detach_grad:
            for name, mid in zip(names, mids):
                mid.detach()
                if advance:
                    self.to_restore[name] = (mid, mid.will_be_last)
        done = [name for name, mid in zip(names, mids) if mid.will_be_last]
        bag = None
        if cfg.do_coalesce_topk or cfg.secondary_target_tied:
            # runner.py:52-54: candidates carry their position among the live sources.  "concat" takes it off again
            # ((idx + i*M) % M == idx, :122); "per-layer" decodes with it: the target's W_dec holds one block of M rows
            # per live source (sparse_coder.py:306-307)
            blocks = cfg.coalesce_topk == "per-layer"
            bag = (torch.cat([mid.latent_indices + (pos * mid.sparse_coder.num_latents if blocks else 0)
                              for pos, mid in enumerate(mids)], dim=1),
                   torch.cat([mid.current_latent_acts for mid in mids], dim=1))

        # ---- the prediction
        if cfg.do_coalesce_topk:
            for mid in mids:
                mid.next()
            out = self._one_bag(cfg, mid_out, bag, len(mids), y, kwargs)
            if cfg.coalesce_topk == "concat" and advance:
                del mid_out.x
        else:
            out = self._sum_of_sources(cfg, mids, y, kwargs)
            if cfg.secondary_target_tied:
                out = self._second_read(mid_out, bag, out, y, kwargs)

        if advance:
            for name in done:
                del self.outputs[name]
        else:  # a dry run (e.g. an evaluation pass): every source stays on the slot it was on
            for mid in mids:
                mid.prev()
        return out

    def __call__(self, x: Tensor, y: Tensor, sparse_coder: SparseCoder, module_name: str, detach_grad: bool = False,
                 dead_mask=None, loss_mask=None, *, encoder_kwargs: dict = {}, decoder_kwargs: dict = {}):
        mid_out = self.encode(x, sparse_coder, dead_mask=dead_mask, **encoder_kwargs)
        return self.decode(mid_out, y, module_name, detach_grad, loss_mask=loss_mask, **decoder_kwargs)

    def restore(self):
        """The deferred encoder backward of the sources that wrote their last target in the hook call just finished."""
        for mid, wrote_last in self.to_restore.values():
            if wrote_last:
                mid.restore(True)
        self.to_restore.clear()

    def reset(self):
        self.outputs.clear()
        self.to_restore.clear()
