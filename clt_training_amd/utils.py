"""Drop-in for the decoder half of ``sparsify.utils`` (sparsify/utils.py:441-563):
``decoder_impl(top_indices, top_acts, W_dec)`` with autograd, on the HIP kernels."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import hip_ops as ops


class SegmentsHolder:
    """Lazily built, shared (token, slot)-by-latent segments of one top-k index tensor.
    One holder travels with the indices of a source layer, so the sort runs once per
    layer and step however many decoders / the encoder backward use it."""

    __slots__ = ("seg",)

    def __init__(self):
        self.seg: Optional[ops.Segments] = None

    def get(self, indices: Tensor, M: int) -> ops.Segments:
        if self.seg is None or self.seg.M != M or self.seg.perm.numel() != indices.numel():
            self.seg = ops.segments_build(indices, M)
        return self.seg


def holder_of(indices: Tensor) -> SegmentsHolder:
    h = getattr(indices, "_clt_seg", None)
    if h is None:
        h = SegmentsHolder()
        try:
            indices._clt_seg = h
        except Exception:  # pragma: no cover
            pass
    return h


class HipDecode(torch.autograd.Function):
    """xFormersEmbeddingBag (xformers.py:234-255) on HIP: out in W.dtype, fp32 accumulate;
    backward returns the dense [M, d] weight grad in W.dtype and the per-sample-weight grad."""

    @staticmethod
    def forward(ctx, indices: Tensor, weight: Tensor, per_sample_weights: Tensor):
        ctx.save_for_backward(indices, weight, per_sample_weights)
        ctx.holder = holder_of(indices)
        return ops.decode_fwd(indices, per_sample_weights, weight)

    @staticmethod
    def backward(ctx, gradient):
        indices, weight, psw = ctx.saved_tensors
        gradient = gradient.contiguous()
        weight_g = psw_g = None
        if ctx.needs_input_grad[2]:
            psw_g, _, _ = ops.decode_bwd_acts(indices, psw, weight, gradient)
            psw_g = psw_g.to(psw.dtype)
        if ctx.needs_input_grad[1]:
            seg = ctx.holder.get(indices, weight.shape[0])
            weight_g = ops.spt_dense(seg, psw.float().contiguous().view(-1), gradient, out_dtype=weight.dtype)
        return None, weight_g, psw_g


def hip_decode(top_indices: Tensor, top_acts: Tensor, W_dec: Tensor) -> Tensor:
    assert top_acts.ndim == 2 and W_dec.ndim == 2
    return HipDecode.apply(top_indices, W_dec, top_acts)


# The reference selects eager / Triton by environment (utils.py:549-563); here there is one
# implementation and it fails loudly without the HIP library.
triton_decode = hip_decode
eager_decode = hip_decode  # (utils.py:442-445 is the reference's CPU / fallback route; there is no CPU route here)
decoder_impl = hip_decode
