"""Drop-in for ``sparsify.fused_encoder`` (sparsify/fused_encoder.py:63-201, 293-422), topk
activation: dense GEMM + bias + ReLU + row top-k under no_grad, then an identity autograd
node whose backward touches only the k active latents."""
from __future__ import annotations

from typing import Literal, NamedTuple

import torch
from torch import Tensor

from . import fused_state
from . import hip_ops as ops
from .utils import SegmentsHolder, holder_of


class EncoderOutput(NamedTuple):
    top_acts: Tensor
    top_indices: Tensor


def _compute_dtype(input: Tensor, weight: Tensor):
    """bf16 under torch.autocast (sparse_coder.py:672-676) or when the weights are bf16."""
    if torch.is_autocast_enabled():
        return torch.get_autocast_dtype("cuda")
    return weight.dtype


class FusedEncoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, weight, bias, values, indices, activation=None):
        fused_state.note_use(weight)
        ctx.save_for_backward(input, weight, bias, values, indices)
        ctx.k = values.shape[-1]
        ctx.activation = activation
        ctx.holder = holder_of(indices)
        return values

    @staticmethod
    @torch.no_grad()
    def backward(ctx, grad_values):
        input, weight, bias, values, indices = ctx.saved_tensors
        grad_input = grad_weight = grad_bias = None
        M = weight.shape[0]
        gv = grad_values.float().contiguous()  # the (values > 0) gate of :97 is applied in-kernel
        if ctx.needs_input_grad[0]:
            gated = gv * (values > 0)
            grad_input = ops.decode_fwd(indices, gated, weight, out_dtype=weight.dtype).to(input.dtype)
        need_b = bias is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or need_b:
            seg = ctx.holder.get(indices, M)
            # grad_W[idx, :] += gv * x (fused_encoder.py:150-155); grad_b[idx] += gv (:140-143)
            x = input if input.dtype in (torch.float32, torch.bfloat16) else input.float()
            h = fused_state.fused_for(weight) if (ctx.needs_input_grad[1] and x.dtype == weight.dtype) else None
            if h is not None:  # the optimizer update happens inside the gradient kernel: no grad_weight
                gw = None
                gb = ops.spt_dense_adam(seg, gv.view(-1), x, weight.data, h["m"], h["v"], lr=h["lr"], betas=h["betas"],
                                        eps=h["eps"], step=h["step"], gate=values.contiguous().view(-1), want_rowsum=True)
            else:
                gw, gb = ops.spt_dense(seg, gv.view(-1), x, gate=values.contiguous().view(-1), out_dtype=weight.dtype,
                                       want_rowsum=True)
            if ctx.needs_input_grad[1]:
                grad_weight = gw
            if need_b:
                grad_bias = gb.to(bias.dtype)
        return grad_input, grad_weight, grad_bias, None, None, None


def fused_encoder(input: Tensor, weight: Tensor, bias: Tensor, k: int,
                  activation: Literal["groupmax", "topk"] = "topk", use_fp8: bool = False) -> EncoderOutput:
    """input (N, D), weight (M, D), bias (M,) -> top-k (or, :373-413, groupmax) values / indices
    (fused_encoder.py:293-309)."""
    if activation not in ("topk", "groupmax", "batchtopk"):
        raise NotImplementedError(f"activation {activation!r}")
    if use_fp8:
        raise NotImplementedError("use_fp8 is out of scope (no BASELINE config uses it)")
    with torch.no_grad():
        cd = _compute_dtype(input, weight)
        x = input if input.dtype == cd else input.to(cd)
        W = weight if weight.dtype == cd else weight.to(cd)
        if activation == "batchtopk":  # fused_encoder.py:315-318, 204-240: row top-4k, then the batch-wide threshold
            values, indices = ops.encode_topk(x, W, bias, 4 * k, out_dtype=cd)
            values = ops.batch_threshold(values, k)
        else:
            values, indices = ops.encode_topk(x, W, bias, k, out_dtype=cd, activation=activation)
    indices._clt_seg = SegmentsHolder()
    values = FusedEncoder.apply(input, weight, bias, values, indices, activation)
    return EncoderOutput(top_acts=values, top_indices=indices)
