"""Drop-in for ``sparsify.kernels`` (sparsify/kernels.py): the same function names and
argument meaning, each ending in a gfx950 HIP kernel of libcltmi355.so instead of Triton."""
from __future__ import annotations

import torch
from torch import Tensor

from . import hip_ops as ops


def triton_sparse_transpose_dense_matmul(sparse_indices: Tensor, sparse_values: Tensor, dense: Tensor, N: int,
                                         BLOCK_SIZE_AK=128) -> Tensor:
    """sparse.T @ dense -> [N, B] fp32 (kernels.py:14-54).  The COO sort is replaced by a
    stable segment build; the atomic run-length kernel by an atomic-free per-row gather."""
    assert sparse_indices.shape == sparse_values.shape
    assert sparse_indices.is_contiguous() and sparse_values.is_contiguous() and dense.is_contiguous()
    assert sparse_indices.shape[0] == dense.shape[0]
    seg = ops.segments_build(sparse_indices, N)
    return ops.spt_dense(seg, sparse_values.float(), dense, out_dtype=torch.float32)


def triton_sparse_dense_matmul(sparse_indices: Tensor, sparse_values: Tensor, dense: Tensor) -> Tensor:
    """sparse @ dense -> [A, B] in sparse_values.dtype (kernels.py:219-260)."""
    assert sparse_indices.shape == sparse_values.shape
    assert dense.is_contiguous()
    return ops.decode_fwd(sparse_indices, sparse_values, dense, out_dtype=sparse_values.dtype)


def triton_dense_dense_sparseout_matmul(dense1: Tensor, dense2: Tensor, at_indices: Tensor) -> Tensor:
    """(dense1 @ dense2).gather(1, at_indices) (kernels.py:332-383); dense2 is [B, N] with
    stride(0) == 1, i.e. the transpose of a row-major [N, B] table."""
    assert dense1.stride(1) == 1
    table = dense2.t()  # [N, B]: row-major when dense2 is the W_dec.T view the reference passes (stride(0) == 1, its
    table = table if table.is_contiguous() else table.contiguous()  # own precondition, kernels.py:345-346)
    dummy = torch.ones(at_indices.shape, dtype=torch.float32, device=dense1.device)
    ga, _, _ = ops.decode_bwd_acts(at_indices, dummy, table, dense1.contiguous())
    return ga.to(dense1.dtype)


class TritonDecoder(torch.autograd.Function):
    """kernels.py:628-662: apply(sparse_indices, sparse_values, decoder_weight) with
    decoder_weight = W_dec.T ([d, M], contiguous when transposed)."""

    @staticmethod
    def forward(ctx, sparse_indices, sparse_values, decoder_weight):
        ctx.save_for_backward(sparse_indices, sparse_values, decoder_weight)
        return triton_sparse_dense_matmul(sparse_indices, sparse_values, decoder_weight.T.contiguous())

    @staticmethod
    def backward(ctx, grad_output):
        sparse_indices, sparse_values, decoder_weight = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        decoder_grad = values_grad = None
        if ctx.needs_input_grad[2]:
            decoder_grad = triton_sparse_transpose_dense_matmul(
                sparse_indices, sparse_values, grad_output, N=decoder_weight.shape[1]).T.to(decoder_weight.dtype)
        if ctx.needs_input_grad[1]:
            values_grad = triton_dense_dense_sparseout_matmul(grad_output, decoder_weight, sparse_indices)
            values_grad = values_grad.to(sparse_values.dtype)
        return None, values_grad, decoder_grad  # (three inputs; the reference returns a stray fourth None, kernels.py:656-662)
