"""Muon for the matrix-valued parameters (``optimizer="muon"``: sparsify/muon.py:12-203, trainer.py:194-222).

Per step and 2-D parameter: decoupled weight decay, momentum (``exp_avg.lerp_(g, 1 - beta)``), Nesterov look-ahead, the
quintic Newton-Schulz iteration in bf16 (muon.py:13-42) and ``p -= lr * 0.2 * sqrt(max(p.shape)) * NS(g)``.

The iteration is three library GEMMs per step (hipBLASLt through torch: plain GEMMs) on the transposed table, so the
Gram matrix is d x d: ``A = X X^T``, ``B = b A + c A A``, ``X = a X + B X``, with the reference's bf16 rounding points.
Latent shards (tp.TPTrainer): a table's rows are split over the ranks, so with ``X = G^T`` [d, M] the Gram
matrix is the SUM of the ranks' ``X_r X_r^T`` -- one all-reduce of d x d values per iteration (and one scalar for the
Frobenius norm); everything else is local to the shard.  The reference instead deals whole parameters round-robin to
the ranks of a data-parallel group and all-gathers the updates (muon.py:150-203), which needs every rank to hold every
parameter."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

NS_COEFFS = (3.4445, -4.7750, 2.0315)  # muon.py:26


def _all_reduce(t: Tensor, comm) -> Tensor:
    return t if comm is None or comm.world == 1 else comm.all_reduce_(t)


@torch.no_grad()
def newton_schulz(G: Tensor, steps: int, comm=None, rows_sharded: bool = False, full_rows: Optional[int] = None) -> Tensor:
    """muon.py:13-42 for a 2-D ``G``; returns bf16 in G's shape.  ``rows_sharded``: G is this rank's row block of a
    [full_rows, cols] matrix whose other blocks live on the other ranks of ``comm``."""
    assert G.ndim == 2
    a, b, c = NS_COEFFS
    rows = full_rows if rows_sharded and full_rows is not None else G.shape[0]
    X = G.bfloat16()
    tall = rows > G.shape[1]
    if rows_sharded and not tall:
        raise NotImplementedError("a row-sharded matrix with fewer rows than columns (no [M, d] table is)")
    if tall:
        X = X.mT  # [cols, rows]: the Gram matrix below is cols x cols
    # spectral norm <= Frobenius norm <= 1 (muon.py:32)
    sq = X.float().pow(2).sum()
    if rows_sharded:
        sq = _all_reduce(sq.view(1), comm)[0]
    X = X / (sq.sqrt().to(X.dtype) + 1e-7)
    sharded = rows_sharded and comm is not None and comm.world > 1
    for _ in range(steps):
        if sharded:  # the shards' partial Gram matrices in fp32 (bf16 operands, fp32 result), summed, rounded ONCE
            from .engine import _mm_f32

            A = _all_reduce(_mm_f32(X, X.mT), comm).to(X.dtype)
        else:
            A = X @ X.mT
        # the reference's operation sequence, every intermediate rounded to bf16 where it rounds (muon.py:36-38): with
        # coefficients of magnitude 2-5 a fused multiply-add epilogue moves the result by 3 % -- parity, not accuracy
        B = b * A + c * A @ A
        X = a * X + B @ X
    return X.mT if tall else X


class Muon(torch.optim.Optimizer):
    """MomentUm Orthogonalized by Newton-schulz for 2-D parameters (muon.py:51-203), one process per GPU.
    ``sharded``: ids of parameters that are row blocks of latent-sharded tables, with ``comm`` the tp group and
    ``full_rows[id(p)]`` the unsharded row count (the update scale uses the FULL shape, muon.py:153-156)."""

    def __init__(self, params, lr: float = 1e-3, momentum: float = 0.95, nesterov: bool = True,
                 weight_decay: float = 0.1, ns_steps: int = 5, comm=None, sharded=(), full_rows: Optional[dict] = None):
        super().__init__(params, dict(lr=lr, momentum=momentum, nesterov=nesterov, ns_steps=ns_steps,
                                      weight_decay=weight_decay))
        self.comm = comm
        self._sharded = set(sharded)
        self._full_rows = dict(full_rows or {})

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        for group in self.param_groups:
            lr, beta = float(group["lr"]), group["momentum"]
            if group["weight_decay"] > 0.0:  # decoupled, before the update (muon.py:137-140)
                for p in group["params"]:
                    p.mul_(1 - lr * group["weight_decay"])
            for p in group["params"]:
                g = p.grad
                assert g is not None, f"parameter of shape {tuple(p.shape)} has no gradient"
                if beta > 0.0:  # muon.py:172-177
                    st = self.state[p]
                    if "exp_avg" not in st:
                        st["exp_avg"] = torch.zeros_like(g)
                    buf = st["exp_avg"].lerp_(g, 1 - beta)
                    g = g.lerp_(buf, beta) if group["nesterov"] else buf
                sh = id(p) in self._sharded
                rows = self._full_rows.get(id(p), p.shape[0])
                u = newton_schulz(g.view(len(g), -1), group["ns_steps"], self.comm, sh, rows)
                scale = 0.2 * max(rows, *p.shape[1:]) ** 0.5  # arXiv 2502.16982 (muon.py:153-156, 197-198)
                p.add_(u.view_as(p).to(p.dtype), alpha=-lr * scale)
