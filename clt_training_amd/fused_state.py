"""How the autograd Functions of the drop-in path (FusedDecode, FusedEncoder) find an optimizer that may consume their
row gradients inside the gradient kernel (clt_decode_bwd_rows_adam / clt_spt_dense_adam): ``Trainer`` registers its
``adam.Adam`` here when every optimizer step sees exactly one micro-step; the Functions then update the ``[M, d]``
matrix in place during backward and return no gradient for it, so ``Adam.step`` skips the tensor (its ``.grad`` stays
None).  A matrix that enters more than one Function in a step (weight tying) is left to the ordinary
``.grad`` -> ``step()`` route."""
from __future__ import annotations

ACTIVE = None      # adam.Adam, or None: no fusion
_uses: dict = {}   # data_ptr -> number of forward uses since begin_step()
_project: set = set()  # data_ptrs of unit-norm SAE decoders: project the gradient row first (sparse_coder.py:696-711)


def begin_step():
    _uses.clear()


def note_use(t):
    if ACTIVE is not None:
        key = t.data_ptr()
        _uses[key] = _uses.get(key, 0) + 1


def set_project(t, on: bool = True):
    (_project.add if on else _project.discard)(t.data_ptr())


def fused_for(t):
    """Optimizer state / hyper-parameters for updating ``t`` inside the gradient kernel, or None."""
    if ACTIVE is None or _uses.get(t.data_ptr(), 0) != 1:
        return None
    h = ACTIVE.fused_rows_by_ptr(t.data_ptr())
    if h is not None and t.data_ptr() in _project:
        h = dict(h, project=True)
    return h
