"""How the autograd Functions of the drop-in path (FusedDecode, FusedEncoder) find an optimizer that may consume their
row gradients inside the gradient kernel (clt_decode_bwd_rows_adam / clt_spt_dense_adam): ``Trainer`` registers its
``adam.Adam`` here when every optimizer step sees exactly one micro-step; the Functions then update the ``[M, d]``
matrix in place during backward and return no gradient for it, so ``Adam.step`` skips the tensor (its ``.grad`` stays
None).  A matrix that enters more than one Function in a step (weight tying) is left to the ordinary
``.grad`` -> ``step()`` route."""
from __future__ import annotations

import weakref

ACTIVE = None      # adam.Adam, or None: no fusion
_uses: dict = {}   # data_ptr -> number of forward uses since begin_step()
# unit-norm SAE decoders: project the gradient row first (sparse_coder.py:696-711).  Keyed by weak references to the
# parameter objects (a freed and reused allocation must not inherit the flag); looked up through their CURRENT storage.
_project: "weakref.WeakSet" = weakref.WeakSet()


def begin_step():
    _uses.clear()


def end_step():
    """After the optimizer step: no Function outside Trainer.step may take the fused branch (an evaluation with
    gradients, a second model) -- it would apply an in-place Adam update and return no weight gradient."""
    global ACTIVE
    ACTIVE = None
    _uses.clear()


def note_use(t):
    if ACTIVE is not None:
        key = t.data_ptr()
        _uses[key] = _uses.get(key, 0) + 1


def set_project(t, on: bool = True):
    (_project.add if on else _project.discard)(t)


def _projected(ptr: int) -> bool:
    return any(p.data_ptr() == ptr for p in _project)


def fused_for(t):
    """Optimizer state / hyper-parameters for updating ``t`` inside the gradient kernel, or None."""
    if ACTIVE is None or _uses.get(t.data_ptr(), 0) != 1:
        return None
    h = ACTIVE.fused_rows_by_ptr(t.data_ptr())
    if h is not None and _projected(t.data_ptr()):
        h = dict(h, project=True)
    return h
