"""Multi-tensor Adam on the HIP kernel `clt_adam_step` -- the MI355X counterpart of the
``torch.optim.Adam(pgs, betas=(b1, b2))`` the reference builds in trainer.py:166-186
(eps 1e-8, no weight decay, no amsgrad; per-group lr, possibly a tensor; LambdaLR-compatible).

One launch updates every parameter of every group: the host refreshes a small table of
(p, g, m, v, numel, lr) records each step (grad pointers change when autograd allocates
new .grad tensors) and uploads it asynchronously; the chunk -> tensor map is static."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .hip_ops import _DT, _call, _stream

import os

CHUNK = int(os.environ.get("CLT_ADAM_CHUNK", 1 << 14))  # elements per workgroup (a multiple of 8)


class Adam(torch.optim.Optimizer):
    HAS_STATE = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        defaults = dict(lr=lr, betas=betas, eps=eps)
        super().__init__(params, defaults)
        self._step = 0
        self._tables = {}  # key: tuple of param ids with grads -> static device tables

    def state_dict(self):
        sd = super().state_dict()
        sd["clt_step"] = self._step  # one step count for all parameters (torch keeps one per parameter)
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        self._step = sd.pop("clt_step", self._step)
        super().load_state_dict(sd)
        self._tables = {}  # the cached pointer tables name the state tensors that were just replaced
        for cache in ("_group_of", "_by_ptr"):
            self.__dict__.pop(cache, None)

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        for cache in ("_group_of", "_by_ptr"):
            self.__dict__.pop(cache, None)
        self._tables = {}

    def _state(self, p):
        st = self.state[p]
        if not st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def fused_rows(self, p):
        """State and hyper-parameters with which a gradient-producing kernel may apply THIS step's update of ``p``
        itself (hip_ops.spt_dense_adam); ``p`` must then reach step() without a .grad.  None if p is not ours."""
        if not hasattr(self, "_group_of"):
            self._group_of = {id(q): gi for gi, g in enumerate(self.param_groups) for q in g["params"]}
        gi = self._group_of.get(id(p))
        if gi is None:
            return None
        st = self._state(p)
        g = self.param_groups[gi]
        return dict(m=st["exp_avg"], v=st["exp_avg_sq"], lr=float(g["lr"]), betas=g["betas"], eps=g["eps"],
                    step=self._step + 1)

    def fused_rows_by_ptr(self, ptr: int):
        """fused_rows for the parameter whose storage starts at ``ptr`` (autograd Functions see saved tensors, not the
        Parameter objects).  Parameters that already hold a .grad are not eligible."""
        if not hasattr(self, "_by_ptr"):
            self._by_ptr = {q.data_ptr(): q for g in self.param_groups for q in g["params"]}
        p = self._by_ptr.get(ptr)
        if p is None or p.grad is not None:
            return None
        return self.fused_rows(p)

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        self._step += 1
        live = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is not None:
                    live.append((gi, p))
        if not live:
            return
        dev = live[0][1].device
        key = tuple(id(p) for _, p in live)
        tb = self._tables.get(key)
        if tb is None:
            ct, cs = [], []
            for ti, (_, p) in enumerate(live):
                n = p.numel()
                starts = np.arange(0, n, CHUNK, dtype=np.int64)
                ct.append(np.full(len(starts), ti, dtype=np.int32))
                cs.append(starts)
            ct = torch.from_numpy(np.concatenate(ct)).to(dev)
            cs = torch.from_numpy(np.concatenate(cs)).to(dev)
            # The upload below is asynchronous and the host may run several steps ahead of the GPU: a ring of pinned
            # staging buffers, each guarded by an event recorded after its copy, so that a buffer is never rewritten
            # (next step's lr, new .grad pointers) before the copy that reads it has executed.
            nbytes = len(live) * C.sizeof(_lib.AdamTensor)
            ring = [[torch.empty(nbytes, dtype=torch.uint8).pin_memory(), None] for _ in range(4)]
            devt = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            tb = self._tables[key] = (ct, cs, ring, devt)
        ct, cs, ring, devt = tb
        slot = ring[self._step % len(ring)]
        if slot[1] is not None:
            slot[1].synchronize()
        host = slot[0]
        recs = (_lib.AdamTensor * len(live)).from_address(host.data_ptr())
        for ti, (gi, p) in enumerate(live):
            g = p.grad
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            assert p.is_contiguous() and g.shape == p.shape
            r = recs[ti]
            if self.HAS_STATE:
                st = self._state(p)
                r.p, r.g, r.m, r.v = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            else:
                r.p, r.g, r.m, r.v = p.data_ptr(), g.data_ptr(), None, None
            r.numel = p.numel()
            r.lr = float(self.param_groups[gi]["lr"])
            r.p_dtype, r.g_dtype = _DT[p.dtype], _DT[g.dtype]
        devt.copy_(host, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        self._launch(devt, len(live), ct, cs)

    def _launch(self, devt, n_tensors, ct, cs):
        betas = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        _call("clt_adam_step", C.c_void_p(devt.data_ptr()), n_tensors, C.c_void_p(ct.data_ptr()),
              C.c_void_p(cs.data_ptr()), ct.numel(), CHUNK, 1.0, float(betas[0]), float(betas[1]), float(eps),
              self._step, _stream(devt))


class SignSGD(Adam):
    """sparsify/sign_sgd.py:5-23 (steepest descent in the L-infinity norm): ``p -= lr * sign(p.grad)`` for every parameter
    that has a gradient, all tensors in one launch of ``clt_signsgd_step``.  No optimizer state."""

    HAS_STATE = False

    def __init__(self, params, lr: float = 1e-3):
        if lr <= 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        torch.optim.Optimizer.__init__(self, params, {"lr": lr})
        self._step = 0
        self._tables = {}

    def fused_rows(self, p):  # no fused row-pass form: the gradient kernels write plain gradients for this optimizer
        return None

    def fused_rows_by_ptr(self, ptr):
        return None

    def _launch(self, devt, n_tensors, ct, cs):
        _call("clt_signsgd_step", C.c_void_p(devt.data_ptr()), n_tensors, C.c_void_p(ct.data_ptr()),
              C.c_void_p(cs.data_ptr()), ct.numel(), CHUNK, 1.0, _stream(devt))
