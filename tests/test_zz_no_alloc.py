"""clt_fvu_fwd_bwd_static -- the entry point that allocates nothing (static device scratch inside the library) --
against clt_fvu_fwd_bwd with a caller workspace, through the C ABI: bit-identical scalars and gradient, on two calls in
a row (the second with a smaller input, so stale partials of the first call in the shared scratch would show), and the
size limit is rejected with a clear error.  Collected last (test_zz_*): a failure here must not hide other tests."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _both(N, d, dtype, g_dtype, masked, seed):
    from clt_training_amd import _lib, hip_ops
    from clt_training_amd.hip_ops import _DT, _ptr, _stream

    lib = _lib.load()
    g = torch.Generator().manual_seed(seed)
    y = torch.randn(N, d, generator=g).to(dtype).to(DEV)
    yhat = (0.7 * y.float().cpu() + 0.3 * torch.randn(N, d, generator=g)).to(dtype).to(DEV)
    mask = (torch.rand(N, generator=g) > 0.2).to(DEV) if masked else None
    # existing entry point: caller workspace
    sc_ref, G_ref = hip_ops.fvu_fwd_bwd(y, yhat, mask, gscale=0.5, g_dtype=g_dtype)
    # new entry point: every buffer preallocated, then the raw C call (no torch allocation between)
    sc = torch.full((3,), float("nan"), dtype=torch.float32, device=DEV)
    G = torch.full((N, d), float("nan"), dtype=g_dtype, device=DEV)
    m8 = None if mask is None else mask.view(torch.uint8)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    rc = lib.clt_fvu_fwd_bwd_static(_ptr(y), _DT[y.dtype], _ptr(yhat), _DT[yhat.dtype], _ptr(m8), N, d, C.c_float(0.5),
                                    _ptr(sc), _ptr(G), _DT[g_dtype], _stream(y))
    assert rc == 0, lib.clt_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.equal(sc, sc_ref), (sc, sc_ref)
    assert torch.equal(G, G_ref)
    assert bool(torch.isfinite(sc).all()) and float(sc[0]) > 0


def test_static_fvu_equals_workspace_fvu_two_calls_in_a_row():
    _both(4096, 768, torch.bfloat16, torch.bfloat16, True, 0)   # fills 512 row blocks of partials
    _both(100, 64, torch.bfloat16, torch.bfloat16, True, 1)     # smaller: 7 row blocks, narrower rows, same scratch
    _both(8192, 2048, torch.float32, torch.float32, False, 2)
    _both(33, 8, torch.float32, torch.bfloat16, True, 3)


def test_static_fvu_rejects_sizes_beyond_its_scratch():
    from clt_training_amd import _lib
    from clt_training_amd.hip_ops import _ptr, _stream

    lib = _lib.load()
    N, d = 8, 8192 + 4
    y = torch.zeros(N, d, device=DEV)
    sc = torch.zeros(3, device=DEV)
    rc = lib.clt_fvu_fwd_bwd_static(_ptr(y), 0, _ptr(y), 0, None, N, d, C.c_float(1.0), _ptr(sc), None, 0, _stream(y))
    assert rc == -2
    assert b"8192" in lib.clt_last_error()
    # the largest supported width still runs
    d = 8192
    y = torch.randn(N, d, device=DEV)
    yh = 0.5 * y
    rc = lib.clt_fvu_fwd_bwd_static(_ptr(y), 0, _ptr(yh), 0, None, N, d, C.c_float(1.0), _ptr(sc), None, 0, _stream(y))
    assert rc == 0, lib.clt_last_error()
    torch.cuda.synchronize()
    assert abs(float(sc[0]) - 0.25 * float((y**2).sum() / ((y - y.mean(0)) ** 2).sum())) < 1e-3
