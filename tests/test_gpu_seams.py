"""Every function of the reference's operator seam that this package mirrors, called the way the
reference's own callers call it, forward and backward through autograd, against the decoder goldens the reference
produced (eager F.embedding_bag route, and its Triton embedding-bag / COO kernels under the interpreter):

  kernels.triton_sparse_transpose_dense_matmul   kernels.py:14-54     <- golden spT
  kernels.triton_sparse_dense_matmul             kernels.py:219-260   <- golden y
  kernels.triton_dense_dense_sparseout_matmul    kernels.py:332-383   <- golden gacts
  kernels.TritonDecoder.apply                    kernels.py:628-662   <- golden y, gW, gacts
  utils.decoder_impl / triton_decode / eager_decode, utils.HipDecode  utils.py:441-457, xformers.py:234-263
                                                                      <- golden y(_xf), gW(_xf), gacts(_xf)
  SparseCoder.decode                             sparse_coder.py:649-661  <- golden y + b_dec
fp32, rtol 1e-5."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP32 = dict(rtol=1e-5, atol=1e-6)
NAMES = ["decoder_d48", "decoder_d64", "decoder_c"]


def dev(t, *names):
    return [t[n].to(DEV) for n in names]


@pytest.mark.parametrize("name", NAMES)
def test_kernels_functions(name):
    from clt_training_amd import kernels as K

    t, meta = load_golden(name)
    W, acts, idx, G = dev(t, "W", "acts", "idx", "G")
    spt = K.triton_sparse_transpose_dense_matmul(idx, acts.float(), G, N=meta["M"])
    assert spt.dtype == torch.float32 and spt.shape == (meta["M"], meta["d"])
    torch.testing.assert_close(spt.cpu(), t["spT"], **FP32)
    y = K.triton_sparse_dense_matmul(idx, acts, W)
    assert y.dtype == acts.dtype
    torch.testing.assert_close(y.cpu(), t["y"], **FP32)
    ga = K.triton_dense_dense_sparseout_matmul(G, W.T, idx)  # dense2 = W_dec.T: [d, M] with stride(0) == 1
    torch.testing.assert_close(ga.cpu(), t["gacts"], **FP32)


@pytest.mark.parametrize("name", NAMES)
def test_triton_decoder_autograd(name):
    from clt_training_amd.kernels import TritonDecoder

    t, meta = load_golden(name)
    W, acts, idx, G = dev(t, "W", "acts", "idx", "G")
    for as_view in (True, False):
        Wl = W.clone().requires_grad_()
        a = acts.clone().requires_grad_()
        # the reference passes the W_dec.T VIEW ([d, M], stride(0) == 1; kernels.py:628-640); a materialised [d, M] works too
        Wt = Wl.T if as_view else Wl.T.contiguous()
        y = TritonDecoder.apply(idx, a, Wt)
        torch.testing.assert_close(y.detach().cpu(), t["y"], **FP32)
        y.backward(G)  # three inputs, three gradients
        torch.testing.assert_close(Wl.grad.cpu(), t["gW"], **FP32)
        torch.testing.assert_close(a.grad.cpu(), t["gacts"], **FP32)


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("fn", ["decoder_impl", "triton_decode", "eager_decode", "hip_decode"])
def test_decoder_impl_autograd(name, fn):
    from clt_training_amd import utils as U

    t, meta = load_golden(name)
    W, acts, idx, G = dev(t, "W", "acts", "idx", "G")
    W = W.clone().requires_grad_()
    a = acts.clone().requires_grad_()
    y = getattr(U, fn)(idx, a, W)
    assert y.dtype == W.dtype and y.shape == (meta["N"], meta["d"])
    for ref in ("y", "y_xf"):
        torch.testing.assert_close(y.detach().cpu(), t[ref], **FP32)
    y.backward(G)
    for sfx in ("", "_xf"):
        torch.testing.assert_close(W.grad.cpu(), t[f"gW{sfx}"], **FP32)  # dense [M, d]: every row written
        torch.testing.assert_close(a.grad.cpu(), t[f"gacts{sfx}"], **FP32)


@pytest.mark.parametrize("name", NAMES)
def test_sparse_coder_decode(name):
    """SparseCoder.decode(top_acts, top_indices, index) = decoder_impl(...) + b_dec (sparse_coder.py:649-661)."""
    from clt_training_amd import SparseCoder, SparseCoderConfig

    t, meta = load_golden(name)
    W, acts, idx, G = dev(t, "W", "acts", "idx", "G")
    cfg = SparseCoderConfig(transcode=True, num_latents=meta["M"], k=meta["k"])
    sae = SparseCoder(meta["d"], cfg, DEV, dtype=torch.float32)
    g = torch.Generator().manual_seed(0)
    b = torch.randn(meta["d"], generator=g)
    with torch.no_grad():
        sae.W_dec.copy_(W)
        sae.b_dec.copy_(b.to(DEV))
    a = acts.clone().requires_grad_()
    y = sae.decode(a, idx)
    torch.testing.assert_close(y.detach().cpu(), t["y"] + b, **FP32)
    y.backward(G)
    torch.testing.assert_close(sae.W_dec.grad.cpu(), t["gW"], **FP32)
    torch.testing.assert_close(sae.b_dec.grad.cpu(), t["G"].sum(0), **FP32)
    torch.testing.assert_close(a.grad.cpu(), t["gacts"], **FP32)
