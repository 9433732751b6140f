"""Tensor-level wrappers over the C ABI: raw device pointers + shapes + the current HIP stream.

PyTorch is used for device memory, streams and autograd plumbing only; every
function here ends in a hand-written gfx950 kernel of libcltmi355.so and raises
if the tensors are not on a GPU."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import CLT_BF16, CLT_DECODE_ACCUMULATE, CLT_DECODE_ROUND_ACTS, CLT_F32, check

_DT = {torch.float32: CLT_F32, torch.bfloat16: CLT_BF16}


def _dt(t: Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}: only float32 / bfloat16") from None


def _ptr(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(t: Tensor):
    if not t.is_cuda:
        raise RuntimeError("clt_training_amd ops need tensors on an MI355X GPU (there is no CPU fallback)")
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _c(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _idx(idx: Tensor) -> Tensor:
    return _c(idx if idx.dtype == torch.int64 else idx.to(torch.int64))


class KernelProfile:
    """Optional HIP-event timing of C-ABI calls on the stream they are launched on (bench.py)."""

    def __init__(self, names=None):
        self.names = None if names is None else set(names)
        self.records = {}
        self.units = {}  # name -> work units (e.g. row-pass jobs) summed over the recorded calls of batched entry points
        self.active = True  # bench.py switches it per step: every event pair costs ~10 us of idle GPU at a call boundary

    def add(self, name, start, end):
        self.records.setdefault(name, []).append((start, end))

    def summary(self):
        """name -> (launches, mean ms).  Call after torch.cuda.synchronize()."""
        out = {}
        for name, evs in self.records.items():
            ms = [s.elapsed_time(e) for s, e in evs]
            out[name] = (len(ms), sum(ms) / max(1, len(ms)))
        return out


PROFILE: Optional[KernelProfile] = None


def _call(name: str, *args):
    fn = getattr(_lib.load(), name)
    prof = PROFILE
    if prof is not None and prof.active and (prof.names is None or name in prof.names):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        prof.add(name, s, e)
    else:
        rc = fn(*args)
    check(rc, name)


_ws_cache: dict = {}


def workspace(device, nbytes: int, tag: str = "default") -> Tensor:
    """Grow-only scratch buffer per (device, tag, CURRENT STREAM): reuse is stream-ordered, so two streams never share
    a buffer (tp.TPTrainer runs the encoder half of the next layer on a side stream), and a buffer that is replaced by
    a larger one was allocated on the only stream that ever used it -- the caching allocator may hand it out again."""
    device = torch.device(device)
    key = (device, tag, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ----------------------------------------------------------------------------- TP candidate exchange format
def candidates_pack(vals: Tensor, idx: Tensor, idx_offset: int = 0) -> Tensor:
    """[..] values (>= 0) and int64 indices -> int64 (float32 bits of the value << 32 | idx + idx_offset): one tensor,
    one collective (engine.CLTEngine._merge_candidates)."""
    vals, idx = _c(vals), _idx(idx)
    assert vals.shape == idx.shape
    out = torch.empty(vals.shape, dtype=torch.int64, device=vals.device)
    _call("clt_candidates_pack", _ptr(vals), _dt(vals), _ptr(idx), int(idx_offset), vals.numel(), _ptr(out), _stream(vals))
    return out


def candidates_unpack(packed: Tensor, groups: int, k: int, dtype) -> tuple:
    """packed [groups * rows, k] as an all-to-all delivers it (block r = what rank r sent) -> values [rows, groups * k]
    in ``dtype`` and int64 indices, row t holding token t's candidates rank by rank."""
    packed = _c(packed)
    assert packed.dtype == torch.int64 and packed.shape[-1] == k and packed.numel() % (groups * k) == 0
    rows = packed.numel() // (groups * k)
    vals = torch.empty(rows, groups * k, dtype=dtype, device=packed.device)
    idx = torch.empty(rows, groups * k, dtype=torch.int64, device=packed.device)
    _call("clt_candidates_unpack", _ptr(packed), groups, rows, k, _ptr(vals), _dt(vals), _ptr(idx), _stream(packed))
    return vals, idx


# ----------------------------------------------------------------------------- encoder
def encoder_gemm_relu(x: Tensor, W: Tensor, b: Optional[Tensor], out_dtype=None, out: Optional[Tensor] = None) -> Tensor:
    x, W = _c(x), _c(W)
    assert x.dtype == W.dtype and x.shape[1] == W.shape[1]
    N, d = x.shape
    M = W.shape[0]
    out_dtype = out_dtype or x.dtype
    if out is None:
        out = torch.empty(N, M, dtype=out_dtype, device=x.device)
    b = None if b is None else _c(b)
    _call("clt_encoder_gemm_relu", _ptr(x), _ptr(W), _dt(x), _ptr(b), _dt(b) if b is not None else 0,
                                            _ptr(out), _DT[out_dtype], N, M, d, _stream(x))
    return out


def topk_rows(vals: Tensor, k: int, cand_idx: Optional[Tensor] = None, counts: Optional[Tensor] = None,
              out_dtype=None):
    assert vals.dim() == 2 and vals.stride(1) == 1
    N, L = vals.shape
    ld = vals.stride(0) if N > 1 else L
    out_dtype = out_dtype or vals.dtype
    ov = torch.empty(N, k, dtype=out_dtype, device=vals.device)
    oi = torch.empty(N, k, dtype=torch.int64, device=vals.device)
    if cand_idx is not None:
        cand_idx = _idx(cand_idx)
        assert cand_idx.shape == vals.shape and vals.is_contiguous()
    if counts is not None:
        counts = _c(counts.to(torch.int32))
    _call("clt_topk_rows", _ptr(vals), _dt(vals), ld, _ptr(cand_idx), _ptr(counts), N, L, k, _ptr(ov),
                                    _DT[out_dtype], _ptr(oi), _stream(vals))
    return ov, oi


def groupmax_rows(vals: Tensor, k: int, out_dtype=None):
    """Largest entry of each of the k consecutive column groups of every row (fused_encoder.py:408-413)."""
    assert vals.dim() == 2 and vals.stride(1) == 1
    N, M = vals.shape
    ld = vals.stride(0) if N > 1 else M
    out_dtype = out_dtype or vals.dtype
    ov = torch.empty(N, k, dtype=out_dtype, device=vals.device)
    oi = torch.empty(N, k, dtype=torch.int64, device=vals.device)
    _call("clt_groupmax_rows", _ptr(vals), _dt(vals), ld, N, M, k, _ptr(ov), _DT[out_dtype], _ptr(oi), _stream(vals))
    return ov, oi


def batch_threshold(v: Tensor, k: int) -> Tensor:
    """batch_topk (fused_encoder.py:204-240) on the per-row top-4k values [N, 4k]: keep the k * N largest of the batch,
    zero the rest (ties with the threshold survive, as ``preacts < threshold`` leaves them).  Small torch glue: one
    selection over N * 4k values and one masked fill, no host sync."""
    n = v.shape[0]
    thr = torch.topk(v.flatten().float(), k * n, sorted=False).values.min()
    return v.masked_fill(v.float() < thr, 0)


def encode_topk(x: Tensor, W: Tensor, b: Optional[Tensor], k: int, out_dtype=None, activation: str = "topk"):
    """relu(x W^T + b) -> row top-k (fused_encoder.py:310-322,372) or groupmax (:373-413)."""
    if activation not in ("topk", "groupmax"):
        raise NotImplementedError(f"activation {activation!r}: 'topk' and 'groupmax' are on the MI355X hot path")
    x, W = _c(x), _c(W)
    assert x.dtype == W.dtype and x.shape[1] == W.shape[1]
    N, d = x.shape
    M = W.shape[0]
    out_dtype = out_dtype or x.dtype
    lib = _lib.load()
    nbytes = lib.clt_encode_topk_workspace_bytes(N, M, _DT[out_dtype])
    ws = workspace(x.device, nbytes, "pre")
    b = None if b is None else _c(b)
    ov = torch.empty(N, k, dtype=out_dtype, device=x.device)
    oi = torch.empty(N, k, dtype=torch.int64, device=x.device)
    # one entry point: the fused sample / filter / merge launches for bf16, GEMM + row top-k otherwise
    _call("clt_encode_topk_fwd" if activation == "topk" else "clt_encode_groupmax_fwd",
          _ptr(x), _ptr(W), _dt(x), _ptr(b), _dt(b) if b is not None else 0, N, M, d, k,
          _ptr(ov), _DT[out_dtype], _ptr(oi), _ptr(ws), ws.numel(), _stream(x))
    return ov, oi


def encode_topk_layers(xbuf: Tensor, Ws, bs, k: int):
    """``encode_topk`` for several layers of one shape in ONE set of launches (clt_encode_topk_fwd_layers).
    ``xbuf`` [N, L * d]: the layers' inputs side by side (layer l = columns [l * d, (l + 1) * d)); ``Ws`` / ``bs``: the
    L encoder tables [M, d] and biases [M] (or None).  Returns values [L, N, k] and int64 indices [L, N, k], bit for bit
    those of L separate calls; None when the stacked path does not take the shape (the caller goes layer by layer)."""
    L = len(Ws)
    N, Ld = xbuf.shape
    d = Ld // L
    M = Ws[0].shape[0]
    if (L < 1 or L > 16 or Ld != L * d or N % 2048 != 0 or xbuf.dtype != torch.bfloat16 or not xbuf.is_contiguous()
            or any(W.shape != (M, d) or W.dtype != xbuf.dtype or not W.is_contiguous() for W in Ws)):
        return None
    have_b = bs is not None and bs[0] is not None
    if have_b and any(b is None or b.dtype != bs[0].dtype or not b.is_contiguous() for b in bs):
        return None
    lib = _lib.load()
    ws = workspace(xbuf.device, lib.clt_encode_topk_workspace_bytes(L * N, M, _DT[xbuf.dtype]), "pre")
    ov = torch.empty(L, N, k, dtype=xbuf.dtype, device=xbuf.device)
    oi = torch.empty(L, N, k, dtype=torch.int64, device=xbuf.device)
    wtab = (C.c_void_p * L)(*[W.data_ptr() for W in Ws])
    btab = (C.c_void_p * L)(*[b.data_ptr() for b in bs]) if have_b else None
    fn = lib.clt_encode_topk_fwd_layers
    prof = PROFILE
    timed = prof is not None and prof.active
    if timed:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    rc = fn(_ptr(xbuf), Ld, d, C.cast(wtab, C.c_void_p), _dt(xbuf), C.cast(btab, C.c_void_p) if have_b else None,
            _dt(bs[0]) if have_b else 0, L, N, M, d, k, _ptr(ov), _DT[xbuf.dtype], _ptr(oi), _ptr(ws), ws.numel(),
            _stream(xbuf))
    if rc == _lib.CLT_ERR_UNSUPPORTED:
        return None
    if timed:
        e.record()
        prof.add("clt_encode_topk_fwd_layers", s, e)
        prof.units["clt_encode_topk_fwd_layers"] = prof.units.get("clt_encode_topk_fwd_layers", 0) + L
    check(rc, "clt_encode_topk_fwd_layers")
    return ov, oi


def encode_topk_fallback_tiles(x: Tensor, M: int, out_dtype=None):
    """Diagnostics (tools/, tests; SYNCHRONISES): after an ``encode_topk(x, W[M, d], ...)`` call on this device, how many
    of its 256-token tiles went through the exact fallback (plain GEMM + row top-k) -> (flagged tiles, all tiles).
    Reads the flag area at the end of the call's workspace (csrc/gemm_fast.h)."""
    N = x.shape[0]
    out_dtype = out_dtype or x.dtype
    nbytes = _lib.load().clt_encode_topk_workspace_bytes(N, M, _DT[out_dtype])
    ntt = (N + 255) // 256
    tail = -(-(4 * (1 + 2 * ntt)) // 4096) * 4096
    ws = workspace(x.device, nbytes, "pre")
    flags = ws[nbytes - tail:nbytes - tail + 4 * (1 + ntt)].view(torch.int32)
    return int(flags[0]), ntt


# ----------------------------------------------------------------------------- decoder
def decode_fwd(idx: Tensor, v: Tensor, W: Tensor, *, post_enc: Optional[Tensor] = None,
               post_scale: Optional[Tensor] = None, bias: Optional[Tensor] = None, out: Optional[Tensor] = None,
               out_dtype=None, accumulate: bool = False, round_acts: bool = False, want_acts: bool = False,
               lat_begin: int = 0):
    idx, v, W = _idx(idx), _c(v), _c(W)
    N, k = idx.shape
    M, d = W.shape
    out_dtype = out_dtype or W.dtype
    if out is None:
        assert not accumulate
        out = torch.empty(N, d, dtype=out_dtype, device=W.device)
    acts = torch.empty(N, k, dtype=torch.float32, device=W.device) if want_acts else None
    flags = (CLT_DECODE_ACCUMULATE if accumulate else 0) | (CLT_DECODE_ROUND_ACTS if round_acts else 0)
    for t in (post_enc, post_scale, bias):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    _call("clt_decode_fwd", _ptr(idx), _ptr(v), _dt(v), _ptr(post_enc), _ptr(post_scale), _ptr(W), _dt(W), M, lat_begin, d,
                                     _ptr(bias), _ptr(out), _dt(out), flags, _ptr(acts), N, k, _stream(W))
    return (out, acts) if want_acts else out


def decode_bwd_acts(idx: Tensor, v: Tensor, W: Tensor, G: Tensor, *, post_enc: Optional[Tensor] = None,
                    post_scale: Optional[Tensor] = None, grad_acts: Optional[Tensor] = None,
                    want_e_post: bool = False, lat_begin: int = 0):
    idx, v, W, G = _idx(idx), _c(v), _c(W), _c(G)
    N, k = idx.shape
    M, d = W.shape
    accumulate = grad_acts is not None
    if grad_acts is None:
        grad_acts = torch.empty(N, k, dtype=torch.float32, device=W.device)
    e_post = torch.empty(N, k, dtype=torch.float32, device=W.device) if want_e_post else None
    e_scale = torch.empty(N, k, dtype=torch.float32, device=W.device) if post_scale is not None else None
    _call("clt_decode_bwd_acts", _ptr(idx), _ptr(v), _dt(v), _ptr(post_enc), _ptr(post_scale), _ptr(W), _dt(W),
                                          M, lat_begin, d, _ptr(G), _dt(G), _ptr(grad_acts), int(accumulate), _ptr(e_post),
                                          _ptr(e_scale), N, k, _stream(W))
    return grad_acts, e_post, e_scale


def _source_table(sources, fields):
    tab = (_lib.DecodeSource * len(sources))()
    keep = []  # the contiguous copies must outlive the launch call
    for ent, src in zip(tab, sources):
        for name in fields:
            t = src.get(name)
            if t is not None:
                t = _idx(t) if name == "idx" else _c(t)
                keep.append(t)
                setattr(ent, name, t.data_ptr())
        ent.lat_begin = int(src.get("lat_begin", 0))
        ent.weight = float(src.get("weight", 1.0))
    return tab, keep


def decode_fwd_multi(sources, *, bias: Optional[Tensor] = None, out: Optional[Tensor] = None, out_dtype=None,
                     accumulate: bool = False, round_acts: bool = False, want_acts: bool = False):
    """One launch for all live sources of a target (clt_decode_fwd_multi).  ``sources``: dicts with idx, v, W and
    optionally post_enc, post_scale, lat_begin, weight.  Returns out (and the list of a' per source)."""
    assert 1 <= len(sources)
    W0, i0, v0 = sources[0]["W"], sources[0]["idx"], sources[0]["v"]
    N, k = i0.shape
    M, d = W0.shape
    for s_ in sources:
        assert s_["W"].shape == (M, d) and s_["W"].dtype == W0.dtype and s_["idx"].shape == (N, k) and s_["v"].dtype == v0.dtype
        for nm in ("post_enc", "post_scale"):
            t = s_.get(nm)
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    out_dtype = out_dtype or W0.dtype
    if out is None:
        assert not accumulate
        out = torch.empty(N, d, dtype=out_dtype, device=W0.device)
    acts = None
    if want_acts:
        acts = [torch.empty(N, k, dtype=torch.float32, device=W0.device) for _ in sources]
        sources = [dict(s_, acts_out=a) for s_, a in zip(sources, acts)]
    flags = (CLT_DECODE_ACCUMULATE if accumulate else 0) | (CLT_DECODE_ROUND_ACTS if round_acts else 0)
    step = _lib.CLT_DECODE_MAX_SOURCES
    for c0 in range(0, len(sources), step):  # more live sources than one launch takes: accumulate group by group
        tab, keep = _source_table(sources[c0:c0 + step], ("idx", "v", "post_enc", "post_scale", "W", "acts_out"))
        _call("clt_decode_fwd_multi", C.cast(tab, C.c_void_p), len(tab), _dt(v0), _dt(W0), M, d,
              _ptr(bias) if c0 == 0 else None, _ptr(out), _dt(out), flags | (CLT_DECODE_ACCUMULATE if c0 else 0), N, k,
              _stream(W0))
        del keep
    return (out, acts) if want_acts else out


def decode_bwd_acts_multi(sources, G: Tensor):
    """One launch for all live sources of a target (clt_decode_bwd_acts_multi).  ``sources``: dicts with idx, v, W,
    optionally post_enc, post_scale, lat_begin, weight (factor on G), grad_acts (accumulated into when given),
    want_e_post.  Returns a list of (grad_acts, e_post, e_scale) per source."""
    G = _c(G)
    W0, i0, v0 = sources[0]["W"], sources[0]["idx"], sources[0]["v"]
    N, k = i0.shape
    M, d = W0.shape
    outs, full = [], []
    for s_ in sources:
        assert s_["W"].shape == (M, d) and s_["W"].dtype == W0.dtype and s_["idx"].shape == (N, k) and s_["v"].dtype == v0.dtype
        ga = s_.get("grad_acts")
        acc = ga is not None
        if ga is None:
            ga = torch.empty(N, k, dtype=torch.float32, device=W0.device)
        assert ga.dtype == torch.float32 and ga.is_contiguous()
        e_post = torch.empty(N, k, dtype=torch.float32, device=W0.device) if s_.get("want_e_post") else None
        e_scale = torch.empty(N, k, dtype=torch.float32, device=W0.device) if s_.get("post_scale") is not None else None
        outs.append((ga, e_post, e_scale))
        full.append(dict(s_, grad_acts=ga, e_post=e_post, e_scale=e_scale, _acc=acc))
    step = _lib.CLT_DECODE_MAX_SOURCES
    for c0 in range(0, len(full), step):
        grp = full[c0:c0 + step]
        tab, keep = _source_table(grp, ("idx", "v", "post_enc", "post_scale", "W", "grad_acts", "e_post", "e_scale"))
        for ent, s_ in zip(tab, grp):
            ent.accumulate_grad_acts = int(s_["_acc"])
        _call("clt_decode_bwd_acts_multi", C.cast(tab, C.c_void_p), len(tab), _dt(v0), _dt(W0), M, d, _ptr(G), _dt(G),
              N, k, _stream(W0))
        del keep
    return outs


# ----------------------------------------------------------------------------- segments
class Segments:
    """Entries (token, slot) grouped by latent: built once per source layer, shared by
    the decoder grad-W of every target and by the encoder backward."""

    __slots__ = ("perm", "offsets", "M", "k")

    def __init__(self, perm, offsets, M, k):
        self.perm, self.offsets, self.M, self.k = perm, offsets, M, k


def segments_build(idx: Tensor, M: int, lat_begin: int = 0) -> Segments:
    idx = _idx(idx)
    n = idx.numel()
    lib = _lib.load()
    perm = torch.empty(n, dtype=torch.int32, device=idx.device)
    offsets = torch.empty(M + 2, dtype=torch.int32, device=idx.device)  # [M + 1] bounds + the long-segment flag
    nbytes = lib.clt_segments_workspace_bytes(n, M)
    ws = workspace(idx.device, nbytes, "seg")
    _call("clt_segments_build", _ptr(idx), n, M, lat_begin, _ptr(perm), _ptr(offsets), _ptr(ws), ws.numel(), _stream(idx))
    return Segments(perm, offsets[:M + 1], M, idx.shape[-1])  # (the kernels read the flag behind the view)


def spt_dense(seg: Segments, vals: Tensor, dense: Tensor, *, gate: Optional[Tensor] = None, out_dtype=None,
              want_rowsum: bool = False, out: Optional[Tensor] = None):
    vals, dense = _c(vals), _c(dense)
    assert vals.dtype == torch.float32 and vals.numel() == seg.perm.numel()
    d = dense.shape[1]
    out_dtype = out_dtype or dense.dtype
    if out is None:
        out = torch.empty(seg.M, d, dtype=out_dtype, device=dense.device)
    rowsum = torch.empty(seg.M, dtype=torch.float32, device=dense.device) if want_rowsum else None
    gate = None if gate is None else _c(gate)
    n = seg.perm.numel()
    ws = workspace(dense.device, _lib.load().clt_rows_workspace_bytes(n, d), "rows")  # long-row partials
    _call("clt_spt_dense", _ptr(seg.perm), _ptr(seg.offsets), seg.M, _ptr(vals), _ptr(gate),
          _dt(gate) if gate is not None else 0, _ptr(dense), _dt(dense), d, seg.k, _ptr(out), _dt(out), _ptr(rowsum), n,
          _ptr(ws), ws.numel(), _stream(dense))
    return (out, rowsum) if want_rowsum else out


def spt_dense_adam(seg: Segments, vals: Tensor, dense: Tensor, p: Tensor, m: Tensor, v: Tensor, *, lr: float,
                   betas, eps: float, step: int, gate: Optional[Tensor] = None, want_rowsum: bool = False):
    """spt_dense whose rows feed the Adam update of (p, m, v) directly (clt_spt_dense_adam)."""
    vals, dense = _c(vals), _c(dense)
    assert vals.dtype == torch.float32 and vals.numel() == seg.perm.numel()
    d = dense.shape[1]
    assert p.shape == (seg.M, d) and p.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    assert m.dtype == p.dtype and v.dtype == p.dtype and m.shape == p.shape and v.shape == p.shape
    rowsum = torch.empty(seg.M, dtype=torch.float32, device=dense.device) if want_rowsum else None
    gate = None if gate is None else _c(gate)
    n = seg.perm.numel()
    ws = workspace(dense.device, _lib.load().clt_rows_workspace_bytes(n, d), "rows")  # long-row partials
    _call("clt_spt_dense_adam", _ptr(seg.perm), _ptr(seg.offsets), seg.M, _ptr(vals), _ptr(gate),
          _dt(gate) if gate is not None else 0, _ptr(dense), _dt(dense), d, seg.k, _ptr(p), _ptr(m), _ptr(v), _dt(p),
          _ptr(rowsum), float(lr), float(betas[0]), float(betas[1]), float(eps), int(step), n, _ptr(ws), ws.numel(),
          _stream(dense))
    return rowsum


def spt_dense_adam_multi(jobs, *, lr: float, betas, eps: float, step: int):
    """Several ``spt_dense_adam`` passes of one shape in ONE launch (clt_spt_dense_adam_multi).  ``jobs``: dicts with
    seg, vals, dense, p, m, v, optionally gate, want_rowsum.  Returns the list of row sums (None where not asked);
    bitwise the results of one spt_dense_adam call per job."""
    j0 = jobs[0]
    M, d = j0["p"].shape
    n = j0["seg"].perm.numel()
    gdt = None
    tabs, outs, keep = [], [], []
    for jb in jobs:
        seg, p, m, v = jb["seg"], jb["p"], jb["m"], jb["v"]
        vals, dense = _c(jb["vals"]), _c(jb["dense"])
        gate = None if jb.get("gate") is None else _c(jb["gate"])
        assert vals.dtype == torch.float32 and vals.numel() == seg.perm.numel() == n and seg.M == M and seg.k == j0["seg"].k
        assert p.shape == (M, d) and dense.shape[1] == d and dense.dtype == _c(j0["dense"]).dtype and p.dtype == j0["p"].dtype
        assert p.is_contiguous() and m.is_contiguous() and v.is_contiguous() and m.dtype == p.dtype == v.dtype
        assert m.shape == p.shape == v.shape
        if gate is not None:
            assert gdt is None or gate.dtype == gdt
            gdt = gate.dtype
        rowsum = torch.empty(M, dtype=torch.float32, device=p.device) if jb.get("want_rowsum") else None
        outs.append(rowsum)
        tabs.append((seg.perm, seg.offsets, vals, gate, dense, p, m, v, rowsum))
        keep.append((vals, dense, gate))
    lib = _lib.load()
    for c0 in range(0, len(tabs), _lib.CLT_ROWS_MAX_JOBS):
        grp = tabs[c0:c0 + _lib.CLT_ROWS_MAX_JOBS]
        tab = (_lib.SptJob * len(grp))()
        for ent, fields in zip(tab, grp):
            for (name, _), t in zip(_lib.SptJob._fields_, fields):
                setattr(ent, name, _ptr(t))
        ws = workspace(j0["p"].device, lib.clt_rows_multi_workspace_bytes(n, d, len(grp)), "rows_multi")
        _call("clt_spt_dense_adam_multi", C.cast(tab, C.c_void_p), len(grp), M, _DT[gdt] if gdt is not None else 0,
              _dt(keep[0][1]), d, j0["seg"].k, _dt(j0["p"]), float(lr), float(betas[0]), float(betas[1]), float(eps),
              int(step), n, _ptr(ws), ws.numel(), _stream(j0["p"]))
        if PROFILE is not None and PROFILE.active:
            PROFILE.units["clt_spt_dense_adam_multi"] = PROFILE.units.get("clt_spt_dense_adam_multi", 0) + len(grp)
    del keep
    return outs


def decode_bwd_rows(seg: Segments, acts: Tensor, v: Tensor, W: Tensor, G: Tensor, grad_acts: Tensor, *,
                    post_enc: Optional[Tensor] = None, post_scale: Optional[Tensor] = None, want_g_post: bool = False,
                    adam: Optional[dict] = None):
    """Decoder backward of one (source, target) pair in one row pass (clt_decode_bwd_rows / _adam): weight-gradient
    rows (returned, or consumed by the Adam update when ``adam`` = dict(m, v, lr, betas, eps, step) is given), the
    gradient wrt the activations accumulated into ``grad_acts`` and the post-encoder sums.
    Returns (grad_W or None, g_post or None, g_scale or None)."""
    acts, v, G = _c(acts), _c(v), _c(G)
    M, d = W.shape
    assert acts.dtype == torch.float32 and acts.numel() == seg.perm.numel() == v.numel() == grad_acts.numel()
    assert W.is_contiguous() and M == seg.M and G.shape[1] == d and G.dtype == W.dtype
    assert grad_acts.dtype == torch.float32 and grad_acts.is_contiguous()
    for t in (post_enc, post_scale):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    g_post = torch.empty(M, dtype=torch.float32, device=W.device) if want_g_post else None
    g_scale = torch.empty(M, dtype=torch.float32, device=W.device) if post_scale is not None else None
    head = (_ptr(seg.perm), _ptr(seg.offsets), M, _ptr(acts), _ptr(v), _dt(v), _ptr(post_enc), _ptr(post_scale), _ptr(G),
            _dt(G), d, seg.k)
    n = seg.perm.numel()
    ws = workspace(W.device, _lib.load().clt_rows_workspace_bytes(n, d), "rows")  # long-row partials
    if adam is None:
        gW = torch.empty(M, d, dtype=W.dtype, device=W.device)
        _call("clt_decode_bwd_rows", *head, _ptr(W), _dt(W), _ptr(gW), _ptr(grad_acts), _ptr(g_post), _ptr(g_scale),
              n, _ptr(ws), ws.numel(), _stream(W))
        return gW, g_post, g_scale
    m_, v_ = adam["m"], adam["v"]
    assert m_.dtype == W.dtype and v_.dtype == W.dtype and m_.shape == W.shape and v_.shape == W.shape
    assert m_.is_contiguous() and v_.is_contiguous()
    _call("clt_decode_bwd_rows_adam", *head, _ptr(W), _ptr(m_), _ptr(v_), _dt(W), _ptr(grad_acts), _ptr(g_post),
          _ptr(g_scale), float(adam["lr"]), float(adam["betas"][0]), float(adam["betas"][1]), float(adam["eps"]),
          int(adam["step"]), int(bool(adam.get("project", False))), n, _ptr(ws), ws.numel(), _stream(W))
    return None, g_post, g_scale


def decode_bwd_rows_adam_multi(jobs, G: Tensor):
    """The decoder backward + Adam update of every live source of one target in ONE launch
    (clt_decode_bwd_rows_adam_multi).  ``jobs``: dicts with seg, acts, v, W, grad_acts, adam (dict(m, v, lr, betas, eps,
    step[, project]) -- the same hyper-parameters in every job), optionally post_enc, post_scale, want_g_post.  All jobs
    share the table shape, k and the dtypes.  Returns a list of (g_post or None, g_scale or None) per job; bitwise the
    results of one decode_bwd_rows(..., adam=...) call per job."""
    G = _c(G)
    j0 = jobs[0]
    M, d = j0["W"].shape
    a0 = j0["adam"]
    n = j0["seg"].perm.numel()
    outs, tabs, keep = [], [], []
    for jb in jobs:
        W, seg, ad = jb["W"], jb["seg"], jb["adam"]
        acts, v, ga = _c(jb["acts"]), _c(jb["v"]), jb["grad_acts"]
        assert W.shape == (M, d) and W.dtype == G.dtype and W.is_contiguous() and seg.M == M and seg.k == j0["seg"].k
        assert acts.dtype == torch.float32 and acts.numel() == seg.perm.numel() == v.numel() == ga.numel() == n
        assert v.dtype == j0["v"].dtype and ga.dtype == torch.float32 and ga.is_contiguous()
        assert (ad["lr"], ad["betas"], ad["eps"], ad["step"], bool(ad.get("project", False))) == (
            a0["lr"], a0["betas"], a0["eps"], a0["step"], bool(a0.get("project", False))), "one optimizer group per launch"
        m_, v_ = ad["m"], ad["v"]
        assert m_.dtype == W.dtype and v_.dtype == W.dtype and m_.shape == W.shape and v_.shape == W.shape
        assert m_.is_contiguous() and v_.is_contiguous()
        pe, ps = jb.get("post_enc"), jb.get("post_scale")
        for t in (pe, ps):
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
        g_post = torch.empty(M, dtype=torch.float32, device=W.device) if jb.get("want_g_post") else None
        g_scale = torch.empty(M, dtype=torch.float32, device=W.device) if ps is not None else None
        outs.append((g_post, g_scale))
        tabs.append((seg.perm, seg.offsets, acts, v, pe, ps, W, m_, v_, ga, g_post, g_scale))
        keep.append((acts, v))
    lib = _lib.load()
    step = _lib.CLT_ROWS_MAX_JOBS
    for c0 in range(0, len(tabs), step):
        grp = tabs[c0:c0 + step]
        tab = (_lib.RowsJob * len(grp))()
        for ent, fields in zip(tab, grp):
            for (name, _), t in zip(_lib.RowsJob._fields_, fields):
                setattr(ent, name, _ptr(t))
        ws = workspace(G.device, lib.clt_rows_multi_workspace_bytes(n, d, len(grp)), "rows_multi")
        _call("clt_decode_bwd_rows_adam_multi", C.cast(tab, C.c_void_p), len(grp), M, _dt(j0["v"]), _ptr(G), _dt(G), d,
              j0["seg"].k, _dt(j0["W"]), float(a0["lr"]), float(a0["betas"][0]), float(a0["betas"][1]), float(a0["eps"]),
              int(a0["step"]), int(bool(a0.get("project", False))), n, _ptr(ws), ws.numel(), _stream(G))
        if PROFILE is not None and PROFILE.active:
            PROFILE.units["clt_decode_bwd_rows_adam_multi"] = PROFILE.units.get("clt_decode_bwd_rows_adam_multi", 0) + len(grp)
    del keep
    return outs


def rows_unit_norm(W: Tensor, eps: Optional[float] = None) -> Tensor:
    """In place W[r,:] /= (||W[r,:]|| + eps) (set_decoder_norm_to_unit_norm, sparse_coder.py:687-694)."""
    assert W.dim() == 2 and W.is_contiguous()
    eps = torch.finfo(W.dtype).eps if eps is None else eps
    _call("clt_rows_unit_norm", _ptr(W), _dt(W), W.shape[0], W.shape[1], float(eps), _stream(W))
    return W


def rows_project_out(G: Tensor, W: Tensor) -> Tensor:
    """In place G[r,:] -= (G[r,:] . W[r,:]) W[r,:] (remove_gradient_parallel_to_decoder_directions, :696-711)."""
    assert G.shape == W.shape and G.dtype == W.dtype and G.is_contiguous() and W.is_contiguous()
    _call("clt_rows_project_out", _ptr(G), _ptr(W), _dt(W), W.shape[0], W.shape[1], _stream(W))
    return G


def segment_sum(seg: Segments, a: Tensor, b: Optional[Tensor] = None):
    a = _c(a)
    assert a.dtype == torch.float32
    oa = torch.empty(seg.M, dtype=torch.float32, device=a.device)
    ob = None
    if b is not None:
        b = _c(b)
        ob = torch.empty(seg.M, dtype=torch.float32, device=a.device)
    _call("clt_segment_sum", _ptr(seg.perm), _ptr(seg.offsets), seg.M, _ptr(a), _ptr(oa), _ptr(b), _ptr(ob),
                                      _stream(a))
    return oa, ob


# ----------------------------------------------------------------------------- FVU
def fvu_fwd_bwd(y: Tensor, yhat: Tensor, mask: Optional[Tensor], gscale: float = 1.0, want_grad: bool = True,
                g_dtype=None):
    """Returns (scalars[3] = fvu, l2, tv on device, G or None)."""
    y, yhat = _c(y), _c(yhat)
    N, d = y.shape
    lib = _lib.load()
    ws = workspace(y.device, lib.clt_fvu_workspace_bytes(N, d), "fvu")
    sc = torch.empty(3, dtype=torch.float32, device=y.device)
    g_dtype = g_dtype or yhat.dtype
    G = torch.empty(N, d, dtype=g_dtype, device=y.device) if want_grad else None
    if mask is not None:
        mask = _c(mask.to(torch.bool)).view(torch.uint8)
    _call("clt_fvu_fwd_bwd", _ptr(y), _dt(y), _ptr(yhat), _dt(yhat), _ptr(mask), N, d, float(gscale), _ptr(sc), _ptr(G),
                              _DT[g_dtype], _ptr(ws), ws.numel(), _stream(y))
    return sc, G


def fvu_fwd_bwd_static(y: Tensor, yhat: Tensor, mask: Optional[Tensor], out_scalars: Tensor, G: Optional[Tensor],
                       gscale: float = 1.0) -> None:
    """clt_fvu_fwd_bwd_static: the caller passes EVERY buffer (scalars[3] fp32, G [N,d] or None, mask as bool / uint8);
    nothing is allocated here or in the library (graph capture, serving loops).  d <= 8192."""
    N, d = y.shape
    assert y.is_contiguous() and yhat.is_contiguous() and out_scalars.is_contiguous()
    assert out_scalars.dtype == torch.float32 and out_scalars.numel() >= 3
    assert mask is None or (mask.is_contiguous() and mask.element_size() == 1)
    assert G is None or (G.is_contiguous() and G.shape == y.shape)
    _call("clt_fvu_fwd_bwd_static", _ptr(y), _dt(y), _ptr(yhat), _dt(yhat), _ptr(mask), N, d, float(gscale),
          _ptr(out_scalars), _ptr(G), _dt(G) if G is not None else 0, _stream(y))


def fvu_colsum(y: Tensor, mask: Optional[Tensor]) -> Tensor:
    """Stage 1 of the token-sharded FVU: local [sum_n m*y (d values), sum_n m]."""
    y = _c(y)
    N, d = y.shape
    lib = _lib.load()
    ws = workspace(y.device, lib.clt_fvu_workspace_bytes(N, d), "fvu")
    out = torch.empty(d + 1, dtype=torch.float32, device=y.device)
    if mask is not None:
        mask = _c(mask.to(torch.bool)).view(torch.uint8)
    _call("clt_fvu_colsum", _ptr(y), _dt(y), _ptr(mask), N, d, _ptr(out), _ptr(ws), ws.numel(), _stream(y))
    return out


def fvu_partial(y: Tensor, yhat: Tensor, mask: Optional[Tensor], colsum_count: Tensor) -> Tensor:
    """Stage 2: local (l2, tv) given the global column sums / count."""
    y, yhat = _c(y), _c(yhat)
    N, d = y.shape
    lib = _lib.load()
    ws = workspace(y.device, lib.clt_fvu_workspace_bytes(N, d), "fvu")
    out = torch.empty(2, dtype=torch.float32, device=y.device)
    if mask is not None:
        mask = _c(mask.to(torch.bool)).view(torch.uint8)
    _call("clt_fvu_partial", _ptr(y), _dt(y), _ptr(yhat), _dt(yhat), _ptr(mask), N, d, _ptr(colsum_count), _ptr(out),
          _ptr(ws), ws.numel(), _stream(y))
    return out


def fvu_grad(y: Tensor, yhat: Tensor, mask: Optional[Tensor], l2tv: Tensor, gscale: float = 1.0, g_dtype=None):
    """Stage 3: scalars (fvu, l2, tv) and G = gscale * (-2 / tv) * m * (y - yhat) from the global (l2, tv)."""
    y, yhat = _c(y), _c(yhat)
    N, d = y.shape
    g_dtype = g_dtype or yhat.dtype
    sc = torch.empty(3, dtype=torch.float32, device=y.device)
    G = torch.empty(N, d, dtype=g_dtype, device=y.device)
    if mask is not None:
        mask = _c(mask.to(torch.bool)).view(torch.uint8)
    _call("clt_fvu_grad", _ptr(y), _dt(y), _ptr(yhat), _dt(yhat), _ptr(mask), N, d, _ptr(l2tv), float(gscale), _ptr(sc),
          _ptr(G), _DT[g_dtype], _stream(y))
    return sc, G
