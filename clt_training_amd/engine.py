"""Explicit (autograd-free) CLT train-step engine with latent sharding over RCCL.

This is the MI355X replacement for the reference's DTensor tensor parallelism
(sparsify/fused_encoder.py:339-370, sparsify/utils.py:479-546, sparsify/trainer.py:526-539):

  * rank r of the ``tp`` group owns latents [r*M/tp, (r+1)*M/tp): rows of ``W_enc`` / ``b_enc``, rows of
    every ``W_decs[t]``, slices of ``post_encs`` / ``post_enc_scales`` and their Adam state;
    ``b_decs`` / ``W_skips`` / ``in_norm`` / ``out_norm`` are replicated;
  * every rank brings its own N_r tokens; the encoder inputs are all-gathered over tp
    (trainer.py:535) so that the group encodes N = tp * N_r tokens;
  * local GEMM + top-k on [N, M/tp], ONE all-gather each of the candidate values and global indices,
    and the same deterministic merge top-k on every rank (fused_encoder.py:347-360);
  * each rank decodes only the selected latents it owns (compacted, not masked) into a full-width
    partial sum over ALL live source layers of a target, then ONE reduce-scatter per target layer
    leaves every rank with the reconstruction of its own tokens (the ``tp_output=False`` /
    ``Partial("sum")`` layout of utils.py:518-530);
  * FVU on token shards with two small all-reduces (column sums + count, then (l2, tv));
  * backward: the loss gradient of the local tokens is all-gathered once per target; everything a rank
    needs for its own latents is then local (no all-reduce of activation gradients: every (token, slot)
    entry is owned by exactly one rank); only the replicated ``b_decs`` / ``W_skips`` gradients are
    all-reduced, once per step, in one flat bucket.

With tp == 1 every collective is the identity and the engine is the single-GPU fast path: the same
arithmetic as CrossLayerRunner + autograd (``coalesce_topk="none"``), without the autograd graph.

The arithmetic backend is injected (``ops``): the product passes ``clt_training_amd.hip_ops`` (HIP kernels,
no CPU fallback); the CPU tests pass an oracle-backed double to exercise the sharding / collective logic
under gloo.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn

from .config import SparseCoderConfig, TrainConfig


@dataclass
class Routes:
    """Which of the engine's equivalent kernel routes a step takes.  Everything is on in the product; the A/B tests
    (fast path == single-pair kernels, fused Adam == grad -> Adam) switch routes off on this object.  Not read from
    the environment."""

    multi: bool = True       # all live sources of a target in one decode launch (clt_decode_fwd_multi)
    rows_pass: bool = True   # decoder backward as one row pass per pair (clt_decode_bwd_rows*)
    fused_adam: bool = True  # Adam applied inside the gradient kernels
    overlap: bool = True     # tp > 1: encoder half of layer j + 1 on a side stream
    overlap_single: bool = False  # the same at world 1 (no collectives to hide; measured)
    pack: bool = True        # per-step packing of the skip matrices / decoder biases of all (source, target) pairs
    rows_multi: bool = True  # the row passes of all live sources of a target in one launch (clt_decode_bwd_rows_adam_multi)
    # world 1: the encoders of all layers in one set of launches (clt_encode_topk_fwd_layers).  Built, bit-identical,
    # and measured at C2: 44.22 ms per step against 44.26 layer by layer -- the per-layer side stages have no tail worth
    # batching -- while its workspace is L times larger; off.
    enc_layers: bool = False


ROUTES = Routes()


_MM_OUT_DTYPE = None  # does torch.mm take out_dtype (bf16 operands -> fp32 result in one library GEMM)?


def _mm_f32(a: Tensor, b: Tensor) -> Tensor:
    """a @ b with an fp32 result."""
    global _MM_OUT_DTYPE
    if a.dtype == torch.float32:
        return a @ b
    if _MM_OUT_DTYPE is None:
        try:
            out = torch.mm(a, b, out_dtype=torch.float32)
            _MM_OUT_DTYPE = True
            return out
        except (TypeError, RuntimeError):
            _MM_OUT_DTYPE = False
    if _MM_OUT_DTYPE:
        return torch.mm(a, b, out_dtype=torch.float32)
    return a.float() @ b.float()  # (no out_dtype on this device / build: still one rounding-free fp32 product)


# --------------------------------------------------------------------------- collectives
class Comm:
    """Thin wrapper over one torch.distributed group (backend "nccl" = RCCL on ROCm; gloo in CPU tests)."""

    def __init__(self, group=None, enabled: Optional[bool] = None):
        self.enabled = dist.is_available() and dist.is_initialized() if enabled is None else enabled
        self.group = group
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.backend = dist.get_backend(group) if self.enabled else "none"
        # bench.py: HIP events around every collective, recorded on the stream that issues it (the stream whose work
        # waits for the result), summed per class of exchange -- the method of scripts/benchmarks/distributed.py:11-57
        # applied per collective.  Off (None) outside the benchmark: an event pair costs ~10 us of idle GPU.
        self.timing: Optional[dict] = None

    def _timed(self, tag: str, t: Tensor, fn):
        if self.timing is None or not t.is_cuda:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.timing.setdefault(tag, []).append((s, e, t.numel() * t.element_size()))
        return out

    def timing_summary(self, steps: int, timing: Optional[dict] = None) -> dict:
        """{class: {calls_per_step, ms_per_step, bytes_per_call}} of the collectives recorded in ``timing`` (default:
        the dict currently attached) over ``steps`` steps; call after a device synchronise."""
        out = {}
        for tag, evs in (timing if timing is not None else self.timing or {}).items():
            ms = sum(s.elapsed_time(e) for s, e, _ in evs)
            out[tag] = {"calls_per_step": round(len(evs) / steps, 2), "ms_per_step": round(ms / steps, 4),
                        "bytes_per_call": int(sum(b for _, _, b in evs) / max(1, len(evs)))}
        return out

    # gloo (CPU tests, and rehearsals of the multi-rank GPU path on a one-GPU box) works on host copies
    def all_gather_rows(self, t: Tensor, tag: str = "all_gather") -> Tensor:
        """[n, ...] per rank -> [world * n, ...] (rank-major)."""
        if self.world == 1:
            return t
        t = t.contiguous()
        return self._timed(tag, t, lambda: self._all_gather_rows(t))

    def _all_gather_rows(self, t: Tensor) -> Tensor:
        if self.backend == "gloo" and t.is_cuda:
            return self._all_gather_rows(t.cpu()).to(t.device)
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out

    def all_gather_cat1(self, t: Tensor, tag: str = "all_gather") -> Tensor:
        """[N, k] per rank -> [N, world * k] (candidates of every shard side by side)."""
        if self.world == 1:
            return t
        g = self.all_gather_rows(t, tag)  # [world * N, k]
        N, k = t.shape
        return g.view(self.world, N, k).permute(1, 0, 2).reshape(N, self.world * k).contiguous()

    def all_to_all_rows(self, t: Tensor, tag: str = "all_to_all") -> Tensor:
        """[world * n, ...] per rank (block r' = what this rank sends to rank r') -> [world * n, ...] whose block r'
        is what rank r' sent here."""
        if self.world == 1:
            return t
        t = t.contiguous()
        return self._timed(tag, t, lambda: self._all_to_all_rows(t))

    def _all_to_all_rows(self, t: Tensor) -> Tensor:
        if self.backend == "gloo" and t.is_cuda:
            return self._all_to_all_rows(t.cpu()).to(t.device)
        out = torch.empty_like(t)
        dist.all_to_all_single(out, t, group=self.group)
        return out

    def reduce_scatter_rows(self, t: Tensor, tag: str = "reduce_scatter") -> Tensor:
        """sum over ranks of [world * n, d], rank r keeps rows [r*n, (r+1)*n)."""
        if self.world == 1:
            return t
        t = t.contiguous()
        return self._timed(tag, t, lambda: self._reduce_scatter_rows(t))

    def _reduce_scatter_rows(self, t: Tensor) -> Tensor:
        n = t.shape[0] // self.world
        if self.backend == "gloo":  # gloo has no reduce_scatter: all-reduce and slice
            h = t.cpu() if t.is_cuda else t
            if h.dtype == torch.bfloat16:  # gloo sums bf16 in bf16: do it in fp32 like RCCL's accumulate
                h = h.float()
            dist.all_reduce(h, group=self.group)
            return h[self.rank * n:(self.rank + 1) * n].to(device=t.device, dtype=t.dtype).clone()
        out = torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.reduce_scatter_tensor(out, t, group=self.group)
        return out

    def all_reduce_(self, t: Tensor, tag: str = "all_reduce") -> Tensor:
        if self.world > 1:
            self._timed(tag, t, lambda: self._all_reduce_(t))
        return t

    def _all_reduce_(self, t: Tensor) -> None:
        if self.backend == "gloo" and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)


# --------------------------------------------------------------------------- sharded parameters
class ShardedCoder(nn.Module):
    """The parameters of one SparseCoder (sparse_coder.py:233-445) with the latent dimension sharded.
    Names follow the reference's state dict; ``W_decs`` etc. always are lists (one entry per target slot,
    a single entry for a plain SAE / transcoder)."""

    def __init__(self, d_in: int, cfg: SparseCoderConfig, device, rank: int = 0, tp: int = 1,
                 dtype: torch.dtype = torch.float32):
        super().__init__()
        if (cfg.coalesce_topk not in ("none", "concat") or cfg.topk_coalesced or cfg.per_source_tied
                or cfg.secondary_target_tied):
            raise NotImplementedError("the sharded engine implements coalesce_topk='none' and 'concat' "
                                      "(topk_coalesced=False); the other modes run on the autograd mirror")
        self.cfg, self.d_in, self.rank, self.tp = cfg, d_in, rank, tp
        self.num_latents = cfg.num_latents or d_in * cfg.expansion_factor
        assert self.num_latents % tp == 0, "latents must divide evenly over the tp ranks"
        self.m_local = self.num_latents // tp
        self.lat_begin = rank * self.m_local
        self.n_slots = max(1, cfg.n_targets if cfg.transcode else 1)  # targets this layer writes to (liveness)
        # "concat" (runner.py:93-127, sparse_coder.py:324-338): ONE decoder / bias / skip matrix per layer -- the TARGET
        # layer's -- and one post-encoder vector pair per target slot of the source
        self.n_dec = 1 if cfg.coalesce_topk == "concat" else self.n_slots
        wd = cfg.torch_dtype or dtype
        Ml, T = self.m_local, self.n_dec
        scaling = 1.0 / d_in**0.5  # nn.Linear default init == U(-1/sqrt(d), 1/sqrt(d)) (sparse_coder.py:271-284)
        w = (torch.rand(Ml, d_in, device=device, dtype=torch.float32) * (2.0 * scaling) - scaling).to(wd)
        self.W_enc = nn.Parameter(w)
        self.b_enc = nn.Parameter(torch.zeros(Ml, device=device, dtype=wd))
        if cfg.transcode:
            self.W_decs = nn.ParameterList([nn.Parameter(torch.zeros(Ml, d_in, device=device, dtype=wd)) for _ in range(T)])
        else:
            wdec = w.float()
            if cfg.normalize_decoder:
                wdec = wdec / (wdec.norm(dim=1, keepdim=True) + torch.finfo(wd).eps)
            self.W_decs = nn.ParameterList([nn.Parameter(wdec.to(wd))])
        self.b_decs = nn.ParameterList([nn.Parameter(torch.zeros(d_in, device=device, dtype=dtype)) for _ in range(T)])
        self.W_skips = nn.ParameterList(
            [nn.Parameter(torch.zeros(d_in, d_in, device=device, dtype=dtype)) for _ in range(T)]
        ) if cfg.skip_connection else None
        self.post_encs = nn.ParameterList(
            [nn.Parameter(torch.zeros(Ml, device=device, dtype=dtype), requires_grad=cfg.train_post_encoder)
             for _ in range(self.n_slots)])
        self.post_enc_scales = nn.ParameterList(
            [nn.Parameter((torch.ones if i == 0 else torch.zeros)(Ml, device=device, dtype=dtype),
                          requires_grad=cfg.train_post_encoder) for i in range(self.n_slots)]
        ) if cfg.post_encoder_scale else None
        if cfg.normalize_io:
            self.register_buffer("in_norm", torch.ones(1, device=device, dtype=dtype))
            self.register_buffer("out_norm", torch.ones(1, device=device, dtype=dtype))

    @staticmethod
    def from_full(sae, rank: int, tp: int, device=None) -> "ShardedCoder":
        """Slice an unsharded ``SparseCoder`` (used by parity tests and for loading checkpoints)."""
        device = device or sae.device
        sc = ShardedCoder(sae.d_in, sae.cfg, device, rank, tp, dtype=torch.float32)
        lo, hi = sc.lat_begin, sc.lat_begin + sc.m_local
        T = sc.n_slots
        multi = hasattr(sae, "W_decs")

        def slot(plural, single, i):
            return getattr(sae, plural)[i] if hasattr(sae, plural) else getattr(sae, single)

        with torch.no_grad():
            sc.W_enc.data = sae.encoder.weight.data[lo:hi].clone().to(device)
            sc.b_enc.data = sae.encoder.bias.data[lo:hi].clone().to(device)
            for i in range(sc.n_dec):
                sc.W_decs[i].data = slot("W_decs", "W_dec", i).data[lo:hi].clone().to(device)
                sc.b_decs[i].data = slot("b_decs", "b_dec", i).data.clone().to(device)
                if sc.W_skips is not None:
                    sc.W_skips[i].data = slot("W_skips", "W_skip", i).data.clone().to(device)
            for i in range(T):
                sc.post_encs[i].data = slot("post_encs", "post_enc", i).data[lo:hi].clone().to(device)
                if sc.post_enc_scales is not None:
                    sc.post_enc_scales[i].data = slot("post_enc_scales", "post_enc_scale", i).data[lo:hi].clone().to(device)
            if sae.cfg.normalize_io:
                sc.in_norm.copy_(sae.in_norm)
                sc.out_norm.copy_(sae.out_norm)
        del multi
        return sc

    def to_full(self, comm: Optional["Comm"] = None):
        """The inverse of ``from_full``: gather the latent shards of the group (all ranks take part in the collectives)
        into an unsharded ``SparseCoder`` on the CPU with the reference's parameter names.  Returned on rank 0 only
        (None elsewhere).  Replaces the TP branch of save_sharded (utils.py:280-360: shards gathered to rank 0)."""
        from .sparse_coder import SparseCoder

        comm = comm or Comm(enabled=False)
        assert comm.world == self.tp

        def rows(t):  # latent-sharded along dim 0
            return comm.all_gather_rows(t.detach()).cpu()

        full = {"encoder.weight": rows(self.W_enc), "encoder.bias": rows(self.b_enc)}
        for i in range(self.n_dec):
            full[f"W_decs.{i}"] = rows(self.W_decs[i])
            full[f"b_decs.{i}"] = self.b_decs[i].detach().cpu()
            if self.W_skips is not None:
                full[f"W_skips.{i}"] = self.W_skips[i].detach().cpu()
        for i in range(self.n_slots):
            full[f"post_encs.{i}"] = rows(self.post_encs[i])
            if self.post_enc_scales is not None:
                full[f"post_enc_scales.{i}"] = rows(self.post_enc_scales[i])
        if comm.rank != 0:
            return None
        sae = SparseCoder(self.d_in, self.cfg, "cpu", dtype=torch.float32)
        sd = sae.state_dict()
        with torch.no_grad():
            for name, t in full.items():
                single = name.rsplit(".", 1)[0][:-1] if name.split(".")[0].endswith("s") and name[-1].isdigit() else None
                target = name if name in sd else (single if single in sd and name.endswith(".0") else None)
                if target is None:
                    raise KeyError(f"{name}: no counterpart in the unsharded module")
                sd[target].copy_(t.to(sd[target].dtype))
            if self.cfg.normalize_io:
                sae.in_norm.copy_(self.in_norm.cpu())
                sae.out_norm.copy_(self.out_norm.cpu())
        return sae

    def save_to_disk(self, path, comm: Optional["Comm"] = None):
        """``sae.safetensors`` + ``cfg.json`` in the reference's unsharded format (sparse_coder.py:593-612)."""
        sae = self.to_full(comm)
        if sae is not None:
            sae.save_to_disk(path)

    @staticmethod
    def load_from_disk(path, device, rank: int = 0, tp: int = 1) -> "ShardedCoder":
        """Every rank reads the unsharded checkpoint (with the reference's imputation rules, SparseCoder.load_state)
        and keeps its latent range."""
        from .sparse_coder import SparseCoder

        return ShardedCoder.from_full(SparseCoder.load_from_disk(path, "cpu"), rank, tp, device)

    def replicated_parameters(self):
        out = list(self.b_decs)
        if self.W_skips is not None:
            out += list(self.W_skips)
        return out


@dataclass
class _Live:
    """One live source layer (the engine's MidDecoder, sparse_coder.py:39-101)."""

    layer: int
    xn_r: Tensor            # normalised input of the LOCAL tokens (skip connection)
    x_enc: Tensor           # encoder input of ALL tokens of the group (encoder backward)
    v: Tensor               # merged top-k values [N, k]
    idx: Tensor             # merged top-k global indices [N, k] int64
    gv: Tensor              # accumulated grad wrt v [N, k] fp32 (the detached leaf's .grad)
    slot: int = 0
    seg: object = None
    acts: dict = field(default_factory=dict)


class CLTEngine:
    """One FVU train step over all layers (trainer.py:598-682, 791-816), explicit forward and backward."""

    def __init__(self, cfg: TrainConfig, coders: list, ops, comm: Optional[Comm] = None):
        self.cfg, self.coders, self.ops = cfg, coders, ops
        self.comm = comm or Comm(enabled=False)
        self.live: list[_Live] = []
        self.last_fvu = {}
        self.last_idx = {}
        self.fused_opt = None  # an optimizer with fused_rows() (adam.Adam): see _grad_rows
        self.keep_outputs = False  # parity tests: keep each target's reconstruction of the local tokens in last_out
        self.last_out = {}

    # ---- helpers
    def _cd(self, lp: ShardedCoder):
        return lp.W_enc.dtype

    def reset(self):
        self.live = []

    @torch.no_grad()
    def begin_step(self, packed: bool = True):
        """Once per step, before the first layer: pack the replicated small parameters of ALL (source, target) pairs.
        ``W_skips`` / ``b_decs`` change only in the optimizer step at the end, so within a step one stack + one
        transposing convert give every target its ``cat_i(W_skip_i^T)`` as a contiguous row block in the compute dtype,
        and one small GEMM with the 0/1 incidence matrix gives every target's summed decoder bias -- 4 launches per
        step instead of ~5 per target.  Off (``packed=False``) in step 0, whose data-dependent init writes ``b_decs``
        layer by layer (trainer.py:547-592), and for configurations the packed route does not cover."""
        self._pack = None
        cs = self.coders
        L = len(cs)
        if not (packed and ROUTES.pack and L > 0 and all(c.W_skips is not None and c.d_in == cs[0].d_in for c in cs)
                and not any(c.cfg.divide_cross_layer or c.cfg.coalesce_topk != "none" for c in cs)):
            return
        d = cs[0].d_in
        order = [(j, i) for j in range(L) for i in range(L) if 0 <= j - i < cs[i].n_slots]  # live sources ascend
        key = tuple(order)
        if getattr(self, "_pack_key", None) != key:
            offs, inc = [], torch.zeros(L, len(order))
            for p, (j, i) in enumerate(order):
                if not offs or order[offs[-1]][0] != j:
                    offs.append(p)
                inc[j, p] = 1.0
            assert len(offs) == L, "every target layer has at least its own source"
            self._pack_key, self._pack_offs = key, offs + [len(order)]
            self._pack_inc = inc.to(cs[0].W_enc.device)
            rows = []
            for j in range(L):
                lo, hi = self._pack_offs[j], self._pack_offs[j + 1]
                p = torch.arange(lo, hi)
                rows.append((p.view(1, -1) * d + torch.arange(d).view(-1, 1)).reshape(-1))  # (b, i) -> p_i * d + b
            self._pack_rows = torch.cat(rows).to(cs[0].W_enc.device)
        cd = self._cd(cs[0])
        # Target j's skip connections as ONE K-contiguous weight [d_out, n_j * d_in] (row b = the b-th rows of its live
        # sources' W_skip side by side), so that sum_i x_i @ W_skip_i^T = X_live @ Wcat_j^T is an "NT" library GEMM on
        # the [N, n_j * d] block of live inputs.  One stack, one cast, one row gather (rows of d elements stay whole:
        # coalesced) build all L of them; the gather index is a constant of the run.
        Wb = torch.stack([cs[i].W_skips[j - i] for j, i in order]).to(cd).view(len(order) * d, d)  # rows (p, b)
        Wcat = Wb.index_select(0, self._pack_rows)                                                  # rows (j, b, i)
        bias = self._pack_inc @ torch.stack([cs[i].b_decs[j - i].float() for j, i in order])        # [L, d] fp32
        self._pack = (Wcat, bias)

    def _keep_input(self, j: int, xn_r: Tensor) -> Tensor:
        """Layer j's (normalised, local) input as a column block of one [N, L*d] buffer, so that the inputs of the
        live layers -- a contiguous range of layers -- form one strided matrix for the batched skip GEMMs."""
        L = len(self.coders)
        d = xn_r.shape[1]
        same = all(c.d_in == d and c.W_skips is not None for c in self.coders)
        if not same or xn_r.dim() != 2:
            self._xbuf = None
            return xn_r
        buf = getattr(self, "_xbuf", None)
        if buf is None or buf.shape != (xn_r.shape[0], L * d) or buf.dtype != xn_r.dtype or buf.device != xn_r.device:
            buf = self._xbuf = torch.empty(xn_r.shape[0], L * d, dtype=xn_r.dtype, device=xn_r.device)
        view = buf[:, j * d:(j + 1) * d]
        view.copy_(xn_r)
        return view

    def _live_inputs(self) -> Optional[Tensor]:
        buf = getattr(self, "_xbuf", None)
        if buf is None or not self.live:
            return None
        layers = [st.layer for st in self.live]
        if layers != list(range(layers[0], layers[0] + len(layers))):
            return None
        d = self.coders[0].d_in
        return buf[:, layers[0] * d:(layers[-1] + 1) * d]

    @torch.no_grad()
    def first_step_init(self, j: int, x_r: Tensor, y_r: Tensor):
        """trainer.py:547-592 on token shards: the means / norms are all-reduced over the group."""
        lp, comm = self.coders[j], self.comm
        d = x_r.shape[-1]
        n = torch.tensor([float(x_r.shape[0])], device=x_r.device)
        comm.all_reduce_(n)
        if lp.cfg.transcode:
            s = x_r.float().sum(0)
            comm.all_reduce_(s)
            mean = -(s / n).to(lp.W_enc.dtype)
            lp.b_enc.data[:] = torch.nn.functional.linear(mean, lp.W_enc.data)
        s = y_r.float().sum(0)
        comm.all_reduce_(s)
        lp.b_decs[0].data[:] = (s / n).to(lp.b_decs[0].dtype)
        if lp.cfg.normalize_io:
            nx = x_r.float().norm(dim=-1).sum()
            ny = y_r.float().norm(dim=-1).sum()
            comm.all_reduce_(nx)
            comm.all_reduce_(ny)
            in_norm, out_norm = nx / n, ny / n
            lp.in_norm.data[:] = in_norm
            lp.out_norm.data[:] = out_norm
            for b in lp.b_decs:
                b.data[:] = b.data * (d**0.5 / out_norm)
            lp.b_enc.data[:] = (lp.b_enc.data.float() * (d**0.5 / in_norm)).to(lp.b_enc.dtype)

    def _merge_candidates(self, v: Tensor, idx: Tensor, k: int, lat_begin: int = 0):
        """The TP top-k merge of fused_encoder.py:347-360 without replicating it: the reference all-gathers the k local
        candidates of every shard to every rank and each rank runs the second top-k over ALL tokens.  Here rank r merges
        only its own N / tp tokens: one all-to-all brings it the tp * k candidates of those tokens (value bits and global
        index packed in one int64: 8 bytes per candidate, one collective), it runs the same deterministic merge on them,
        and one all-gather of the packed winners gives every rank the merged [N, k].  Per rank ~2 * N * k * 8 * (tp-1)/tp
        bytes received instead of N * k * 10 * (tp - 1), and 1/tp of the merge work."""
        comm, ops = self.comm, self.ops
        tp = comm.world
        N = v.shape[0]
        assert N % tp == 0
        vdt = v.dtype
        if hasattr(ops, "candidates_pack"):  # two small kernels either side of each collective (clt_candidates_*)
            mine = comm.all_to_all_rows(ops.candidates_pack(v, idx, lat_begin), "candidate_all_to_all")  # [tp * N/tp, k]: block r' from rank r'
            cand_v, cand_i = ops.candidates_unpack(mine, tp, k, vdt)         # [N/tp, tp * k], rank by rank
            mv, mi = ops.topk_rows(cand_v, k, cand_idx=cand_i)
            win = comm.all_gather_rows(ops.candidates_pack(mv, mi), "candidate_all_gather")          # [N, k]
            return ops.candidates_unpack(win, 1, k, vdt)
        # (the oracle-backed ops double of the CPU tests: the same format in torch ops)
        packed = (v.float().view(torch.int32).to(torch.int64) << 32) | (idx + lat_begin)  # post-ReLU: sign bit clear
        mine = comm.all_to_all_rows(packed, "candidate_all_to_all")
        n = N // tp
        cand = mine.view(tp, n, k).permute(1, 0, 2).reshape(n, tp * k)
        cand_v = (cand >> 32).to(torch.int32).view(torch.float32).to(vdt)
        cand_i = cand & 0xFFFFFFFF
        mv, mi = ops.topk_rows(cand_v.contiguous(), k, cand_idx=cand_i.contiguous())
        win = comm.all_gather_rows((mv.float().view(torch.int32).to(torch.int64) << 32) | mi, "candidate_all_gather")  # [N, k]
        return (win >> 32).to(torch.int32).view(torch.float32).to(vdt), win & 0xFFFFFFFF

    @torch.no_grad()
    def encode_all(self, xs: list) -> Optional[list]:
        """The encoder half of EVERY layer of a step at once (world 1, transcoders of one shape, top-k, bf16): all layer
        inputs are known when the step starts and no encoder reads what a decoder or the optimizer writes within the
        step, so the L fused encode + top-k calls go out stacked as one (clt_encode_topk_fwd_layers reads the inputs in
        place from the side-by-side block that the skip GEMMs use anyway).  Returns what ``encode_part`` would return
        per layer, or None when the stacked path does not apply (the caller then encodes layer by layer)."""
        ops, cs = self.ops, self.coders
        c0 = cs[0]
        cd = self._cd(c0)
        if not (ROUTES.enc_layers and self.comm.world == 1 and len(cs) > 1 and hasattr(ops, "encode_topk_layers")
                and cd == torch.bfloat16
                and all(c.cfg.transcode and c.cfg.activation == "topk" and c.cfg.k == c0.cfg.k and c.d_in == c0.d_in
                        and c.W_enc.shape == c0.W_enc.shape and c.W_enc.dtype == cd and c.W_skips is not None
                        and (c.b_enc is None) == (c0.b_enc is None) for c in cs)
                and all(x.dim() == 2 and x.shape == xs[0].shape for x in xs) and xs[0].shape[0] % 2048 == 0):
            return None
        d = c0.d_in
        views, x_encs = [], []
        for j, (lp, x_r) in enumerate(zip(cs, xs)):
            xn_r = x_r.float() * (d**0.5 / lp.in_norm) if lp.cfg.normalize_io else x_r
            x_enc = xn_r.to(cd)
            x_encs.append(x_enc)
            views.append(self._keep_input(j, x_enc))
        buf = getattr(self, "_xbuf", None)
        if buf is None:
            return None
        out = ops.encode_topk_layers(buf, [c.W_enc for c in cs], None if c0.b_enc is None else [c.b_enc for c in cs],
                                     c0.cfg.k)
        if out is None:
            return None
        v, idx = out
        return [dict(xn_r=views[j], x_enc=x_encs[j], v=v[j], idx=idx[j], fired=None) for j in range(len(cs))]

    # ---- one hook call (trainer.py:492-682)
    @torch.no_grad()
    def encode_part(self, j: int, x_r: Tensor, y_r: Tensor) -> dict:
        """The encoder half of a hook call (sparse_coder.py:632-647, fused_encoder.py:293-372): normalise, all-gather
        the inputs of the group, local GEMM + top-k, TP merge.  It depends on nothing the decoder / backward half of
        EARLIER layers produces, so a caller may run it one layer ahead on a second stream (tp.TPTrainer does, to hide
        the collectives of either half behind the kernels of the other)."""
        ops, comm = self.ops, self.comm
        lp = self.coders[j]
        cfg = lp.cfg
        cd = self._cd(lp)
        d, k = lp.d_in, cfg.k
        if not cfg.transcode:
            x_r = y_r
        xn_r = x_r.float() * (d**0.5 / lp.in_norm) if cfg.normalize_io else x_r
        x_enc_r = xn_r if cfg.transcode else xn_r.float() - lp.b_decs[0]
        x_enc = comm.all_gather_rows(x_enc_r.to(cd), "input_all_gather")
        fired = None
        if cfg.activation == "groupmax":
            # fused_encoder.py:373-413: the k groups are runs of consecutive latents, so each shard owns k / tp whole
            # groups and the exchange is a plain concatenation (one all-gather of value bits | global index)
            assert k % comm.world == 0, "groupmax: k must be a multiple of the tp degree (fused_encoder.py:377)"
            v, idx = ops.encode_topk(x_enc, lp.W_enc, lp.b_enc, k // comm.world, out_dtype=cd, activation="groupmax")
            if comm.world > 1:
                packed = (v.float().view(torch.int32).to(torch.int64) << 32) | (idx + lp.lat_begin)
                win = comm.all_gather_cat1(packed, "candidate_all_gather")
                v, idx = (win >> 32).to(torch.int32).view(torch.float32).to(cd), win & 0xFFFFFFFF
        else:
            if cfg.activation not in ("topk", "batchtopk"):
                raise NotImplementedError(f"activation {cfg.activation!r}")
            kk = 4 * k if cfg.activation == "batchtopk" else k  # fused_encoder.py:315-318: row top-4k first
            v, idx = ops.encode_topk(x_enc, lp.W_enc, lp.b_enc, kk, out_dtype=cd)
            if comm.world > 1:
                v, idx = self._merge_candidates(v, idx, kk, lp.lat_begin)
            if cfg.activation == "batchtopk":
                # the batch-wide threshold of fused_encoder.py:204-240 on the MERGED rows (the single-device rule; every
                # rank holds the same [N, 4k] and takes the same threshold)
                v = ops.batch_threshold(v, k)
                # The zeroed slots (3 of 4) carry nothing forward or backward: hand the kernels an out-of-range index for
                # them, which the segment builder and the decode kernels drop like another shard's entries.  The
                # fired-latent bookkeeping keeps the indices as the reference would mark them (trainer.py:622-628).
                fired, idx = idx, torch.where(v > 0, idx, torch.full_like(idx, -1))
        return dict(xn_r=self._keep_input(j, xn_r.to(cd)), x_enc=x_enc, v=v, idx=idx, fired=fired)

    @torch.no_grad()
    def layer_step(self, j: int, x_r: Tensor, y_r: Tensor, mask_r: Optional[Tensor], acc_steps: int = 1,
                   enc: Optional[dict] = None) -> Tensor:
        ops, comm = self.ops, self.comm
        lp = self.coders[j]
        cfg = lp.cfg
        cd = self._cd(lp)
        d, k = lp.d_in, cfg.k
        if enc is None:
            enc = self.encode_part(j, x_r, y_r)
        x_enc, v, idx = enc["x_enc"], enc["v"], enc["idx"]
        N, k = v.shape  # (4 * cfg.k slots per row under batchtopk)
        me = _Live(j, enc["xn_r"], x_enc, v, idx, torch.zeros(N, k, dtype=torch.float32, device=v.device))
        self.live.append(me)
        if cfg.coalesce_topk == "concat":
            fvu = self._concat_target(j, me, y_r, mask_r, acc_steps)
            self._advance(acc_steps)
            self.last_fvu[j] = fvu
            self.last_idx[j] = idx
            return fvu
        div = max(1, len(self.live) - 1) if cfg.divide_cross_layer else 1  # runner.py:46-49

        # ---------------- forward: partial decode of every live source into this target
        # The skip connections and decoder biases of all live sources (sparse_coder.py:186-190, summed by
        # runner.py:46-49) are ONE GEMM: the live layers' inputs sit side by side in a persistent [N, L*d] buffer,
        # so sum_i x_i @ W_skip_i^T = X_live @ cat_i(W_skip_i^T), with 1/div folded into the other sources' blocks.
        scale = [1.0 if st is me else 1.0 / div for st in self.live]
        Wt = [self.coders[st.layer].W_decs[st.slot] for st in self.live]
        X_live = self._live_inputs()
        pack = getattr(self, "_pack", None) if X_live is not None else None
        denorm = (lp.out_norm / d**0.5) if cfg.normalize_io else None  # sparse_coder.py:627-630
        S = bias_sum = None
        if pack is not None:
            lo, hi = self._pack_offs[j], self._pack_offs[j + 1]
            assert hi - lo == len(self.live)
            # sum_i x_i @ W_skip_i^T: [N_r, d], written in fp32 by the library GEMM (bf16 operands, fp32 accumulate and
            # output), so nothing of the reconstruction is rounded to bf16 before the loss
            S = _mm_f32(X_live, pack[0][lo * d:hi * d].view(d, (hi - lo) * d).t())
            bias_sum = pack[1][j]
        multi = (hasattr(ops, "decode_fwd_multi") and all(w.shape == Wt[0].shape and w.dtype == Wt[0].dtype for w in Wt)
                 and all(st.idx.shape == me.idx.shape and st.v.dtype == me.v.dtype for st in self.live)  # per_layer_k
                 and ROUTES.multi)
        yhat_r = None
        if multi:  # every live source in ONE launch: fp32 sum over all sources, one rounding, one output write
            srcs = []
            for st, sc_ in zip(self.live, scale):
                li = self.coders[st.layer]
                t = st.slot
                srcs.append(dict(idx=st.idx, v=st.v, W=li.W_decs[t], post_enc=li.post_encs[t],
                                 post_scale=li.post_enc_scales[t] if li.post_enc_scales is not None else None,
                                 lat_begin=li.lat_begin, weight=sc_))
            if S is not None and comm.world == 1 and denorm is None:
                # one GPU: the decode launch accumulates ONTO the fp32 skip-connection output and adds the summed
                # biases -- the whole reconstruction in fp32 inside one kernel; no elementwise kernels at all
                yhat_r, acts_l = ops.decode_fwd_multi(srcs, bias=bias_sum, out=S, accumulate=True,
                                                      round_acts=Wt[0].dtype == torch.bfloat16, want_acts=True)
            else:
                P, acts_l = ops.decode_fwd_multi(srcs, round_acts=Wt[0].dtype == torch.bfloat16, want_acts=True)
            for st, a in zip(self.live, acts_l):
                st.acts[st.slot] = a
        else:
            P_cur = P_oth = None
            for st in self.live:
                li = self.coders[st.layer]
                t = st.slot
                ps = li.post_enc_scales[t] if li.post_enc_scales is not None else None
                is_cur = st is me
                acc_into = None if is_cur else P_oth
                out, acts = ops.decode_fwd(st.idx, st.v, li.W_decs[t], post_enc=li.post_encs[t], post_scale=ps,
                                           round_acts=li.W_decs[t].dtype == torch.bfloat16, want_acts=True,
                                           lat_begin=li.lat_begin, out=acc_into, accumulate=acc_into is not None)
                st.acts[t] = acts
                if is_cur:
                    P_cur = out
                else:
                    P_oth = out
            P = P_cur if P_oth is None else (P_cur + P_oth if div == 1 else P_cur + P_oth / div)
        if yhat_r is None:
            if S is not None:
                E = S + bias_sum
            else:
                bs = torch.stack([self.coders[st.layer].b_decs[st.slot].float() for st in self.live])  # [n_live, d]
                if div != 1:
                    bs = bs * torch.tensor(scale, dtype=torch.float32, device=bs.device).unsqueeze(1)
                E = bs.sum(0, keepdim=True)
                if X_live is not None:
                    Ws = torch.cat([self.coders[st.layer].W_skips[st.slot].mT for st in self.live], dim=0)  # [n_live*d, d]
                    if div != 1:
                        Ws = Ws * torch.tensor(scale, dtype=Ws.dtype, device=Ws.device).repeat_interleave(d).unsqueeze(1)
                    E = E + (X_live @ Ws.to(cd)).float()
                else:
                    for st, sc_ in zip(self.live, scale):
                        li = self.coders[st.layer]
                        if li.W_skips is not None:
                            E = E + (st.xn_r @ li.W_skips[st.slot].to(cd).mT).float() * sc_  # sparse_coder.py:190
            yhat_r = comm.reduce_scatter_rows(P, "output_reduce_scatter").float() + E
            if denorm is not None:
                yhat_r = yhat_r * denorm
        if self.keep_outputs:
            self.last_out[j] = yhat_r

        # ---------------- FVU on token shards (sparse_coder.py:207-221)
        colsum = comm.all_reduce_(ops.fvu_colsum(y_r, mask_r), "fvu_all_reduce")
        l2tv = comm.all_reduce_(ops.fvu_partial(y_r, yhat_r, mask_r, colsum), "fvu_all_reduce")
        sc, G_r = ops.fvu_grad(y_r, yhat_r, mask_r, l2tv, gscale=1.0 / acc_steps, g_dtype=cd)
        fvu = sc[0]

        # ---------------- backward
        if denorm is not None:
            G_r = (G_r.float() * denorm).to(cd)
        G = comm.all_gather_rows(G_r, "loss_gradient_all_gather")
        G_div = G if div == 1 else (G.float() / div).to(cd)
        G_r_div = G_r if div == 1 else (G_r.float() / div).to(cd)
        gb = G_r.sum(0, dtype=torch.float32)  # d b_dec: the same column sums for every live source (x 1/div)
        gWs = (G_r.mT @ X_live) if X_live is not None else None  # [d, n_live*d]: block i is d W_skip of source i
        gWs32 = None
        if gWs is not None and div == 1:  # one [n_live, d, d] fp32 copy per target instead of a strided cast per source
            nl = len(self.live)
            gWs32 = torch.empty(nl, d, d, dtype=torch.float32, device=gWs.device).copy_(gWs.view(d, nl, d).permute(1, 0, 2))
        # Decoder backward.  Preferred: ONE row pass per (source, target) pair (ops.decode_bwd_rows) that yields the
        # weight-gradient rows (consumed by the Adam update on the spot when the optimizer allows it), the gradient
        # wrt the activations and the post-encoder sums -- no second gather of decoder rows per token.  Otherwise:
        # the gradient wrt the activations of every live source in one launch, then grad-W rows and segment sums.
        rows_pass = (hasattr(ops, "decode_bwd_rows") and all(w.dtype == G.dtype for w in Wt)
                     and ROUTES.rows_pass)
        bwd_multi = None
        if not rows_pass and multi and hasattr(ops, "decode_bwd_acts_multi"):  # grad wrt the activations of every live source: ONE launch
            srcs = []
            for st, sc_ in zip(self.live, scale):
                li = self.coders[st.layer]
                t = st.slot
                srcs.append(dict(idx=st.idx, v=st.v, W=li.W_decs[t], post_enc=li.post_encs[t],
                                 post_scale=li.post_enc_scales[t] if li.post_enc_scales is not None else None,
                                 lat_begin=li.lat_begin, weight=sc_, grad_acts=st.gv,
                                 want_e_post=li.post_encs[t].requires_grad))
            bwd_multi = ops.decode_bwd_acts_multi(srcs, G)
        # the row passes of all live sources in ONE launch when they share everything but their tables: same G (no
        # 1/div), same table shape / dtype, same k, and every table's update is fused into the pass
        batch_rows = (rows_pass and multi and div == 1 and len(self.live) > 1 and ROUTES.rows_multi
                      and hasattr(ops, "decode_bwd_rows_adam_multi"))
        jobs = []
        for n_st, st in enumerate(self.live):
            li = self.coders[st.layer]
            t = st.slot
            is_cur = st is me
            Gi, Gi_r = (G, G_r) if is_cur else (G_div, G_r_div)
            self._acc_grad(li.b_decs[t], gb if is_cur or div == 1 else gb / div)
            if li.W_skips is not None:
                if gWs32 is not None:
                    self._acc_grad(li.W_skips[t], gWs32[n_st])
                elif gWs is not None:
                    blk = gWs[:, n_st * d:(n_st + 1) * d]
                    self._acc_grad(li.W_skips[t], blk if is_cur or div == 1 else blk / div)
                else:
                    self._acc_grad(li.W_skips[t], (Gi_r.mT @ st.xn_r).float())
            ps = li.post_enc_scales[t] if li.post_enc_scales is not None else None
            need_pe = li.post_encs[t].requires_grad
            if st.seg is None:
                st.seg = ops.segments_build(st.idx, li.m_local, lat_begin=li.lat_begin)
            if rows_pass:
                W = li.W_decs[t]
                h = self._fused_state(W, acc_steps)
                if h is not None and li.cfg.normalize_decoder and not li.cfg.transcode:
                    # unit-norm SAE decoder (sparse_coder.py:696-711): the finished gradient row is projected inside the
                    # pass; without the fused optimizer the caller projects W.grad before its step (tp.TPTrainer.step)
                    h = dict(h, project=True)
                if batch_rows and h is not None:
                    jobs.append(dict(seg=st.seg, acts=st.acts.pop(t).view(-1), v=st.v, W=W.data, grad_acts=st.gv,
                                     post_enc=li.post_encs[t], post_scale=ps, want_g_post=need_pe, adam=h,
                                     _pe=li.post_encs[t] if need_pe else None,
                                     _ps=ps if need_pe and ps is not None and ps.requires_grad else None))
                    continue
                gW, gp, gs = ops.decode_bwd_rows(st.seg, st.acts.pop(t).view(-1), st.v, W.data, Gi, st.gv,
                                                 post_enc=li.post_encs[t], post_scale=ps, want_g_post=need_pe, adam=h)
                if gW is not None:
                    self._acc_grad(W, gW)
            else:
                if bwd_multi is not None:
                    _, e_post, e_scale = bwd_multi[n_st]
                else:
                    _, e_post, e_scale = ops.decode_bwd_acts(st.idx, st.v, li.W_decs[t], Gi, post_enc=li.post_encs[t],
                                                             post_scale=ps, grad_acts=st.gv, want_e_post=need_pe,
                                                             lat_begin=li.lat_begin)
                self._grad_rows(li.W_decs[t], st.seg, st.acts.pop(t).view(-1), Gi, acc_steps)
                gp = gs = None
                if need_pe:
                    if ps is not None and ps.requires_grad:
                        gp, gs = ops.segment_sum(st.seg, e_post.view(-1), e_scale.view(-1))
                    else:
                        gp, _ = ops.segment_sum(st.seg, e_post.view(-1))
            if need_pe:
                if ps is not None and ps.requires_grad:
                    self._acc_grad(ps, gs)
                self._acc_grad(li.post_encs[t], gp)
        # one launch per optimizer setting: every source is a parameter group of its own (its lr may differ, and a
        # resumed run restores it per group), and a launch carries ONE set of Adam hyper-parameters
        by_group: dict = {}
        for jb in jobs:
            ad = jb["adam"]
            key = (float(ad["lr"]), tuple(ad["betas"]), float(ad["eps"]), int(ad["step"]), bool(ad.get("project", False)))
            by_group.setdefault(key, []).append(jb)
        for grp in by_group.values():
            for jb, (gp, gs) in zip(grp, ops.decode_bwd_rows_adam_multi(grp, G)):
                if jb["_ps"] is not None:
                    self._acc_grad(jb["_ps"], gs)
                if jb["_pe"] is not None:
                    self._acc_grad(jb["_pe"], gp)

        self._advance(acc_steps)
        self.last_fvu[j] = fvu
        self.last_idx[j] = idx if enc.get("fired") is None else enc["fired"]  # merged global indices (trainer.py:622-628)
        return fvu

    @torch.no_grad()
    def _concat_target(self, j: int, me: _Live, y_r: Tensor, mask_r: Optional[Tensor], acc_steps: int) -> Tensor:
        """``coalesce_topk="concat"`` with ``topk_coalesced=False`` (runner.py:52-55, 93-127; BASELINE configs[4]): the
        candidates of every live source -- post-encoder affine of the source's slot for this target applied -- form ONE
        bag of k * n_live entries per token, decoded with the TARGET layer's single ``W_dec``; only the target's
        ``b_dec`` / ``W_skip`` apply.  Forward: one fan-out launch whose sources all name the same table.  Backward: the
        concatenated entries get one segment build and ONE row pass over the table (weight gradient -> Adam in the
        pass, and the gradient wrt every entry); the per-source affine backward is elementwise on [N, k]."""
        ops, comm = self.ops, self.comm
        lp = self.coders[j]
        cfg = lp.cfg
        cd = self._cd(lp)
        d = lp.d_in
        W = lp.W_decs[0]
        denorm = (lp.out_norm / d**0.5) if cfg.normalize_io else None
        srcs = []
        for st in self.live:
            li = self.coders[st.layer]
            t = st.slot
            srcs.append(dict(idx=st.idx, v=st.v, W=W, post_enc=li.post_encs[t],
                             post_scale=li.post_enc_scales[t] if li.post_enc_scales is not None else None,
                             lat_begin=lp.lat_begin, weight=1.0))
        S = _mm_f32(me.xn_r, lp.W_skips[0].to(cd).t()) if lp.W_skips is not None else None  # sparse_coder.py:184-190
        bias = lp.b_decs[0].float()
        rb = W.dtype == torch.bfloat16
        if S is not None and comm.world == 1 and denorm is None:
            yhat_r, acts_l = ops.decode_fwd_multi(srcs, bias=bias, out=S, accumulate=True, round_acts=rb, want_acts=True)
        else:
            P, acts_l = ops.decode_fwd_multi(srcs, round_acts=rb, want_acts=True)
            yhat_r = comm.reduce_scatter_rows(P, "output_reduce_scatter").float() + (bias if S is None else S + bias)
            if denorm is not None:
                yhat_r = yhat_r * denorm
        if self.keep_outputs:
            self.last_out[j] = yhat_r
        colsum = comm.all_reduce_(ops.fvu_colsum(y_r, mask_r), "fvu_all_reduce")
        l2tv = comm.all_reduce_(ops.fvu_partial(y_r, yhat_r, mask_r, colsum), "fvu_all_reduce")
        sc, G_r = ops.fvu_grad(y_r, yhat_r, mask_r, l2tv, gscale=1.0 / acc_steps, g_dtype=cd)
        if denorm is not None:
            G_r = (G_r.float() * denorm).to(cd)
        G = comm.all_gather_rows(G_r, "loss_gradient_all_gather")
        self._acc_grad(lp.b_decs[0], G_r.sum(0, dtype=torch.float32))
        if lp.W_skips is not None:
            self._acc_grad(lp.W_skips[0], _mm_f32(G_r.mT, me.xn_r))
        # ---- one bag per token: [N, k * n_live]
        idx_cat = torch.cat([st.idx for st in self.live], 1)
        acts_cat = torch.cat(acts_l, 1)
        v_cat = torch.cat([st.v for st in self.live], 1)
        seg = ops.segments_build(idx_cat, lp.m_local, lat_begin=lp.lat_begin)
        ga_cat = torch.zeros(idx_cat.shape, dtype=torch.float32, device=idx_cat.device)  # stays 0 for other shards' entries
        if W.dtype == G.dtype and ROUTES.rows_pass and hasattr(ops, "decode_bwd_rows"):
            gW, _, _ = ops.decode_bwd_rows(seg, acts_cat.view(-1), v_cat, W.data, G, ga_cat, adam=self._fused_state(W, acc_steps))
            if gW is not None:
                self._acc_grad(W, gW)
        else:
            ops.decode_bwd_acts(idx_cat, v_cat, W, G, grad_acts=ga_cat, lat_begin=lp.lat_begin)
            self._grad_rows(W, seg, acts_cat.view(-1), G, acc_steps)
        # ---- per source: backward of a' = (v + p[idx] * (v > 0)) * s[idx]  (sparse_coder.py:136-140)
        off = 0
        for st in self.live:
            li = self.coders[st.layer]
            t = st.slot
            k = st.idx.shape[1]
            ga = ga_cat[:, off:off + k]
            off += k
            loc = (st.idx - li.lat_begin).clamp_(0, li.m_local - 1)  # (entries of other shards carry ga = 0)
            vf = st.v.float()
            pos = (vf > 0).float()
            ps = li.post_enc_scales[t] if li.post_enc_scales is not None else None
            u = ga * ps[loc] if ps is not None else ga
            st.gv += u
            if li.post_encs[t].requires_grad:
                if st.seg is None:
                    st.seg = ops.segments_build(st.idx, li.m_local, lat_begin=li.lat_begin)
                if ps is not None and ps.requires_grad:
                    gp, gs = ops.segment_sum(st.seg, (u * pos).view(-1), (ga * (vf + li.post_encs[t][loc] * pos)).reshape(-1))
                    self._acc_grad(ps, gs)
                else:
                    gp, _ = ops.segment_sum(st.seg, (u * pos).view(-1))
                self._acc_grad(li.post_encs[t], gp)
        return sc[0]

    @torch.no_grad()
    def _advance(self, acc_steps: int):
        """Advance every live source to its next target; deferred encoder backward of the sources that just wrote their
        last one (runner.py:56-59, 193-197; fused_encoder.py:92-155)."""
        ops = self.ops
        still = []
        enc_jobs = []  # encoder row passes of the sources that finish now, batched into one launch when several do
        fo = self.fused_opt
        batch_enc = (fo is not None and acc_steps == 1 and ROUTES.rows_multi and hasattr(ops, "spt_dense_adam_multi"))
        for st in self.live:
            li = self.coders[st.layer]
            st.slot += 1
            if st.slot >= li.n_slots:
                if not li.cfg.transcode:
                    # SAE: the encoder input is xn - b_dec (sparse_coder.py:637-638), so b_dec also gets -sum_n grad_x.
                    # grad_x[n,:] = sum_j gv[n,j] W_enc[idx[n,j],:] (fused_encoder.py:100-105) is never formed: its
                    # column sum is  sum_r (sum of the gated gv over latent r's segment) * W_enc[r,:] = grad_b_enc @ W_enc,
                    # one pass over W_enc instead of a gather of k rows per token (before W_enc's rows are updated in
                    # place by the fused optimizer step below).  bf16 tables: the fp32 vector goes in as hi + lo parts.
                    gbe, _ = ops.segment_sum(st.seg, (st.gv * (st.v > 0)).view(-1))
                    if li.W_enc.dtype == torch.float32:
                        sx = gbe.unsqueeze(0) @ li.W_enc
                    else:
                        hi = gbe.to(li.W_enc.dtype)
                        lo = (gbe - hi.float()).to(li.W_enc.dtype)
                        sx = (torch.stack([hi, lo]) @ li.W_enc).float().sum(0, keepdim=True)
                    self._acc_grad(li.b_decs[0], -sx.view(-1).float())
                h = fo.fused_rows(li.W_enc) if batch_enc and li.W_enc.grad is None else None
                if h is not None:
                    enc_jobs.append((li, h, dict(seg=st.seg, vals=st.gv.view(-1), dense=st.x_enc, p=li.W_enc.data, m=h["m"],
                                                 v=h["v"], gate=st.v.contiguous().view(-1), want_rowsum=True)))
                    continue
                gb = self._grad_rows(li.W_enc, st.seg, st.gv.view(-1), st.x_enc, acc_steps,
                                     gate=st.v.contiguous().view(-1), want_rowsum=True)
                self._acc_grad(li.b_enc, gb.to(li.b_enc.dtype))
            else:
                still.append(st)
        self.live = still
        if enc_jobs:
            same = all(j[2]["p"].shape == enc_jobs[0][2]["p"].shape and j[2]["seg"].k == enc_jobs[0][2]["seg"].k
                       and (j[1]["lr"], j[1]["betas"], j[1]["eps"]) == (enc_jobs[0][1]["lr"], enc_jobs[0][1]["betas"],
                                                                        enc_jobs[0][1]["eps"]) for j in enc_jobs)
            if len(enc_jobs) > 1 and same:
                h0 = enc_jobs[0][1]
                sums = ops.spt_dense_adam_multi([j[2] for j in enc_jobs], lr=h0["lr"], betas=h0["betas"], eps=h0["eps"],
                                                step=h0["step"])
            else:
                sums = [ops.spt_dense_adam(j["seg"], j["vals"], j["dense"], j["p"], j["m"], j["v"], lr=h["lr"],
                                           betas=h["betas"], eps=h["eps"], step=h["step"], gate=j["gate"], want_rowsum=True)
                        for _, h, j in enc_jobs]
            for (li, _, _), gb in zip(enc_jobs, sums):
                self._acc_grad(li.b_enc, gb.to(li.b_enc.dtype))

    def _fused_state(self, p: Tensor, acc_steps: int):
        """Optimizer state / hyper-parameters for applying this step's update of ``p`` inside the gradient kernel, or
        None when the gradient has another consumer (accumulation over micro-steps, an existing .grad, no optimizer)."""
        fo = self.fused_opt
        if fo is None or acc_steps != 1 or p.grad is not None:
            return None
        return fo.fused_rows(p)

    def _grad_rows(self, p: Tensor, seg, vals: Tensor, dense: Tensor, acc_steps: int, gate: Optional[Tensor] = None,
                   want_rowsum: bool = False):
        """Row gradient sparse^T @ dense of a [M_local, d] parameter.  When an optimizer that can take it is attached
        (self.fused_opt) and this is the step's only contribution, the rows go straight into the Adam update and
        no .grad is materialised; otherwise into p.grad.  Returns the row sums if asked."""
        fo = self.fused_opt
        if fo is not None and acc_steps == 1 and p.grad is None and hasattr(self.ops, "spt_dense_adam"):
            h = fo.fused_rows(p)
            if h is not None:
                return self.ops.spt_dense_adam(seg, vals, dense, p.data, h["m"], h["v"], lr=h["lr"], betas=h["betas"],
                                               eps=h["eps"], step=h["step"], gate=gate, want_rowsum=want_rowsum)
        if want_rowsum:
            gW, rs = self.ops.spt_dense(seg, vals, dense, gate=gate, out_dtype=p.dtype, want_rowsum=True)
        else:
            gW, rs = self.ops.spt_dense(seg, vals, dense, gate=gate, out_dtype=p.dtype), None
        self._acc_grad(p, gW)
        return rs

    @staticmethod
    def _acc_grad(p: Tensor, g: Tensor):
        if g.dtype != p.dtype:
            g = g.to(p.dtype)
        if p.grad is None:
            p.grad = g
        else:
            p.grad += g

    @torch.no_grad()
    def sync_replicated_grads(self):
        """One flat all-reduce for the gradients of the replicated parameters (b_decs, W_skips)."""
        if self.comm.world == 1:
            return
        grads = [p.grad for c in self.coders for p in c.replicated_parameters() if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1).float() for g in grads])
        self.comm.all_reduce_(flat, "replicated_grads_all_reduce")
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
