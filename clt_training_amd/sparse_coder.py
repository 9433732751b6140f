"""SparseCoder / MidDecoder / ForwardOutput: the MI355X mirror of sparsify/sparse_coder.py.

Same constructor, parameter names, shapes, dtypes and initialisation
(sparse_coder.py:233-445), same ``encode`` / ``decode`` / ``forward`` and the same
MidDecoder protocol (``detach`` / ``restore`` / ``next`` / ``__call__``,
sparse_coder.py:39-229), so state dicts and call sites are interchangeable.  The arithmetic
runs in the HIP kernels: the post-encoder affine, the sparse decode and the decoder bias are
one fused kernel (forward and backward), FVU is a fused loss kernel.  Tensor parallelism is
provided by ``clt_training_amd.tp`` (RCCL), not by DTensor: ``mesh`` must be None here."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import torch
from torch import Tensor, nn

from . import fused_state
from . import hip_ops as ops
from .config import SparseCoderConfig
from .fused_encoder import EncoderOutput, fused_encoder
from .utils import decoder_impl, holder_of


@dataclass
class ForwardOutput:
    """sparse_coder.py:22-36."""

    sae_out: Tensor
    latent_acts: Tensor
    latent_indices: Tensor
    fvu: Tensor
    is_last: bool = False


# The reference wraps SparseCoder.forward and MidDecoder.__call__ in torch.autocast("cuda", bfloat16)
# whenever the GPU supports bf16 (sparse_coder.py:143-147, 672-676).  Set to False (or export
# CLT_NO_AUTOCAST=1) to run fp32 parameters through exact-fp32 kernels, e.g. for 1e-5 parity checks.
AUTOCAST_ENABLED = os.environ.get("CLT_NO_AUTOCAST", "0") != "1"


def _autocast_dtype(default: torch.dtype) -> torch.dtype:
    return torch.bfloat16 if AUTOCAST_ENABLED else default


class FusedDecode(torch.autograd.Function):
    """post-encoder affine (sparse_coder.py:136-140) + decoder_impl (:660) + b_dec (:661)
    as one kernel each way.  Gradients: latent_acts (leaf of MidDecoder.detach), post_enc,
    post_enc_scale, W_dec (dense [M, d], atomic-free via the shared segments), b_dec."""

    @staticmethod
    def forward(ctx, latent_acts, indices, post_enc, post_scale, W_dec, b_dec, round_acts: bool):
        out, acts = ops.decode_fwd(indices, latent_acts, W_dec, post_enc=post_enc, post_scale=post_scale, bias=b_dec,
                                   round_acts=round_acts, want_acts=True)
        fused_state.note_use(W_dec)
        ctx.save_for_backward(latent_acts, indices, post_enc, post_scale, W_dec, acts)
        ctx.holder = holder_of(indices)
        ctx.has_bias = b_dec is not None
        return out

    @staticmethod
    @torch.no_grad()
    def backward(ctx, G):
        latent_acts, indices, post_enc, post_scale, W_dec, acts = ctx.saved_tensors
        G = G.contiguous()
        need_acts, _, need_pe, need_ps, need_W, need_b, _ = ctx.needs_input_grad
        g_acts = g_pe = g_ps = g_W = g_b = None
        seg = None
        if need_W and need_acts and G.dtype == W_dec.dtype and W_dec.is_contiguous():
            # ONE row pass (clt_decode_bwd_rows): weight gradient, gradient wrt the activations and the post-encoder
            # sums, without gathering the k decoder rows per token a second time; long segments are load-balanced
            seg = ctx.holder.get(indices, W_dec.shape[0])
            ga = torch.zeros(indices.shape, dtype=torch.float32, device=G.device)
            g_W, g_pe, g_ps = ops.decode_bwd_rows(seg, acts.view(-1), latent_acts, W_dec.data, G, ga, post_enc=post_enc,
                                                  post_scale=post_scale,
                                                  want_g_post=bool(need_pe and post_enc is not None),
                                                  adam=fused_state.fused_for(W_dec))  # fused: g_W is None
            g_acts = ga if ga.dtype == latent_acts.dtype else ga.to(latent_acts.dtype)
            if not need_ps:
                g_ps = None
            if ctx.has_bias and need_b:
                g_b = G.sum(0, dtype=torch.float32)
            return g_acts, None, g_pe, g_ps, g_W, g_b, None
        if need_acts or need_pe or need_ps:
            ga, e_post, e_scale = ops.decode_bwd_acts(indices, latent_acts, W_dec, G, post_enc=post_enc,
                                                      post_scale=post_scale,
                                                      want_e_post=bool(need_pe and post_enc is not None))
            if need_acts:
                g_acts = ga if ga.dtype == latent_acts.dtype else ga.to(latent_acts.dtype)
            if e_post is not None or (e_scale is not None and need_ps):
                seg = ctx.holder.get(indices, W_dec.shape[0])
                if e_post is not None and need_ps and e_scale is not None:
                    g_pe, g_ps = ops.segment_sum(seg, e_post.view(-1), e_scale.view(-1))
                elif e_post is not None:
                    g_pe, _ = ops.segment_sum(seg, e_post.view(-1))
                else:
                    g_ps, _ = ops.segment_sum(seg, e_scale.view(-1))
        if need_W:
            seg = seg or ctx.holder.get(indices, W_dec.shape[0])
            g_W = ops.spt_dense(seg, acts.view(-1), G, out_dtype=W_dec.dtype)
        if ctx.has_bias and need_b:
            g_b = G.sum(0, dtype=torch.float32)
        return g_acts, None, g_pe, g_ps, g_W, g_b, None


class FvuLoss(torch.autograd.Function):
    """sparse_coder.py:207-221 as one fused forward+gradient pass."""

    @staticmethod
    def forward(ctx, y, sae_out, loss_mask):
        sc, G = ops.fvu_fwd_bwd(y, sae_out, loss_mask, 1.0, want_grad=True, g_dtype=sae_out.dtype)
        ctx.save_for_backward(G)
        return sc[0]

    @staticmethod
    def backward(ctx, grad_fvu):
        (G,) = ctx.saved_tensors
        return None, G * grad_fvu.to(G.dtype), None


class MidDecoder:
    """sparse_coder.py:39-229."""

    def __init__(self, sparse_coder: "SparseCoder", x: Tensor, activations: Tensor, indices: Tensor,
                 dead_mask: Optional[Tensor] = None):
        self.sparse_coder = sparse_coder
        self.x = x
        self.latent_acts = activations
        self.latent_indices = indices
        self.dead_mask = dead_mask
        self.index = 0

    def copy(self, x=None, activations=None, indices=None, dead_mask=None):
        return MidDecoder(self.sparse_coder, self.x if x is None else x,
                          self.latent_acts if activations is None else activations,
                          self.latent_indices if indices is None else indices,
                          self.dead_mask if dead_mask is None else dead_mask)

    def detach(self):
        if not hasattr(self, "original_activations"):
            self.original_activations = self.latent_acts
            self.latent_acts = self.latent_acts.detach()
            self.latent_acts.requires_grad = True

    def restore(self, is_last: bool = False):
        grad = self.latent_acts.grad
        assert grad is not None, "Activations have no gradient."
        self.latent_acts = self.original_activations
        self.latent_acts.backward(grad, retain_graph=not is_last)
        del self.original_activations

    def next(self):
        self.index += 1

    def prev(self):
        self.index = max(0, self.index - 1)

    @property
    def will_be_last(self):
        return self.index + 1 >= self.sparse_coder.cfg.n_targets

    @property
    def current_w_dec(self):
        sc = self.sparse_coder
        return sc.W_decs[self.index] if hasattr(sc, "W_decs") else sc.W_dec

    def _post_params(self, slot: int):
        sc = self.sparse_coder
        post_enc = sc.post_encs[slot] if sc.multi_target else sc.post_enc
        post_scale = None
        if sc.cfg.post_encoder_scale:
            post_scale = sc.post_enc_scales[slot] if sc.multi_target else sc.post_enc_scale
        return post_enc, post_scale

    @property
    def current_latent_acts(self):
        """(a + post_enc[idx] * (a > 0)) * post_enc_scale[idx] -- sparse_coder.py:103-141.
        Stand-alone (un-fused) form, used by the ``concat`` coalescing mode (runner.py:55)."""
        post_enc, post_scale = self._post_params(self.index)
        latent_acts = self.latent_acts
        latent_acts = latent_acts + post_enc[self.latent_indices] * (latent_acts > 0)
        if post_scale is not None:
            latent_acts = latent_acts * post_scale[self.latent_indices]
        return latent_acts

    def __call__(self, y: Optional[Tensor], index: Optional[int] = None, addition=0, no_extras: bool = False,
                 denormalize: bool = True, add_post_enc: bool = True, loss_mask: Optional[Tensor] = None) -> ForwardOutput:
        sc = self.sparse_coder
        if y is None:
            y = self.x
        assert isinstance(y, Tensor), "y must be a tensor."
        slot = self.index  # the post-encoder slot is read before next() (sparse_coder.py:167-173)
        if index is None:
            index = self.index
            self.next()
        elif sc.cfg.n_targets > 0:
            assert 0 <= index < sc.cfg.n_targets, "Index out of bounds."
        is_last = self.index >= sc.cfg.n_targets

        W_dec = sc.W_decs[index] if hasattr(sc, "W_decs") else sc.W_dec
        b_dec = sc.b_decs[index] if hasattr(sc, "b_decs") else sc.b_dec
        assert W_dec is not None, "Decoder weight was not initialized."
        if add_post_enc:
            post_enc, post_scale = self._post_params(slot)
        else:
            post_enc = post_scale = None
        # top_acts.to(self.dtype) (sparse_coder.py:660): a' is rounded to the weight dtype
        round_acts = W_dec.dtype == torch.bfloat16
        sae_out = FusedDecode.apply(self.latent_acts, self.latent_indices, post_enc, post_scale, W_dec, b_dec, round_acts)

        W_skip = sc.W_skips[index] if hasattr(sc, "W_skips") else sc.W_skip
        if W_skip is not None:
            cd = _autocast_dtype(sc.dtype)
            skip = self.x.to(cd) @ W_skip.to(cd).mT  # sparse_coder.py:190 (bf16 GEMM under autocast)
            sae_out = sae_out + skip.to(sae_out.dtype)
        if not (isinstance(addition, (int, float)) and addition == 0):
            sae_out = sae_out + addition
        if denormalize:
            sae_out = sc.denormalize_output(sae_out)

        if no_extras:
            zero = sae_out.new_zeros(())
            return ForwardOutput(sae_out, self.latent_acts, self.latent_indices, zero, is_last)
        fvu = FvuLoss.apply(y, sae_out, loss_mask)
        return ForwardOutput(sae_out, self.latent_acts, self.latent_indices, fvu, is_last)


class SparseCoder(nn.Module):
    """sparse_coder.py:232-711 (single device; see clt_training_amd.tp for latent sharding)."""

    def __init__(self, d_in: int, cfg: SparseCoderConfig, device="cpu", dtype: Optional[torch.dtype] = None, *,
                 decoder: bool = True, mesh=None):
        super().__init__()
        if mesh is not None:
            raise NotImplementedError("DTensor meshes are replaced by clt_training_amd.tp (RCCL latent sharding)")
        self.cfg = cfg
        self.d_in = d_in
        self.num_latents = cfg.num_latents or d_in * cfg.expansion_factor
        self.multi_target = cfg.n_targets > 0 and cfg.transcode
        self.mesh = None
        weight_dtype = cfg.torch_dtype
        if weight_dtype is None:
            weight_dtype = dtype
        # the reference constructs nn.Linear twice (sparse_coder.py:254-262); the second draw is kept
        self.encoder = nn.Linear(d_in, self.num_latents, device=device, dtype=weight_dtype)
        self.encoder.bias.data.zero_()
        self.encoder = nn.Linear(d_in, self.num_latents, device=device, dtype=weight_dtype)
        self.encoder.bias.data.zero_()

        if decoder:
            if cfg.transcode:
                def create_W_dec():
                    n = self.num_latents
                    if cfg.coalesce_topk == "per-layer":
                        n *= max(1, cfg.n_sources)
                    return nn.Parameter(torch.zeros(n, d_in, device=device, dtype=weight_dtype))

                if self.multi_target and cfg.coalesce_topk not in ("concat", "per-layer") and not cfg.per_source_tied:
                    self.W_decs = nn.ParameterList([create_W_dec() for _ in range(cfg.n_targets)])
                    self.W_dec = self.W_decs[0]
                else:
                    self.W_dec = create_W_dec()
            else:
                self.W_dec = nn.Parameter(self.encoder.weight.data.clone())
                if cfg.normalize_decoder:
                    self.set_decoder_norm_to_unit_norm()
        else:
            self.W_dec = None

        def create_bias():
            return nn.Parameter(torch.zeros(d_in, device=device, dtype=dtype))

        def create_W_skip():
            if not cfg.skip_connection:
                return None
            return nn.Parameter(torch.zeros(d_in, d_in, device=device, dtype=dtype))

        if self.multi_target and cfg.coalesce_topk not in ("concat", "per-layer"):
            self.b_decs = nn.ParameterList()
            self.W_skips = nn.ParameterList()
            for _ in range(cfg.n_targets):
                self.b_decs.append(create_bias())
                self.W_skips.append(create_W_skip())
            self.W_skip = self.W_skips[0]
            self.b_dec = self.b_decs[0]
        else:
            self.b_dec = create_bias()
            self.W_skip = create_W_skip()

        if cfg.normalize_io:
            self.register_buffer("in_norm", torch.ones(1, device=device, dtype=dtype))
            self.register_buffer("out_norm", torch.ones(1, device=device, dtype=dtype))

        def make_post_enc(is_zeros: bool = True):
            t = (torch.zeros if is_zeros else torch.ones)(self.num_latents, device=device, dtype=dtype)
            return nn.Parameter(t, requires_grad=cfg.train_post_encoder)

        if self.multi_target:
            self.post_encs = nn.ParameterList([make_post_enc() for _ in range(cfg.n_targets)])
        else:
            self.post_enc = make_post_enc()
        if cfg.post_encoder_scale:
            if self.multi_target:
                self.post_enc_scales = nn.ParameterList(
                    [make_post_enc(is_zeros=i > 0) for i in range(cfg.n_targets)])
            else:
                self.post_enc_scale = make_post_enc(is_zeros=False)

    # ------------------------------------------------------------------ persistence (sparse_coder.py:448-612)
    @staticmethod
    def load_many(name: str, local: bool = False, layers: Optional[list] = None, device="cpu", *, decoder: bool = True,
                  pattern: Optional[str] = None) -> dict:
        """sparse_coder.py:448-479: one SparseCoder per hookpoint directory of a run (``local=True``: a directory on
        disk; otherwise a Hub snapshot, which needs network access), in natural order."""
        from fnmatch import fnmatch

        from .activations import natural_key

        pattern = pattern + "/*" if pattern is not None else None
        if local:
            repo_path = Path(name)
        else:
            from huggingface_hub import snapshot_download

            repo_path = Path(snapshot_download(name, allow_patterns=pattern))
        if layers is not None:
            return {layer: SparseCoder.load_from_disk(repo_path / layer, device=device, decoder=decoder)
                    for layer in sorted(layers, key=natural_key)}
        files = [f for f in repo_path.iterdir() if f.is_dir() and (pattern is None or fnmatch(f.name, pattern))]
        return {f.name: SparseCoder.load_from_disk(f, device=device, decoder=decoder)
                for f in sorted(files, key=lambda f: natural_key(f.name))}

    @staticmethod
    def load_from_hub(name: str, hookpoint: Optional[str] = None, device="cpu", *, decoder: bool = True) -> "SparseCoder":
        """sparse_coder.py:481-503 (needs network access)."""
        from huggingface_hub import snapshot_download

        repo_path = Path(snapshot_download(name, allow_patterns=f"{hookpoint}/*" if hookpoint is not None else None))
        if hookpoint is not None:
            repo_path = repo_path / hookpoint
        elif not repo_path.joinpath("cfg.json").exists():
            raise FileNotFoundError("No config file found; try specifying a layer.")
        return SparseCoder.load_from_disk(repo_path, device=device, decoder=decoder)

    @staticmethod
    def load_from_disk(path, device="cpu", *, decoder: bool = True, mesh=None) -> "SparseCoder":
        from safetensors.torch import load_file

        path = Path(path)
        with open(path / "cfg.json") as f:
            cfg_dict = json.load(f)
            d_in = cfg_dict.pop("d_in")
            cfg = SparseCoderConfig.from_dict(cfg_dict, drop_extra_fields=True)
        sae = SparseCoder(d_in, cfg, device=device, decoder=decoder, dtype=torch.float32)
        sae.load_state(path)
        return sae

    def load_state(self, path, strict: bool = False):
        """sparse_coder.py:518-592: read ``sae.safetensors`` and IMPUTE what an older / single-target checkpoint lacks:
        missing post-encoder vectors keep this module's initial values; a checkpoint with one ``W_dec`` / ``b_dec`` /
        ``W_skip`` loaded into a per-target module fills slot 0 and leaves the other slots at their initial values."""
        from safetensors.torch import load_file

        mine = self.state_dict()
        state = load_file(str(Path(path) / "sae.safetensors"), device=str(self.device))

        def keep_initial(stem):  # "post_enc" / "post_enc_scale": single vector, or one per target
            if hasattr(self, stem) and not hasattr(self, stem + "s") and stem not in state:
                state[stem] = getattr(self, stem).detach().clone()
            plural = getattr(self, stem + "s", None)
            if plural is not None and not any(f"{stem}s.{i}" in state for i in range(len(plural))):
                for i, t in enumerate(plural):
                    state[f"{stem}s.{i}"] = t.detach().clone()

        def spread_single(stem):  # "W_dec" / "b_dec" / "W_skip" -> slot 0 of the per-target list
            plural = getattr(self, stem + "s", None)
            if plural is None or any(f"{stem}s.{i}" in state for i in range(len(plural))):
                return
            if stem in state:
                state[f"{stem}s.0"] = state[stem]
            for i, t in enumerate(plural):
                if i > 0 and t is not None:
                    state[f"{stem}s.{i}"] = t.detach().clone()

        keep_initial("post_enc")
        keep_initial("post_enc_scale")
        spread_single("W_dec")
        if self.cfg.skip_connection:
            spread_single("W_skip")
        spread_single("b_dec")
        ordered = {k: state[k] for k in mine if k in state}
        # the singular names alias slot 0 (self.W_dec = self.W_decs[0]): keep the two views of one storage consistent
        for stem in ("W_dec", "b_dec", "W_skip"):
            if f"{stem}s.0" in ordered and stem in mine:
                ordered[stem] = ordered[f"{stem}s.0"]
        self.load_state_dict(ordered, strict=strict)
        return self

    def save_to_disk(self, path):
        from safetensors.torch import save_file

        path = Path(path)
        path.mkdir(parents=True, exist_ok=True)
        sd = {}
        seen = {}
        for k, v in self.state_dict().items():  # W_dec aliases W_decs.0: safetensors wants no shared storage
            key = v.data_ptr()
            sd[k] = v.detach().clone().contiguous() if key in seen else v.detach().contiguous()
            seen[key] = k
        save_file(sd, str(path / "sae.safetensors"))
        with open(path / "cfg.json", "w") as f:
            json.dump({**self.cfg.to_dict(), "d_in": self.d_in}, f)

    @property
    def device(self):
        return self.encoder.weight.device

    @property
    def dtype(self):
        return self.encoder.weight.dtype

    def normalize_input(self, x: Tensor) -> Tensor:
        if self.cfg.normalize_io:
            return x * (x.shape[-1] ** 0.5 / self.in_norm)
        return x

    def denormalize_output(self, x: Tensor) -> Tensor:
        if self.cfg.normalize_io:
            return x * (self.out_norm / (x.shape[-1] ** 0.5))
        return x

    def encode(self, x: Tensor) -> EncoderOutput:
        """sparse_coder.py:632-647."""
        x = self.normalize_input(x)
        if not self.cfg.transcode:
            x = x - self.b_dec
        return fused_encoder(x, self.encoder.weight, self.encoder.bias, self.cfg.k, self.cfg.activation,
                             self.cfg.use_fp8)

    def decode(self, top_acts: Tensor, top_indices: Tensor, index: int = 0) -> Tensor:
        """sparse_coder.py:649-661."""
        W_dec = self.W_decs[index] if hasattr(self, "W_decs") else self.W_dec
        b_dec = self.b_decs[index] if hasattr(self, "b_decs") else self.b_dec
        assert W_dec is not None, "Decoder weight was not initialized."
        y = decoder_impl(top_indices, top_acts.to(self.dtype), W_dec)
        return y + b_dec

    def forward(self, x: Tensor, y: Optional[Tensor] = None, *, dead_mask: Optional[Tensor] = None,
                return_mid_decoder: bool = False, loss_mask: Optional[Tensor] = None):
        """sparse_coder.py:663-685: bf16 autocast on the GPU unless cfg.dtype says float16."""
        if self.cfg.dtype == "float16":
            raise NotImplementedError("float16 sparse coders are out of scope (bf16 / fp32 only)")
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=x.is_cuda and AUTOCAST_ENABLED):
            top_acts, top_indices = self.encode(x)
            x = self.normalize_input(x)
            mid_decoder = MidDecoder(self, x, top_acts, top_indices, dead_mask)
            if self.multi_target or return_mid_decoder:
                return mid_decoder
            return mid_decoder(y, 0, loss_mask=loss_mask)

    @torch.no_grad()
    def set_decoder_norm_to_unit_norm(self):
        for W_dec in self.W_decs if hasattr(self, "W_decs") else (self.W_dec,):
            assert W_dec is not None, "Decoder weight was not initialized."
            if W_dec.is_cuda and W_dec.data.is_contiguous():  # one HIP pass (clt_rows_unit_norm)
                ops.rows_unit_norm(W_dec.data)
                continue
            eps = torch.finfo(W_dec.dtype).eps  # CPU construction (tests, checkpoint conversion)
            norm = torch.norm(W_dec.data, dim=1, keepdim=True)
            W_dec.data /= norm + eps

    @torch.no_grad()
    def remove_gradient_parallel_to_decoder_directions(self):
        for W_dec in self.W_decs if hasattr(self, "W_decs") else (self.W_dec,):
            assert W_dec is not None
            if W_dec.grad is None:
                continue  # projected and consumed inside the row pass (fused_state)
            if W_dec.is_cuda and W_dec.grad.dtype == W_dec.dtype:  # one HIP pass (clt_rows_project_out)
                if not W_dec.grad.is_contiguous():
                    W_dec.grad = W_dec.grad.contiguous()
                ops.rows_project_out(W_dec.grad, W_dec.data)
                continue
            parallel = (W_dec.grad * W_dec.data).sum(1, keepdim=True)
            W_dec.grad -= parallel * W_dec.data


Sae = SparseCoder
