"""Generate golden vectors by running the REFERENCE itself (CPU, this container only).

Usage (build container, where /root/reference exists):
    python tests/golden/make_golden.py

The reference never travels to the GPU box; only the small ``*.safetensors`` /
``*.json`` fixtures written next to this script do.  The harness follows
SURVEY.md section 8c:
  * register an empty ``sparsify`` package whose ``__path__`` is the reference dir
    (bypasses ``sparsify/__init__`` -> trainer -> schedulefree for the op goldens),
  * in-memory import stand-ins for the absent ``simple_parsing`` / ``natsort`` /
    ``schedulefree`` (dataclass plumbing only, no arithmetic),
  * SPARSIFY_NO_COMPILE=1, TRITON_INTERPRET=1 (the encoder grad-W Triton kernel
    then runs on CPU), eager decoder via SPARSIFY_DISABLE_TRITON=1,
  * ``ForwardOutput`` constructor that drops the two stray positional zeros of
    sparse_coder.py:196-205 (reference bug).
"""
import dataclasses
import json
import os
import sys
import types
from dataclasses import replace
from pathlib import Path

os.environ["SPARSIFY_NO_COMPILE"] = "1"
os.environ["SPARSIFY_DISABLE_TRITON"] = "1"
os.environ["TRITON_INTERPRET"] = "1"

REF = "/root/reference"
OUT = Path(__file__).resolve().parent


def install_harness():
    pkg = types.ModuleType("sparsify")
    pkg.__path__ = [f"{REF}/sparsify"]
    sys.modules["sparsify"] = pkg

    sp = types.ModuleType("simple_parsing")

    class Serializable:
        def to_dict(self):
            return dataclasses.asdict(self)

        @classmethod
        def from_dict(cls, d, drop_extra_fields=False):
            names = {f.name for f in dataclasses.fields(cls)}
            return cls(**{k: v for k, v in d.items() if k in names})

        def save_json(self, path):  # (trainer.py:949 writes config.json with it)
            with open(path, "w") as f:
                json.dump(self.to_dict(), f)

    def list_field(*a, **k):
        return dataclasses.field(default_factory=lambda: list(a))

    sp.Serializable = Serializable
    sp.list_field = list_field
    sp.field = dataclasses.field
    sys.modules["simple_parsing"] = sp

    ns = types.ModuleType("natsort")
    ns.natsorted = sorted
    sys.modules["natsort"] = ns

    sf = types.ModuleType("schedulefree")
    sf.ScheduleFreeWrapper = type("ScheduleFreeWrapper", (), {})
    sf.ScheduleFreeWrapperReference = type("ScheduleFreeWrapperReference", (), {})
    sys.modules["schedulefree"] = sf


install_harness()

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

import sparsify.sparse_coder as ref_sc  # noqa: E402
from sparsify.config import SparseCoderConfig  # noqa: E402
from sparsify.fused_encoder import fused_encoder  # noqa: E402
from sparsify.kernels import triton_sparse_transpose_dense_matmul  # noqa: E402
from sparsify.runner import CrossLayerRunner  # noqa: E402
from sparsify.sparse_coder import SparseCoder  # noqa: E402
from sparsify.utils import eager_decode  # noqa: E402
from sparsify.xformers import xformers_embedding_bag  # noqa: E402

_RefForwardOutput = ref_sc.ForwardOutput


def _forward_output(sae_out, latent_acts, latent_indices, fvu, *rest):
    # sparse_coder.py:196-205 passes 7 positionals to a 5-field dataclass.
    return _RefForwardOutput(sae_out, latent_acts, latent_indices, fvu, bool(rest[-1]) if rest else False)


ref_sc.ForwardOutput = _forward_output


def save(name, tensors, meta):
    tensors = {k: v.detach().contiguous().clone() for k, v in tensors.items()}
    save_file(tensors, str(OUT / f"{name}.safetensors"))
    with open(OUT / f"{name}.json", "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", name, sum(t.numel() * t.element_size() for t in tensors.values()), "bytes")


def canon(vals, idx):
    """Canonical order for an unsorted top-k: value desc, index asc."""
    o1 = torch.argsort(idx, dim=1, stable=True)
    v1, i1 = torch.gather(vals, 1, o1), torch.gather(idx, 1, o1)
    o2 = torch.argsort(v1, dim=1, descending=True, stable=True)
    return torch.gather(v1, 1, o2), torch.gather(i1, 1, o2), torch.gather(o1, 1, o2)


# ---------------------------------------------------------------- encoder
def golden_encoder(name, N, d, M, k, seed, activation="topk"):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, d, generator=g).requires_grad_()
    W = (torch.randn(M, d, generator=g) / d**0.5).requires_grad_()
    b = (0.1 * torch.randn(M, generator=g)).requires_grad_()
    up = torch.randn(N, 4 * k if activation == "batchtopk" else k, generator=g)
    out = fused_encoder(x, W, b, k, activation)
    vals, idx = out.top_acts, out.top_indices
    # loss 1: sum of values (tests/test_encode.py:46)
    vals.sum().backward(retain_graph=True)
    g1 = dict(gx=x.grad.clone(), gW=W.grad.clone(), gb=b.grad.clone())
    x.grad = W.grad = b.grad = None
    if activation == "groupmax":  # the output order IS the group order: keep it
        cv, ci, perm = vals.detach(), idx, torch.arange(k).expand(N, k)
    else:
        cv, ci, perm = canon(vals.detach(), idx)
    # loss 2: random upstream grad, expressed in canonical order
    (vals * torch.zeros_like(vals).scatter_(1, perm, up)).sum().backward()
    g2 = dict(gx=x.grad.clone(), gW=W.grad.clone(), gb=b.grad.clone())
    save(
        name,
        dict(x=x, W=W, b=b, up=up, vals=cv, idx=ci,
             gx_sum=g1["gx"], gW_sum=g1["gW"], gb_sum=g1["gb"],
             gx_up=g2["gx"], gW_up=g2["gW"], gb_up=g2["gb"]),
        dict(N=N, d=d, M=M, k=k, seed=seed, activation=activation, ref="fused_encoder.py:293-422"),
    )


# ---------------------------------------------------------------- decoder
def golden_decoder(name, N, d, M, k, seed):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(M, d, generator=g).requires_grad_()
    lat = torch.rand(N, M, generator=g)
    acts, idx = lat.topk(k)
    acts = acts.clone().requires_grad_()
    G = torch.randn(N, d, generator=g)
    y = eager_decode(idx, acts, W)
    y.backward(G)
    gW_e, ga_e = W.grad.clone(), acts.grad.clone()
    W.grad = acts.grad = None
    y2 = xformers_embedding_bag(idx, W, acts)  # Triton kernels under the interpreter
    y2.backward(G)
    spt = triton_sparse_transpose_dense_matmul(idx, acts.detach().float(), G, N=M)
    save(
        name,
        dict(W=W, acts=acts, idx=idx, G=G, y=y, gW=gW_e, gacts=ga_e,
             y_xf=y2, gW_xf=W.grad, gacts_xf=acts.grad, spT=spt),
        dict(N=N, d=d, M=M, k=k, seed=seed, ref="utils.py:442-457; xformers.py; kernels.py:14-54"),
    )


# ---------------------------------------------------------------- single layer
def warm(sae, g):
    """Non-trivial parameter values (transcoder decoders start at zero)."""
    with torch.no_grad():
        for n, p in sae.named_parameters():
            if n.startswith("encoder"):
                continue
            if "post_enc_scale" in n:
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif "post_enc" in n:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif "W_dec" in n:
                p.copy_(torch.randn(p.shape, generator=g) / p.shape[1] ** 0.5)
            elif "W_skip" in n:
                p.copy_(0.1 * torch.randn(p.shape, generator=g) / p.shape[1] ** 0.5)
            elif "b_dec" in n:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        sae.encoder.bias.copy_(0.05 * torch.randn(sae.encoder.bias.shape, generator=g))
        if hasattr(sae, "in_norm"):
            sae.in_norm.fill_(1.7)
            sae.out_norm.fill_(2.3)


def golden_single(name, cfg_kw, N, d, seed, masked):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    cfg = SparseCoderConfig(**cfg_kw)
    sae = SparseCoder(d, cfg, dtype=torch.float32)
    warm(sae, g)
    if not cfg.transcode and cfg.normalize_decoder:
        sae.set_decoder_norm_to_unit_norm()
    x = torch.randn(N, d, generator=g) + 0.1 * torch.randn(d, generator=g)
    y = x if not cfg.transcode else torch.randn(N, d, generator=g)
    mask = None
    if masked:
        mask = torch.ones(N, dtype=torch.bool)
        mask[::5] = False
    params = {f"p.{n}": p.detach().clone() for n, p in sae.named_parameters()}
    for n, b_ in sae.named_buffers():
        params[f"p.{n}"] = b_.detach().clone()
    runner = CrossLayerRunner()
    enc = runner.encode(x, sae)
    out = runner.decode(enc, y, module_name="l0", detach_grad=True, loss_mask=mask)
    out.fvu.backward()
    runner.restore()
    cv, ci, _ = canon(enc.latent_acts.detach(), enc.latent_indices)
    t = dict(x=x, y=y, fvu=out.fvu.detach(), sae_out=out.sae_out.detach(), vals=cv, idx=ci, **params)
    if mask is not None:
        t["mask"] = mask
    for n, p in sae.named_parameters():
        if p.grad is not None:
            t[f"g.{n}"] = p.grad.clone()
    if not cfg.transcode:
        sae.remove_gradient_parallel_to_decoder_directions()
        t["g_proj.W_dec"] = sae.W_dec.grad.clone()
    save(name, t, dict(cfg=cfg_kw, N=N, d=d, seed=seed, masked=masked,
                       ref="sparse_coder.py:143-229,632-685; runner.py:25-201"))


# ---------------------------------------------------------------- CLT
def golden_clt(name, cfg_kw, L, cross, N, d, seed):
    g = torch.Generator().manual_seed(seed)
    base = SparseCoderConfig(**cfg_kw)
    saes = []
    sources = []
    for pos in range(L):  # trainer.py:117-146
        torch.manual_seed(0)
        n_targets = min(cross, L - pos)
        sources.append(pos + n_targets)
        while sources[0] < pos:
            sources.pop(0)
        cfg = replace(base, n_targets=n_targets, n_sources=len(sources))
        sae = SparseCoder(d, cfg, dtype=torch.float32)
        warm(sae, g)
        saes.append(sae)
    mask = torch.ones(N, dtype=torch.bool)
    mask[::7] = False
    t = dict(mask=mask)
    xs, ys = [], []
    for l in range(L):
        x = torch.randn(N, d, generator=g) + 0.1 * torch.randn(d, generator=g)
        y = 0.5 * x + torch.randn(N, d, generator=g)
        xs.append(x)
        ys.append(y)
        t[f"x.{l}"], t[f"y.{l}"] = x, y
        for n, p in saes[l].named_parameters():
            t[f"p.{l}.{n}"] = p.detach().clone()
        for n, b_ in saes[l].named_buffers():
            t[f"p.{l}.{n}"] = b_.detach().clone()
    runner = CrossLayerRunner()
    for l in range(L):  # trainer.py:598-682
        enc = runner.encode(xs[l], saes[l])
        out = runner.decode(enc, ys[l], module_name=f"layers.{l}", detach_grad=True, loss_mask=mask)
        t[f"fvu.{l}"] = out.fvu.detach().clone()
        t[f"sae_out.{l}"] = out.sae_out.detach().clone()
        cv, ci, _ = canon(enc.original_activations.detach() if hasattr(enc, "original_activations")
                          else enc.latent_acts.detach(), enc.latent_indices)
        t[f"vals.{l}"], t[f"idx.{l}"] = cv, ci
        out.fvu.backward()
        runner.restore()
    assert len(runner.outputs) == 0
    for l in range(L):
        for n, p in saes[l].named_parameters():
            assert p.grad is not None, (l, n)
            t[f"g.{l}.{n}"] = p.grad.clone()
    save(name, t, dict(cfg=cfg_kw, L=L, cross_layer=cross, N=N, d=d, seed=seed,
                       ref="runner.py:25-201; trainer.py:117-146,598-682"))


# ---------------------------------------------------------------- the reference Trainer itself
def golden_muon(name="muon_steps"):
    """The reference's Muon (sparsify/muon.py:51-203, ddp=False) on three matrices -- tall, square, wide -- for three
    steps with recorded gradients, and its quintic Newton-Schulz iteration (muon.py:13-42) on a tall matrix.  The
    function is decorated with torch.compile; dynamo is switched off so that it runs its Python body eagerly."""
    import torch._dynamo as dynamo

    dynamo.config.disable = True
    from sparsify.muon import Muon, quintic_newtonschulz

    g = torch.Generator().manual_seed(31)
    shapes = [(96, 16), (16, 16), (8, 24)]
    ps = [torch.nn.Parameter(0.1 * torch.randn(s, generator=g)) for s in shapes]
    t = {f"p0.{i}": p.detach().clone() for i, p in enumerate(ps)}
    opt = Muon(ps, lr=2e-3, ddp=False)
    for s in range(3):
        for i, p in enumerate(ps):
            gr = torch.randn(p.shape, generator=g)
            t[f"g{s}.{i}"] = gr.clone()  # (the optimizer's Nesterov lerp_ overwrites .grad)
            p.grad = gr
        opt.step()
        for i, p in enumerate(ps):
            t[f"p{s + 1}.{i}"] = p.detach().clone()
    G = torch.randn(96, 16, generator=g)
    t["ns_in"] = G.clone()
    t["ns_out"] = quintic_newtonschulz(G, 5).float()
    G2 = torch.randn(8, 24, generator=g)
    t["ns_in_wide"] = G2.clone()
    t["ns_out_wide"] = quintic_newtonschulz(G2, 5).float()
    save(name, t, dict(lr=2e-3, momentum=0.95, nesterov=True, weight_decay=0.1, ns_steps=5, steps=3, n_params=3,
                       ref="muon.py:13-42, 51-203 (ddp=False)"))


def golden_trainer(name, sae_kw, cross, b1, seed, L=3, d=32, T=16, B=2, steps=3, warmup=2, optimizer="adam",
                   loss_fn="fvu"):
    """Drive the reference ``Trainer.fit`` (trainer.py:61-254, 376-865; mesh=None, optimizer="adam", FVU loss) on a
    random-init GPT-2 built from a config (no download) for ``steps`` batches, and record what pins SURVEY.md rows
    a18-a20, a22: the coders' parameters as constructed, after the data-dependent step-0 init (captured at the first
    ``optimizer.step``, with that step's gradients), after every optimizer step, the learning rate each step ran with,
    ``num_tokens_since_fired`` at the end, and the activations the hooks saw (so the counterpart trainers can be fed
    cached activations).  The checkpoint ``fit`` writes at its end (sae.safetensors + cfg.json per hookpoint,
    optimizer_0.pt in the flattened layout of utils.py:193-277, lr_scheduler_0.pt, state.pt, rank_0_state.pt) is kept
    under tests/golden/<name>_ckpt/: reference-written files for the checkpoint-format tests."""
    import shutil
    import tempfile

    from datasets import Dataset
    from transformers import GPT2Config, GPT2LMHeadModel

    import sparsify.trainer as ref_tr
    from sparsify.config import TrainConfig

    ref_tr.ForwardOutput = _RefForwardOutput  # (isinstance check at trainer.py:617)
    torch.manual_seed(seed)
    gcfg = GPT2Config(n_layer=L, n_embd=d, n_head=2, vocab_size=96, n_positions=T, bos_token_id=0, eos_token_id=0)
    model = GPT2LMHeadModel(gcfg).eval()
    g = torch.Generator().manual_seed(seed + 1)
    ids = torch.randint(1, 96, (B * steps, T), generator=g)
    ds = Dataset.from_dict({"input_ids": ids.tolist()}).with_format("torch")
    tmp = tempfile.mkdtemp()
    if optimizer == "muon":
        import torch._dynamo as dynamo

        dynamo.config.disable = True  # muon.py:12: torch.compile'd Newton-Schulz, run eagerly
    cfg = TrainConfig(sae=SparseCoderConfig(**sae_kw), batch_size=B, optimizer=optimizer, lr=1e-2, lr_warmup_steps=warmup,
                      b1=b1, hookpoints=["h.*.mlp"], cross_layer=cross, remove_first_token=True, save_every=10**9,
                      # filter_bos=True skips `bos_mask &= 0` (trainer.py:687-688), which torch 2.10 rejects (Long into
                      # Bool); no token id equals bos_token_id here, so the mask is the same: first tokens only
                      filter_bos=True,
                      log_to_wandb=False, save_dir=tmp, run_name="run", dead_feature_threshold=B * T, loss_fn=loss_fn)
    tr = ref_tr.Trainer(cfg, ds, model)
    hooks = list(tr.saes)
    t = {}
    for l, h in enumerate(hooks):
        for n, p in tr.saes[h].state_dict().items():
            t[f"p0.{l}.{n}"] = p.detach().clone()
    # record the activations the hooks consume (transcoder: module input -> module output)
    acts = []

    def rec(l):
        def f(module, inputs, outputs):
            x = inputs[0] if isinstance(inputs, tuple) else inputs
            y = outputs[0] if isinstance(outputs, tuple) else outputs
            acts[-1][l] = (x.detach().flatten(0, 1).clone(), y.detach().flatten(0, 1).clone())
        return f

    handles = [model.base_model.get_submodule(h).register_forward_hook(rec(l)) for l, h in enumerate(hooks)]
    orig_forward = model.forward

    def forward(*a, **k):
        acts.append({})
        return orig_forward(*a, **k)

    model.forward = forward
    opt, last = tr.optimizers[0], tr.optimizers[-1]  # (optimizer="muon": Muon steps first, then Adam)
    orig_step, orig_last = opt.step, last.step
    n_step = [0]

    def before():
        s = n_step[0]
        for l, h in enumerate(hooks):
            if s == 0:
                for n, p in tr.saes[h].state_dict().items():
                    t[f"init.{l}.{n}"] = p.detach().clone()
            for n, p in tr.saes[h].named_parameters():
                t[f"grad{s}.{l}.{n}"] = p.grad.detach().clone()
        t[f"lr{s}"] = torch.tensor([float(pg["lr"]) for o in tr.optimizers for pg in o.param_groups])

    def after():
        s = n_step[0]
        for l, h in enumerate(hooks):
            for n, p in tr.saes[h].state_dict().items():
                t[f"after{s}.{l}.{n}"] = p.detach().clone()
        n_step[0] += 1

    def step(*a, **k):
        before()
        out = orig_step(*a, **k)
        if last is opt:
            after()
        return out

    def step_last(*a, **k):
        out = orig_last(*a, **k)
        after()
        return out

    opt.step = step
    if last is not opt:
        last.step = step_last
    ckpt_note = "written by Trainer.fit's final save (trainer.py:861, 910-964)"
    try:
        tr.fit()
    except RuntimeError as e:
        # With the safetensors of this image the reference's own save_to_disk refuses a CLT's state dict (W_dec and
        # W_decs.0 are one tensor under two names; the pinned older safetensors wrote it).  Training is complete by
        # then; write what the rest of _checkpoint would have (trainer.py:923-962) with the reference's own functions.
        if n_step[0] != steps or "share memory" not in str(e):
            raise
        from sparsify.utils import save_sharded

        path = f"{tmp}/run"
        os.makedirs(path, exist_ok=True)
        for i, (o, sch) in enumerate(zip(tr.optimizers, tr.lr_schedulers)):
            save_sharded(o.state_dict(), f"{path}/optimizer_{i}.pt", None, save_st=False, unflatten=True)
            torch.save(sch.state_dict(), f"{path}/lr_scheduler_{i}.pt")
        torch.save({"global_step": tr.global_step}, f"{path}/state.pt")
        torch.save({"num_tokens_since_fired": tr.num_tokens_since_fired, "best_loss": tr.best_loss},
                   f"{path}/rank_0_state.pt")
        ckpt_note = ("sae.safetensors NOT written: the reference's save_to_disk raised under this image's safetensors "
                     "(shared tensors W_dec / W_decs.0); optimizer_0.pt, lr_scheduler_0.pt, state.pt, rank_0_state.pt "
                     "written with the reference's own save_sharded / torch.save calls of trainer.py:923-962")
    for hd in handles:
        hd.remove()
    assert n_step[0] == steps
    if loss_fn == "fvu":  # (the end-to-end losses run the model more than once per step, on changing activations)
        fit_acts = [a for a in acts if len(a) == L][-steps:]  # (resolve_widths ran the model once on dummy inputs)
        for s, a in enumerate(fit_acts):
            for l in range(L):
                t[f"x{s}.{l}"], t[f"y{s}.{l}"] = a[l]
    for l, h in enumerate(hooks):
        t[f"since_fired.{l}"] = tr.num_tokens_since_fired[h].clone()
    mask = torch.ones(B, T, dtype=torch.bool)
    mask[:, 0] = False  # remove_first_token (trainer.py:689-690, 609)
    t["mask"] = mask.flatten()
    save(name, t, dict(cfg=sae_kw, L=L, d=d, cross_layer=cross, b1=b1, lr=1e-2, warmup=warmup, steps=steps,
                       tokens_per_step=B * T, hooks=hooks, seed=seed, num_batches=steps, optimizer=optimizer,
                       dead_feature_threshold=B * T, checkpoint=ckpt_note, loss_fn=loss_fn,
                       ref="trainer.py:61-254, 376-865 (fit, mesh=None)"))
    dst = OUT / f"{name}_ckpt"
    if dst.exists():
        shutil.rmtree(dst)
    shutil.copytree(f"{tmp}/run", dst)
    shutil.rmtree(tmp)
    print("kept the reference-written checkpoint:", sorted(str(q.relative_to(dst)) for q in dst.rglob("*") if q.is_file()))


def main(only=None):
    if only == "trainer":
        clt = dict(expansion_factor=8, k=4, transcode=True, skip_connection=True, post_encoder_scale=True)
        golden_trainer("trainer_clt_b1_09", clt, cross=2, b1=0.9, seed=21)
        golden_trainer("trainer_clt_b1_00", dict(clt, normalize_io=True), cross=3, b1=0.0, seed=22)
        golden_trainer("trainer_sae_b1_09", dict(expansion_factor=8, k=4, transcode=False), cross=0, b1=0.9, seed=23)
        return
    if only == "e2e":  # end-to-end losses (trainer.py:739-785): the SAE output replaces the module output in the forward
        clt = dict(expansion_factor=8, k=4, transcode=True, skip_connection=True)
        if not (OUT / "trainer_clt_ce.safetensors").exists():
            golden_trainer("trainer_clt_ce", clt, cross=2, b1=0.9, seed=31, loss_fn="ce")
        if not (OUT / "trainer_clt_kl.safetensors").exists():
            golden_trainer("trainer_clt_kl", dict(clt, post_encoder_scale=True), cross=3, b1=0.9, seed=32, loss_fn="kl")
        # loss_fn="kl-fvu" cannot be driven with mesh=None: trainer.py:764-769 wraps kl in DTensor.from_local(kl, self.mesh,
        # ...) unconditionally and torch raises "No device mesh is currently active!" (tried here) -- no golden for it.
        # Round 3 tried a world-1 CPU mesh (tests/golden/attempt_klfvu_mesh.py): AssertionError at sparse_coder.py:276
        # (dtensor.rand on a CPU mesh) with the mesh in the constructor, "mixed torch.Tensor and DTensor" at
        # trainer.py:565 with the mesh attached afterwards.  Still no golden: kl-fvu stays parity unpinned.
        golden_trainer("trainer_sae_kl", dict(expansion_factor=8, k=4, transcode=False), cross=0, b1=0.9, seed=34,
                       loss_fn="kl")
        return
    if only == "muon":
        golden_muon()
        golden_trainer("trainer_clt_muon", dict(expansion_factor=8, k=4, transcode=True, skip_connection=True), cross=2,
                       b1=0.9, seed=24, optimizer="muon")
        return
    if only == "groupmax":  # added later: leaves the earlier fixtures untouched
        golden_encoder("encoder_groupmax", N=24, d=48, M=512, k=8, seed=12, activation="groupmax")
        golden_single("single_transcoder_groupmax",
                      dict(expansion_factor=8, k=8, transcode=True, skip_connection=True, activation="groupmax"),
                      N=40, d=32, seed=13, masked=True)
        return
    if only == "batchtopk":
        golden_encoder("encoder_batchtopk", N=24, d=48, M=512, k=4, seed=15, activation="batchtopk")
        golden_single("single_transcoder_batchtopk",
                      dict(expansion_factor=8, k=4, transcode=True, skip_connection=True, activation="batchtopk"),
                      N=40, d=32, seed=16, masked=True)
        return
    if only == "per_layer":
        golden_clt("clt_perlayer_L3",
                   dict(expansion_factor=8, k=4, transcode=True, skip_connection=True, coalesce_topk="per-layer",
                        post_encoder_scale=True), L=3, cross=3, N=28, d=24, seed=17)
        return
    if only == "per_source_tied":
        golden_clt("clt_none_per_source_tied_L3",
                   dict(expansion_factor=8, k=4, transcode=True, skip_connection=True, per_source_tied=True,
                        post_encoder_scale=True), L=3, cross=3, N=28, d=24, seed=18)
        return
    if only == "secondary_target_tied":
        golden_clt("clt_none_secondary_tied_L3",
                   dict(expansion_factor=8, k=4, transcode=True, skip_connection=True, secondary_target_tied=True,
                        post_encoder_scale=True, divide_cross_layer=True), L=3, cross=3, N=28, d=24, seed=19)
        return
    if only == "topk_coalesced":
        golden_clt("clt_concat_topkcoal_L3",
                   dict(expansion_factor=8, k=4, transcode=True, skip_connection=True, coalesce_topk="concat",
                        topk_coalesced=True, post_encoder_scale=True), L=3, cross=3, N=28, d=24, seed=14)
        return
    golden_encoder("encoder_a", N=32, d=48, M=512, k=8, seed=1)
    golden_encoder("encoder_b", N=16, d=64, M=256, k=32, seed=2)
    golden_decoder("decoder_d48", N=2, d=48, M=128, k=10, seed=3)  # tests/test_decode.py shapes
    golden_decoder("decoder_d64", N=2, d=64, M=128, k=10, seed=4)
    golden_decoder("decoder_c", N=24, d=40, M=64, k=6, seed=5)
    golden_single("single_transcoder_skip", dict(expansion_factor=8, k=8, transcode=True, skip_connection=True),
                  N=40, d=32, seed=6, masked=True)
    golden_single("single_transcoder_scale_norm",
                  dict(expansion_factor=8, k=4, transcode=True, skip_connection=True,
                       post_encoder_scale=True, normalize_io=True), N=30, d=24, seed=7, masked=False)
    golden_single("single_sae", dict(expansion_factor=8, k=8, transcode=False), N=40, d=32, seed=8, masked=True)
    golden_clt("clt_none_L3", dict(expansion_factor=8, k=4, transcode=True, skip_connection=True),
               L=3, cross=3, N=28, d=24, seed=9)
    golden_clt("clt_none_div_scale_norm_L4",
               dict(expansion_factor=8, k=4, transcode=True, skip_connection=True, divide_cross_layer=True,
                    post_encoder_scale=True, normalize_io=True), L=4, cross=3, N=28, d=24, seed=10)
    golden_clt("clt_concat_L3",
               dict(expansion_factor=8, k=4, transcode=True, skip_connection=True, coalesce_topk="concat",
                    post_encoder_scale=True, normalize_io=True), L=3, cross=3, N=28, d=24, seed=11)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
