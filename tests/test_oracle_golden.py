"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import pytest
import torch

from conftest import golden_grad_name, golden_layers, load_golden
from oracle import clt_oracle as O

FP32 = dict(rtol=1e-5, atol=1e-6)  # BASELINE.json: rtol 1e-5 in fp32


def scatter_up(t):
    return t["up"]


def assert_same_selection(got_v, got_i, ref_v, ref_i):
    """Values to rtol 1e-5; indices bit-exact wherever the activation is positive (batchtopk zeroes slots whose index
    the reference leaves to torch.topk's order among equal zeros; their activation and gradient are 0)."""
    torch.testing.assert_close(got_v, ref_v, **FP32)
    live = ref_v > 0
    assert torch.equal(got_v > 0, live)
    assert torch.equal(got_i[live], ref_i[live])
    if bool(live.all()):
        assert torch.equal(got_i, ref_i)


@pytest.mark.parametrize("name", ["encoder_a", "encoder_b", "encoder_groupmax", "encoder_batchtopk"])
def test_encoder_golden(name):
    t, meta = load_golden(name)
    k = meta["k"]
    vals, idx = O.encoder_forward(t["x"], t["W"], t["b"], k, activation=meta.get("activation", "topk"))
    assert_same_selection(vals, idx, t["vals"], t["idx"])  # bit-exact index sets, canonical order
    if meta.get("activation") == "batchtopk":
        assert vals.shape[1] == 4 * k and int((vals > 0).sum()) == k * vals.shape[0]  # k * N survivors in the batch
    for tag, g in (("sum", torch.ones_like(vals)), ("up", t["up"])):
        gW, gb, gx, _ = O.encoder_backward(t["x"], t["W"], vals, idx, g, need_grad_x=True)
        torch.testing.assert_close(gW, t[f"gW_{tag}"], **FP32)
        torch.testing.assert_close(gb, t[f"gb_{tag}"], **FP32)
        torch.testing.assert_close(gx, t[f"gx_{tag}"], **FP32)


@pytest.mark.parametrize("name", ["decoder_d48", "decoder_d64", "decoder_c"])
def test_decoder_golden(name):
    t, meta = load_golden(name)
    y = O.decode_fwd(t["idx"], t["acts"], t["W"])
    torch.testing.assert_close(y, t["y"], **FP32)
    torch.testing.assert_close(y, t["y_xf"], **FP32)  # reference Triton path == eager path
    gW, ga = O.decode_bwd(t["idx"], t["acts"], t["W"], t["G"])
    torch.testing.assert_close(gW, t["gW"], **FP32)
    torch.testing.assert_close(ga, t["gacts"], **FP32)
    torch.testing.assert_close(gW, t["gW_xf"], **FP32)
    torch.testing.assert_close(ga, t["gacts_xf"], **FP32)
    spt = O.sparse_transpose_dense_matmul(t["idx"], t["acts"], t["G"], meta["M"])
    torch.testing.assert_close(spt, t["spT"], **FP32)
    order, offsets = O.segments_build(t["idx"], meta["M"])
    flat = t["idx"].reshape(-1)
    assert torch.equal(flat[order], torch.sort(flat, stable=True).values)
    assert offsets[-1] == flat.numel()


def _check_step(t, meta, layers, xs, ys, mask, single):
    cfg = meta["cfg"]
    res = O.clt_train_step(
        layers, xs, ys, mask, cfg["k"],
        coalesce=cfg.get("coalesce_topk", "none"),
        normalize_io=cfg.get("normalize_io", False),
        divide_cross_layer=cfg.get("divide_cross_layer", False),
        activation=cfg.get("activation", "topk"),
        topk_coalesced=cfg.get("topk_coalesced", False),
        secondary_target_tied=cfg.get("secondary_target_tied", False),
    )
    for l in range(len(layers)):
        sfx = "" if single else f".{l}"
        got_v, got_i = res.top_acts[l].float(), res.top_idx[l]
        if cfg.get("activation") == "groupmax":  # group order -> the goldens' canonical (value desc, index asc) order
            got_v, got_i = O.topk_merge(got_v, got_i, got_v.shape[1])
        assert_same_selection(got_v, got_i, t[f"vals{sfx}"], t[f"idx{sfx}"])
        torch.testing.assert_close(res.sae_outs[l], t[f"sae_out{sfx}"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(res.fvus[l], t[f"fvu{sfx}"], **FP32)
        gpre = "g." if single else f"g.{l}."
        seen = 0
        for key, ref in t.items():
            if not key.startswith(gpre):
                continue
            name = golden_grad_name(key[len(gpre):])
            got = res.grads[l][name].float()
            torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-7, msg=lambda m: f"{key}: {m}")
            seen += 1
        assert seen >= 4
    return res


@pytest.mark.parametrize("name", ["single_transcoder_skip", "single_transcoder_scale_norm", "single_sae",
                                  "single_transcoder_groupmax", "single_transcoder_batchtopk"])
def test_single_layer_golden(name):
    t, meta = load_golden(name)
    layers = golden_layers(t, meta)
    res = _check_step(t, meta, layers, [t["x"]], [t["y"]], t.get("mask"), single=True)
    if name == "single_sae":
        g = O.remove_parallel_grad(layers[0].W_decs[0].double(), res.grads[0]["W_decs.0"]).float()
        torch.testing.assert_close(g, t["g_proj.W_dec"], rtol=1e-5, atol=2e-7)
        W = O.unit_norm_decoder(layers[0].W_decs[0])
        torch.testing.assert_close(W, layers[0].W_decs[0], **FP32)  # golden was unit-normed already


@pytest.mark.parametrize("name", ["clt_none_L3", "clt_none_div_scale_norm_L4", "clt_concat_L3", "clt_concat_topkcoal_L3",
                                  "clt_perlayer_L3", "clt_none_per_source_tied_L3",
                                  "clt_none_secondary_tied_L3"])
def test_clt_golden(name):
    t, meta = load_golden(name)
    L = meta["L"]
    layers = golden_layers(t, meta, L)
    xs = [t[f"x.{l}"] for l in range(L)]
    ys = [t[f"y.{l}"] for l in range(L)]
    _check_step(t, meta, layers, xs, ys, t["mask"], single=False)


def test_groupmax_tie_rule():
    """fused_encoder.py:408-413: per group the first maximum; an all-zero (post-ReLU) group reports its first column."""
    pre = torch.tensor([[0.0, 2.0, 2.0, 1.0, 0.0, 0.0, 0.0, 0.0, 5.0, 1.0, 5.0, 5.0]])
    v, i = O.groupmax(pre, 3)
    assert v.tolist() == [[2.0, 0.0, 5.0]] and i.tolist() == [[1, 4, 8]]
    rv, ri = pre.unflatten(-1, (3, -1)).max(dim=-1)  # the reference's expression, on CPU
    assert torch.equal(v, rv) and torch.equal(i, ri + torch.arange(0, 12, 4))


def test_topk_tie_rule():
    pre = torch.tensor([[0.0, 2.0, 2.0, 1.0, 2.0, 0.0], [0.0, 0.0, 0.0, 0.0, 3.0, 0.0]])
    v, i = O.topk_lowest_index(pre, 2)
    assert i.tolist() == [[1, 2], [4, 0]]
    v, i = O.topk_merge(torch.tensor([[1.0, 3.0, 3.0, 2.0]]), torch.tensor([[9, 7, 5, 1]]), 2)
    assert i.tolist() == [[5, 7]]


def test_adam_matches_torch():
    torch.manual_seed(0)
    p0 = torch.randn(50)
    p = torch.nn.Parameter(p0.clone())
    for b1 in (0.9, 0.0):
        p.data.copy_(p0)
        opt = torch.optim.Adam([p], lr=torch.tensor(1e-2), betas=(b1, 0.999))
        pp, m, v = p0.double(), torch.zeros(50).double(), torch.zeros(50).double()
        for step in range(1, 4):
            g = torch.randn(50)
            p.grad = g.clone()
            opt.step()
            pp, m, v = O.adam_step(pp, g, m, v, step, 1e-2, b1, 0.999)
            torch.testing.assert_close(pp.float(), p.data, **FP32)


def test_schedule():
    assert O.warmup_linear_factor(0, 50, 100) == 0.0  # first optimizer step runs with lr 0
    assert O.warmup_linear_factor(25, 50, 100) == 0.5
    assert O.warmup_linear_factor(75, 50, 100) == 0.5
    assert abs(O.auto_lr(98304) - 2e-4 / 6**0.5) < 1e-12
