"""CPU-side checks: the C-ABI library loads and exports every symbol include/clt_mi355.h
declares; host logic (config, construction rule, schedule); loud failure without a GPU."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "clt_mi355.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(clt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from clt_training_amd import _lib

    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/clt_mi355.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)
    _lib.load()
    assert _lib.load().clt_version() >= 100


def test_workspace_queries_need_no_gpu():
    from clt_training_amd import _lib

    lib = _lib.load()
    assert lib.clt_encode_topk_workspace_bytes(8192, 98304, 1) >= 8192 * 98304 * 2
    assert lib.clt_segments_workspace_bytes(8192 * 32, 98304) > 3 * 8192 * 32 * 4
    assert lib.clt_fvu_workspace_bytes(8192, 768) > 0


def test_bad_arguments_are_reported_not_thrown():
    from clt_training_amd import _lib

    lib = _lib.load()
    rc = lib.clt_topk_rows(None, 0, 8, None, None, 1, 8, 4, None, 0, None, None)
    assert rc == -1 and b"null" in lib.clt_last_error()
    with pytest.raises(RuntimeError, match="clt_topk_rows"):
        _lib.check(rc, "clt_topk_rows")


def test_ops_fail_loudly_without_gpu():
    from clt_training_amd import SparseCoder, SparseCoderConfig

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sc = SparseCoder(32, SparseCoderConfig(transcode=True, k=4), dtype=torch.float32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sc(torch.randn(4, 32), torch.randn(4, 32))


def test_no_oracle_in_product():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|importlib.*oracle|__import__.*oracle", re.M)
    for f in list((ROOT / "clt_training_amd").glob("*.py")) + list((ROOT / "clt_training_amd" / "csrc").glob("*")):
        assert not pat.search(f.read_text()), f"{f} must not import or call the oracle"


def test_config_roundtrip_and_defaults():
    from clt_training_amd import SparseCoderConfig, TrainConfig

    c = SparseCoderConfig()
    assert (c.expansion_factor, c.k, c.activation, c.tp_output, c.normalize_decoder) == (32, 32, "topk", True, True)
    assert c.torch_dtype is None and SparseCoderConfig(dtype="bfloat16").torch_dtype == torch.bfloat16
    d = c.to_dict()
    assert SparseCoderConfig.from_dict({**d, "junk": 1}, drop_extra_fields=True) == c
    t = TrainConfig(sae=c, hookpoints=["a"])
    assert (t.optimizer, t.lr_warmup_steps, t.b1, t.b2, t.dead_feature_threshold) == ("signum", 1000, 0.9, 0.999, 10_000_000)
    assert TrainConfig.from_dict(t.to_dict()).sae == c
    with pytest.raises(ValueError):
        TrainConfig(sae=c, layers=[1], layer_stride=2)


def test_param_names_match_reference_state_dict():
    """Names / shapes of sparse_coder.py:233-445 (W_dec aliases W_decs[0] etc.)."""
    from clt_training_amd import SparseCoder, SparseCoderConfig

    sc = SparseCoder(16, SparseCoderConfig(transcode=True, n_targets=3, skip_connection=True, post_encoder_scale=True,
                                           expansion_factor=4, normalize_io=True), dtype=torch.float32)
    names = dict(sc.named_parameters())
    assert set(names) == {"W_dec", "W_skip", "b_dec", "encoder.weight", "encoder.bias", "W_decs.1", "W_decs.2",
                          "b_decs.1", "b_decs.2", "W_skips.1", "W_skips.2", "post_encs.0", "post_encs.1", "post_encs.2",
                          "post_enc_scales.0", "post_enc_scales.1", "post_enc_scales.2"}
    assert sc.W_dec is sc.W_decs[0] and names["W_dec"].shape == (64, 16) and float(names["W_dec"].abs().sum()) == 0
    assert float(sc.post_enc_scales[0].sum()) == 64 and float(sc.post_enc_scales[1].sum()) == 0
    assert "in_norm" in dict(sc.named_buffers())
    tied = SparseCoder(16, SparseCoderConfig(transcode=True, n_targets=3, coalesce_topk="concat", expansion_factor=4),
                       dtype=torch.float32)
    assert not hasattr(tied, "W_decs") and len(tied.post_encs) == 3
    sae = SparseCoder(16, SparseCoderConfig(expansion_factor=4), dtype=torch.float32)
    torch.testing.assert_close(sae.W_dec.norm(dim=1), torch.ones(64), rtol=1e-5, atol=1e-5)


def test_trainer_construction_rule():
    from clt_training_amd import SparseCoderConfig, TrainConfig, Trainer

    hooks = [f"h.{i}.mlp" for i in range(5)]
    cfg = TrainConfig(sae=SparseCoderConfig(transcode=True, expansion_factor=2, k=4), hookpoints=hooks, cross_layer=3,
                      optimizer="adam", lr_warmup_steps=2)
    tr = Trainer(cfg, [None] * 10, device="cpu", input_widths={h: 8 for h in hooks})
    assert [tr.saes[h].cfg.n_targets for h in hooks] == [3, 3, 3, 2, 1]  # trainer.py:125-127
    assert [tr.saes[h].cfg.n_sources for h in hooks] == [1, 2, 3, 4, 4]  # trainer.py:128-131 as written
    lr = float(tr.optimizers[0].param_groups[0]["lr"])
    assert lr == 0.0  # LambdaLR applies factor(0) = 0 at construction: first step runs with lr 0
    sch = tr.lr_schedulers[0]
    assert sch.lr_lambda(1) == 0.5 and sch.lr_lambda(2) == 1.0 and sch.lr_lambda(6) == 0.5


def test_encoder_gemm_kernels_use_no_scratch():
    """The persistent encoder GEMM kernels run at the edge of their register budget (256 VGPRs at two waves per SIMD,
    512 at one).  A spilled register is more than a slow access there: its reload is a vector-memory load, so hipcc puts
    ``s_waitcnt vmcnt(0)`` in front of it and the kernel's LDS-DMA prefetch ring drains once per tile (seen in round 3
    with two registers of the eight-wave kernel).  hipcc's own resource report must show no scratch for them."""
    import re
    import subprocess

    from clt_training_amd.build import CSRC, FLAGS, HIPCC

    # (sample_topk_kernel keeps a whole row in registers through fully unrolled loops: a version whose loops hipcc
    # declined to unroll indexed the row dynamically and put it in scratch, 208 bytes per lane)
    for src, kernels in (("encoder_gemm_w8.hip", ("gemm_w8_filter_kernel",)), ("encoder_gemm_fast16.hip", ("gemm_fast16_kernel",)),
                         ("encoder_gemm_fast.hip", ("sample_topk_kernel",))):
        out = subprocess.run([HIPCC, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", str(CSRC / src), "-o", "/dev/null"],
                             capture_output=True, text=True, check=True).stderr
        blocks = re.split(r"remark: Function Name: ", out)[1:]
        seen = 0
        for b in blocks:
            name = b.split()[0]
            if any(k in name for k in kernels):
                seen += 1
                scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
                spills = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
                assert scratch == 0 and spills == 0, (src, name, scratch, spills)
        assert seen >= 1, (src, "kernel not found in the resource report")


def test_hot_loops_do_not_wait_in_front_of_every_load():
    """Round 3 found three kernels in which hipcc had put ``s_waitcnt vmcnt(0)`` directly in front of every load of a loop
    (loads under a per-lane or a dtype branch: each sits in a block of its own and its destination register "may" still
    be pending) -- the gathered rows of the row passes and the rows of the FVU kernels arrived one memory round trip
    after another.  The ISA of those kernels must not show that pattern again."""
    import re
    import subprocess

    from clt_training_amd.build import CSRC, FLAGS, HIPCC

    def serialised_loads(src, kernel_substrings):
        asm = subprocess.run([HIPCC, *FLAGS, "-S", "--cuda-device-only", "-o", "-", str(CSRC / src)],
                             capture_output=True, text=True, check=True).stdout.split("\n")
        name, counts = None, {}
        for i, line in enumerate(asm):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                name = m.group(1)
            if name and "s_waitcnt vmcnt(0)" in line:
                nxt = [x for x in asm[i + 1:i + 4] if x.strip() and not x.strip().startswith(";")][:2]
                if any("global_load" in x or "flat_load" in x for x in nxt):
                    counts[name] = counts.get(name, 0) + 1
        names = {m.group(1) for m in (re.match(r"^(_Z\w+):", ln) for ln in asm) if m}
        seen = [n for n in names if any(k in n for k in kernel_substrings)]
        assert seen, (src, "kernels not found")
        return {n: counts.get(n, 0) for n in seen}

    # (one or two such waits are legitimate: a dependent load behind the load of its index)
    for src, kernels in (("decode_rows.hip", ("rows_lds_multi_kernelIDF16bLi2ELi4E", "rows_lds_kernelIDF16bLi2ELi4E")),
                         ("fvu.hip", ("fvu_partial_kernel", "fvu_colsum_kernel", "fvu_grad_kernel"))):
        for name, n in serialised_loads(src, kernels).items():
            assert n <= 2, (src, name, n)
