import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def load_golden(name):
    import json

    from safetensors.torch import load_file

    t = load_file(str(GOLDEN / f"{name}.safetensors"))
    with open(GOLDEN / f"{name}.json") as f:
        meta = json.load(f)
    return t, meta


def golden_layers(t, meta, L=None, prefix="p"):
    """Build oracle LayerParams from a single_* / clt_* golden (``prefix``: which parameter snapshot, e.g. "p0")."""
    import torch

    from oracle.clt_oracle import LayerParams

    cfg = meta["cfg"]
    single = L is None
    layers = []
    for l in range(1 if single else L):
        pre = f"{prefix}." if single else f"{prefix}.{l}."
        P = {k[len(pre):]: v for k, v in t.items() if k.startswith(pre)}

        def lst(stem, plural):
            # slot 0 is registered under the singular alias (self.W_dec = self.W_decs[0])
            out = []
            if f"{plural}.0" in P:
                out.append(P[f"{plural}.0"].clone())
            elif stem in P:
                out.append(P[stem].clone())
            else:
                return None
            i = 1
            while f"{plural}.{i}" in P:
                out.append(P[f"{plural}.{i}"].clone())
                i += 1
            return out

        W_decs = lst("W_dec", "W_decs")
        b_decs = lst("b_dec", "b_decs")
        W_skips = lst("W_skip", "W_skips") or [None] * len(W_decs)
        post_encs = lst("post_enc", "post_encs")
        scales = lst("post_enc_scale", "post_enc_scales")
        if single:
            n_targets = 0
        else:
            n_targets = min(meta["cross_layer"], meta["L"] - l)
        layers.append(
            LayerParams(
                P["encoder.weight"].clone(), P["encoder.bias"].clone(), W_decs, b_decs, W_skips, post_encs,
                scales, n_targets, P.get("in_norm"), P.get("out_norm"), transcode=cfg.get("transcode", False),
            )
        )
    return layers


def golden_grad_name(n):
    """Map a reference parameter name to the oracle's grad-dict key."""
    n = n.replace("encoder.weight", "W_enc").replace("encoder.bias", "b_enc")
    for stem, plural in (("W_dec", "W_decs"), ("b_dec", "b_decs"), ("W_skip", "W_skips"),
                         ("post_enc_scale", "post_enc_scales"), ("post_enc", "post_encs")):
        if n == stem:
            return f"{plural}.0"
    return n
