"""Fused encode + top-k timing at the C2 layer shape, on two dictionaries:

  randn    i.i.d. Gaussian encoder rows, small random bias: exchangeable latents, the best case for the strided
           sample threshold;
  hostile  a 'trained-like' dictionary: Zipf firing frequencies through the bias (a handful of latents fire on more than
           half of the tokens, a long tail almost never), and latents of one 256-row tile share a direction (correlation
           rho), so a token aligned with a tile's direction fires many latents of THAT tile at once -- dense candidate
           lists, a skewed sample, the worst case for both.

For each: ms per call, TFLOP/s, and the share of 256-token tiles that went through the exact fallback
(hip_ops.encode_topk_fallback_tiles).  Results are exact either way; this measures what robustness costs.
-DCLT_TUNING builds: CLT_GEMM_DEBUG ablations (1 no hit stores, 2 no filter scan, 32 plain path)."""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from clt_training_amd import hip_ops as ops
from tools.bench_kernels import timeit


def dictionary(kind, M, d, rho=0.5, zipf_a=1.1, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    W = torch.randn(M, d, device="cuda", generator=g) / d**0.5
    b = 0.1 * torch.randn(M, device="cuda", generator=g)
    if kind == "hostile":
        nt = (M + 255) // 256
        u = torch.randn(nt, d, device="cuda", generator=g) / d**0.5
        W = (1 - rho) ** 0.5 * W + rho**0.5 * u.repeat_interleave(256, 0)[:M]
        # Zipf through the bias: rank r latent gets bias ~ c * r^-a sigma; the first few dominate every token
        perm = torch.randperm(M, device="cuda", generator=g)
        rank = torch.empty(M, device="cuda")
        rank[perm] = torch.arange(1, M + 1, device="cuda", dtype=torch.float32)
        b = 6.0 * rank.pow(-zipf_a / 4) - 1.5  # ~4.5 sigma for the top ranks, negative for the tail
    return W.bfloat16(), b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=int(os.environ.get("BN", 8192)))
    ap.add_argument("--M", type=int, default=int(os.environ.get("BM", 98304)))
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--kinds", default="randn")
    ap.add_argument("--rho", type=float, default=0.5)
    args = ap.parse_args()
    N, M, d, k = args.N, args.M, args.d, args.k
    # (a seed of its own: with the dictionary's seed the first N rows of W would be the tokens themselves)
    gx = torch.Generator(device="cuda").manual_seed(12345)
    x = (torch.randn(N, d, device="cuda", generator=gx) + 0.1 * torch.randn(d, device="cuda", generator=gx)).bfloat16()
    for kind in args.kinds.split(","):
        W, b = dictionary(kind, M, d, args.rho)
        v, i = ops.encode_topk(x, W, b, k)
        flagged, tiles = ops.encode_topk_fallback_tiles(x, M)
        if kind != "randn":  # exactness against GEMM + row top-k on this dictionary
            vr, ir = ops.topk_rows(ops.encoder_gemm_relu(x, W, b), k)
            assert torch.equal(i, ir) and torch.equal(v, vr), "fused != plain"
            freq = torch.bincount(i.flatten(), minlength=M).float() / N
            print(f"  {kind}: {int((freq > 0.5).sum())} latents fire on > 50 % of the tokens, "
                  f"{int((freq == 0).sum())} of {M} never; busiest 256-latent tile holds "
                  f"{float(torch.bincount(i.flatten() // 256).max()) / N:.1f} of the {k} per token")
        t = timeit(lambda: ops.encode_topk(x, W, b, k), iters=20)
        print(f"encode_topk[{kind}] {t * 1e3:.3f} ms {2.0 * N * M * d / t / 1e12:.1f} TF; "
              f"fallback tiles {flagged}/{tiles}")


if __name__ == "__main__":
    main()
