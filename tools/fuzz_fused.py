"""Randomised check of the fused encode + top-k call against plain GEMM + row top-k (bit for bit), over shapes the fixed
test list does not name: ragged N, M at and around the fused path's limits, every d that takes the 16x16x32 kernels and
some that do not, k in [1, 64] and a few wider, bias dtypes, three data kinds (randn, integer-valued with ties, rows with
few positives).  python tools/fuzz_fused.py [--cases 40] [--seed 0]; prints one line per case and fails on a mismatch."""
import argparse
import random
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from clt_training_amd import hip_ops as ops  # noqa: E402


def run(cases: int, seed: int, verbose: bool = True) -> int:
    """-> number of cases whose fused result differs from GEMM + row top-k."""
    rng = random.Random(seed)
    dev = torch.device("cuda", 0)
    bad = 0
    for c in range(cases):
        d = rng.choice([64, 96, 128, 192, 256, 320, 512, 768, 1024])
        k = rng.choice([1, 2, 7, 16, 31, 32, 33, 48, 64, 64, 100, 192])
        N = rng.choice([1, 63, 64, 255, 256, 257, 300, 1000, 2048, 3000, 4097])
        m_min = 80 * k  # the fused path needs a sample stride of at least 2: M >= 80 k
        M = rng.choice([m_min, m_min + 16, 2 * m_min + 8 * rng.randrange(0, 64), 16 * rng.randrange(m_min // 16 + 1, m_min // 16 + 2000)])
        M = max(M, k + 8) // 8 * 8
        if N * M > 3e8:
            M = max(m_min, int(3e8 / N) // 16 * 16)
        kind = rng.choice(["randn", "int", "sparse"])
        bias = rng.choice([None, "f32", "bf16"])
        g = torch.Generator(device=dev).manual_seed(1000 + c)
        if kind == "int":  # integer-valued operands: exact products, many exact ties
            x = torch.randint(-3, 4, (N, d), device=dev, generator=g).to(torch.bfloat16)
            W = torch.randint(-2, 3, (M, d), device=dev, generator=g).to(torch.bfloat16)
            b = torch.randint(-4, 5, (M,), device=dev, generator=g).float()
        else:
            x = torch.randn(N, d, device=dev, generator=g).to(torch.bfloat16)
            W = (torch.randn(M, d, device=dev, generator=g) / d**0.5).to(torch.bfloat16)
            b = 0.1 * torch.randn(M, device=dev, generator=g)
            if kind == "sparse":  # most rows have fewer than k positive pre-activations
                b = b - 4.0
        b = None if bias is None else (b.to(torch.bfloat16) if bias == "bf16" else b)
        v, i = ops.encode_topk(x, W, b, k)
        flagged, tiles = ops.encode_topk_fallback_tiles(x, M)
        vr, ir = ops.topk_rows(ops.encoder_gemm_relu(x, W, b), k)
        ok = torch.equal(v, vr) and torch.equal(i, ir)
        bad += 0 if ok else 1
        if verbose or not ok:  # (the fallback count is meaningful only where the fused path ran: bf16, M >= 80 k)
            print(f"case {c:3d} N={N:5d} M={M:7d} d={d:4d} k={k:3d} {kind:6s} bias={bias}  fallback {flagged}/{tiles}  "
                  f"{'ok' if ok else 'MISMATCH'}", flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    bad = run(a.cases, a.seed)
    print("all cases bit-identical" if bad == 0 else f"{bad} MISMATCHES")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
