"""Same-process A/B of the fused encode + top-k call under the tuning switches of the -DCLT_TUNING library
(CLT_LIB=.../libcltmi355_tuning.so): --arms lists CLT_GEMM_DEBUG values, interleaved rounds (0 = as shipped, 128 = the
sample's top-k through the generic clt_topk_rows kernel + tau_kernel instead of sample_topk_kernel).  Prints the median
and the minimum of each arm in ms per call and the TFLOP/s of the faster figure."""
import argparse
import os
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from clt_training_amd import hip_ops as ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=8192)
    ap.add_argument("--M", type=int, default=98304)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--arms", default="0,128")
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    W = (torch.randn(a.M, a.d, device="cuda", generator=g) / a.d**0.5).bfloat16()
    b = 0.1 * torch.randn(a.M, device="cuda", generator=g)
    gx = torch.Generator(device="cuda").manual_seed(12345)
    x = (torch.randn(a.N, a.d, device="cuda", generator=gx) + 0.1 * torch.randn(a.d, device="cuda", generator=gx)).bfloat16()
    arms = [int(s) for s in a.arms.split(",")]
    ref = None
    times = {arm: [] for arm in arms}
    for arm in arms:  # results agree bit for bit
        os.environ["CLT_GEMM_DEBUG"] = str(arm)
        v, i = ops.encode_topk(x, W, b, a.k)
        flagged, tiles = ops.encode_topk_fallback_tiles(x, a.M)
        print(f"arm {arm}: fallback tiles {flagged}/{tiles}", flush=True)
        if ref is None:
            ref = (v, i)
        else:
            assert torch.equal(v, ref[0]) and torch.equal(i, ref[1]), f"arm {arm} differs from arm {arms[0]}"
    for r in range(a.rounds):
        for arm in arms:
            os.environ["CLT_GEMM_DEBUG"] = str(arm)
            times[arm].append(timeit(lambda: ops.encode_topk(x, W, b, a.k), iters=20) * 1e3)
    fl = 2.0 * a.N * a.M * a.d
    for arm in arms:
        t = times[arm]
        print(f"arm {arm}: median {statistics.median(t):.4f} ms  min {min(t):.4f} ms  "
              f"{fl / statistics.median(t) / 1e9:.1f} TF  frac {fl / statistics.median(t) / 1e9 / 2500:.3f}", flush=True)


if __name__ == "__main__":
    main()
