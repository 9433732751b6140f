"""clt_decode_fwd_multi at the C2 shape over cold tables: ns live sources of 98 304 x 768 bf16 rows each, 8192 tokens, k = 32,
fp32 output accumulated (as the engine calls it).  python tools/bench_fanout.py [--ns 7] [--sets 3]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from clt_training_amd import hip_ops as ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=8192)
ap.add_argument("--d", type=int, default=768)
ap.add_argument("--M", type=int, default=98304)
ap.add_argument("--k", type=int, default=32)
ap.add_argument("--ns", type=int, default=7)
ap.add_argument("--sets", type=int, default=3, help="independent table sets walked in turn (beyond the Infinity Cache)")
a = ap.parse_args()
dev = "cuda"
torch.manual_seed(0)
sets = []
for s in range(a.sets):
    src = []
    for j in range(a.ns):
        W = (torch.randn(a.M, a.d, device=dev) / a.d**0.5).bfloat16()
        idx = torch.randint(0, a.M, (a.N, a.k), device=dev)
        v = torch.rand(a.N, a.k, device=dev).bfloat16()
        src.append(dict(idx=idx, v=v, W=W, post_enc=torch.zeros(a.M, device=dev)))
    sets.append(src)
out = torch.zeros(a.N, a.d, device=dev)
bias = torch.zeros(a.d, device=dev)


def run():
    for src in sets:
        ops.decode_fwd_multi(src, bias=bias, out=out, accumulate=True, round_acts=True)


t = timeit(run, iters=8, warmup=2) / a.sets
by = 2 * a.N * (a.ns * a.k + 1) * a.d
print(f"decode_fwd_multi ns={a.ns}  {t*1e6:8.1f} us  {by/t/1e9:8.1f} GB/s")
