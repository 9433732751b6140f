"""clt_decode_bwd_rows_adam / clt_spt_dense_adam over T cold decoder tables (beyond the Infinity Cache), C2 shapes."""
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
ap.add_argument("--T", type=int, default=8)
ap.add_argument("--tp", type=int, default=1, help="own 1/tp of the latents of N*tp tokens")
ap.add_argument("--dense-latents", type=int, default=0, help="this many latents are active on EVERY token (long segments)")
a = ap.parse_args()
N, d, M, k, T = a.N * a.tp, a.d, a.M // a.tp, a.k, a.T
dev = "cuda"
torch.manual_seed(0)
idx = torch.stack([torch.randperm(a.M, device=dev)[:k] for _ in range(N)]) if N <= 8192 else torch.randint(0, a.M, (N, k), device=dev)
for j in range(a.dense_latents):
    idx[:, j] = 100 + j
v = torch.rand(N, k, device=dev).bfloat16()
G = torch.randn(N, d, device=dev).bfloat16()
acts = torch.rand(N, k, device=dev)
Ws = [(torch.randn(M, d, device=dev) / d**0.5).bfloat16() for _ in range(T)]
ms = [torch.zeros_like(w) for w in Ws]
vs = [torch.zeros_like(w) for w in Ws]
pe = torch.zeros(M, device=dev)
seg = ops.segments_build(idx, M)
gv = torch.zeros(N, k, device=dev)
hp = dict(lr=1e-4, betas=(0.9, 0.999), eps=1e-8)
by = 2 * (6 * M * d + (N * k // a.tp) * d)
step = [0]


def run_rows():
    step[0] += 1
    for W, m_, v_ in zip(Ws, ms, vs):
        ops.decode_bwd_rows(seg, acts.view(-1), v, W, G, gv, post_enc=pe, want_g_post=True, adam=dict(m=m_, v=v_, step=step[0], **hp))


def run_spt():
    step[0] += 1
    for W, m_, v_ in zip(Ws, ms, vs):
        ops.spt_dense_adam(seg, acts.view(-1), G, W, m_, v_, step=step[0], **hp)


t = timeit(run_spt, iters=5, warmup=2) / T
print(f"spt_dense_adam        {t*1e6:8.1f} us  {by/t/1e9:8.1f} GB/s")
t = timeit(run_rows, iters=5, warmup=2) / T
print(f"decode_bwd_rows_adam  {t*1e6:8.1f} us  {by/t/1e9:8.1f} GB/s")
