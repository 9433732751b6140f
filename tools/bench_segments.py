"""clt_segments_build timing at the C2 shape (N*k = 262144 entries, M = 98304) and at the tp8 shard shape
(2097152 entries, 12288 owned latents of 98304).  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from clt_training_amd import hip_ops as ops
from tools.bench_kernels import timeit

g = torch.Generator(device="cuda").manual_seed(0)
for N, k, M, Ml, lo in [(8192, 32, 98304, 98304, 0), (65536, 32, 98304, 12288, 12288)]:
    idx = torch.stack([torch.randperm(M, device="cuda", generator=g)[:k] for _ in range(256)])
    idx = idx.repeat(N // 256, 1)
    idx = (idx + torch.arange(N, device="cuda")[:, None] * 7919) % M
    sec = timeit(lambda: ops.segments_build(idx, Ml, lat_begin=lo), iters=20)
    print(f"segments_build N={N} k={k} M={Ml} (of {M}): {sec * 1e6:.1f} us")
