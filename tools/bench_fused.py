"""Fused encode+top-k timing (optionally with CLT_GEMM_DEBUG ablations: 1 no hit stores, 2 no filter scan, 32 plain path)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from clt_training_amd import hip_ops as ops
from tools.bench_kernels import timeit

import os
N, d, M, k = int(os.environ.get('BN', 8192)), 768, int(os.environ.get('BM', 98304)), 32
torch.manual_seed(0)
x = (torch.randn(N, d, device="cuda") + 0.1 * torch.randn(d, device="cuda")).bfloat16()
W = (torch.randn(M, d, device="cuda") / d**0.5).bfloat16()
b = 0.1 * torch.randn(M, device="cuda")
t = timeit(lambda: ops.encode_topk(x, W, b, k), iters=20)
print(f"encode_topk {t*1e3:.3f} ms {2.0*N*M*d/t/1e12:.1f} TF")
