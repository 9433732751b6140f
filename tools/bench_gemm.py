"""GEMM-only timing (optionally with CLT_GEMM_DEBUG ablations)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from clt_training_amd import hip_ops as ops
from tools.bench_kernels import timeit

N, d, M = 8192, 768, 98304
torch.manual_seed(0)
x = torch.randn(N, d, device="cuda").bfloat16()
W = (torch.randn(M, d, device="cuda") / d**0.5).bfloat16()
b = 0.1 * torch.randn(M, device="cuda")
pre = torch.empty(N, M, dtype=torch.bfloat16, device="cuda")
t = timeit(lambda: ops.encoder_gemm_relu(x, W, b, out=pre), iters=20)
fl = 2.0 * N * M * d
print(f"gemm {t*1e3:.3f} ms {fl/t/1e12:.1f} TF")
