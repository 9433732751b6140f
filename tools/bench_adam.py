"""Adam over several encoder-sized tensors (beyond the Infinity Cache): GB/s of the 7 streams."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from clt_training_amd.adam import Adam
from tools.bench_kernels import timeit

M, d, T = 98304, 768, 10
ps = []
for _ in range(T):
    p = torch.nn.Parameter(torch.randn(M, d, device="cuda").bfloat16())
    p.grad = torch.randn_like(p)
    ps.append(p)
opt = Adam(ps, lr=1e-4)
t = timeit(lambda: opt.step(), iters=5, warmup=2)
by = 7 * 2 * M * d * T
print(f"adam {T} x {M*d/1e6:.0f}M bf16: {t*1e3:.3f} ms {by/t/1e9:.0f} GB/s")
