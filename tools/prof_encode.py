"""Run the fused encode+top-k a few times (for rocprofv3 --kernel-trace --stats)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from clt_training_amd import hip_ops as ops

N, d, M, k = 8192, 768, 98304, 32
torch.manual_seed(0)
x = (torch.randn(N, d, device="cuda") + 0.1 * torch.randn(d, device="cuda")).bfloat16()
W = (torch.randn(M, d, device="cuda") / d**0.5).bfloat16()
b = 0.1 * torch.randn(M, device="cuda")
for _ in range(12):
    ops.encode_topk(x, W, b, k)
torch.cuda.synchronize()
