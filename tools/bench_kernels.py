"""Per-kernel micro-benchmarks at BASELINE config-2 shapes (GPT-2 CLT: d=768, M=98304, k=32, N=8192).
Prints achieved TFLOP/s / GB/s against MI355X peaks (2.5 PFLOP/s bf16 dense, 8 TB/s HBM)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from clt_training_amd import hip_ops as ops  # noqa: E402


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=8192)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--M", type=int, default=98304)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    N, d, M, k = a.N, a.d, a.M, a.k
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    es = 2 if dt == torch.bfloat16 else 4
    dev = "cuda"
    torch.manual_seed(0)
    x = torch.randn(N, d, device=dev).to(dt)
    W = (torch.randn(M, d, device=dev) / d**0.5).to(dt)
    b = (0.1 * torch.randn(M, device=dev))
    pre = torch.empty(N, M, dtype=dt, device=dev)

    t = timeit(lambda: ops.encoder_gemm_relu(x, W, b, out=pre))
    fl = 2.0 * N * M * d
    print(f"encoder_gemm_relu  {t*1e3:8.3f} ms  {fl/t/1e12:8.1f} TFLOP/s  ({fl/t/2.5e15*100:.1f}% of 2.5 PF)")
    t = timeit(lambda: ops.topk_rows(pre, k))
    by = N * M * es
    print(f"topk_rows          {t*1e3:8.3f} ms  {by/t/1e9:8.1f} GB/s     ({by/t/8e12*100:.1f}% of 8 TB/s)")
    t = timeit(lambda: ops.encode_topk(x, W, b, k))
    print(f"encode_topk        {t*1e3:8.3f} ms  {fl/t/1e12:8.1f} TFLOP/s")
    v, idx = ops.encode_topk(x, W, b, k)
    Wd = (torch.randn(M, d, device=dev) / d**0.5).to(dt)
    G = torch.randn(N, d, device=dev).to(dt)
    by = N * (k + 1) * d * es
    t = timeit(lambda: ops.decode_fwd(idx, v, Wd))
    print(f"decode_fwd         {t*1e3:8.3f} ms  {by/t/1e9:8.1f} GB/s     ({by/t/8e12*100:.1f}% of 8 TB/s)")
    pe = torch.zeros(M, device=dev); ps = torch.ones(M, device=dev); bd = torch.zeros(d, device=dev)
    t = timeit(lambda: ops.decode_fwd(idx, v, Wd, post_enc=pe, post_scale=ps, bias=bd, want_acts=True))
    print(f"decode_fwd fused   {t*1e3:8.3f} ms  {by/t/1e9:8.1f} GB/s     ({by/t/8e12*100:.1f}% of 8 TB/s)")
    t = timeit(lambda: ops.decode_bwd_acts(idx, v, Wd, G, post_enc=pe, post_scale=ps, want_e_post=True))
    print(f"decode_bwd_acts    {t*1e3:8.3f} ms  {by/t/1e9:8.1f} GB/s     ({by/t/8e12*100:.1f}% of 8 TB/s)")
    t = timeit(lambda: ops.segments_build(idx, M))
    print(f"segments_build     {t*1e3:8.3f} ms")
    seg = ops.segments_build(idx, M)
    vf = v.float()
    by2 = (M * d + N * k * d) * es
    t = timeit(lambda: ops.spt_dense(seg, vf, G))
    print(f"spt_dense          {t*1e3:8.3f} ms  {by2/t/1e9:8.1f} GB/s     ({by2/t/8e12*100:.1f}% of 8 TB/s)")
    y = torch.randn(N, d, device=dev).to(dt)
    t = timeit(lambda: ops.fvu_fwd_bwd(y, G, None))
    print(f"fvu_fwd_bwd        {t*1e3:8.3f} ms")
    # Adam over one encoder-sized tensor
    from clt_training_amd.adam import Adam
    p = torch.nn.Parameter(W.clone()); p.grad = torch.randn_like(p)
    opt = Adam([p], lr=1e-4)
    t = timeit(lambda: opt.step())
    by3 = 7 * M * d * es
    print(f"adam (M*d params)  {t*1e3:8.3f} ms  {by3/t/1e9:8.1f} GB/s     ({by3/t/8e12*100:.1f}% of 8 TB/s)")
    # hipBLASLt reference for the GEMM alone
    t = timeit(lambda: torch.mm(x, W.t()))
    print(f"torch.mm (hipBLASLt){t*1e3:7.3f} ms  {fl/t/1e12:8.1f} TFLOP/s")
    t = timeit(lambda: torch.topk(pre, k, dim=1, sorted=False))
    print(f"torch.topk         {t*1e3:8.3f} ms")


if __name__ == "__main__":
    main()
