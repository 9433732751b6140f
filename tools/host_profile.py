"""Host-side (Python) cost of a train step: cProfile over a few steps at a tiny token count, so GPU time is negligible.
python tools/host_profile.py [engine|autograd]"""
import cProfile
import pstats
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import synth_batches  # noqa: E402
from clt_training_amd import SparseCoderConfig, TrainConfig, Trainer  # noqa: E402
from clt_training_amd.tp import TPTrainer  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else "autograd"
dev = torch.device("cuda", 0)
L, d, N = 12, 768, 256
hooks = [f"h.{i}.mlp" for i in range(L)]
cfg = TrainConfig(sae=SparseCoderConfig(dtype="bfloat16", expansion_factor=8, k=32, transcode=True, skip_connection=True),
                  hookpoints=hooks, cross_layer=L, optimizer="adam", lr=1e-3, lr_warmup_steps=20, log_to_wandb=False)
batches = synth_batches(hooks, N, d, 2, torch.bfloat16, dev)
tr = (TPTrainer if path == "engine" else Trainer)(cfg, batches, device=dev, num_batches=100)
for s in range(3):
    tr.step(batches[s % 2])
torch.cuda.synchronize()
import time  # noqa: E402

t0 = time.perf_counter()
for s in range(10):
    tr.step(batches[s % 2])
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue {t_issue * 100:.2f} ms/step, wall {t_all * 100:.2f} ms/step (same launch sequence as C2: {L} layers, "
      f"{L * (L + 1) // 2} pairs; tables 16x smaller and {N} tokens, so the GPU never back-pressures the host)")
pr = cProfile.Profile()
pr.enable()
for s in range(5):
    tr.step(batches[s % 2])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
