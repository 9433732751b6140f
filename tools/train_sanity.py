"""End-to-end sanity beyond the parity tests: a small CLT trained for a few hundred steps on the synthetic task of
bench.py (y = x R + noise) must drive FVU down.  python tools/train_sanity.py [--path engine|autograd]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import synth_batches  # noqa: E402
from clt_training_amd import SparseCoderConfig, TrainConfig, Trainer  # noqa: E402
from clt_training_amd.tp import TPTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--path", default="engine")
ap.add_argument("--steps", type=int, default=300)
a = ap.parse_args()
dev = torch.device("cuda", 0)
L, d, N = 3, 768, 4096
hooks = [f"h.{i}.mlp" for i in range(L)]
cfg = TrainConfig(sae=SparseCoderConfig(dtype="bfloat16", expansion_factor=16, k=32, transcode=True, skip_connection=True),
                  hookpoints=hooks, cross_layer=L, optimizer="adam", lr=1e-3, lr_warmup_steps=20, log_to_wandb=False)
batches = synth_batches(hooks, N, d, 16, torch.bfloat16, dev)
tr = (TPTrainer if a.path == "engine" else Trainer)(cfg, batches, device=dev, num_batches=a.steps)
hist = []
for s in range(a.steps):
    fv = tr.step(batches[s % len(batches)])
    if s % 25 == 0 or s == a.steps - 1:
        hist.append((s, [round(float(v), 4) for v in fv.values()]))
        print(hist[-1], flush=True)
first, last = hist[1][1], hist[-1][1]
assert all(b < 0.8 * a_ for a_, b in zip(first, last)) and all(b == b for b in last), (first, last)
print("ok: FVU fell on every layer")
