"""Print every host synchronisation the autograd path performs in one step (torch sync debug mode)."""
import sys
import warnings
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import synth_batches  # noqa: E402
from clt_training_amd import SparseCoderConfig, TrainConfig, Trainer  # noqa: E402
from clt_training_amd.tp import TPTrainer  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else "autograd"
dev = torch.device("cuda", 0)
L, d, N = 3, 768, 256
hooks = [f"h.{i}.mlp" for i in range(L)]
cfg = TrainConfig(sae=SparseCoderConfig(dtype="bfloat16", expansion_factor=8, k=32, transcode=True, skip_connection=True),
                  hookpoints=hooks, cross_layer=L, optimizer="adam", lr=1e-3, lr_warmup_steps=20, log_to_wandb=False)
batches = synth_batches(hooks, N, d, 2, torch.bfloat16, dev)
tr = (TPTrainer if path == "engine" else Trainer)(cfg, batches, device=dev, num_batches=100)
for s in range(2):
    tr.step(batches[s % 2])
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
warnings.simplefilter("always")
tr.step(batches[0])
torch.cuda.set_sync_debug_mode("default")
print("done")
