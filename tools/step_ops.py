"""Which torch operators (not C-ABI calls) run GPU kernels inside one engine step at the C2 shape, by device time:
torch.profiler over one step, aten operators with their input shapes.  python tools/step_ops.py [layers]"""
import sys
from pathlib import Path

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import synth_batches  # noqa: E402
from clt_training_amd import SparseCoderConfig, TrainConfig  # noqa: E402
from clt_training_amd.tp import TPTrainer  # noqa: E402

dev = torch.device("cuda", 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 12
d, N = 768, 8192
hooks = [f"h.{i}.mlp" for i in range(L)]
cfg = TrainConfig(sae=SparseCoderConfig(dtype="bfloat16", expansion_factor=128, k=32, transcode=True, skip_connection=True),
                  hookpoints=hooks, cross_layer=L, optimizer="adam", lr=2e-4, lr_warmup_steps=50, log_to_wandb=False)
batches = synth_batches(hooks, N, d, 2, torch.bfloat16, dev)
tr = TPTrainer(cfg, batches, device=dev, num_batches=10_000)
for s in range(3):
    tr.step(batches[s % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(batches[1])
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="device_time_total", row_limit=40, max_name_column_width=40,
                                                        max_shapes_column_width=70))
