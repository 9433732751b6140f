"""Ad-hoc check: a model-driven fit whose last batch is short (the token count changes between steps), both trainers."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import transformers  # noqa: E402

from clt_training_amd import SparseCoderConfig, TrainConfig, Trainer  # noqa: E402
from clt_training_amd.tp import TPTrainer  # noqa: E402

torch.manual_seed(0)
mcfg = transformers.GPT2Config(n_layer=3, n_embd=64, n_head=2, vocab_size=50, n_positions=64, bos_token_id=1, eos_token_id=1)
model = transformers.GPT2LMHeadModel(mcfg).eval().cuda()
ids = torch.randint(2, 50, (10, 64))
rows = [{"input_ids": r} for r in ids]
for cls in (Trainer, TPTrainer):
    cfg = TrainConfig(sae=SparseCoderConfig(dtype="bfloat16", transcode=True, expansion_factor=32, k=8, skip_connection=True),
                      hookpoints=["h.*.mlp"], cross_layer=3, optimizer="adam", batch_size=4, log_to_wandb=False,
                      save_dir="/tmp/ckpt_short", save_every=100)
    tr = cls(cfg, rows, model)
    tr.fit(log=lambda s, fv: print(cls.__name__, s, {k: round(float(v), 4) for k, v in fv.items()}))
    assert tr.global_step == 3
print("ok")
