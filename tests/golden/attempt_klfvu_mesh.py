"""One documented attempt to pin ``loss_fn="kl-fvu"`` on the reference: drive the reference
``Trainer`` with a world-1 CPU ``DeviceMesh`` (gloo, one process) so that ``DTensor.from_local(kl, self.mesh, ...)``
(trainer.py:764-769) has a mesh.  Run from the repository root in the build container (needs /root/reference):

    python tests/golden/attempt_klfvu_mesh.py

Outcome here (torch 2.10, CPU), both ordinary Python errors -- no golden could be made, the row stays "parity unpinned":
  * mesh passed to the constructor   -> AssertionError at sparse_coder.py:276 (``dtensor.rand`` wants a device RNG
    tracker that a CPU mesh does not have; the whole Trainer then also takes its DTensor branches, whose
    ``ReduceOp.AVG`` gloo lacks, utils.py:465);
  * mesh attached after construction -> RuntimeError "aten.addmm.default got mixed torch.Tensor and DTensor" at
    trainer.py:565 (with a mesh the hook wraps the activations in DTensors, trainer.py:528-536, while the coders hold
    plain parameters).
"""
import os, sys, traceback
os.environ.update(SPARSIFY_NO_COMPILE="1", TRITON_INTERPRET="1", SPARSIFY_DISABLE_TRITON="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, torch.distributed as dist
import make_golden as mg
mg.install_harness()
dist.init_process_group("gloo", rank=0, world_size=1)
from torch.distributed.device_mesh import init_device_mesh
mesh = init_device_mesh("cpu", (1, 1), mesh_dim_names=("dp", "tp"))
from datasets import Dataset
from transformers import GPT2Config, GPT2LMHeadModel
import sparsify.trainer as ref_tr
from sparsify.config import TrainConfig, SparseCoderConfig
ref_tr.ForwardOutput = mg._RefForwardOutput
for variant in ("mesh_to_constructor", "mesh_set_after_construction"):
    torch.manual_seed(33)
    L, d, T, B, steps = 3, 32, 16, 2, 3
    gcfg = GPT2Config(n_layer=L, n_embd=d, n_head=2, vocab_size=96, n_positions=T, bos_token_id=0, eos_token_id=0)
    model = GPT2LMHeadModel(gcfg).eval()
    g = torch.Generator().manual_seed(34)
    ids = torch.randint(1, 96, (B * steps, T), generator=g)
    ds = Dataset.from_dict({"input_ids": ids.tolist()}).with_format("torch")
    cfg = TrainConfig(sae=SparseCoderConfig(expansion_factor=8, k=4, transcode=True, skip_connection=True), batch_size=B,
                      optimizer="adam", lr=1e-2, lr_warmup_steps=2, b1=0.9, hookpoints=["h.*.mlp"], cross_layer=2,
                      remove_first_token=True, save_every=10**9, filter_bos=True, log_to_wandb=False, save_dir="/tmp/klfvu_run",
                      run_name="run", dead_feature_threshold=B * T, loss_fn="kl-fvu")
    try:
        if variant == "mesh_to_constructor":
            tr = ref_tr.Trainer(cfg, ds, model, mesh)
        else:
            tr = ref_tr.Trainer(cfg, ds, model)
            tr.mesh = mesh
        tr.fit()
        print(variant, "-> fit completed")
    except Exception as e:
        tb = traceback.extract_tb(e.__traceback__)
        where = [f for f in tb if "reference/sparsify" in f.filename]
        print(variant, "->", type(e).__name__, str(e)[:300].replace("\n", " "), "| at", (where[-1].filename.split("/")[-1] + ":" + str(where[-1].lineno)) if where else tb[-1])
dist.destroy_process_group()
