"""Configuration dataclasses: same field names and defaults as sparsify/config.py:9-218
(the reference parses them with simple_parsing; here they are plain dataclasses with the
same ``to_dict`` / ``from_dict`` surface, so ``cfg.json`` files are interchangeable)."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from functools import partial
from typing import Literal, Optional

import torch


class Serializable:
    def to_dict(self) -> dict:
        return dataclasses.asdict(self)

    @classmethod
    def from_dict(cls, d: dict, drop_extra_fields: bool = False):
        names = {f.name for f in dataclasses.fields(cls)}
        extra = set(d) - names
        if extra and not drop_extra_fields:
            raise ValueError(f"unknown fields for {cls.__name__}: {sorted(extra)}")
        kw = {k: v for k, v in d.items() if k in names}
        for f in dataclasses.fields(cls):
            if f.name in kw and isinstance(kw[f.name], dict) and dataclasses.is_dataclass(f.type if isinstance(f.type, type) else None):
                kw[f.name] = f.type.from_dict(kw[f.name], drop_extra_fields)
        return cls(**kw)


@dataclass
class SparseCoderConfig(Serializable):
    """sparsify/config.py:9-88."""

    dtype: Literal["none", "float32", "float16", "bfloat16"] = "none"
    activation: Literal["groupmax", "topk", "batchtopk"] = "topk"
    expansion_factor: int = 32
    normalize_decoder: bool = True
    num_latents: int = 0
    k: int = 32
    skip_connection: bool = False
    transcode: bool = False
    tp_output: bool = True
    n_targets: int = 0
    n_sources: int = 0
    normalize_io: bool = False
    divide_cross_layer: bool = False
    train_post_encoder: bool = True
    post_encoder_scale: bool = False
    per_source_tied: bool = False
    secondary_target_tied: bool = False
    coalesce_topk: Literal["none", "concat", "per-layer", "group"] = "none"
    topk_coalesced: bool = False
    use_fp8: bool = False

    @property
    def torch_dtype(self):
        if self.dtype == "none":
            return None
        return dict(float32=torch.float32, float16=torch.float16, bfloat16=torch.bfloat16)[self.dtype]

    @property
    def do_coalesce_topk(self):
        return self.coalesce_topk != "none"


SaeConfig = SparseCoderConfig
TranscoderConfig = partial(SparseCoderConfig, transcode=True)


@dataclass
class TrainConfig(Serializable):
    """sparsify/config.py:96-218."""

    sae: SparseCoderConfig = field(default_factory=SparseCoderConfig)
    batch_size: int = 32
    grad_acc_steps: int = 1
    micro_acc_steps: int = 1
    loss_fn: Literal["ce", "fvu", "kl", "kl-fvu"] = "fvu"
    kl_coeff: float = 1.0
    filter_bos: bool = False
    remove_first_token: bool = False
    remove_transcoded_modules: bool = False
    optimizer: Literal["adam", "adam8", "muon", "signum"] = "signum"
    lr: Optional[float] = None
    lr_warmup_steps: int = 1000
    b1: float = 0.9
    b2: float = 0.999
    force_lr_warmup: bool = False
    k_decay_steps: int = 0
    k_anneal_mul: int = 10
    dead_feature_threshold: int = 10_000_000
    dead_latent_penalty: float = 0.0
    hookpoints: list = field(default_factory=list)
    hookpoints_in: list = field(default_factory=list)
    init_seeds: list = field(default_factory=lambda: [0])
    layers: list = field(default_factory=list)
    per_layer_k: list = field(default_factory=list)
    layer_stride: int = 1
    cross_layer: int = 0
    tp: int = 1
    save_every: int = 1000
    save_best: bool = False
    finetune: Optional[str] = None
    restart_epoch: bool = False
    log_to_wandb: bool = True
    run_name: Optional[str] = None
    wandb_log_frequency: int = 1
    save_dir: str = "checkpoints"

    @property
    def distribute_modules(self) -> bool:
        return self.tp > 1

    def __post_init__(self):
        if isinstance(self.sae, dict):
            self.sae = SparseCoderConfig.from_dict(self.sae, drop_extra_fields=True)
        if self.layers and self.layer_stride != 1:
            raise ValueError("Cannot specify both `layers` and `layer_stride`.")
        if not self.init_seeds:
            raise ValueError("Must specify at least one random seed.")
