"""Optimizer checkpoint layout of the reference (sparsify/utils.py:193-277, trainer.py:923-937, 337-348).

The reference writes ``optimizer_{i}.pt`` with ``save_sharded(optimizer.state_dict(), ..., save_st=False, unflatten=True)``:
the nested state dict FLATTENED into one dict whose keys are tuples -- the path to each leaf, with list positions
written as ``("list", i)`` -- and reads it back with ``unflatten_dict``.  Latent-sharded tensors are gathered to rank 0
first, so the file never depends on the tp degree.  These helpers produce and consume exactly that layout (the torchao
8-bit state branch excepted: that optimizer is not built here), so a run checkpointed by the reference resumes here and
vice versa."""
from __future__ import annotations

from typing import Any

import torch

# the keys torch.optim.Adam keeps in every param group; written so that torch's Adam (the reference's optimizer) can
# step after loading a file saved here (its load_state_dict takes the SAVED group dict)
TORCH_ADAM_GROUP_DEFAULTS = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                                 differentiable=False, fused=None, decoupled_weight_decay=False)


def flatten_state(d: Any) -> dict:
    """Nested dicts / lists -> {path tuple: leaf} (utils.py:193-234)."""
    out = {}

    def walk(node, path):
        if isinstance(node, list):
            for i, x in enumerate(node):
                walk(x, path + (("list", i),))
        elif isinstance(node, dict):
            for k, v in node.items():
                walk(v, path + (k,))
        else:
            out[path] = node

    walk(d, ())
    return out


def is_flat(d: dict) -> bool:
    return bool(d) and all(isinstance(k, tuple) for k in d)


def unflatten_state(flat: dict) -> Any:
    """The inverse of ``flatten_state`` (utils.py:237-277)."""
    if len(flat) == 1 and () in flat:
        return flat[()]
    groups: dict = {}
    for k, v in flat.items():
        groups.setdefault(k[0], {})[k[1:]] = v
    first = next(iter(groups))
    if isinstance(first, tuple) and len(first) == 2 and first[0] == "list":
        return [unflatten_state(groups[k]) for k in sorted(groups, key=lambda t: t[1])]
    return {k: unflatten_state(v) for k, v in groups.items()}


def to_reference_layout(state_dict: dict, step: int) -> dict:
    """``adam.Adam.state_dict()`` -> the flattened dict the reference writes: torch.optim.Adam's per-parameter ``step``
    tensors and group keys are filled in; our single step counter travels under a key the reference ignores."""
    sd = dict(state_dict)
    sd.pop("clt_step", None)
    state = {}
    for i, st in sd["state"].items():
        st = dict(st)
        st.setdefault("step", torch.tensor(float(step)))
        state[i] = st
    groups = [{**TORCH_ADAM_GROUP_DEFAULTS, **g} for g in sd["param_groups"]]
    flat = flatten_state({"state": state, "param_groups": groups})
    flat[("clt_step",)] = int(step)
    return flat


def from_file_layout(blob: dict):
    """What ``torch.load(optimizer_0.pt)`` returned (the reference's flattened layout, or the nested dict older
    checkpoints of this package hold) -> (nested state dict for ``Optimizer.load_state_dict``, step count or None)."""
    if "state" in blob and not is_flat(blob):  # nested, as saved by this package before the flattened layout
        nested, step = dict(blob), blob.get("clt_step")
    else:
        nested = unflatten_state(blob)
        step = nested.pop("clt_step", None)
    if step is None:  # a reference-written file: torch.optim.Adam keeps one step tensor per parameter
        steps = [int(st["step"]) for st in nested.get("state", {}).values() if "step" in st]
        step = max(steps) if steps else None
    # flatten_state keeps leaves only, so an EMPTY state (SignSGD keeps none; Adam before its first step) leaves no
    # "state" key behind in the file: Optimizer.load_state_dict needs it
    nested.setdefault("state", {})
    nested.setdefault("param_groups", [])
    for g in nested.get("param_groups", []):
        if isinstance(g.get("betas"), list):
            g["betas"] = tuple(g["betas"])
    return nested, step
