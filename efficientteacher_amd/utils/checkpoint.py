"""Checkpoint interchange with the reference.

The reference stores PICKLED modules: ``{'epoch', 'best_fitness', 'model': Model.half(), 'ema': Model.half(),
'updates', 'optimizer', 'wandb_id'}`` (reference trainer/trainer.py:475-481) and resumes with
``ckpt['model'].float().state_dict()`` (:135) / ``ckpt['ema'].float().state_dict()`` (:174).  The parameter and
buffer names of this package's models are identical to the reference's, so interchange is a state_dict hand-over:

* ``load_reference_checkpoint``: needs the reference tree importable (un-pickling looks its classes up), which is
  the situation of a reference user switching trainers.  State-dict-only files written by
  ``save_checkpoint(..., reference_model_factory=None)`` load anywhere.
* ``save_checkpoint``: with ``reference_model_factory`` (a callable building the REFERENCE ``Model(cfg)``) the file
  is exactly what the reference's resume / val / export code expects; without it the modules are replaced by their
  state_dicts (same keys) and ``'format': 'state_dict'`` marks the difference.
"""
import copy

import torch


def _to_state_dict(obj):
    if obj is None:
        return None
    if isinstance(obj, dict):
        return {k: v.float() if torch.is_tensor(v) and v.is_floating_point() else v for k, v in obj.items()}
    return obj.float().state_dict()          # a pickled reference module


def load_reference_checkpoint(path, model=None, ema=None, map_location="cpu", strict=True):
    """Read a reference (or state-dict) checkpoint; fill ``model`` / ``ema.ema`` (this package's Models) from its
    'model' / 'ema' entries.  Returns the checkpoint dict with those two entries replaced by fp32 state_dicts."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    msd, esd = _to_state_dict(ckpt.get("model")), _to_state_dict(ckpt.get("ema"))
    if model is not None and msd is not None:
        model.load_state_dict(msd, strict=strict)
    if ema is not None:
        src = esd if esd is not None else msd
        if src is not None:
            ema.ema.load_state_dict(src, strict=strict)
        if ckpt.get("updates") is not None:
            ema.updates = ckpt["updates"]
    out = dict(ckpt)
    out["model"], out["ema"] = msd, esd
    return out


def save_checkpoint(path, model, ema=None, optimizer=None, epoch=-1, best_fitness=None, reference_model_factory=None):
    """Write a checkpoint in the reference's layout (trainer/trainer.py:475-481)."""
    def export(m):
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        if reference_model_factory is None:
            return {k: v.half() if v.is_floating_point() else v for k, v in sd.items()}
        ref = reference_model_factory()
        ref.load_state_dict(sd, strict=True)
        return copy.deepcopy(ref).half()
    ckpt = {"epoch": epoch, "best_fitness": best_fitness, "model": export(model),
            "ema": export(ema.ema) if ema is not None else None,
            "updates": ema.updates if ema is not None else None,
            "optimizer": optimizer.state_dict() if optimizer is not None else None, "wandb_id": None}
    if reference_model_factory is None:
        ckpt["format"] = "state_dict"
    torch.save(ckpt, path)
    return ckpt
