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
import importlib

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


# ---- pickling this package's models (what the reference's own save code does: torch.save({'model': deepcopy(model).half()})) ----
def _reference_model_class(et_model):
    """the reference's Model class of the same module path (models.detector.yolo / yolo_ssod), if its tree is importable"""
    name = type(et_model).__module__.replace("efficientteacher_amd.", "", 1)
    try:
        mod = importlib.import_module(name)
    except ImportError:
        return None
    cls = getattr(mod, "Model", None)
    if cls is None or cls.__module__.startswith("efficientteacher_amd"):
        return None
    return cls


def _rebuild_et_model(module_name, cfg, state_dict, half):
    m = importlib.import_module(module_name).Model(cfg)
    m.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in state_dict.items()}, strict=True)
    m._compute_dtype = torch.bfloat16 if half else torch.float32
    return m


def reduce_model(m, protocol):
    """``Model.__reduce_ex__``: the arenas, slots and weak references of a live model are not state -- its state_dict is.
    Where the reference tree is importable (a reference ``train.py`` running the adapters) the pickle is that of the
    REFERENCE's own Model carrying these weights, fp16 after ``.half()`` as the reference stores them: the files its
    ``after_epoch`` writes (trainer.py:475-481, ssod_trainer.py:391-407) stay loadable by the reference alone.  Elsewhere the
    pickle rebuilds this package's Model from (cfg, state_dict)."""
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    half = m._compute_dtype == torch.bfloat16
    Ref = _reference_model_class(m)
    if Ref is not None:
        ref = Ref(m.cfg)
        ref.load_state_dict(sd, strict=True)
        for k in ("nc", "names", "hyp", "class_weights", "yaml"):
            if hasattr(m, k):
                setattr(ref, k, getattr(m, k))
        if half:
            ref.half()
        # pickle refuses __newobj__ of another class for this object; a stdlib callable applied to the reference module keeps
        # the stream free of names from this package
        return copy.copy, (ref,)
    if half:
        sd = {k: v.half() if v.is_floating_point() else v for k, v in sd.items()}
    return _rebuild_et_model, (type(m).__module__, m.cfg, sd, half)
