"""What `--resume 1` carries besides the model weights: best-so-far / early-stopping bookkeeping, the host RNG streams the
reference's samplers draw from (`random`, `numpy.random`: macr_mf/load_data.py:543-566, utility/load_data.py:174-254) and
the device samplers' positions.

Plain JSON -- numbers and lists only, nothing that executes on load -- written by the main rank next to the checkpoint.
On a resume ONLY the main rank reads files (a node-local checkpoint directory exists on one node); the bookkeeping and the
weights reach the other ranks through the process group, so every rank continues from the same sampler and RNG positions.
"""
import json
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from . import sharding


def _rng_state():
    ver, internal, gauss = random.getstate()
    kind, keys, pos, has_gauss, cached = np.random.get_state()
    return {"py": [ver, list(internal), gauss],
            "np": [kind, [int(x) for x in keys], int(pos), int(has_gauss), float(cached)]}


def _set_rng_state(s):
    ver, internal, gauss = s["py"]
    random.setstate((ver, tuple(internal), gauss))
    kind, keys, pos, has_gauss, cached = s["np"]
    np.random.set_state((kind, np.asarray(keys, dtype=np.uint32), pos, has_gauss, cached))


def save(path, bookkeeping):
    """bookkeeping: a dict of numbers / strings / lists (the CLI's own scalars).  The host RNG states are added here."""
    doc = {"format": 1, "bookkeeping": bookkeeping, "rng": _rng_state()}
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(doc, f)
    os.replace(tmp, path)


def _share(obj):
    """rank 0's python object on every rank"""
    if sharding.world()[1] == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def load_everywhere(model, path):
    """A weights file read by the MAIN rank only reaches every rank's model: the main rank shares the layout of the state
    dict (names, shapes, dtypes, the non-tensor entries), every tensor is broadcast from it, and every rank calls its own
    `model.load_state_dict` on the complete dict -- a replica copies it, a row-sharded model (mf.ShardedBPRMF) takes the
    rows it owns.  Never goes through the model's `state_dict()`: that one is collective for a row-sharded model and not
    every rank has a model to reassemble yet."""
    sd = None
    if sharding.is_main():
        sd = torch.load(path, map_location="cpu", weights_only=True)
    if sharding.world()[1] == 1:
        model.load_state_dict({k: (v.to(model.device) if isinstance(v, torch.Tensor) else v) for k, v in sd.items()})
        return
    layout = _share(None if sd is None else
                    [(k, tuple(v.shape), str(v.dtype).split(".")[-1]) if isinstance(v, torch.Tensor) else (k, None, v)
                     for k, v in sd.items()])
    full = {}
    for k, shape, meta in layout:
        if shape is None:
            full[k] = meta
            continue
        if sharding.is_main():
            t = sd[k].to(model.device).contiguous()
        else:
            t = torch.empty(shape, dtype=getattr(torch, meta), device=model.device)
        sharding.broadcast_params([t])
        full[k] = t
    model.load_state_dict(full)


def resume(model, latest_checkpoint, side_path_of, warn=print):
    """The resume step of both CLIs.  latest_checkpoint() -> (epoch, weights path) or None, evaluated on the main rank
    only; side_path_of(epoch) -> the JSON written by save().  The main rank reads the weights and their tensors are
    broadcast (load_everywhere), the host RNG streams are restored on every rank; returns (epoch or None, bookkeeping dict
    or None).  Weights without a usable side file (written before the JSON format, or by --save_flag alone) resume the model
    only: the caller is told, because best-so-far / early-stopping state and the sampler positions start over."""
    found, doc = None, None
    if sharding.is_main():
        found = latest_checkpoint()
        if found is not None:
            side = side_path_of(found[0])
            if os.path.exists(side):
                with open(side) as f:
                    doc = json.load(f)
                if doc.get("format") != 1:
                    warn("WARNING: --resume: %s has format %r (this build reads format 1): resuming the weights only -- bests, "
                         "early-stopping count, RNG and sampler positions start over" % (side, doc.get("format")))
                    doc = None
            else:
                legacy = side[:-len(".json")] + ".pt" if side.endswith(".json") else None
                warn("WARNING: --resume: no %s next to %s%s: resuming the weights only -- bests, early-stopping count, RNG and "
                     "sampler positions start over" % (os.path.basename(side), os.path.basename(found[1]),
                                                       " (a legacy %s exists; that pickle format is no longer read)"
                                                       % os.path.basename(legacy) if legacy and os.path.exists(legacy) else ""))
    epoch, path, doc = _share((None, None, None) if found is None else (found[0], found[1], doc))
    if epoch is None:
        return None, None
    load_everywhere(model, path)
    if doc is None:
        return epoch, None
    _set_rng_state(doc["rng"])
    return epoch, doc["bookkeeping"]
