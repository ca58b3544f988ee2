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


def resume(model, latest_checkpoint, side_path_of):
    """The resume step of both CLIs.  latest_checkpoint() -> (epoch, weights path) or None, evaluated on the main rank
    only; side_path_of(epoch) -> the JSON written by save().  Loads the weights on the main rank, broadcasts them, restores
    the host RNG streams on every rank and returns (epoch or None, bookkeeping dict or None)."""
    found, doc = None, None
    if sharding.is_main():
        found = latest_checkpoint()
        if found is not None:
            model.load_state_dict(torch.load(found[1], map_location=model.device, weights_only=True))
            side = side_path_of(found[0])
            if os.path.exists(side):
                with open(side) as f:
                    doc = json.load(f)
                if doc.get("format") != 1:
                    doc = None
    epoch, doc = _share((None if found is None else found[0], doc))
    if epoch is None:
        return None, None
    if sharding.world()[1] > 1:
        sd = model.state_dict()
        sharding.broadcast_params([t for t in sd.values() if isinstance(t, torch.Tensor)])
        scalars = _share({k: v for k, v in sd.items() if not isinstance(v, torch.Tensor)})
        if not sharding.is_main():
            sd.update(scalars)
            model.load_state_dict(sd)
    if doc is None:
        return epoch, None
    _set_rng_state(doc["rng"])
    return epoch, doc["bookkeeping"]
