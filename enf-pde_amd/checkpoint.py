"""Weight / latent import-export as flat ``.npz`` archives of the Flax parameter tree.

The reference checkpoints with orbax (experiments/fitting/trainers/_base_pde_trainer.py:192-237), which is not
available here; a maintainer exports a trained tree once with
    np.savez(path, **{"/".join(k): v for k, v in flax.traverse_util.flatten_dict(params).items()})
and ``load_tree`` / ``EquivariantCrossAttentionNeF.load_params`` take it from there.  Keys are the tree paths
joined by "/" ("params/cross_attention_blocks_0/attn/a_to_k/kernel", ...), values fp32 arrays.
"""
import numpy as np
import torch


def flatten_tree(tree, prefix=()):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(flatten_tree(v, prefix + (k,)))
        else:
            out["/".join(prefix + (k,))] = v
    return out


def unflatten_tree(flat):
    tree = {}
    for key, v in flat.items():
        node = tree
        parts = key.split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return tree


def save_tree(path, tree):
    """Write a (nested dict of tensors / arrays) tree -- nef params, a latent dict, learning rates -- to ``path``."""
    flat = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in flatten_tree(tree).items()}
    np.savez(path, **flat)


def load_tree(path, device="cpu", dtype=torch.float32):
    """Inverse of save_tree; also reads an archive exported from a Flax tree as described in the module docstring.
    Floating-point entries are cast to ``dtype`` (None = as stored); integer entries (counters) keep their type."""
    with np.load(path) as z:
        flat = {}
        for k in z.files:
            t = torch.as_tensor(z[k])
            flat[k] = t.to(device=device, dtype=dtype) if (dtype is not None and t.is_floating_point()) else t.to(device=device)
    return unflatten_tree(flat)


# --------------------------------------------------------------------------------------------------------------------
# Training-state checkpoints (experiments/fitting/trainers/_base_pde_trainer.py:192-237: the reference saves the whole
# TrainState -- params, every optimiser's count / mu / nu -- plus the config through orbax, and restores it into a freshly
# initialised state).  One ``.npz`` per checkpoint, every array in its own dtype:
#   params/...                         the parameter tree (nef weights, meta-init latents, inner rates, ODE weights)
#   <name>_opt_state/count (int64), <name>_opt_state/mu/<i>, <name>_opt_state/nu/<i>     for every optimiser state
#   step, epoch (int64), rng_state (uint8: torch.Generator.get_state()), config_json (the config as JSON)
def _to_plain(cfg):
    if cfg is None or isinstance(cfg, (bool, int, float, str)):
        return cfg
    if isinstance(cfg, dict):
        return {k: _to_plain(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [_to_plain(v) for v in cfg]
    if hasattr(cfg, "__dict__"):
        return {k: _to_plain(v) for k, v in vars(cfg).items()}
    return str(cfg)


def save_train_state(path, state, config=None, epoch=0):
    """Write a TrainState / NonMetaTrainState (fitting/trainers) to ``path`` (.npz)."""
    import json
    flat = {}
    arr = lambda v: v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    for k, v in flatten_tree(state.params, ("params",)).items():
        flat[k] = arr(v)
    for name, opt in vars(state).items():
        if not name.endswith("opt_state") or opt is None:
            continue
        flat[f"{name}/count"] = np.asarray(int(opt["count"]), dtype=np.int64)
        for part in ("mu", "nu"):
            for i, t in enumerate(opt[part]):
                flat[f"{name}/{part}/{i}"] = arr(t)
    flat["step"] = np.asarray(int(state.step), dtype=np.int64)
    flat["epoch"] = np.asarray(int(epoch), dtype=np.int64)
    flat["rng_state"] = state.rng.get_state().numpy()
    flat["config_json"] = np.frombuffer(json.dumps(_to_plain(config)).encode(), dtype=np.uint8)
    with open(path, "wb") as f:          # (np.savez would append ".npz" to a bare path)
        np.savez(f, **flat)


def load_train_state(path, template, device=None):
    """Restore a checkpoint into a copy of ``template`` (a freshly initialised state of the same trainer, as the reference
    does with its surrogate state): shapes and the set of entries must match.  Returns (state, epoch, config dict)."""
    import copy
    import json
    with np.load(path) as z:
        flat = {k: z[k] for k in z.files}
    state = copy.copy(template)
    tflat = flatten_tree(template.params, ("params",))
    if set(tflat) != {k for k in flat if k.startswith("params/")}:
        raise ValueError("checkpoint parameter tree does not match this trainer's: "
                         f"{sorted(set(tflat) ^ {k for k in flat if k.startswith('params/')})[:6]}")
    to = lambda a, like: torch.as_tensor(a).to(device=device or like.device, dtype=like.dtype)
    new = {}
    for k, like in tflat.items():
        if tuple(flat[k].shape) != tuple(like.shape):
            raise ValueError(f"checkpoint entry {k} has shape {flat[k].shape}, expected {tuple(like.shape)}")
        new[k] = to(flat[k], like)
    state.params = unflatten_tree(new)["params"]
    for name, opt in vars(template).items():
        if not name.endswith("opt_state"):
            continue
        if opt is None:
            if f"{name}/count" in flat:
                raise ValueError(f"checkpoint has {name} but this trainer does not")
            continue
        if f"{name}/count" not in flat:
            raise ValueError(f"checkpoint lacks {name}")
        setattr(state, name, {"count": int(flat[f"{name}/count"]),
                              "mu": [to(flat[f"{name}/mu/{i}"], t) for i, t in enumerate(opt["mu"])],
                              "nu": [to(flat[f"{name}/nu/{i}"], t) for i, t in enumerate(opt["nu"])]})
    state.step = int(flat["step"])
    rng = torch.Generator()
    rng.set_state(torch.as_tensor(flat["rng_state"]))
    state.rng = rng
    return state, int(flat["epoch"]), json.loads(bytes(flat["config_json"]).decode())
