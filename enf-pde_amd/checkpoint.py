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
    """Inverse of save_tree; also reads an archive exported from a Flax tree as described in the module docstring."""
    with np.load(path) as z:
        flat = {k: torch.as_tensor(z[k]).to(device=device, dtype=dtype) for k in z.files}
    return unflatten_tree(flat)
