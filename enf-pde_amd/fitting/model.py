"""get_model_pde: config -> EquivariantCrossAttentionNeF (experiments/fitting/__init__.py:14-38)."""
import math

from ..enf.models import EquivariantCrossAttentionNeF
from ..enf.steerable_attention.invariant import get_sa_invariant, get_ca_invariant
from .ode_models import MLPODE, PonitaODEGen


def get_model_pde(cfg, precision="bf16"):
    """Returns ``(nef, ode_model)`` like the reference.  ``cfg`` is any attribute-style config with the reference's
    ``nef`` keys (config_navier_stokes.yaml:33-55) and, for the latent ODE, ``node`` keys (:57-69); without a ``node``
    section ``ode_model`` is None."""
    self_attn_invariant = get_sa_invariant(cfg.nef)
    cross_attn_invariant = get_ca_invariant(cfg.nef)
    assert math.sqrt(cfg.nef.num_latents)
    nef = EquivariantCrossAttentionNeF(
        num_hidden=cfg.nef.num_hidden, num_heads=cfg.nef.num_heads, num_layers=cfg.nef.num_layers,
        num_out=cfg.nef.num_out, latent_dim=cfg.nef.latent_dim,
        self_attn_invariant=self_attn_invariant, cross_attn_invariant=cross_attn_invariant,
        embedding_type=cfg.nef.embedding_type,
        embedding_freq_multiplier=[cfg.nef.embedding_freq_multiplier_invariant, cfg.nef.embedding_freq_multiplier_value],
        condition_value_transform=cfg.nef.condition_value_transform,
        use_gaussian_window=cfg.nef.use_gaussian_window, precision=precision)
    node = getattr(cfg, "node", None)
    if node is None:
        return nef, None
    if node.name == "mlp":                                              # experiments/fitting/__init__.py:41-47
        ode_model = MLPODE(num_hidden=node.num_hidden, num_layers=node.num_layers, scalar_num_out=cfg.nef.latent_dim, vec_num_out=1)
    elif node.name == "ponita":                                         # :48-61 (kernel_size is fixed to "global" there)
        ode_model = PonitaODEGen(num_hidden=node.num_hidden, num_layers=node.num_layers, scalar_num_out=cfg.nef.latent_dim,
                                 invariant=self_attn_invariant, vec_num_out=1, basis_dim=node.basis_dim, degree=node.degree,
                                 widening_factor=node.widening_factor, kernel_size="global", global_pool=False)
    else:
        raise ValueError(f"Unknown ODE model: {node.name}")
    return nef, ode_model
