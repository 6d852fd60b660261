"""get_model_pde: config -> EquivariantCrossAttentionNeF (experiments/fitting/__init__.py:14-38)."""
import math

from ..enf.models import EquivariantCrossAttentionNeF
from ..enf.steerable_attention.invariant import get_sa_invariant, get_ca_invariant


def get_model_pde(cfg, precision="bf16"):
    """Returns ``(nef, ode_model)`` like the reference; ``ode_model`` is None here (the latent ODE
    is a "next" row).  ``cfg`` is any attribute-style config with the reference's
    ``nef`` keys (config_navier_stokes.yaml:33-55)."""
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
    return nef, None
