"""Latent ODE models, mirroring experiments/fitting/ode_models."""
from .ponita_ode_g import PonitaGen, PonitaODEGen, PolynomialFeatures, sep_gconv
from .mlp_ode import MLPODE

__all__ = ["PonitaGen", "PonitaODEGen", "PolynomialFeatures", "sep_gconv", "MLPODE"]
