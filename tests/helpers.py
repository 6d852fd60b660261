"""Shared builders for the parity tests: seeded inputs following SURVEY.md 8d."""
import numpy as np

from oracle import enf_ref_np as R


def make_cfg(invariant="rel_pos_periodic", D=128, H=2, C=16, O=1, num_in=2, freq=(0.05, 0.1), use_window=True):
    if invariant in ("ball", "ball_lat"):
        num_in = 3
    return dict(num_hidden=D, num_heads=H, latent_dim=C, num_out=O, invariant=invariant, num_in=num_in,
                embedding_freq_multiplier=tuple(freq), use_gaussian_window=use_window, num_layers=0,
                condition_value_transform=True)


def make_inputs(cfg, B, N, Z, seed=0, sigma_scale=1.0):
    """x, p, a, sigma (float64 numpy) for an invariant: poses = reference init + jitter, a = 1 + 0.1 N(0,1)."""
    rng = np.random.default_rng(seed)
    name = cfg["invariant"]
    spec = R.invariant_spec(name, cfg.get("num_in", 2))
    dx = spec["dx"]
    if name in ("latitude_periodic", "polar_periodic"):
        x = np.stack([rng.uniform(0, 2 * np.pi, (B, N)), rng.uniform(0.05, np.pi - 0.05, (B, N))], -1)
        p = np.stack([rng.uniform(0, 2 * np.pi, (B, Z)), rng.uniform(0.2, np.pi - 0.2, (B, Z))], -1)
        sigma = np.full((B, Z, 1), 0.8) * sigma_scale
    elif name in ("ball", "ball_lat"):        # x = (phi, theta, r); p = Euler angles (alpha, beta, gamma) + radius
        x = np.stack([rng.uniform(0, 2 * np.pi, (B, N)), rng.uniform(0.05, np.pi - 0.05, (B, N)), rng.uniform(0.1, 1.0, (B, N))], -1)
        p = np.stack([rng.uniform(0, 2 * np.pi, (B, Z)), rng.uniform(0.2, np.pi - 0.2, (B, Z)),
                      rng.uniform(0, 2 * np.pi, (B, Z)), rng.uniform(0.5, 1.0, (B, Z))], -1)
        sigma = np.full((B, Z, 1), 1.0) * sigma_scale
    else:
        x = rng.uniform(-1, 1, (B, N, dx))
        p = rng.uniform(-1, 1, (B, Z, spec["z_pos"]))
        if spec["z_ori"]:
            p = np.concatenate([p, rng.uniform(-np.pi, np.pi, (B, Z, 1))], -1)
        k = max(1.0, round(Z ** (1.0 / spec["z_pos"])))
        sigma = np.full((B, Z, 1), spec["z_pos"] / k) * sigma_scale
    sigma = sigma * (1 + 0.1 * rng.uniform(-1, 1, sigma.shape))
    a = 1 + 0.1 * rng.standard_normal((B, Z, cfg["latent_dim"]))
    return x, p, a, sigma


def build_nef(cfg, precision):
    """Product-side module for an oracle cfg dict."""
    from types import SimpleNamespace as NS
    from enf_pde_amd.enf.models import EquivariantCrossAttentionNeF
    from enf_pde_amd.enf.steerable_attention.invariant import get_ca_invariant
    inv = get_ca_invariant(NS(invariant_type=cfg["invariant"], num_in=cfg.get("num_in", 2)))
    return EquivariantCrossAttentionNeF(
        num_hidden=cfg["num_hidden"], num_heads=cfg["num_heads"], num_layers=0, num_out=cfg["num_out"],
        latent_dim=cfg["latent_dim"], cross_attn_invariant=inv, self_attn_invariant=inv, embedding_type="rff",
        embedding_freq_multiplier=cfg["embedding_freq_multiplier"], condition_value_transform=True,
        use_gaussian_window=cfg.get("use_gaussian_window", True), precision=precision)
