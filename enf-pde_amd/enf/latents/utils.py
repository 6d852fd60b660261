"""Latent pose initialisers, mirroring enf/latents/utils.py:36-109 (torch, float32)."""
import math

import torch


def init_positions_grid(num_signals, num_latents, num_dims):
    """Cell-centred grid on [-1, 1]^d, 'ij' order (utils.py:73-103)."""
    k = int(round(num_latents ** (1.0 / num_dims)))
    assert abs(round(num_latents ** (1.0 / num_dims), 5) % 1) < 1e-5, \
        "num_latents must be a power of the number of position dimensions"
    ax = torch.linspace(-1 + 1 / k, 1 - 1 / k, k, dtype=torch.float64)
    g = torch.stack(torch.meshgrid(*[ax] * num_dims, indexing="ij"), dim=-1).reshape(-1, num_dims)
    return g[None].repeat(num_signals, 1, 1).float()


def init_positions_polar(num_signals, num_latents, num_dims=2):
    """(phi, theta) grid with twice the resolution along phi (utils.py:36-70)."""
    n = num_latents // 2
    assert abs(round(n ** (1.0 / num_dims), 5) % 1) < 1e-5, \
        "num_latents must be a power of the number of position dimensions"
    k = int(round(n ** (1.0 / num_dims)))
    gphi = torch.linspace(math.pi / (2 * k), 2 * math.pi - math.pi / (2 * k), 2 * k, dtype=torch.float64)
    gth = torch.linspace((math.pi / 2) / k, math.pi - (math.pi / 2) / k, k, dtype=torch.float64)
    g = torch.stack(torch.meshgrid(gphi, gth, indexing="ij"), dim=-1).reshape(-1, num_dims)
    return g[None].repeat(num_signals, 1, 1).float()


def init_ori_rotation_invariant_s2(num_signals, num_latents, num_dims):
    """One orientation per latent, atan2(pos0, pos1) (utils.py:106-109)."""
    pos = init_positions_grid(num_signals, num_latents, num_dims)
    return torch.atan2(pos[:, :, 0], pos[:, :, 1])[:, :, None]


def init_positions_ball(num_signals, num_latents):
    """Euler angles (alpha, beta, gamma) on a Fibonacci lattice plus the radius 0.75 (utils.py:4-33): shape (S, Z, 4)."""
    i = torch.arange(1, num_latents + 1, dtype=torch.float32)
    alpha = torch.arccos(1 - 2 * i / (num_latents + 1))
    beta = (math.pi * (1 + 5 ** 0.5)) * i
    gamma = torch.arange(num_latents, dtype=torch.float32) * (2 * math.pi / num_latents)
    pos = torch.stack([alpha, beta, gamma, torch.full_like(alpha, 0.75)], dim=-1)
    return pos[None].repeat(num_signals, 1, 1)
