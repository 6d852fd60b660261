"""Latent container, mirroring enf/latents/autodecoder.py:8-73.

``init()`` returns the same ``{'params': {'p_pos', 'a', 'gaussian_window'[, 'p_ori']}}`` dict the
Flax module produces (autodecoder.py:21-33,56); ``apply(params, idx)`` returns ``(p, a, window)``.
"""
import math

import torch

from .utils import init_positions_grid, init_positions_polar, init_positions_ball, init_ori_rotation_invariant_s2


class PositionOrientationFeatureAutodecoder:
    def __init__(self, num_signals, num_latents, latent_dim, num_pos_dims, num_ori_dims,
                 gaussian_window_size=None, frequency_parameter=None, coordinate_system="cartesian"):
        self.num_signals, self.num_latents, self.latent_dim = num_signals, num_latents, latent_dim
        self.num_pos_dims, self.num_ori_dims = num_pos_dims, num_ori_dims
        self.gaussian_window_size = gaussian_window_size
        self.frequency_parameter = frequency_parameter
        self.coordinate_system = coordinate_system

    def init(self, key=None, device="cuda"):
        S, Z, d = self.num_signals, self.num_latents, self.num_pos_dims
        if self.coordinate_system == "cartesian":
            p_pos = init_positions_grid(S, Z, d)
            k = int(round(Z ** (1.0 / d), 5))
            gw = d / k                                                   # autodecoder.py:38-43
        elif self.coordinate_system == "polar":
            p_pos = init_positions_polar(S, Z, d)
            k = int(round((Z // 2) ** (1.0 / d), 5))
            gw = d * math.pi / k                                         # autodecoder.py:45-51
        elif self.coordinate_system == "ball":
            p_pos = init_positions_ball(S, Z)
            gw = 1.0                                                     # autodecoder.py:53-54
        else:
            raise ValueError(f"unknown coordinate system {self.coordinate_system}")
        P = {"p_pos": p_pos}
        if self.num_ori_dims > 0:
            assert d == 2, "Only implemented for 2D"                     # autodecoder.py:28
            P["p_ori"] = init_ori_rotation_invariant_s2(S, Z, d)
        P["a"] = torch.ones(S, Z, self.latent_dim)                       # autodecoder.py:33
        P["gaussian_window"] = torch.full((S, Z, 1), float(gw))          # autodecoder.py:56
        return {"params": {k_: v.to(device=device, dtype=torch.float32) for k_, v in P.items()}}

    def apply(self, params, idx):
        P = params["params"]
        p = torch.cat((P["p_pos"][idx], P["p_ori"][idx]), dim=-1) if self.num_ori_dims > 0 else P["p_pos"][idx]
        return p, P["a"][idx], P["gaussian_window"][idx]                 # autodecoder.py:58-73
