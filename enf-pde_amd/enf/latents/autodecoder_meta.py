"""Meta (shared-initialisation) latent container, mirroring enf/latents/autodecoder_meta.py:6-25."""
import torch

from .autodecoder import PositionOrientationFeatureAutodecoder


class PositionOrientationFeatureAutodecoderMeta(PositionOrientationFeatureAutodecoder):
    def apply(self, params):
        P = params["params"]
        p = torch.cat((P["p_pos"], P["p_ori"]), dim=-1) if self.num_ori_dims > 0 else P["p_pos"]
        window = P["gaussian_window"] if self.gaussian_window_size is not None else None   # autodecoder_meta.py:21-24
        return p, P["a"], window
