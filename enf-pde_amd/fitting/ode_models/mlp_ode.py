"""MLPODE: the non-equivariant baseline latent ODE (experiments/fitting/ode_models/mlp_ode.py:5-42): two 4-layer MLPs on
[p | a - 1] per latent -- B Z rows of plain GEMMs, run by the library."""
import math

import torch
import torch.nn.functional as Fnn


class MLPODE:
    def __init__(self, num_hidden, num_layers, scalar_num_out, vec_num_out):
        self.num_hidden, self.num_layers, self.scalar_num_out, self.vec_num_out = num_hidden, num_layers, scalar_num_out, vec_num_out

    def init(self, key, latents, device=None):
        p, a, _ = latents
        device = device or a.device
        gen = torch.Generator().manual_seed(int(key))
        out = {}
        for net, n_out in (("mlp_a", self.scalar_num_out), ("mlp_p", 2 * self.vec_num_out)):
            dims = [p.shape[-1] + a.shape[-1], self.num_hidden, self.num_hidden, self.num_hidden, n_out]
            out[net] = {}
            for i in range(4):
                k = torch.empty(dims[i], dims[i + 1])
                torch.nn.init.trunc_normal_(k, 0.0, 1.0, -2.0, 2.0, generator=gen)
                out[net][f"layers_{2 * i}"] = {"kernel": (k * (math.sqrt(1.0 / dims[i]) / 0.87962566103423978)).to(device),
                                               "bias": torch.zeros(dims[i + 1], device=device)}
        return {"params": out}

    def load_params(self, tree, device="cuda"):
        conv = lambda t: {k: conv(v) for k, v in t.items()} if isinstance(t, dict) else \
            torch.as_tensor(t, dtype=torch.float32).to(device).contiguous()
        return conv(tree)

    def apply(self, params, latents):
        p, a, window = latents
        h = torch.cat([p, a - 1], -1)                                     # a has mean 1 (mlp_ode.py:35)
        out = []
        for net in ("mlp_p", "mlp_a"):
            x = h
            for i in (0, 2, 4):
                L = params["params"][net][f"layers_{i}"]
                x = Fnn.gelu(x @ L["kernel"] + L["bias"], approximate="tanh")
            L = params["params"][net]["layers_6"]
            out.append(x @ L["kernel"] + L["bias"])
        return out[0], out[1], torch.zeros_like(window)
