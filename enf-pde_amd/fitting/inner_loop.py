"""MAML inner loop and decode on the HIP path (trainers/pde_trainer.py:122-235, 389-405)."""
import ctypes

import torch

from .. import _lib


def default_meta_sgd_lrs(latent_dim, lr_p=1.0, lr_a=5.0, lr_window=0.0, with_ori=False, device="cuda"):
    """Initial inner learning rates (pde_trainer.py:83-97): scalar for poses/window, (C,) for a."""
    lrs = {"p_pos": torch.full((1,), lr_p, device=device), "a": torch.full((latent_dim,), lr_a, device=device),
           "gaussian_window": torch.full((1,), lr_window, device=device)}
    if with_ori:
        lrs["p_ori"] = torch.full((1,), lr_p, device=device)
    return lrs


def make_masks(num_coords, num_sampled, num_inner_steps, generator=None, device="cuda"):
    """(N_s, S+1) independent column permutations truncated to N_s rows (pde_trainer.py:148-154).
    jax.random.permutation cannot be reproduced; parity tests pass masks explicitly."""
    cols = [torch.randperm(num_coords, generator=generator)[:num_sampled] for _ in range(num_inner_steps + 1)]
    return torch.stack(cols, dim=1).to(device)


FUSED_FIT_INPUTS = __import__("os").environ.get("ENF_FIT_INPUTS") != "0"


def _fit_inputs(latents0, coords, img, masks):
    """enf_fit_inputs (include/enf_hip.h): (lat, xs_all, ys_all, losses) of inner_loop in one launch, or None where the arguments are not
    what the kernel takes (fp32, contiguous, on one GPU, at most four latent components of leading dimension 1)."""
    ts = list(latents0.values()) + [coords, img]
    if not (img.is_cuda and masks.is_cuda and masks.dtype == torch.int64 and masks.dim() == 2 and masks.is_contiguous() and coords.dim() == 2
            and img.dim() == 3 and 1 <= len(latents0) <= _lib.ENF_SGD_MAX_SEGMENTS and masks.shape[0] > 0
            and all(t.dtype == torch.float32 and t.is_contiguous() and t.device == img.device for t in ts)
            and all(v.dim() == 3 and v.shape[0] == 1 for v in latents0.values())
            and len({v.shape[1] for v in latents0.values()}) == 1):
        return None
    B, N, O = img.shape
    Ns, S1 = masks.shape
    Z = next(iter(latents0.values())).shape[1]
    dev = img.device
    lat = {k: torch.empty((B, Z, v.shape[2]), device=dev, dtype=torch.float32) for k, v in latents0.items()}
    xs = torch.empty((S1, Ns, coords.shape[1]), device=dev, dtype=torch.float32)
    ys = torch.empty((S1, B, Ns, O), device=dev, dtype=torch.float32)
    losses = torch.empty(S1, device=dev, dtype=torch.float32)
    comps = (_lib.EnfFitComponent * _lib.ENF_SGD_MAX_SEGMENTS)()
    keep = []
    for i, (k, v) in enumerate(latents0.items()):
        src = v.detach()
        keep.append(src)
        comps[i] = _lib.EnfFitComponent(src.data_ptr(), lat[k].data_ptr(), v.shape[2], 0)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.launch(dev, _lib.load().enf_fit_inputs, len(latents0), comps, B, Z, N, Ns, S1, coords.shape[1], O, coords.data_ptr(), img.data_ptr(),
                masks.data_ptr(), xs.data_ptr(), ys.data_ptr(), losses.data_ptr(), st)
    return lat, xs, ys, losses


def _pose(lat, num_ori_dims):
    return torch.cat((lat["p_pos"], lat["p_ori"]), dim=-1) if num_ori_dims > 0 else lat["p_pos"]


def meta_sgd_update(lat, grads, lrs, scale):
    """One meta-SGD update of every latent component in a single HIP launch (pde_trainer.py:206-219):
    ``lat[k] - lrs[k] * (scale * grads[k])`` for the keys of ``grads``; the other entries of ``lat`` are passed through.
    A gradient may be a column slice of a wider (..., P) array (the pose gradient split into p_pos / p_ori)."""
    lib = _lib.load()
    new = dict(lat)
    segs = (_lib.EnfSgdSegment * _lib.ENF_SGD_MAX_SEGMENTS)()
    keep = []
    for i, (k, g) in enumerate(grads.items()):
        x = lat[k].float().contiguous()
        w = x.shape[-1]
        if g.dtype != torch.float32 or g.shape != x.shape or g.stride(-1) != 1 or \
                any(g.stride(d) != g.stride(d + 1) * g.shape[d + 1] for d in range(g.dim() - 2)):
            g = g.float().contiguous()
        lr = lrs[k].detach().float().contiguous()
        out = torch.empty_like(x)
        keep += [x, g, lr]
        segs[i] = _lib.EnfSgdSegment(x.data_ptr(), g.data_ptr(), lr.data_ptr(), out.data_ptr(), x.numel(), w,
                                     g.stride(-2) if g.dim() > 1 else w, lr.numel(), 0)
        new[k] = out
    st = ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)
    _lib.launch(out.device, lib.enf_meta_sgd_update, len(grads), segs, float(scale), st)
    return new


def inner_loop(nef, nef_params, latents0, lrs, coords, img, masks, optimize_gaussian_window=False,
               noise_pos=0.0, generator=None):
    """Fit per-signal latents with S steps of meta-SGD (pde_trainer.py:156-235).

    latents0 : {'p_pos','a','gaussian_window'[,'p_ori']} with leading dim 1 (the meta-init)
    lrs      : inner learning rates, same keys
    coords   : (N, dx) grid;  img: (B, N, O) targets;  masks: (N_s, S+1) long
    Each step is one HIP forward, the fused loss/d-out kernel and one HIP backward-to-latents
    (nef.mse_value_and_latent_grads: no autograd graph); the gradient of the batch-mean loss is multiplied by B
    (pde_trainer.py:207) so signals are independent.
    Returns (loss on the last mask, fitted latents dict with leading dim B).
    """
    B = img.shape[0]
    S = masks.shape[1] - 1
    n_ori = nef.cross_attn_invariant.num_z_ori_dims
    fused = _fit_inputs(latents0, coords, img, masks) if FUSED_FIT_INPUTS else None
    if fused is not None:
        # the signals' copies of the latent initialisation (pde_trainer.py:157-159), the coordinates and targets of all S+1 steps
        # gathered once (:193-197) and the zeroed loss accumulators: ONE launch (enf_fit_inputs) instead of eight framework kernels
        lat, xs_all, ys_all, losses = fused
    else:
        lat = {k: v.detach().repeat_interleave(B, dim=0) for k, v in latents0.items()}       # pde_trainer.py:157-159 (a fresh tensor)
        masks_t = masks.t().contiguous()                                     # (a gather inherits the strides of a transposed index)
        xs_all = coords[masks_t]                                             # (S+1, N_s, dx)
        ys_all = img[:, masks_t].transpose(0, 1).float().contiguous()        # (S+1, B, N_s, O)
        losses = torch.zeros(S + 1, device=img.device, dtype=torch.float32)  # one accumulator per step, zeroed in one fill
    if noise_pos:                                                                             # pde_trainer.py:162-167
        lat["p_pos"] = lat["p_pos"] + torch.randn(lat["p_pos"].shape, generator=generator,
                                                  device="cpu").to(lat["p_pos"].device) * noise_pos
    n_pos = lat["p_pos"].shape[-1]
    for s in range(S):                                                  # pde_trainer.py:191
        xs = xs_all[s][None].expand(B, -1, -1)                          # stride-0 batch
        _, dp, da, dsig = nef.mse_value_and_latent_grads(nef_params, xs, _pose(lat, n_ori), lat["a"],
                                                         lat.get("gaussian_window"), ys_all[s], loss_out=losses[s:s + 1])
        # the gradient of the batch-mean loss times B (pde_trainer.py:206), scaled by the learned rates (:215-219);
        # sigma only moves when asked to (:209-212)
        grads = {"p_pos": dp[..., :n_pos], "a": da}
        if n_ori > 0:
            grads["p_ori"] = dp[..., n_pos:]
        if optimize_gaussian_window and dsig is not None:
            grads["gaussian_window"] = dsig
        lat = meta_sgd_update(lat, grads, lrs, B)
    with torch.no_grad():                                               # pde_trainer.py:225-235
        xs = xs_all[S][None].expand(B, -1, -1)
        out = nef.apply(nef_params, xs, _pose(lat, n_ori), lat["a"], lat.get("gaussian_window")).float().contiguous()
        st = ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)
        if out.shape != ys_all[S].shape:
            raise AssertionError(f"targets have shape {tuple(ys_all[S].shape)}, expected {tuple(out.shape)}")
        _lib.launch(out.device, _lib.load().enf_mse_value_grad, out.data_ptr(), ys_all[S].data_ptr(), out.numel(), 1.0, None,
                    losses[S:].data_ptr(), st)
    return losses[S], lat


@torch.no_grad()
def decode(nef, nef_params, coords, p, a, window, chunk=None):
    """Reconstruct (B, N, O) on the full grid (pde_trainer.py:393-402).  The fused kernel tiles over
    queries itself, so ``chunk`` is optional (the reference chunks by max_num_sampled_points)."""
    B = p.shape[0]
    x = coords[None].expand(B, -1, -1) if coords.dim() == 2 else coords
    if chunk is None:
        return nef.apply(nef_params, x, p, a, window)
    outs = [nef.apply(nef_params, x[:, i:i + chunk], p, a, window) for i in range(0, x.shape[1], chunk)]
    return torch.cat(outs, dim=1)
