"""PonitaODEGen: the equivariant message-passing network that gives the latents' time derivative, mirroring
experiments/fitting/ode_models/ponita_ode_g.py (PolynomialFeatures :15-26, ConvBlock :29-49, SepGconv :52-83,
PonitaGen :86-195, PonitaODEGen :198-258).

Work split (per ODE evaluation, B signals x Z latents):
  * pair-wise (B Z^2 rows): invariants of (p, p) and their Kronecker powers (element-wise device ops), the kernel-basis
    MLP (two plain GEMMs: library), and per layer the separable group convolution -- the HIP kernels of
    csrc/enf_ode.hip (``sep_gconv``: fp32 MFMA, the (B, Z, Z, C) kernel tensor is never materialised), forward and
    backward;
  * per-latent (B Z rows): stem, LayerNorm, the widening MLP and the readouts: library GEMMs.
Everything is differentiable (d/d p, d/d a, d/d weights) so that `ode_loss` (pde_trainer.py:411-500) can be trained
through the solver.  Parameters: the reference's flax tree ``{'params': {'ponita': {...}}}`` with device tensors.
"""
import ctypes
import math
import os

import torch
import torch.nn.functional as Fnn

from ... import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


SPLIT_K = 4096      # rows per slice of the pair axis in a weight-gradient GEMM X^T G: (K x P)(P x N) has only K N / tile^2
                    # output tiles, so the library GEMM runs batched over slices of P and the slices are summed


def _xt_dot(X, G):
    """X^T @ G for X (P, K), G (P, N) with P >> K, N: split-K over the pair axis."""
    P = X.shape[0]
    n = P // SPLIT_K
    if n < 2:
        return X.t() @ G
    out = torch.bmm(X[:n * SPLIT_K].view(n, SPLIT_K, -1).transpose(1, 2), G[:n * SPLIT_K].view(n, SPLIT_K, -1)).sum(0)
    if n * SPLIT_K < P:
        out = out + X[n * SPLIT_K:].t() @ G[n * SPLIT_K:]
    return out


class _PairDense(torch.autograd.Function):
    """x @ W + b over the B Z^2 pair rows; the weight gradient is a split-K GEMM (a plain X^T G call picks a tile shape
    for a square problem and runs 20x slower at P = 65536)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return torch.addmm(b, x.reshape(-1, x.shape[-1]), W).view(*x.shape[:-1], W.shape[1])

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        dx = (g2 @ W.t()).view(x.shape) if ctx.needs_input_grad[0] else None
        dW = _xt_dot(x.reshape(-1, x.shape[-1]), g2) if ctx.needs_input_grad[1] else None
        db = g2.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dW, db


class _SepGconv(torch.autograd.Function):
    """out[b,r,:] = bias + sum_s a[b,s,:] * (kb[b,r,s,:] @ W)   (SepGconv.__call__, ponita_ode_g.py:63-83)."""

    @staticmethod
    def forward(ctx, a, kb, W, bias):
        lib = _lib.load()
        a, kb, W = a.contiguous(), kb.contiguous(), W.contiguous()
        B, Z, C = a.shape
        J = kb.shape[-1]
        out = torch.empty_like(a)
        _lib.launch(a.device, lib.enf_ode_conv_forward, B, Z, J, C, _ptr(a), _ptr(kb), Z * J, J, _ptr(W),
                                            _ptr(bias.contiguous() if bias is not None else None), _ptr(out), _stream(a.device))
        ctx.save_for_backward(a, kb, W)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, kb, W = ctx.saved_tensors
        g = g.contiguous()
        B, Z, C = a.shape
        J = kb.shape[-1]
        st = _stream(a.device)
        da = dkb = dW = db = None
        if ctx.needs_input_grad[0]:       # d a[b,s,:] = sum_r g[b,r,:] * kernel[b,r,s,:]: the same contraction, (r, s) swapped
            da = torch.empty_like(a)
            _lib.launch(a.device, lib.enf_ode_conv_forward, B, Z, J, C, _ptr(g), _ptr(kb), J, Z * J, _ptr(W), _ptr(None), _ptr(da), st)
        if ctx.needs_input_grad[1]:
            dkb = torch.empty_like(kb)
            _lib.launch(a.device, lib.enf_ode_conv_backward_basis, B, Z, J, C, _ptr(a), _ptr(g), _ptr(W), _ptr(dkb), st)
        if ctx.needs_input_grad[2]:       # d W = kb^T (g (x) a) over the pair axis, g (x) a formed in registers; d bias rides along
            buf = torch.empty(J * C + C, device=a.device, dtype=torch.float32)
            n = lib.enf_ode_conv_backward_weight_scratch_bytes(B, Z, J, C)
            sc = torch.empty(n // 4, device=a.device, dtype=torch.float32)
            _lib.launch(a.device, lib.enf_ode_conv_backward_weight, B, Z, J, C, _ptr(a), _ptr(kb), _ptr(g), _ptr(buf), _ptr(sc), n, st)
            dW = buf[:J * C].view(J, C)
            if ctx.has_bias and ctx.needs_input_grad[3]:
                db = buf[J * C:]
        elif ctx.has_bias and ctx.needs_input_grad[3]:
            db = g.sum((0, 1))
        return da, dkb, dW, db


def sep_gconv(a, kb, W, bias=None):
    if not a.is_cuda:
        raise RuntimeError("sep_gconv runs on the HIP kernels of libenf_hip.so only (no CPU path)")
    if a.dtype != torch.float32 or kb.dtype != torch.float32:
        raise TypeError("sep_gconv computes in fp32")
    return _SepGconv.apply(a, kb, W, bias)


class _PolyFeatures(torch.autograd.Function):
    """Kronecker powers of the pair invariants in one HIP launch each way (csrc/enf_ode.hip: enf_ode_poly_*)."""

    @staticmethod
    def forward(ctx, x, degree):
        lib = _lib.load()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        P, I = x2.shape
        F = lib.enf_ode_poly_num_features(I, degree)
        _lib.check(F if F < 0 else 0)
        out = torch.empty((P, F), device=x.device, dtype=torch.float32)
        _lib.launch(x.device, lib.enf_ode_poly_forward, P, I, degree, _ptr(x2), _ptr(out), _stream(x.device))
        ctx.save_for_backward(x2)
        ctx.degree, ctx.shape = degree, x.shape
        return out.view(*x.shape[:-1], F)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x2,) = ctx.saved_tensors
        P, I = x2.shape
        g2 = g.reshape(P, -1).contiguous()
        dx = torch.empty_like(x2)
        _lib.launch(x2.device, lib.enf_ode_poly_backward, P, I, ctx.degree, _ptr(x2), _ptr(g2), _ptr(dx), _stream(x2.device))
        return dx.view(ctx.shape), None


class PolynomialFeatures:
    """[x, x(x)x, ...]: degree + 1 Kronecker powers, flattened and concatenated (ponita_ode_g.py:15-26).  Device fp32
    tensors go through the fused HIP kernels; host tensors (tests of the host logic) through the definition below."""

    def __init__(self, degree):
        self.degree = degree

    def __call__(self, x):
        if x.is_cuda and x.dtype == torch.float32 and x.shape[-1] <= 8:
            return _PolyFeatures.apply(x, self.degree)
        out = [x]
        for _ in range(self.degree):
            out.append((out[-1][..., :, None] * x[..., None, :]).flatten(-2))
        return torch.cat(out, -1)

    def num_features(self, dim):
        return sum(dim ** k for k in range(1, self.degree + 2))


class _KernelBasis(torch.autograd.Function):
    """kb = gelu(gelu(poly(inv) W1 + b1) W3 + b3) over the B Z^2 pairs in ONE HIP kernel each way (csrc/enf_ode_basis.hip):
    neither the (B Z^2, F) feature tensor nor the hidden layer exist; the backward recomputes the forward per tile and
    returns d inv and the four weight gradients (kernel_basis MLP + PolynomialFeatures, ponita_ode_g.py:15-26, 128-131)."""

    @staticmethod
    def supported(inv, degree, W1, W3, backward):
        return (inv.is_cuda and inv.dtype == torch.float32 and
                bool(_lib.load().enf_ode_basis_supported(inv.shape[-1], degree, W1.shape[1], W3.shape[1], int(backward))))

    @staticmethod
    def _scratch(lib, P, I, H1, J, backward, dev):
        n = lib.enf_ode_basis_scratch_bytes(P, I, H1, J, backward)
        return torch.empty(n // 4, device=dev, dtype=torch.float32), n

    @staticmethod
    def forward(ctx, inv, degree, W1, b1, W3, b3):
        lib = _lib.load()
        x = inv.reshape(-1, inv.shape[-1]).contiguous()
        W1, b1, W3, b3 = W1.contiguous(), b1.contiguous(), W3.contiguous(), b3.contiguous()
        P, I = x.shape
        H1, J = W1.shape[1], W3.shape[1]
        kb = torch.empty((P, J), device=x.device, dtype=torch.float32)
        sc, n = _KernelBasis._scratch(lib, P, I, H1, J, 0, x.device)
        _lib.launch(x.device, lib.enf_ode_basis_forward, P, I, degree, H1, J, _ptr(x), _ptr(W1), _ptr(b1), _ptr(W3), _ptr(b3),
                    _ptr(kb), _ptr(sc), n, _stream(x.device))
        ctx.save_for_backward(x, W1, b1, W3, b3)
        ctx.degree, ctx.shape = degree, inv.shape
        return kb.view(*inv.shape[:-1], J)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, W1, b1, W3, b3 = ctx.saved_tensors
        P, I = x.shape
        H1, J = W1.shape[1], W3.shape[1]
        if not lib.enf_ode_basis_supported(I, ctx.degree, H1, J, 1):
            raise NotImplementedError(f"fused kernel-basis backward: hidden width {H1} is forward-only")
        g2 = g.reshape(P, J).contiguous()
        dx, dW1, db1, dW3, db3 = (torch.empty_like(t) for t in (x, W1, b1, W3, b3))
        sc, n = _KernelBasis._scratch(lib, P, I, H1, J, 1, x.device)
        _lib.launch(x.device, lib.enf_ode_basis_backward, P, I, ctx.degree, H1, J, _ptr(x), _ptr(W1), _ptr(b1), _ptr(W3),
                    _ptr(b3), _ptr(g2), _ptr(dx), _ptr(dW1), _ptr(db1), _ptr(dW3), _ptr(db3), _ptr(sc), n, _stream(x.device))
        return dx.view(ctx.shape), None, dW1, db1, dW3, db3


def kernel_basis(inv, degree, K1, K3, poly=None):
    """The kernel-basis MLP over the pair invariants: fused where the HIP kernels cover the shape (I <= 4, degree 3, hidden
    32..128 -- 256 for inference --, basis 32..128), else PolynomialFeatures + two pair-wise Dense layers."""
    W1, b1, W3, b3 = K1["kernel"], K1["bias"], K3["kernel"], K3["bias"]
    train = torch.is_grad_enabled() and any(t.requires_grad for t in (inv, W1, b1, W3, b3))
    if FUSED_BASIS and _KernelBasis.supported(inv, degree, W1, W3, train):
        return _KernelBasis.apply(inv, degree, W1, b1, W3, b3)
    poly = poly or PolynomialFeatures(degree)
    return _gelu(_PairDense.apply(_gelu(_PairDense.apply(poly(inv), W1, b1)), W3, b3))


FUSED_BASIS = os.environ.get("ENF_ODE_UNFUSED_BASIS", "0") != "1"      # diagnostic switch (tests, scripts/bench_ode.py): the unfused path


def _gelu(x):
    return Fnn.gelu(x, approximate="tanh")                 # flax nn.gelu default


def _dense(x, p):
    y = x @ p["kernel"]
    return y + p["bias"] if "bias" in p else y


class _LatentMLP(torch.autograd.Function):
    """ConvBlock after the convolution (ponita_ode_g.py:44-48): LayerNorm(eps 1e-6) -> Dense -> gelu -> Dense over the B Z latent
    rows.  Widths the HIP kernels cover (csrc/enf_ode_block.hip: hidden 32 / 64 / 128, widening factor 2): ONE launch forward,
    two backward.  Other widths: library GEMMs with the backward written out (4 launches forward, 10 backward instead of the
    ~25 of the op-by-op autograd graph) -- the evaluation is launch-bound."""
    EPS = 1e-6

    @staticmethod
    def forward(ctx, x, gamma, beta, W1, b1, W2, b2):
        H, M = W1.shape
        lib = _lib.load()
        ctx.fused = bool(FUSED_BLOCK and x.is_cuda and x.dtype == torch.float32 and lib.enf_ode_block_supported(H, M))
        if ctx.fused:
            x2 = x.reshape(-1, H).contiguous()
            gamma, beta, W1, b1, W2, b2 = (t.contiguous() for t in (gamma, beta, W1, b1, W2, b2))
            R = x2.shape[0]
            out, pre = torch.empty_like(x2), torch.empty((R, M), device=x.device, dtype=torch.float32)
            _lib.launch(x.device, lib.enf_ode_block_forward, R, H, M, _ptr(x2), _ptr(gamma), _ptr(beta), _ptr(W1), _ptr(b1),
                        _ptr(W2), _ptr(b2), _LatentMLP.EPS, _ptr(out), _ptr(pre), _stream(x.device))
            ctx.save_for_backward(x2, gamma, beta, W1, W2, pre)
            ctx.shape = x.shape
            return out.view(*x.shape[:-1], H)
        xn, mean, rstd = torch.native_layer_norm(x, (H,), gamma, beta, _LatentMLP.EPS)
        pre = torch.addmm(b1, xn.reshape(-1, H), W1)
        h = Fnn.gelu(pre, approximate="tanh")
        out = torch.addmm(b2, h, W2)
        ctx.save_for_backward(x, mean, rstd, gamma, beta, xn, pre, h, W1, W2)
        return out.view(*x.shape[:-1], W2.shape[1])

    @staticmethod
    def backward(ctx, g):
        if ctx.fused:
            lib = _lib.load()
            x2, gamma, beta, W1, W2, pre = ctx.saved_tensors
            (R, H), M = x2.shape, W1.shape[1]
            g2 = g.reshape(R, H).contiguous()
            dx = torch.empty_like(x2)
            dpar = torch.empty(2 * H * M + M + 3 * H, device=x2.device, dtype=torch.float32)
            n = lib.enf_ode_block_scratch_bytes(R, H, M)
            sc = torch.empty(n // 4, device=x2.device, dtype=torch.float32)
            _lib.launch(x2.device, lib.enf_ode_block_backward, R, H, M, _ptr(x2), _ptr(gamma), _ptr(beta), _ptr(W1), _ptr(W2),
                        _ptr(pre), _ptr(g2), _LatentMLP.EPS, _ptr(dx), _ptr(dpar), _ptr(sc), n, _stream(x2.device))
            dW1, dW2, db1, db2, dgamma, dbeta = torch.split(dpar, [H * M, M * H, M, H, H, H])
            return dx.view(ctx.shape), dgamma, dbeta, dW1.view(H, M), db1, dW2.view(M, H), db2
        x, mean, rstd, gamma, beta, xn, pre, h, W1, W2 = ctx.saved_tensors
        H = x.shape[-1]
        g2 = g.reshape(-1, g.shape[-1])
        dW2, db2 = h.t() @ g2, g2.sum(0)
        dpre = torch.ops.aten.gelu_backward(g2 @ W2.t(), pre, approximate="tanh")
        dW1, db1 = xn.reshape(-1, H).t() @ dpre, dpre.sum(0)
        dx, dgamma, dbeta = torch.ops.aten.native_layer_norm_backward((dpre @ W1.t()).view(x.shape), x, (H,), mean, rstd, gamma, beta,
                                                                      [True, True, True])
        return dx, dgamma, dbeta, dW1, db1, dW2, db2


FUSED_BLOCK = os.environ.get("ENF_ODE_UNFUSED_BLOCK", "0") != "1"      # diagnostic switch, like FUSED_BASIS


class _VecReadout(torch.autograd.Function):
    """out[b,r,:] = mean_s (inv[b,r,s,:] . Wi + aw[b,s]) * (cr u[b,r,:] + cs w[b,s,:]): the vector readout (ponita_ode_g.py:176-193)
    in one HIP launch each way (csrc/enf_ode.hip: enf_ode_vec_readout_*) instead of ~10 + ~25 element-wise / reduction launches."""

    @staticmethod
    def supported(inv, u):
        return FUSED_READOUT and inv.is_cuda and inv.dtype == torch.float32 and inv.shape[-1] <= 6 and u.shape[-1] in (2, 3)

    @staticmethod
    def forward(ctx, inv, aw, u, w, Wi, cr, cs):
        lib = _lib.load()
        inv, aw, u, w, Wi = (t.contiguous() for t in (inv, aw, u, w, Wi))
        B, Z, _, I = inv.shape
        D = u.shape[-1]
        out = torch.empty((B, Z, D), device=inv.device, dtype=torch.float32)
        _lib.launch(inv.device, lib.enf_ode_vec_readout_forward, B, Z, I, D, _ptr(inv), _ptr(aw), _ptr(u), _ptr(w), cr, cs, _ptr(Wi),
                    _ptr(out), _stream(inv.device))
        ctx.save_for_backward(inv, aw, u, w, Wi)
        ctx.c = (cr, cs)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        inv, aw, u, w, Wi = ctx.saved_tensors
        B, Z, _, I = inv.shape
        D = u.shape[-1]
        g = g.contiguous()
        dinv, daw, du, dw = torch.empty_like(inv), torch.empty_like(aw), torch.empty_like(u), torch.empty_like(w)
        part = torch.empty((B * ((Z + 63) // 64), I), device=inv.device, dtype=torch.float32)
        _lib.launch(inv.device, lib.enf_ode_vec_readout_backward, B, Z, I, D, _ptr(inv), _ptr(aw), _ptr(u), _ptr(w), ctx.c[0], ctx.c[1],
                    _ptr(Wi), _ptr(g), _ptr(dinv), _ptr(daw), _ptr(du), _ptr(dw), _ptr(part), _stream(inv.device))
        return dinv, daw, du, dw, part.sum(0), None, None


FUSED_READOUT = os.environ.get("ENF_ODE_UNFUSED_READOUT", "0") != "1"      # diagnostic switch, like FUSED_BASIS


def _trunc_normal(gen, shape, std, device):
    t = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(t, 0.0, 1.0, -2.0, 2.0, generator=gen)
    return (t * (std / 0.87962566103423978)).to(device)


def _lecun(gen, n_in, n_out, device, bias=True):
    p = {"kernel": _trunc_normal(gen, (n_in, n_out), math.sqrt(1.0 / n_in), device)}      # flax Dense default
    if bias:
        p["bias"] = torch.zeros(n_out, device=device)
    return p


def _flatten_tree(tree):
    """Leaves of a nested dict in sorted-key order, and the function that rebuilds the dict from such a list."""
    def leaves(t):
        out = []
        for k in sorted(t):
            out += leaves(t[k]) if isinstance(t[k], dict) else [t[k]]
        return out

    def build(lv):
        it = iter(lv)

        def rec(t):
            return {k: (rec(t[k]) if isinstance(t[k], dict) else next(it)) for k in sorted(t)}
        return rec(tree)
    return leaves(tree), build


class PonitaGen:
    def __init__(self, num_hidden, num_layers, scalar_num_out, vec_num_out, invariant, basis_dim, degree, widening_factor,
                 global_pool, kernel_size="global"):
        assert kernel_size == "global" or kernel_size > 0, "kernel_size must be 'global' or a positive number."   # :100
        self.num_hidden, self.num_layers, self.scalar_num_out, self.vec_num_out = num_hidden, num_layers, scalar_num_out, vec_num_out
        self.invariant, self.basis_dim, self.degree, self.widening_factor = invariant, basis_dim, degree, widening_factor
        self.global_pool, self.kernel_size = global_pool, kernel_size
        self.poly = PolynomialFeatures(degree)

    # ---- parameters (shapes ponita_ode_g.py:97-134)
    def init(self, key, latent_dim, device="cuda"):
        gen = torch.Generator().manual_seed(int(key))
        inv, H, J = self.invariant, self.num_hidden, self.basis_dim
        P = {"kernel_basis": {"layers_1": _lecun(gen, self.poly.num_features(inv.dim), H, device),
                              "layers_3": _lecun(gen, H, J, device)},
             "a_stem": _lecun(gen, latent_dim, H, device, bias=False)}
        lim = math.sqrt(2.0 / (J + H) * J)                                                # chang_xavier_uniform, :9-13
        for i in range(self.num_layers):
            P[f"interaction_layers_{i}"] = {
                "conv": {"kernel": {"kernel": ((torch.rand((J, H), generator=gen) * 2 - 1) * lim).to(device)},
                         "bias": torch.zeros(H, device=device)},
                "norm": {"scale": torch.ones(H, device=device), "bias": torch.zeros(H, device=device)},
                "linear_1": _lecun(gen, H, self.widening_factor * H, device),
                "linear_2": _lecun(gen, self.widening_factor * H, H, device)}
        ro = lambda n_in, n_out: {"kernel": _trunc_normal(gen, (n_in, n_out), math.sqrt(1e-6 / n_in), device)}   # :124
        P["readout_scalar"] = {"layers_0": ro(H, self.scalar_num_out)}
        if self.vec_num_out > 0:
            P["readout_vec_rel"] = ro(inv.dim + H, self.vec_num_out)
            if inv.num_z_ori_dims > 0:
                P["readout_vec_ori"] = ro(inv.dim + H, self.vec_num_out)
        return P

    def __call__(self, P, latent):
        p, a, _ = latent
        inv = self.invariant
        zp = inv.num_z_pos_dims
        if inv.num_z_ori_dims > 0:                                                        # :152-155
            p = torch.cat((p[..., :zp], torch.cos(p[..., zp:]), torch.sin(p[..., zp:])), -1)
        invariants = inv(p, p)                                                            # (B, Z, Z, I)
        K1, K3 = P["kernel_basis"]["layers_1"], P["kernel_basis"]["layers_3"]
        kb = kernel_basis(invariants, self.degree, K1, K3, self.poly)
        if self.kernel_size != "global":                                                  # :162-164
            kb = kb * torch.exp(-torch.linalg.norm(p[:, :, None, :] - p[:, None, :, :], dim=-1) / self.kernel_size)[..., None]
        a = _dense(a, P["a_stem"])
        for i in range(self.num_layers):                                                  # ConvBlock, :42-49
            L = P[f"interaction_layers_{i}"]
            x = sep_gconv(a, kb, L["conv"]["kernel"]["kernel"], L["conv"]["bias"])
            if x.is_cuda and x.dtype == torch.float32:
                a = _LatentMLP.apply(x, L["norm"]["scale"], L["norm"]["bias"], L["linear_1"]["kernel"], L["linear_1"]["bias"],
                                     L["linear_2"]["kernel"], L["linear_2"]["bias"])
            else:                                                                     # host tensors (tests of the host logic)
                x = Fnn.layer_norm(x, (x.shape[-1],), L["norm"]["scale"], L["norm"]["bias"], 1e-6)
                a = _dense(_gelu(_dense(x, L["linear_1"])), L["linear_2"])
        scalar_out = _dense(a, P["readout_scalar"]["layers_0"])
        vec_out = None
        if self.vec_num_out > 0:                                                          # :176-193
            # Dense([invariants | a_s]) = invariants @ W[:I] + (a @ W[I:]) of the sender, without the (B, Z, Z, I + H) concat
            I = invariants.shape[-1]

            def readout(Wk):
                # (B Z^2 x I)(I x V) with I <= 6, V = 1: as a library GEMM this picks a 16x16 tile and takes 350 us (and its
                # weight gradient the same again); a broadcast multiply + sum over I is a few us
                inv_part = (invariants[..., None] * Wk[:I]).sum(-2) if Wk.shape[1] <= 4 else invariants @ Wk[:I]
                return inv_part + (a @ Wk[I:])[:, None, :, :]
            fused = P["readout_vec_rel"]["kernel"].shape[1] == 1 and _VecReadout.supported(invariants, p[..., :zp])

            def fused_readout(Wk, u, w, cr, cs):
                return _VecReadout.apply(invariants, (a @ Wk[I:])[..., 0], u, w, Wk[:I, 0], cr, cs)
            if fused:
                vec_out = fused_readout(P["readout_vec_rel"]["kernel"], p[..., :zp], p[..., :zp], 1.0, -1.0)
            else:
                vec_out = (readout(P["readout_vec_rel"]["kernel"]) * (p[:, :, None, :zp] - p[:, None, :, :zp])).mean(-2)
            if inv.num_z_ori_dims > 0:
                if fused and p.shape[-1] - zp in (2, 3):
                    vec_out = vec_out + fused_readout(P["readout_vec_ori"]["kernel"], p[..., zp:], p[..., zp:], 0.0, 1.0)
                else:
                    vec_out = vec_out + (readout(P["readout_vec_ori"]["kernel"]) * p[:, None, :, zp:]).mean(-2)
        if self.global_pool:
            scalar_out = scalar_out.mean(1)
            vec_out = vec_out.mean(1) if vec_out is not None else None
        return scalar_out, vec_out


class PonitaODEGen:
    """``init(key, latents) -> params``, ``apply(params, latents) -> (dp/dt, da/dt, dwindow/dt)`` (flax calling style)."""

    def __init__(self, num_hidden, num_layers, scalar_num_out, vec_num_out, invariant, basis_dim, degree, widening_factor,
                 global_pool, kernel_size="global"):
        self.invariant = invariant
        self.scalar_num_out = scalar_num_out
        n_sc = scalar_num_out + 1 if invariant.num_z_ori_dims > 0 else scalar_num_out          # :212-217 (angle update)
        self.ponita = PonitaGen(num_hidden, num_layers, n_sc, vec_num_out, invariant, basis_dim, degree, widening_factor,
                                global_pool, kernel_size)

    def init(self, key, latents, device=None):
        p, a, _ = latents
        return {"params": {"ponita": self.ponita.init(key, a.shape[-1], device or a.device)}}

    def load_params(self, tree, device="cuda"):
        """A reference parameter tree (numpy leaves, flax names) -> device tensors."""
        conv = lambda t: {k: conv(v) for k, v in t.items()} if isinstance(t, dict) else \
            torch.as_tensor(t, dtype=torch.float32).to(device).contiguous()
        return conv(tree)

    def graphed(self, params, latents):
        """Inference-only derivative function ``f(latents) -> (dp, da, dwindow)`` replaying ONE captured hipGraph of
        ``apply`` (an evaluation is ~60 small launches and launch-bound: 0.51 -> 0.36 ms at the bench shape).  Shapes are
        those of the sample ``latents``; ``params`` are read in place (later in-place updates are seen, new tensors are
        not).  No autograd: use ``apply`` for training."""
        static_in = tuple(None if v is None else v.detach().clone() for v in latents)
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                                   # warm up (allocations, kernel loading) off-capture
                    self.apply(params, static_in)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.apply(params, static_in)

        def f(z):
            for dst, src in zip(static_in, z):
                if dst is not None:
                    dst.copy_(src)
            graph.replay()
            return tuple(None if v is None else v.clone() for v in static_out)
        f.graph = graph
        return f

    def graphed_train(self, params, latents, n):
        """``n`` training evaluations ``f_k(latents) -> (dp, da, dwindow)``, each replaying its own captured pair of hipGraphs
        (forward, backward; torch.cuda.make_graphed_callables, one memory pool): an Euler / RK4 roll-out calls every
        derivative evaluation once per step, and an eager evaluation is ~150 launches and host-bound (2.3-3 ms against 1.3 ms
        of kernels).  Bitwise equal to ``apply``.  The leaves of ``params`` are read IN PLACE by the graphs and must be
        leaf tensors that require grad (the trainer keeps persistent ones and copies the current values in); gradients
        reach them and the input latents through autograd as usual.  Shapes are those of the sample ``latents``."""
        leaves, build = _flatten_tree(params)
        p0, a0, w0 = latents

        def fn(p, a, *lv):
            dp, da, _ = self.apply(build(lv), (p, a, w0))
            return dp, da

        fresh = lambda t: t.detach().clone().requires_grad_(True)       # own input buffers per capture: a backward may read them
        samples = tuple((fresh(p0), fresh(a0), *leaves) for _ in range(n))
        graphed = torch.cuda.make_graphed_callables((fn,) * n, samples, allow_unused_input=True)
        graphed = graphed if isinstance(graphed, tuple) else (graphed,)

        def wrap(g):
            def f(z):
                p, a, w = z
                dp, da = g(p if p.requires_grad else p.detach().requires_grad_(True),
                           a if a.requires_grad else a.detach().requires_grad_(True), *leaves)
                return dp, da, (torch.zeros_like(w) if w is not None else None)
            return f
        return [wrap(g) for g in graphed]

    def apply(self, params, latents):
        p, a, window = latents
        scalar, vec = self.ponita(params["params"]["ponita"], (p, a - 1, window))            # a has mean 1 (:233)
        if self.invariant.num_z_ori_dims > 0:
            da, dp = scalar[..., :-1], torch.cat([vec, scalar[..., -1:]], -1)
        else:
            da, dp = scalar, vec
        return dp, da, (torch.zeros_like(window) if window is not None else None)
