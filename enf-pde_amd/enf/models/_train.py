"""Training path of the decoder: gradients w.r.t. the network weights.

The reference differentiates ``nef.apply`` w.r.t. ``params['nef']`` with ``jax.value_and_grad``
(experiments/fitting/trainers/pde_trainer.py:255, nonmaml_pde_trainer.py:304-339).  Here the
per-pair chain -- all but ~3 % of the arithmetic -- stays in the HIP kernels
(``enf_pair_forward`` / ``enf_backward_weights``, include/enf_hip.h); what is per-latent or per-query
(the weight folds, the latent prologue, the tail after the softmax-weighted sum) is expressed as
ordinary differentiable device ops around it so that autograd carries the gradient from the
"effective" per-pair parameters and the latent table back to the Flax-named tensors.

``enf_backward_weights`` is one library call: the backward pair kernel materialises each per-pair layer's input and delta
(bf16 in bf16 mode) in the call's scratch and ``enf_xtd_kernel`` turns them into every  dW = X^T delta  and bias sum.
"""
import ctypes
import math

import torch
import torch.nn.functional as Fnn

from ... import _lib
from . import _pad

LN_EPS = 1e-6          # flax.linen.LayerNorm default (NEF:56, ECA:19)
STORE_BUDGET_BYTES = 6 << 30   # bound on the materialised activations of one backward chunk


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def lt_layout(desc):
    lib = _lib.load()
    v = [ctypes.c_int(0) for _ in range(6)]
    _lib.check(lib.enf_lt_layout(ctypes.byref(desc), *[ctypes.byref(t) for t in v]))
    lay = dict(zip(("stride", "u", "v0", "pose", "wcoef", "c"), [t.value for t in v]))
    e = [ctypes.c_int(0) for _ in range(3)]
    _lib.check(lib.enf_lt_layout_ext(ctypes.byref(desc), *[ctypes.byref(t) for t in e]))
    lay.update(zip(("ext", "phq", "phv"), [t.value for t in e]))
    return lay


SPLIT_K = 4096      # rows per slice of the pair axis: a (D x P)(P x D) product has only D^2/tile^2 output tiles, so the
                    # library GEMM is run batched over slices of P (split-K) and the slices are summed in fp32


_NO_SPLIT = __import__("os").environ.get("ENF_TRAIN_SPLIT") == "0"      # A/B switch for the split-row reductions below


def _col_sum(Dl):
    """Column sums of Dl (P, D) in fp32.  For P >> D the generic reduction kernel runs at a quarter of the memory rate; a
    batched ones-vector product over 4096-row slices reads Dl at GEMM speed."""
    P, D = Dl.shape
    n = P // SPLIT_K
    if n < 4 or _NO_SPLIT:
        return Dl.sum(0, dtype=torch.float32)
    ones = Dl.new_ones((n, 1, SPLIT_K))
    Db = Dl[:n * SPLIT_K].view(n, SPLIT_K, D)
    if Dl.dtype == torch.float32:
        out = torch.bmm(ones, Db).sum((0, 1))
    else:
        try:
            out = torch.bmm(ones, Db, out_dtype=torch.float32).sum((0, 1))
        except (TypeError, RuntimeError, NotImplementedError):
            return Dl.sum(0, dtype=torch.float32)
    if n * SPLIT_K < P:
        out = out + Dl[n * SPLIT_K:].sum(0, dtype=torch.float32)
    return out


def _xt_dot2(X, G):
    """X^T @ G for X (P, K), G (P, N), fp32, split over the row axis when it is long (few output tiles otherwise)."""
    P = X.shape[0]
    n = P // SPLIT_K
    if n < 2:
        return X.t() @ G
    out = torch.bmm(X[:n * SPLIT_K].view(n, SPLIT_K, -1).transpose(1, 2), G[:n * SPLIT_K].view(n, SPLIT_K, -1)).sum(0)
    if n * SPLIT_K < P:
        out = out + X[n * SPLIT_K:].t() @ G[n * SPLIT_K:]
    return out


class _RowDense(torch.autograd.Function):
    """x @ W + b over many rows (the per-query tail: B N rows); weight / bias gradients as split-row products."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return torch.addmm(b, x.reshape(-1, x.shape[-1]), W).view(*x.shape[:-1], W.shape[1])

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        dx = (g2 @ W.t()).view(x.shape) if ctx.needs_input_grad[0] else None
        dW = _xt_dot2(x.reshape(-1, x.shape[-1]), g2) if ctx.needs_input_grad[1] else None
        db = _col_sum(g2) if ctx.needs_input_grad[2] else None
        return dx, dW, db


def _dense(x, W, b):
    return _RowDense.apply(x, W, b) if x.numel() // x.shape[-1] >= 2 * SPLIT_K and not _NO_SPLIT else Fnn.linear(x, W.t(), b)


class _PairFunction(torch.autograd.Function):
    """ybar = softmax-weighted sum of the per-pair value chain; HIP forward and backward."""

    @staticmethod
    def forward(ctx, x, lt, model, *eff):
        lib = _lib.load()
        B, N, _ = x.shape
        Z = lt.shape[0] // B
        H, D = model._Hp, model._Dp
        dev = lt.device
        ctx.masks = getattr(model, "_masks", None)       # relu masks of this call (model.relu_masks); its backward replays them
        desc = model._desc(B, N, Z, masks=ctx.masks) if ctx.masks is not None else model._desc(B, N, Z)
        xb, xstride = model._x_arg(x)
        lt_ = lt.detach().contiguous()
        st = _stream(dev)
        # the packed panels of the effective tensors: the outer step runs several passes on the SAME weights (new tensor
        # objects of unchanged storage), so the blob is cached on the identity + version of the leaf weights
        # (apply_train: model._pair_key) and the ~20 pack launches run once per weight update, not once per pass
        key, leaves = getattr(model, "_pair_key", None) or (None, None)
        hit = getattr(model, "_pair_blob", None)
        if key is not None and hit is not None and hit[0] == key:
            blob, effc = hit[1], hit[2]
        else:
            effc = [t.detach().to(torch.float32).contiguous() for t in eff]
            blob = torch.empty(int(lib.enf_packed_weight_bytes(ctypes.byref(desc))), device=dev, dtype=torch.uint8)
            arr = (ctypes.c_void_p * len(effc))(*[t.data_ptr() for t in effc])
            _lib.launch(dev, lib.enf_pack_pair, ctypes.byref(desc), arr, _ptr(blob), st)
            if key is not None:      # (the leaves are kept alive with the entry: a freed weight's address cannot come back
                model._pair_blob = (key, blob, effc, leaves)          #  under the same key with other values)
        ybar = torch.empty((B, N, H * D), device=dev, dtype=torch.float32)
        lse = torch.empty((B, N, H), device=dev, dtype=torch.float32)
        nscr = int(lib.enf_pair_scratch_bytes(ctypes.byref(desc)))
        scratch = torch.empty(nscr, device=dev, dtype=torch.uint8) if nscr else None
        _lib.launch(dev, lib.enf_pair_forward, ctypes.byref(desc), _ptr(xb), xstride, _ptr(lt_), _ptr(blob), _ptr(ybar),
                                        _ptr(lse), _ptr(scratch), nscr, st)
        ctx.model, ctx.xstride, ctx.dims = model, xstride, (B, N, Z)
        ctx.need_w = any(ctx.needs_input_grad[3:])
        ctx.x_shape = tuple(x.shape)
        ctx.save_for_backward(xb, lt_, blob, ybar, lse)
        ctx._keep = effc          # fp32 sources stay alive until the pack kernels have run
        return ybar

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dybar):
        lib = _lib.load()
        model = ctx.model
        xb, lt, blob, ybar, lse = ctx.saved_tensors
        B, N, Z = ctx.dims
        H, D = model._Hp, model._Dp
        HD = H * D
        dev = lt.device
        st = _stream(dev)
        dybar = dybar.contiguous().float()
        delta = (dybar * ybar).view(B, N, H, D).sum(-1).contiguous()
        dlt = torch.empty_like(lt)
        stride = lt.shape[1]
        # gradient w.r.t. the query coordinates (self-attention blocks: the queries are the latent poses), on request
        dxq = torch.zeros((B, N, ctx.x_shape[-1]), device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None

        def dx_out():
            return dxq          # (B, N, dx) also for a broadcast grid: autograd sums over the expand itself
        if not ctx.need_w:
            desc = model._desc(B, N, Z)
            _lib.launch(dev, lib.enf_pair_backward_ex, ctypes.byref(desc), _ptr(xb), ctx.xstride, _ptr(lt), _ptr(blob), _ptr(lse),
                                                _ptr(dybar), _ptr(delta), _ptr(dlt), None, _ptr(dxq), st)
            return (dx_out(), dlt, None) + (None,) * _lib.ENF_NUM_PAIR_TENSORS

        # weight gradients: ONE library call (include/enf_hip.h: enf_backward_weights) -- K3 writes the layer inputs / deltas
        # of a chunk of signals into the scratch, K4 (csrc/enf_xtd.hip) forms every X^T delta and bias sum from it
        desc = model._desc(B, N, Z, masks=ctx.masks) if ctx.masks is not None and ctx.masks[1] == "read" else model._desc(B, N, Z)
        group = ctx.masks[2] if ctx.masks is not None and ctx.masks[1] == "read" else 1
        cb = B
        while cb > group and int(lib.enf_backward_weights_scratch_bytes(ctypes.byref(desc), cb)) > STORE_BUDGET_BYTES:
            cb = max(group, (cb - 1) // group * group)
        scratch = torch.empty(int(lib.enf_backward_weights_scratch_bytes(ctypes.byref(desc), cb)), device=dev, dtype=torch.uint8)
        f32 = dict(device=dev, dtype=torch.float32)
        shapes = [(D, D), (D,), (D, D), (D,), (D, D), (D,), (D, 2 * HD), (2 * HD,), (D, D), (D,)]      # ENF_P_AQ1 .. ENF_P_BM
        grads = [torch.empty(sh, **f32) for sh in shapes]
        arr = (ctypes.c_void_p * _lib.ENF_NUM_PAIR_TENSORS)(*([g.data_ptr() for g in grads] + [None, None]))
        _lib.launch(dev, lib.enf_backward_weights, ctypes.byref(desc), _ptr(xb), ctx.xstride, _ptr(lt), _ptr(blob), _ptr(lse),
                    _ptr(dybar), _ptr(delta), _ptr(dlt), arr, _ptr(dxq), _ptr(scratch), scratch.numel(), st)
        assert dlt.shape[1] == stride
        # ENF_P_* order: AQ1,BQ1, AV1,BV1, AF,BF, AGB,BGB, AM,BM, COEFQ,COEFV (frozen: RFF:87-90)
        return (dx_out(), dlt, None, *grads, None, None)


def _ln(x, g, b, n_true):
    """LayerNorm over the last axis whose first-n_true-per-block features are real and the rest zero padding
    (the affine's zero-padded scale / bias clears the padded outputs): statistics divide by n_true."""
    if n_true == x.shape[-1]:
        # no padding: the framework's fused LayerNorm (one kernel forward, two backward, instead of ~13 / ~25 element-wise and
        # reduction launches: the outer step is bound by exactly those, DESIGN.md "Training path")
        return Fnn.layer_norm(x, (n_true,), g, b, LN_EPS)
    mu = x.sum(-1, keepdim=True) / n_true
    var = (x * x).sum(-1, keepdim=True) / n_true - mu * mu
    return (x - mu) * torch.rsqrt(var.clamp_min(0) + LN_EPS) * g + b


def _gelu(x):
    return Fnn.gelu(x, approximate="tanh")      # flax nn.gelu default


def latent_table(model, W, p, a, sigma, lay, stem=True, inv=None):
    """K1 as differentiable ops: stem, LayerNorm, k / v0, the logit fold (u, c), pose embedding
    and window coefficient, laid out as the pair kernels' latent table (enf_lt_layout).
    stem=False: ``a`` is already in the hidden space (layers after the stem, NEF:223-229)."""
    H, D = model._Hp, model._Dp
    B, Z = p.shape[:2]
    inv = inv if inv is not None else model.cross_attn_invariant
    lin = lambda t, w, b_: Fnn.linear(t, w.t(), b_)                       # x @ W + b as ONE kernel (W is (in, out))
    s = lin(a, W["stem_w"], W["stem_b"]) if stem else a                   # NEF:220
    an = _ln(s, W["lna_g"], W["lna_b"], model.num_hidden)                 # NEF:56 / ECB
    k = lin(an, W["k_w"], W["k_b"]).view(B, Z, H, D)                      # ECA:93
    v0 = lin(an, W["v_w"], W["v_b"])                                      # ECA:94
    scale = 1.0 / math.sqrt(model.num_hidden)                             # ECA:59 (the true width, not a padded one)
    qw = W["q_w"].view(D, H, D)
    mu = scale * torch.einsum("ij,jhd->hid", W["rq_w2"], qw)              # (H, D_in, D) : logits = h1 . (mu_h k_h)
    cvec = scale * (torch.einsum("j,jhd->hd", W["rq_b2"], qw) + W["q_b"].view(H, D))
    u = torch.einsum("hid,bzhd->bzhi", mu, k).reshape(B, Z, H * D)
    c = torch.einsum("hd,bzhd->bzh", cvec, k)
    name = inv.name
    zero = p.new_zeros(B, Z, 1)
    if name in ("ponita", "ponita_full"):
        pose = torch.cat([p[..., :2], torch.cos(p[..., 2:3]), torch.sin(p[..., 2:3])], -1)
    elif name in ("latitude_periodic", "polar_periodic", "ball", "ball_lat"):
        pose = torch.cat([p[..., :2], torch.sin(p[..., 1:2]), torch.cos(p[..., 1:2])], -1)
    else:
        pose = torch.cat([p[..., :3]] + [zero] * (4 - min(p.shape[-1], 3)), -1)
    if sigma is None:
        wc = p.new_ones(B, Z, 1)
    elif name in ("latitude_periodic", "polar_periodic", "ball", "ball_lat"):
        wc = 1.0 / (2.0 * sigma * sigma)
    else:
        wc = 1.0 / (sigma * sigma)
    parts = [(lay["u"], u), (lay["v0"], v0), (lay["pose"], pose), (lay["wcoef"], wc), (lay["c"], c)]
    if name in ("ball", "ball_lat"):
        # ext = [R (9) | latent-only invariants (2)]: the backward kernel returns d R and d(latent-only) in the same slots;
        # phases = coeff[latent-only rows]^T latent-only invariants, per RFFNet (csrc/enf_layout.h: enf_inv_rows)
        if name == "ball":
            al, be, ga = p[..., 0], p[..., 1], p[..., 2]
            ca, sa, cb, sb, cg, sg = al.cos(), al.sin(), be.cos(), be.sin(), ga.cos(), ga.sin()
            R = torch.stack([ca * cb, ca * sb * sg - sa * cg, ca * sb * cg + sa * sg,
                             sa * cb, sa * sb * sg + ca * cg, sa * sb * cg - ca * sg,
                             -sb, cb * sg, cb * cg], -1)                                # ball.py:76-84
            lat, rows = torch.stack([p[..., 3], torch.zeros_like(p[..., 3])], -1), [4]
        else:
            R = p.new_zeros(B, Z, 9)
            lat, rows = torch.stack([p[..., 1], p[..., 3]], -1), [1, 5]                 # ball_lat.py:66-88: th_p, r_p
        ld = lat.detach()[..., :len(rows)]      # the phases' gradient comes back through the ext slots, not the phase slots
        parts += [(lay["ext"], torch.cat([R, lat], -1)),
                  (lay["phq"], ld @ W["rq_coef"].detach()[rows]), (lay["phv"], ld @ W["rv_coef"].detach()[rows])]
    parts = sorted(parts, key=lambda t: t[0])
    out, pos = [], 0
    for off, t in parts:
        if off > pos:
            out.append(p.new_zeros(B, Z, off - pos))
        out.append(t)
        pos = off + t.shape[-1]
    if pos < lay["stride"]:
        out.append(p.new_zeros(B, Z, lay["stride"] - pos))
    return torch.cat(out, -1).reshape(B * Z, lay["stride"])


def effective_pair_params(model, W):
    """The twelve ENF_P_* tensors from the Flax-named weights (exact folds)."""
    return [W["rq_w1"], W["rq_b1"], W["rv_w1"], W["rv_b1"],
            W["rv_w2"] @ W["f1_w0"], W["rv_b2"] @ W["f1_w0"] + W["f1_b0"],
            W["f1_g"][:, None] * W["f1_w1"], W["f1_be"] @ W["f1_w1"] + W["f1_b1"],
            W["mx_w0"], W["mx_b0"], W["rq_coef"].detach(), W["rv_coef"].detach()]


def tail(model, W, ybar):
    """Everything after the softmax-weighted sum.  The mixer's LayerNorm affine and Dense_1 are
    applied after the sum (attention weights sum to one, so this equals the reference's order)."""
    H, D = model._Hp, model._Dp
    B, N, _ = ybar.shape
    y = ybar.view(B, N, H, D) * W["mx_g"] + W["mx_be"]
    y = Fnn.linear(y, W["mx_w1"].t(), W["mx_b1"]).reshape(B, N, H * D)    # ECA:16-21 (mixer Dense_1)
    y = _dense(y, W["ao_w"], W["ao_b"])                                   # ECA out_proj
    f = _dense(_ln(_gelu(_dense(y, W["ff_w0"], W["ff_b0"])), W["ff_g"], W["ff_be"], model.num_heads * model.num_hidden),
               W["ff_w1"], W["ff_b1"])
    o = _gelu(f)                                                          # NEF:227-233
    o = _gelu(_dense(o, W["o0_w"], W["o0_b"]))
    o = _gelu(_dense(o, W["o2_w"], W["o2_b"]))
    return _dense(o, W["o4_w"], W["o4_b"])


W_NAMES = ["stem_w", "stem_b", "lna_g", "lna_b",
           "rq_coef", "rq_w1", "rq_b1", "rq_w2", "rq_b2", "rv_coef", "rv_w1", "rv_b1", "rv_w2", "rv_b2",
           "q_w", "q_b", "k_w", "k_b", "v_w", "v_b",
           "f1_w0", "f1_b0", "f1_g", "f1_be", "f1_w1", "f1_b1", "mx_w0", "mx_b0", "mx_g", "mx_be", "mx_w1", "mx_b1",
           "ao_w", "ao_b", "ff_w0", "ff_b0", "ff_g", "ff_be", "ff_w1", "ff_b1",
           "o0_w", "o0_b", "o2_w", "o2_b", "o4_w", "o4_b"]      # ENF_W_* order
assert len(W_NAMES) == _lib.ENF_NUM_TENSORS



NATIVE_BACKWARD = __import__("os").environ.get("ENF_TRAIN_COMPOSED") != "1"      # apply_train through ONE library call per direction (enf_forward_stages / enf_backward_all); False = the
                            # older composition of differentiable device ops around the pair kernels (kept for the layered model and
                            # as a cross-check in the tests)
FROZEN = (W_NAMES.index("rq_coef"), W_NAMES.index("rv_coef"))      # RFF coefficients: no gradient (rff.py:87-90)


class _TrainAllFunction(torch.autograd.Function):
    """nef.apply differentiable w.r.t. all 46 weight tensors, the latents and the query coordinates: the forward is the library's
    enf_forward_stages on the blob enf_pack_weights builds (folds included), the backward ONE call of enf_backward_all -- tail,
    pair chain, prologue and the chain rule through the folds all inside the library (include/enf_hip.h)."""

    @staticmethod
    def forward(ctx, x, p, a, sigma, model, key, *tensors):
        lib = _lib.load()
        B, Z, N, dev = p.shape[0], p.shape[1], x.shape[1], p.device
        ctx.masks = getattr(model, "_masks", None)
        desc = model._desc(B, N, Z, masks=ctx.masks)
        xb, xstride = model._x_arg(x)
        p_, a_ = p.detach().contiguous(), a.detach().contiguous()
        s_ = sigma.detach().contiguous() if sigma is not None else None
        st = _stream(dev)
        ts = [t.detach().to(torch.float32).contiguous() for t in tensors]
        hit = getattr(model, "_train_blob", None)
        if key is not None and hit is not None and hit[0] == key:
            blob, ts = hit[1], hit[2]
        else:
            blob = torch.empty(int(lib.enf_packed_weight_bytes(ctypes.byref(desc))), device=dev, dtype=torch.uint8)
            arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
            _lib.launch(dev, lib.enf_pack_weights, ctypes.byref(desc), arr, _ptr(blob), st)
            if key is not None:
                model._train_blob = (key, blob, ts)
        HD = model._Hp * model._Dp
        out = torch.empty((B, N, model.num_out), device=dev, dtype=torch.float32)
        ybar = torch.empty((B, N, HD), device=dev, dtype=torch.float32)
        lse = torch.empty((B, N, model._Hp), device=dev, dtype=torch.float32)
        ws = model._workspace(desc, dev)
        _lib.launch(dev, lib.enf_forward_stages, ctypes.byref(desc), _ptr(xb), xstride, _ptr(p_), _ptr(a_), _ptr(s_), _ptr(blob),
                    _ptr(out), _ptr(ybar), _ptr(lse), _ptr(ws), ws.numel(), 15 | 16, st)          # prologue, fold, pair, tail + stash
        ctx.ws_tag = model._ws_touch(ws)
        ctx.model, ctx.xstride, ctx.dims, ctx.has_sigma = model, xstride, (B, N, Z), sigma is not None
        ctx.x_shape = tuple(x.shape)
        ctx.ts = ts                      # the fp32 sources the blob was packed from (the fold backward reads them)
        ctx.save_for_backward(xb, p_, a_, s_ if s_ is not None else p_.new_empty(0), blob, ybar, lse)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        lib = _lib.load()
        model = ctx.model
        xb, p_, a_, s_, blob, ybar, lse = ctx.saved_tensors
        sigma = s_ if ctx.has_sigma else None
        B, N, Z = ctx.dims
        dev = p_.device
        st = _stream(dev)
        read = ctx.masks is not None and ctx.masks[1] == "read"
        desc = model._desc(B, N, Z, masks=ctx.masks) if read else model._desc(B, N, Z)
        group = ctx.masks[2] if read else 1
        cb = B
        while cb > group and int(lib.enf_backward_all_scratch_bytes(ctypes.byref(desc), cb)) > STORE_BUDGET_BYTES:
            cb = max(group, (cb - 1) // group * group)
        nscr = int(lib.enf_backward_all_scratch_bytes(ctypes.byref(desc), cb))
        scratch = torch.empty(nscr, device=dev, dtype=torch.uint8)
        f32 = dict(device=dev, dtype=torch.float32)
        grads = [None if i in FROZEN else torch.empty(t.shape, **f32) for i, t in enumerate(ctx.ts)]
        arrT = (ctypes.c_void_p * len(ctx.ts))(*[t.data_ptr() for t in ctx.ts])
        arrG = (ctypes.c_void_p * len(grads))(*[None if g is None else g.data_ptr() for g in grads])
        dp, da = torch.empty_like(p_), torch.empty_like(a_)
        dsig = torch.empty((B, Z, 1), **f32)
        dxq = torch.zeros((B, N, ctx.x_shape[-1]), **f32) if ctx.needs_input_grad[0] else None
        ws = model._workspace(desc, dev)
        flags = 3 if model._ws_tag(ws) == ctx.ws_tag else 0       # latent table + tail stash still the forward's
        _lib.launch(dev, lib.enf_backward_all, ctypes.byref(desc), _ptr(xb), ctx.xstride, _ptr(p_), _ptr(a_), _ptr(sigma), arrT,
                    _ptr(blob), _ptr(ybar), _ptr(lse), _ptr(dout.contiguous().float()), _ptr(dp), _ptr(da), _ptr(dsig), arrG, _ptr(dxq),
                    _ptr(ws), ws.numel(), _ptr(scratch), nscr, flags, st)
        model._ws_touch(ws)
        return (dxq, dp, da, dsig if ctx.has_sigma else None, None, None, *grads)

class _SelfAttnView:
    """The pair kernels' descriptor for a latent self-attention block: queries = the latents' own positions (and, for
    Ponita2D, orientations), the self-attention invariant (NEF:223-226)."""

    def __init__(self, model):
        self._m = model
        self._Hp, self._Dp, self.precision = model._Hp, model._Dp, model.precision
        self.num_hidden, self.num_heads = model.num_hidden, model.num_heads
        self._x_arg = model._x_arg
        self.pair_variants, self.default_pair_variants = model.pair_variants, model.default_pair_variants

    def _desc(self, B, N, Z, masks=None):
        m, inv = self._m, self._m.self_attn_invariant
        return _lib.make_desc(B, N, Z, m._Hp, m._Dp, m.latent_dim, m.num_out, inv.num_x_pos_dims + inv.num_x_ori_dims,
                              inv.kernel_id, m.use_gaussian_window, _lib.PREC[m.precision],
                              d_true=m.num_hidden if m._Dp != m.num_hidden else 0, h_true=m.num_heads if m._Hp != m.num_heads else 0,
                              variants=tuple(_lib.VARIANT[v] for v in (m.pair_variants or m.default_pair_variants)), masks=masks)


def apply_layers(model, tensors, x, p, a, sigma):
    """nef.apply with num_layers > 0 (NEF:204-235): stem, the latent self-attention blocks -- each one the attention
    operator over (p, p) on the HIP pair kernels between differentiable per-latent ops --, then the cross-attention
    block on the hidden latents and the output MLP.  Differentiable w.r.t. the latents (poses included: the pair backward
    also returns the query-side gradient, the queries of a self-attention block being the poses) and every weight.
    A narrow model runs zero-padded in the kernels' width like the layer-free one (_pad.py)."""
    n0 = _lib.ENF_NUM_TENSORS
    Dt, Ht, D, H = model.num_hidden, model.num_heads, model._Dp, model._Hp
    blocks = [tensors[n0 + 38 * i:n0 + 38 * (i + 1)] for i in range(model.num_layers)]
    head = tensors[:n0]
    if D != Dt or H != Ht:
        head = _pad.pad_tensors(head, Dt, D, Ht, H)
        blocks = [_pad.pad_block_tensors(b, Dt, D, Ht, H) for b in blocks]
    W = dict(zip(W_NAMES, head))
    blk = W_NAMES[2:40]                                                    # the 38 tensors of one attention block
    HD = H * D
    B, Z = p.shape[:2]
    sa = model.self_attn_invariant
    s = a @ W["stem_w"] + W["stem_b"]                                     # NEF:220
    view = _SelfAttnView(model)
    xq = p[..., :sa.num_x_pos_dims + sa.num_x_ori_dims]                   # queries of a self-attention block: x = p (differentiable;
                                                                          # Ponita2D: position and the raw angle, embedded in-kernel)
    lay = lt_layout(view._desc(B, Z, Z))
    for i in range(model.num_layers):
        Wi = dict(zip(blk, blocks[i]))
        lt = latent_table(view, Wi, p, s, sigma, lay, stem=False, inv=sa)
        ybar = _PairFunction.apply(xq, lt, view, *effective_pair_params(view, Wi))            # (B, Z, HD)
        y = ybar.view(B, Z, H, D) * Wi["mx_g"] + Wi["mx_be"]
        y = (y @ Wi["mx_w1"] + Wi["mx_b1"]).reshape(B, Z, HD)                                  # ECA:16-21 (mixer Dense_1)
        a_attn = y @ Wi["ao_w"] + Wi["ao_b"]                                                   # ECA:150 (project_heads: HD -> D)
        r = s + a_attn                                                                         # NEF:62-64 (residual)
        f = _ln(_gelu(r @ Wi["ff_w0"] + Wi["ff_b0"]), Wi["ff_g"], Wi["ff_be"], Dt) @ Wi["ff_w1"] + Wi["ff_b1"]
        s = _gelu(s + f)                                                                       # NEF:225-226
    desc = model._desc(B, x.shape[1], Z)
    _lib.check(_lib.load().enf_check_desc(ctypes.byref(desc)))
    lt = latent_table(model, W, p, s, sigma, lt_layout(desc), stem=False)
    ybar = _PairFunction.apply(x, lt, model, *effective_pair_params(model, W))
    return tail(model, W, ybar)


def apply_train(model, tensors, x, p, a, sigma):
    """nef.apply differentiable w.r.t. every weight tensor and the latents."""
    if NATIVE_BACKWARD:
        key = (model.precision, str(p.device), tuple((t.data_ptr(), t._version) for t in tensors))
        if model._Dp != model.num_hidden or model._Hp != model.num_heads:  # run in the kernels' shape (differentiable zero padding)
            tensors = _pad.pad_tensors(tensors, model.num_hidden, model._Dp, model.num_heads, model._Hp)
        _lib.check(_lib.load().enf_check_desc(ctypes.byref(model._desc(p.shape[0], x.shape[1], p.shape[1]))))
        return _TrainAllFunction.apply(x, p, a, sigma, model, key, *tensors)
    if model._Dp != model.num_hidden or model._Hp != model.num_heads:      # run in the kernels' shape (differentiable zero padding)
        tensors = _pad.pad_tensors(tensors, model.num_hidden, model._Dp, model.num_heads, model._Hp)
    W = dict(zip(W_NAMES, tensors))
    desc = model._desc(p.shape[0], x.shape[1], p.shape[1])
    _lib.check(_lib.load().enf_check_desc(ctypes.byref(desc)))
    lay = lt_layout(desc)
    lt = latent_table(model, W, p, a, sigma, lay)
    eff = effective_pair_params(model, W)
    model._pair_key = ((model.precision, str(p.device), tuple((t.data_ptr(), t._version) for t in tensors)), list(tensors))
    try:
        ybar = _PairFunction.apply(x, lt, model, *eff)
    finally:
        model._pair_key = None
    return tail(model, W, ybar)
