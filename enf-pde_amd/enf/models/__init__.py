"""EquivariantCrossAttentionNeF -- host-side mirror of enf/models/equivariant_cross_attention_nef.py:70-235.

Same constructor keywords (experiments/fitting/__init__.py:25-38), same ``init`` / ``apply``
call shapes as the Flax module, same parameter tree (names follow Flax's naming rules, SURVEY.md
8a), same latent conventions.  ``apply`` runs the fused HIP path through the C-ABI
(include/enf_hip.h); gradients w.r.t. the latents (p, a, gaussian_window) come from the
hand-written HIP backward; when the weights require grad, ``apply`` takes the training path of
``_train.py`` (same HIP pair kernels, weight gradients as well).  There is no eager / CPU path.
"""
import ctypes
import math

import torch

from .. import steerable_attention  # noqa: F401  (package layout parity)
from ..steerable_attention.invariant import BaseInvariant
from ... import _lib
from . import _pad

__all__ = ["EquivariantCrossAttentionNeF", "TENSOR_PATHS", "tensor_paths"]

_BLK = "cross_attention_blocks_0"
# ENF_W_* order of include/enf_hip.h -> path in the Flax parameter tree
TENSOR_PATHS = [
    ("latent_stem", "kernel"), ("latent_stem", "bias"),
    (_BLK, "layer_norm_attn", "scale"), (_BLK, "layer_norm_attn", "bias"),
    (_BLK, "attn", "invariant_embedding_query", "encoding", "coefficients"),
    (_BLK, "attn", "invariant_embedding_query", "layers_0", "linear", "kernel"),
    (_BLK, "attn", "invariant_embedding_query", "layers_0", "linear", "bias"),
    (_BLK, "attn", "invariant_embedding_query", "linear_final", "kernel"),
    (_BLK, "attn", "invariant_embedding_query", "linear_final", "bias"),
    (_BLK, "attn", "invariant_embedding_value", "encoding", "coefficients"),
    (_BLK, "attn", "invariant_embedding_value", "layers_0", "linear", "kernel"),
    (_BLK, "attn", "invariant_embedding_value", "layers_0", "linear", "bias"),
    (_BLK, "attn", "invariant_embedding_value", "linear_final", "kernel"),
    (_BLK, "attn", "invariant_embedding_value", "linear_final", "bias"),
    (_BLK, "attn", "inv_emb_to_q", "kernel"), (_BLK, "attn", "inv_emb_to_q", "bias"),
    (_BLK, "attn", "a_to_k", "kernel"), (_BLK, "attn", "a_to_k", "bias"),
    (_BLK, "attn", "a_to_v", "kernel"), (_BLK, "attn", "a_to_v", "bias"),
    (_BLK, "attn", "inv_emb_to_v", "Dense_0", "kernel"), (_BLK, "attn", "inv_emb_to_v", "Dense_0", "bias"),
    (_BLK, "attn", "inv_emb_to_v", "LayerNorm_0", "scale"), (_BLK, "attn", "inv_emb_to_v", "LayerNorm_0", "bias"),
    (_BLK, "attn", "inv_emb_to_v", "Dense_1", "kernel"), (_BLK, "attn", "inv_emb_to_v", "Dense_1", "bias"),
    (_BLK, "attn", "inv_emb_cond_mixer", "Dense_0", "kernel"), (_BLK, "attn", "inv_emb_cond_mixer", "Dense_0", "bias"),
    (_BLK, "attn", "inv_emb_cond_mixer", "LayerNorm_0", "scale"), (_BLK, "attn", "inv_emb_cond_mixer", "LayerNorm_0", "bias"),
    (_BLK, "attn", "inv_emb_cond_mixer", "Dense_1", "kernel"), (_BLK, "attn", "inv_emb_cond_mixer", "Dense_1", "bias"),
    (_BLK, "attn", "out_proj", "kernel"), (_BLK, "attn", "out_proj", "bias"),
    (_BLK, "pointwise_ffn", "Dense_0", "kernel"), (_BLK, "pointwise_ffn", "Dense_0", "bias"),
    (_BLK, "pointwise_ffn", "LayerNorm_0", "scale"), (_BLK, "pointwise_ffn", "LayerNorm_0", "bias"),
    (_BLK, "pointwise_ffn", "Dense_1", "kernel"), (_BLK, "pointwise_ffn", "Dense_1", "bias"),
    ("out_proj", "layers_0", "kernel"), ("out_proj", "layers_0", "bias"),
    ("out_proj", "layers_2", "kernel"), ("out_proj", "layers_2", "bias"),
    ("out_proj", "layers_4", "kernel"), ("out_proj", "layers_4", "bias"),
]
assert len(TENSOR_PATHS) == _lib.ENF_NUM_TENSORS
BLOCK_PATHS = [t[1:] for t in TENSOR_PATHS if t[0] == _BLK]       # the 38 tensors of one attention block


def tensor_paths(num_layers=0):
    """TENSOR_PATHS followed by the tensors of the latent self-attention blocks (NEF:137-167), layer by layer."""
    return TENSOR_PATHS + [(f"self_attention_blocks_{i}",) + t for i in range(num_layers) for t in BLOCK_PATHS]


def _get(tree, path):
    for k in path:
        tree = tree[k]
    return tree


def _set(tree, path, value):
    for k in path[:-1]:
        tree = tree.setdefault(k, {})
    tree[path[-1]] = value


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


FUSED_FIT_STEP = __import__("os").environ.get("ENF_FIT_STEP") != "0"      # mse_value_and_latent_grads through enf_fit_step (one call)


class _EnfFunction(torch.autograd.Function):
    """nef.apply with the HIP forward (enf_forward) and backward-to-latents (enf_backward_latents)."""

    @staticmethod
    def forward(ctx, x, p, a, sigma, model, packed):
        lib = _lib.load()
        B, Z = p.shape[0], p.shape[1]
        N = x.shape[1]
        desc = model._desc(B, N, Z, masks=model._masks)
        xb, xstride = model._x_arg(x)
        p_, a_ = p.contiguous(), a.contiguous()
        s_ = sigma.contiguous() if sigma is not None else None
        dev = p.device
        out = torch.empty((B, N, model.num_out), device=dev, dtype=torch.float32)
        HD = model._Hp * model._Dp
        # no input needs a gradient (a decode): nothing will read this call's ybar / lse -- they stay in the workspace, and the pair
        # kernel may hand ybar to the tail as bf16 (ENF_STAGE_YBAR_HALF, include/enf_hip.h: same `out`, half the bytes)
        no_grad = not any(ctx.needs_input_grad[:4])
        ybar = None if no_grad else torch.empty((B, N, HD), device=dev, dtype=torch.float32)
        lse = None if no_grad else torch.empty((B, N, model._Hp), device=dev, dtype=torch.float32)
        ws = model._workspace(desc, dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        # a backward follows when an input needs a gradient: stash the tail's pre-activations for it (ENF_STAGE_TAIL_SAVE)
        ctx.tail_saved = any(ctx.needs_input_grad[1:4])
        # the latent table depends on (p, a, sigma, weights) only and sits at the head of the workspace whatever N is: a
        # forward on the SAME latent tensors and weights as the last call on this workspace (the decode that follows a fit's
        # final-loss forward, pde_trainer.py:225-235 then :393-402) skips the prologue kernel
        lt_key = (ws.data_ptr(), packed.data_ptr(), B, Z) + tuple((t.data_ptr(), t._version) for t in (p_, a_) + ((s_,) if s_ is not None else ()))
        held = model._lt_held.get(ws.data_ptr())
        reuse_lt = held is not None and held[0] == lt_key and held[1] == model._ws_tags.get(ws.data_ptr())
        stages = (14 if reuse_lt else 15) | (16 if ctx.tail_saved else 0) | (64 if no_grad else 0)
        _lib.launch(dev, lib.enf_forward_stages, ctypes.byref(desc), _ptr(xb), xstride, _ptr(p_), _ptr(a_), _ptr(s_), _ptr(packed),
                                          _ptr(out), _ptr(ybar), _ptr(lse), _ptr(ws), ws.numel(), stages, st)
        ctx.ws_tag = model._ws_touch(ws)      # backward may reuse the latent table if nothing else used the workspace
        # (the tensors are held with the entry: their addresses cannot come back under the same key with other values)
        model._lt_held[ws.data_ptr()] = (lt_key, ctx.ws_tag[1], (p_, a_, s_, packed))
        ctx.model = model
        ctx.has_sigma = sigma is not None
        ctx.xstride = xstride
        if not no_grad:
            ctx.save_for_backward(xb, p_, a_, s_ if s_ is not None else p_.new_empty(0), packed, ybar, lse)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        lib = _lib.load()
        model = ctx.model
        xb, p_, a_, s_, packed, ybar, lse = ctx.saved_tensors
        sigma = s_ if ctx.has_sigma else None
        B, Z = p_.shape[0], p_.shape[1]
        N = ybar.shape[1]
        desc = model._desc(B, N, Z)
        dev = p_.device
        dout = dout.contiguous().float()
        dp = torch.empty_like(p_)
        da = torch.empty_like(a_)
        dsig = torch.empty((B, Z, 1), device=dev, dtype=torch.float32)
        ws = model._workspace(desc, dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        reuse = 1 if model._ws_tag(ws) == ctx.ws_tag else 0          # ENF_BWD_REUSE_PROLOGUE
        if reuse and ctx.tail_saved:
            reuse |= 2                                               # ENF_BWD_REUSE_TAIL
        _lib.launch(dev, lib.enf_backward_latents_ex, ctypes.byref(desc), _ptr(xb), ctx.xstride, _ptr(p_), _ptr(a_), _ptr(sigma),
                                               _ptr(packed), _ptr(ybar), _ptr(lse), _ptr(dout), _ptr(dp), _ptr(da),
                                               _ptr(dsig), _ptr(ws), ws.numel(), reuse, st)
        model._ws_touch(ws)
        return None, dp, da, (dsig if ctx.has_sigma else None), None, None


class EquivariantCrossAttentionNeF:
    """Equivariant cross-attention neural field (NEF:70-235) on the fused gfx950 path.

    Args mirror the Flax module's fields (NEF:85-96); ``precision`` ("bf16" | "f32") selects
    the MFMA arithmetic of the per-pair contractions (ENF_PREC_*).
    """

    default_pair_variants = ("auto", "auto")

    def __init__(self, num_hidden, num_heads, num_layers, num_out, latent_dim, cross_attn_invariant,
                 self_attn_invariant=None, embedding_type="rff", embedding_freq_multiplier=(0.05, 0.1),
                 condition_value_transform=True, use_gaussian_window=True, precision="bf16"):
        if not isinstance(cross_attn_invariant, BaseInvariant):
            raise TypeError("cross_attn_invariant must come from enf.steerable_attention.invariant.get_ca_invariant")
        if embedding_type != "rff":
            if embedding_type in ("ffn", "polynomial"):
                raise NotImplementedError(f"embedding type '{embedding_type}' is outside the accelerated path "
                                          "(no shipped config selects it; SURVEY.md 2, row 2)")
            raise ValueError(f"Unknown embedding type: {embedding_type}.")          # EMB:33
        if not condition_value_transform:
            raise NotImplementedError("condition_value_transform=False is not on the accelerated path")
        assert not num_hidden % 2, "For the Fourier Features hidden_dim should be even to calculate them correctly."  # RFF:75-77
        if precision not in _lib.PREC:
            raise ValueError(f"unknown precision {precision!r}")
        self.num_hidden, self.num_heads, self.num_layers = int(num_hidden), int(num_heads), int(num_layers)
        self._Dp = _pad.padded_width(self.num_hidden)      # width of the kernels that run it (zero-padded if wider)
        self._Hp = _pad.padded_heads(self.num_heads)       # heads of the kernels that run it (3 -> 4, one zero head)
        if self._Hp == 4 and self._Dp != 64:
            raise NotImplementedError("3 or 4 heads are built for num_hidden <= 64 only")
        self.num_out, self.latent_dim = int(num_out), int(latent_dim)
        self.cross_attn_invariant = cross_attn_invariant
        self.self_attn_invariant = self_attn_invariant if self_attn_invariant is not None else cross_attn_invariant
        if self.num_layers > 0:
            # latent self-attention (NEF:223-226) runs the same pair kernels with the latents' own positions as queries
            # (enf/models/_train.py: apply_layers); dormant in every shipped config, so only the plain shapes are served
            if self.self_attn_invariant.name not in _lib.INVARIANT_IDS:
                raise NotImplementedError(f"self-attention with the '{self.self_attn_invariant.name}' invariant is not built")
        self.embedding_type = embedding_type
        self.embedding_freq_multiplier = tuple(embedding_freq_multiplier)
        self.condition_value_transform = condition_value_transform
        self.use_gaussian_window = bool(use_gaussian_window)
        self.precision = precision
        self._pack_cache = {}
        self._ws_cache = {}
        self._ws_gen = 0
        self._ws_tags = {}
        self._lt_held = {}                      # workspace -> (key of the latent table it holds, its use tag, the tensors)
        self.pair_variants = None               # (forward, backward) pair-kernel variant of this model's calls (_lib.VARIANT
                                                # keys); None = the class default below (tests flip it to cover both kernels)
        self._masks = None                      # (buffer, "write" | "read", signals) inside relu_masks(), else None

    def with_precision(self, precision):
        """The same model (fields, invariants) running its per-pair contractions in another arithmetic ("f32" | "bf16"), with
        caches of its own; parameters are plain tensors, so both models take the same parameter tree."""
        import copy
        if precision not in _lib.PREC:
            raise ValueError(f"unknown precision {precision!r}")
        m = copy.copy(self)
        m.precision = precision
        m._pack_cache, m._ws_cache, m._ws_gen, m._ws_tags, m._masks, m._lt_held = {}, {}, 0, {}, None, {}
        return m

    # ------------------------------------------------------------------ descriptors / buffers
    def _desc(self, B, N, Z, masks=None):
        """The call descriptor; ``masks`` = a (buffer, mode, signals) triple for calls whose pair kernels take relu masks."""
        inv = self.cross_attn_invariant
        return _lib.make_desc(B, N, Z, self._Hp, self._Dp, self.latent_dim, self.num_out,
                              inv.num_x_pos_dims, inv.kernel_id, self.use_gaussian_window, _lib.PREC[self.precision],
                              d_true=self.num_hidden if self._Dp != self.num_hidden else 0,
                              h_true=self.num_heads if self._Hp != self.num_heads else 0,
                              variants=tuple(_lib.VARIANT[v] for v in (self.pair_variants or self.default_pair_variants)),
                              masks=masks)

    def _workspace(self, desc, device):
        # one cached scratch buffer per (shape, stream); the autograd graph never keeps it alive
        lib = _lib.load()
        nbytes = lib.enf_workspace_bytes(ctypes.byref(desc))
        if nbytes == 0:
            _lib.check(lib.enf_check_desc(ctypes.byref(desc)))
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws_cache.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(int(nbytes), device=device, dtype=torch.uint8)
            self._ws_cache[key] = ws
        return ws

    def _ws_touch(self, ws):
        """Mark a use of workspace `ws`; returns the tag identifying this use."""
        self._ws_gen += 1
        self._ws_tags[ws.data_ptr()] = self._ws_gen
        return (ws.data_ptr(), self._ws_gen)

    def _ws_tag(self, ws):
        return (ws.data_ptr(), self._ws_tags.get(ws.data_ptr()))

    @staticmethod
    def _x_arg(x):
        """(tensor to pass, batch stride in elements); a stride-0 batch (broadcast grid) is legal (TR:197,393)."""
        if x.dim() != 3:
            raise ValueError("x must be (batch, num_coords, coord_dim)")
        if x.stride(0) == 0:
            return x[0].contiguous(), 0
        xc = x.contiguous()
        return xc, xc.shape[1] * xc.shape[2]

    # ------------------------------------------------------------------ parameters
    def init(self, key, x=None, p=None, a=None, gaussian_window_size=None, device=None):
        """Random parameters with the reference's initialisers.  ``key`` is an int
        seed or a torch.Generator (JAX PRNG keys cannot be reproduced).  The sample inputs are
        accepted for call-shape parity with ``nef.init(key, x, p, a, window)`` (TR:101) and only
        supply the device."""
        if device is None:
            device = p.device if p is not None else torch.device("cuda")
        g = key if isinstance(key, torch.Generator) else torch.Generator().manual_seed(int(key))
        D, H, C, O = self.num_hidden, self.num_heads, self.latent_dim, self.num_out
        I = self.cross_attn_invariant.dim
        HD = H * D

        def normal(shape, std):
            return torch.randn(shape, generator=g) * std

        def lecun(n_in, n_out):     # flax Dense default: truncated normal, std sqrt(1/fan_in) / .8796
            t = torch.empty(n_in, n_out)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=g)
            return {"kernel": t * (math.sqrt(1.0 / n_in) / 0.87962566103423978), "bias": torch.zeros(n_out)}

        def ln(n):
            return {"scale": torch.ones(n), "bias": torch.zeros(n)}

        def ffn(n_in, n_hid, n_out):
            return {"Dense_0": lecun(n_in, n_hid), "LayerNorm_0": ln(n_hid), "Dense_1": lecun(n_hid, n_out)}

        def rff(std):
            lim = math.sqrt(3 * 2.0 / D)
            return {"encoding": {"coefficients": normal((I, D // 2), std)},                                   # RFF:83
                    "layers_0": {"linear": {"kernel": normal((D, D), math.sqrt(2.0 / D)), "bias": normal((D,), 1e-6)}},  # RFF:55-60
                    "linear_final": {"kernel": (torch.rand((D, D), generator=g) * 2 - 1) * lim, "bias": normal((D,), 1e-6)}}  # RFF:35-40

        fq, fv = self.embedding_freq_multiplier
        attn = {"invariant_embedding_query": rff(fq), "invariant_embedding_value": rff(fv),
                "inv_emb_to_q": lecun(D, HD), "a_to_k": lecun(D, HD), "a_to_v": lecun(D, HD),
                "inv_emb_to_v": ffn(D, D, 2 * HD), "inv_emb_cond_mixer": ffn(D, D, D), "out_proj": lecun(HD, HD)}
        P = {"latent_stem": lecun(C, D),
             _BLK: {"layer_norm_attn": ln(D), "attn": attn, "pointwise_ffn": ffn(HD, HD, HD)},
             "out_proj": {"layers_0": lecun(HD, D), "layers_2": lecun(D, D), "layers_4": lecun(D, O)}}
        Is = self.self_attn_invariant.dim
        for i in range(self.num_layers):                                 # NEF:137-167 (project_heads=True: widths D)
            def rff_s(std):
                r = rff(std)
                r["encoding"]["coefficients"] = normal((Is, D // 2), std)
                return r
            P[f"self_attention_blocks_{i}"] = {
                "layer_norm_attn": ln(D),
                "attn": {"invariant_embedding_query": rff_s(fq), "invariant_embedding_value": rff_s(fv),
                         "inv_emb_to_q": lecun(D, HD), "a_to_k": lecun(D, HD), "a_to_v": lecun(D, HD),
                         "inv_emb_to_v": ffn(D, D, 2 * HD), "inv_emb_cond_mixer": ffn(D, D, D), "out_proj": lecun(HD, D)},
                "pointwise_ffn": ffn(D, D, D)}

        def to_dev(t):
            return {k: to_dev(v) for k, v in t.items()} if isinstance(t, dict) else t.to(device=device, dtype=torch.float32)
        return {"params": to_dev(P)}

    def param_tensors(self, params):
        """The ENF_NUM_TENSORS weight tensors in C-ABI order (then, for num_layers > 0, 38 per self-attention block)."""
        P = params["params"] if "params" in params else params
        return [_get(P, path) for path in tensor_paths(self.num_layers)]

    def invalidate_caches(self):
        """Forget the packed-weight blob, the packed pair panels of the training path and the latent tables held in workspaces.

        CONTRACT of the caches: reuse is decided from ``(tensor.data_ptr(), tensor._version)`` of the weight / latent tensors,
        so an update is seen when it goes through torch (in-place ops bump ``_version``; the optimisers and
        ``meta_sgd_update`` of this package return fresh tensors).  Writes torch cannot see -- through ``.data``, through a
        detached alias made before the call, by a raw-pointer kernel, ``hipMemcpy`` or a collective on an alias -- leave
        ``_version`` unchanged: call this method after such a write (``load_params`` does, for the tree it returns is new)."""
        self._pack_cache.clear()
        self._lt_held.clear()
        self._pair_key = None
        self._pair_blob = None
        self._train_blob = None

    def load_params(self, tree, device="cuda"):
        """Build a parameter tree from nested numpy / torch arrays (e.g. an exported Flax tree)."""
        self.invalidate_caches()
        P = tree["params"] if "params" in tree else tree
        out = {}
        for path in tensor_paths(self.num_layers):
            _set(out, path, torch.as_tensor(_get(P, path)).to(device=device, dtype=torch.float32).contiguous())
        return {"params": out}

    def pack(self, params):
        """Packed weight blob (device uint8 tensor) for ``params``; cached until a tensor changes."""
        lib = _lib.load()
        ts = self.param_tensors(params)[:_lib.ENF_NUM_TENSORS]
        dev = ts[0].device
        if dev.type != "cuda":
            raise _lib.EnfError("parameters must live on the GPU: the decoder has no CPU path")
        key = (self.precision, tuple((t.data_ptr(), t._version) for t in ts))
        hit = self._pack_cache.get("k")
        if hit is not None and hit[0] == key:
            return hit[1]
        desc = self._desc(1, 1, 1)
        _lib.check(lib.enf_check_desc(ctypes.byref(desc)))
        ts = [t.detach().to(torch.float32).contiguous() for t in ts]
        self._check_shapes(ts)
        if self._Dp != self.num_hidden or self._Hp != self.num_heads:
            ts = [t.contiguous() for t in _pad.pad_tensors(ts, self.num_hidden, self._Dp, self.num_heads, self._Hp)]
        nbytes = lib.enf_packed_weight_bytes(ctypes.byref(desc))
        blob = torch.empty(int(nbytes), device=dev, dtype=torch.uint8)
        arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.launch(dev, lib.enf_pack_weights, ctypes.byref(desc), arr, _ptr(blob), st)
        self._pack_cache["k"] = (key, blob, ts)   # keep the fp32 sources alive until the pack kernels ran
        return blob

    def _check_shapes(self, ts):
        for t, shp, path in zip(ts, self._expected_shapes(), TENSOR_PATHS):
            if tuple(t.shape) != shp:
                raise ValueError(f"parameter {'/'.join(path)} has shape {tuple(t.shape)}, expected {shp}")

    def _expected_shapes(self):
        D, H, C, O, I = self.num_hidden, self.num_heads, self.latent_dim, self.num_out, self.cross_attn_invariant.dim
        HD = H * D
        rff = [(I, D // 2), (D, D), (D,), (D, D), (D,)]
        return ([(C, D), (D,), (D,), (D,)] + rff + rff + [(D, HD), (HD,)] * 3 +
                [(D, D), (D,), (D,), (D,), (D, 2 * HD), (2 * HD,)] + [(D, D), (D,), (D,), (D,), (D, D), (D,)] +
                [(HD, HD), (HD,)] + [(HD, HD), (HD,), (HD,), (HD,), (HD, HD), (HD,)] +
                [(HD, D), (D,), (D, D), (D,), (D, O), (O,)])

    # ------------------------------------------------------------------ forward
    def apply(self, params, x, p, a, gaussian_window_size=None):
        """nef.apply(params, x, p, a, gaussian_window) -> (B, N, num_out)   (NEF:204-235).

        x (B,N,dx) [stride-0 batch allowed], p (B,Z,z_pos+z_ori), a (B,Z,latent_dim),
        gaussian_window_size (B,Z,1).  Differentiable w.r.t. p, a, gaussian_window_size, the weights and x.
        """
        inv = self.cross_attn_invariant
        if not (x.is_cuda and p.is_cuda and a.is_cuda):
            raise _lib.EnfError("EquivariantCrossAttentionNeF.apply needs CUDA/HIP tensors: there is no CPU path")
        if x.shape[-1] != inv.num_x_pos_dims:
            raise AssertionError(f"x has coordinate width {x.shape[-1]}, invariant '{inv.name}' expects {inv.num_x_pos_dims}")
        if p.shape[-1] != inv.num_z_pos_dims + inv.num_z_ori_dims:
            raise AssertionError(f"p has width {p.shape[-1]}, expected {inv.num_z_pos_dims + inv.num_z_ori_dims}")
        if a.shape[-1] != self.latent_dim:
            raise AssertionError(f"a has width {a.shape[-1]}, expected latent_dim={self.latent_dim}")
        if x.shape[0] != p.shape[0] or a.shape[:2] != p.shape[:2]:
            raise AssertionError("batch / latent dimensions of x, p, a disagree")
        sigma = gaussian_window_size if self.use_gaussian_window else None
        if self.use_gaussian_window and sigma is None:
            raise AssertionError("gaussian_window_size is required when use_gaussian_window=True")
        if sigma is not None and not torch.is_tensor(sigma):
            sigma = torch.full((p.shape[0], p.shape[1], 1), float(sigma), device=p.device)
        x, p, a = x.float(), p.float(), a.float()
        if sigma is not None:
            sigma = sigma.float().reshape(p.shape[0], p.shape[1], 1)
        ts = self.param_tensors(params)
        if self.num_layers > 0:
            from . import _train
            self._check_shapes(ts)
            return _train.apply_layers(self, ts, x, p, a, sigma)
        if torch.is_grad_enabled() and (x.requires_grad or any(t.requires_grad for t in ts)):
            # training path: gradients w.r.t. the weights (TR:255, NTR:304-339) and / or the query coordinates
            from . import _train
            self._check_shapes(ts)
            return _train.apply_train(self, ts, x, p, a, sigma)
        packed = self.pack(params)
        return _EnfFunction.apply(x, p, a, sigma, self, packed)

    __call__ = apply

    # ------------------------------------------------------------------ relu masks (second-order terms by differences)
    def relu_mask_buffer(self, B, N, Z, device):
        """Device buffer for the relu masks of a (B, N, Z) problem (enf_relu_mask_bytes)."""
        desc = self._desc(B, N, Z)
        return torch.empty(int(_lib.load().enf_relu_mask_bytes(ctypes.byref(desc))) // 4, device=device, dtype=torch.int32)

    def relu_masks(self, buf, mode, signals):
        """Context manager: inside it, THIS model's pair-kernel forwards (any path) WRITE ("write") the relu masks of their
        pre-activations into ``buf``, or its forwards and the weight-gradient backwards of those forwards READ ("read") them
        -- relu linearised at the point the masks were taken, for signals b, b + signals, ... alike (include/enf_hip.h:
        EnfDesc.mask_mode).  The masks travel in each call's descriptor: other models and streams are not affected."""
        import contextlib
        if mode not in ("write", "read"):
            raise ValueError("mode must be 'write' or 'read'")

        @contextlib.contextmanager
        def cm():
            prev, self._masks = self._masks, (buf, mode, int(signals))
            try:
                yield buf
            finally:
                self._masks = prev
        return cm()

    @torch.no_grad()
    def mse_value_and_latent_grads(self, params, x, p, a, gaussian_window_size, target, grad_scale=1.0, loss_out=None):
        """loss = mean((nef.apply(params, x, p, a, window) - target)^2) and grad_scale * d loss / d(p, a, window) in one
        sequence of HIP launches (forward, loss + d out, backward), without building an autograd graph: what one
        inner step of the MAML loop computes (pde_trainer.py:175-207; grad_scale = B there).
        ``loss_out``: an already ZEROED float32 (1,) tensor to accumulate the loss into (saves the fill per call).
        Returns (loss (1,), dp, da, dwindow or None)."""
        lib = _lib.load()
        sigma = gaussian_window_size if self.use_gaussian_window else None
        if self.num_layers > 0:           # no fused sequence for the layered model: autograd through apply()
            with torch.enable_grad():
                leaves = [t.detach().float().requires_grad_(True) for t in (p, a)] + \
                         ([sigma.detach().float().requires_grad_(True)] if sigma is not None else [])
                out = self.apply(params, x, leaves[0], leaves[1], leaves[2] if sigma is not None else None)
                loss = ((out - target) ** 2).mean()
                g = torch.autograd.grad(loss * grad_scale, leaves, allow_unused=True)
            g = [torch.zeros_like(t) if gi is None else gi for t, gi in zip(leaves, g)]
            if loss_out is not None:
                loss_out.add_(loss.detach().reshape(1))
            return loss.detach().reshape(1), g[0], g[1], (g[2] if sigma is not None else None)
        packed = self.pack(params)
        x, p_, a_ = x.float(), p.float().contiguous(), a.float().contiguous()
        s_ = sigma.float().reshape(p_.shape[0], p_.shape[1], 1).contiguous() if sigma is not None else None
        B, Z, N, dev = p_.shape[0], p_.shape[1], x.shape[1], p_.device
        desc = self._desc(B, N, Z, masks=self._masks)
        xb, xstride = self._x_arg(x)
        ws = self._workspace(desc, dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        tgt = target.float().contiguous()
        if tuple(tgt.shape) != (B, N, self.num_out):
            raise AssertionError(f"target has shape {tuple(tgt.shape)}, expected {(B, N, self.num_out)}")
        loss = loss_out if loss_out is not None else torch.zeros(1, device=dev, dtype=torch.float32)
        dp, da = torch.empty_like(p_), torch.empty_like(a_)
        dsig = torch.empty((B, Z, 1), device=dev, dtype=torch.float32)
        if not FUSED_FIT_STEP:       # the same step as three library calls (cross-check in the tests, A/B in scripts/)
            HD = self._Hp * self._Dp
            out = torch.empty((B, N, self.num_out), device=dev, dtype=torch.float32)
            ybar = torch.empty((B, N, HD), device=dev, dtype=torch.float32)
            lse = torch.empty((B, N, self._Hp), device=dev, dtype=torch.float32)
            _lib.launch(dev, lib.enf_forward_stages, ctypes.byref(desc), _ptr(xb), xstride, _ptr(p_), _ptr(a_), _ptr(s_), _ptr(packed),
                        _ptr(out), _ptr(ybar), _ptr(lse), _ptr(ws), ws.numel(), 15 | 16 | 32, st)       # + TAIL_SAVE + PREPARE_BWD
            dout = torch.empty_like(out)
            _lib.launch(dev, lib.enf_mse_value_grad, _ptr(out), _ptr(tgt), out.numel(), float(grad_scale), _ptr(dout), _ptr(loss), st)
            _lib.launch(dev, lib.enf_backward_latents_ex, ctypes.byref(desc), _ptr(xb), xstride, _ptr(p_), _ptr(a_), _ptr(s_),
                        _ptr(packed), _ptr(ybar), _ptr(lse), _ptr(dout), _ptr(dp), _ptr(da), _ptr(dsig), _ptr(ws), ws.numel(), 1 | 2 | 4, st)
            self._ws_touch(ws)
            return loss, dp, da, (dsig if sigma is not None else None)
        # ONE library call per inner step (include/enf_hip.h: enf_fit_step): prologue, pair forward, the tail as a single kernel with
        # the loss and its gradient formed in registers, pair backward, prologue backward
        _lib.launch(dev, lib.enf_fit_step, ctypes.byref(desc), _ptr(xb), xstride, _ptr(p_), _ptr(a_), _ptr(s_), _ptr(packed), _ptr(tgt),
                    float(grad_scale), _ptr(loss), _ptr(dp), _ptr(da), _ptr(dsig), _ptr(ws), ws.numel(), st)
        self._ws_touch(ws)
        return loss, dp, da, (dsig if sigma is not None else None)
