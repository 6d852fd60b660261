"""ctypes binding of the C-ABI in include/enf_hip.h (libenf_hip.so, built in-tree by
``make -C enf-pde_amd/csrc`` / ``__graft_entry__.build()``).  Fails loudly when the library is
missing: there is no fallback path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ENF_HIP_LIB") or os.path.join(_HERE, "libenf_hip.so")   # ENF_HIP_LIB: A/B builds (scripts/build_variant.sh)
TEST_LIB_PATH = os.environ.get("ENF_HIP_TEST_LIB") or os.path.join(_HERE, "libenf_hip_test.so")    # same ABI + test hooks (csrc/Makefile); tests only

ENF_NUM_TENSORS = 46
PREC = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}
INVARIANT_IDS = {"rel_pos_periodic": 0, "latitude_periodic": 1, "polar_periodic": 2, "ponita": 3,
                 "abs_pos": 4, "rel_pos": 5, "norm_rel_pos": 6, "ball": 7, "ball_lat": 8, "ponita_full": 9}

EXPORTS = ["enf_abi_version", "enf_strerror", "enf_invariant_dim", "enf_invariant_pose_dim", "enf_check_desc",
           "enf_packed_weight_bytes", "enf_pack_weights", "enf_workspace_bytes", "enf_forward",
           "enf_backward_latents", "enf_backward_latents_ex", "enf_forward_stages", "enf_lt_layout", "enf_lt_layout_ext", "enf_pack_pair", "enf_pair_forward",
           "enf_pair_backward", "enf_pair_backward_ex", "enf_pair_scratch_bytes", "enf_pair_variant", "enf_pair_partition", "enf_backward_weights", "enf_backward_weights_scratch_bytes",
           "enf_backward_all", "enf_backward_all_scratch_bytes", "enf_fit_step", "enf_fit_inputs",
           "enf_mse_value_grad",
           "enf_ode_conv_forward", "enf_ode_conv_backward_basis", "enf_ode_conv_backward_weight", "enf_ode_conv_backward_weight_scratch_bytes", "enf_ode_poly_num_features", "enf_ode_poly_forward",
           "enf_ode_poly_backward", "enf_ode_vec_readout_forward", "enf_ode_vec_readout_backward", "enf_ode_block_supported", "enf_ode_block_scratch_bytes", "enf_ode_block_forward", "enf_ode_block_backward",
           "enf_ode_basis_supported", "enf_ode_basis_scratch_bytes", "enf_ode_basis_forward",
           "enf_ode_basis_backward", "enf_relu_mask_bytes", "enf_meta_sgd_update"]
ENF_NUM_PAIR_TENSORS = 12          # ENF_P_* of include/enf_hip.h
(ENF_S_EQ, ENF_S_EV, ENF_S_G1, ENF_S_NH, ENF_S_DA1, ENF_S_DA2, ENF_S_DA3, ENF_S_HEAD0) = range(8)


def num_store(H):
    """ENF_NUM_STORE(H)"""
    return 7 + 4 * H


class EnfDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("B", "N", "Z", "H", "D", "C", "O", "dx", "invariant_id", "use_window", "precision")] + \
               [("h_true", ctypes.c_int32), ("d_true", ctypes.c_int32),
                # per-call options (include/enf_hip.h): the library keeps no settings
                ("pair_fwd_variant", ctypes.c_int32), ("pair_bwd_variant", ctypes.c_int32), ("mask_mode", ctypes.c_int32),
                ("mask_signals", ctypes.c_int32), ("reserved", ctypes.c_int32), ("relu_masks", ctypes.c_void_p)]


VARIANT = {"auto": 0, "latent_split": 1, "z_fold": 2, "z_fold_zsplit": 3}       # ENF_VARIANT_*
MASK_MODE = {"off": 0, "write": 1, "read": 2}                # ENF_MASK_*


class EnfSgdSegment(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("g", ctypes.c_void_p), ("lr", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("n", ctypes.c_int64), ("width", ctypes.c_int32), ("g_stride", ctypes.c_int32), ("lr_len", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


ENF_SGD_MAX_SEGMENTS = 4


class EnfFitComponent(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("width", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class EnfError(RuntimeError):
    pass


_lib = None
_test_lib = None


def load():
    """Load libenf_hip.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _lib = _bind(LIB_PATH, test_hooks=False)
    return _lib


def load_test():
    """The test library (libenf_hip_test.so): the whole product ABI plus the test-only entry points.  Tests only."""
    global _test_lib
    if _test_lib is None:
        _test_lib = _bind(TEST_LIB_PATH, test_hooks=True)
    return _test_lib


class using:
    """Context manager for tests: route the package's calls through another build of the library (e.g. load_test())."""

    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        global _lib
        self.prev, _lib = _lib, self.lib
        return self.lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev


def _bind(path, test_hooks):
    if not os.path.exists(path):
        raise EnfError(f"{path} not found: build it with `make -C enf-pde_amd/csrc` "
                       "(or python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path.")
    lib = ctypes.CDLL(path)
    vp, i64, sz, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_size_t, ctypes.c_int
    dp = ctypes.POINTER(EnfDesc)
    lib.enf_abi_version.restype = ci
    lib.enf_strerror.restype = ctypes.c_char_p
    lib.enf_strerror.argtypes = [ci]
    lib.enf_invariant_dim.argtypes = [ci, ci]
    lib.enf_invariant_pose_dim.argtypes = [ci, ci]
    lib.enf_check_desc.argtypes = [dp]
    lib.enf_packed_weight_bytes.restype = sz
    lib.enf_packed_weight_bytes.argtypes = [dp]
    lib.enf_workspace_bytes.restype = sz
    lib.enf_workspace_bytes.argtypes = [dp]
    lib.enf_pack_weights.argtypes = [dp, ctypes.POINTER(vp), vp, vp]
    lib.enf_forward.argtypes = [dp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.enf_forward_stages.argtypes = [dp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, sz, ctypes.c_uint, vp]
    lib.enf_backward_latents.argtypes = [dp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.enf_backward_latents_ex.argtypes = [dp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, ctypes.c_uint, vp]
    ip = ctypes.POINTER(ci)
    lib.enf_lt_layout.argtypes = [dp, ip, ip, ip, ip, ip, ip]
    lib.enf_pack_pair.argtypes = [dp, ctypes.POINTER(vp), vp, vp]
    lib.enf_mse_value_grad.argtypes = [vp, vp, sz, ctypes.c_float, vp, vp, vp]
    lib.enf_meta_sgd_update.argtypes = [ctypes.c_int, ctypes.POINTER(EnfSgdSegment), ctypes.c_float, vp]
    lib.enf_fit_inputs.argtypes = [ctypes.c_int, ctypes.POINTER(EnfFitComponent)] + [ctypes.c_int32] * 7 + [vp] * 7
    lib.enf_ode_conv_forward.argtypes = [ci, ci, ci, ci, vp, vp, i64, i64, vp, vp, vp, vp]
    lib.enf_ode_conv_backward_basis.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp]
    lib.enf_ode_conv_backward_weight_scratch_bytes.restype = sz
    lib.enf_ode_conv_backward_weight_scratch_bytes.argtypes = [ci, ci, ci, ci]
    lib.enf_ode_conv_backward_weight.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, sz, vp]
    lib.enf_ode_poly_num_features.argtypes = [ci, ci]
    lib.enf_ode_poly_forward.argtypes = [i64, ci, ci, vp, vp, vp]
    lib.enf_ode_poly_backward.argtypes = [i64, ci, ci, vp, vp, vp, vp]
    lib.enf_ode_basis_supported.argtypes = [ci, ci, ci, ci, ci]
    cf = ctypes.c_float
    lib.enf_ode_vec_readout_forward.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, cf, cf, vp, vp, vp]
    lib.enf_ode_vec_readout_backward.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, cf, cf, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.enf_ode_block_supported.argtypes = [ci, ci]
    lib.enf_ode_block_scratch_bytes.restype = sz
    lib.enf_ode_block_scratch_bytes.argtypes = [i64, ci, ci]
    lib.enf_ode_block_forward.argtypes = [i64, ci, ci, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp]
    lib.enf_ode_block_backward.argtypes = [i64, ci, ci, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, sz, vp]
    lib.enf_ode_basis_scratch_bytes.restype = sz
    lib.enf_ode_basis_scratch_bytes.argtypes = [i64, ci, ci, ci, ci]
    lib.enf_ode_basis_forward.argtypes = [i64, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.enf_ode_basis_backward.argtypes = [i64, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.enf_relu_mask_bytes.restype = sz
    lib.enf_relu_mask_bytes.argtypes = [dp]
    lib.enf_pair_variant.argtypes = [dp, ci]
    lib.enf_pair_partition.argtypes = [dp] + [ctypes.POINTER(ctypes.c_int32)] * 3
    lib.enf_backward_weights_scratch_bytes.restype = sz
    lib.enf_backward_weights_scratch_bytes.argtypes = [dp, ci]
    lib.enf_backward_weights.argtypes = [dp, vp, i64, vp, vp, vp, vp, vp, vp, ctypes.POINTER(vp), vp, vp, sz, vp]
    lib.enf_fit_step.argtypes = [dp, vp, i64, vp, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, vp, vp, sz, vp]
    lib.enf_backward_all_scratch_bytes.restype = sz
    lib.enf_backward_all_scratch_bytes.argtypes = [dp, ci]
    lib.enf_backward_all.argtypes = [dp, vp, i64, vp, vp, vp, ctypes.POINTER(vp), vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(vp), vp,
                                     vp, sz, vp, sz, ctypes.c_uint, vp]
    lib.enf_pair_scratch_bytes.restype = sz
    lib.enf_pair_scratch_bytes.argtypes = [dp]
    lib.enf_pair_forward.argtypes = [dp, vp, i64, vp, vp, vp, vp, vp, sz, vp]
    lib.enf_pair_backward.argtypes = [dp, vp, i64, vp, vp, vp, vp, vp, vp, ctypes.POINTER(vp), vp]
    lib.enf_pair_backward_ex.argtypes = [dp, vp, i64, vp, vp, vp, vp, vp, vp, ctypes.POINTER(vp), vp, vp]
    if test_hooks:
        for name in ("enf_debug_gemm", "enf_debug_pack", "enf_test_read_wave_sums"):
            getattr(lib, name).restype = ci
        lib.enf_debug_gemm.argtypes = [vp, vp, vp, ci, ci, ci, vp]
        lib.enf_debug_pack.argtypes = [vp, vp, ci, ci, ci, vp]
        lib.enf_test_read_wave_sums.argtypes = [vp]
    if lib.enf_abi_version() != 2:
        raise EnfError(f"{path}: ABI version mismatch")
    return lib


def check(rc):
    """Map a negative return code to the exception the reference would raise."""
    if rc == 0:
        return
    msg = load().enf_strerror(rc).decode()
    if rc == -2:   # ENF_EINVARIANT: reference raises ValueError (invariant/__init__.py:44,78)
        raise ValueError(msg)
    if rc == -6:   # ENF_EDIM: reference asserts (invariant/__init__.py:28,31,62,65)
        raise AssertionError(msg)
    if rc == -3:
        raise NotImplementedError(msg)
    raise EnfError(f"libenf_hip error {rc}: {msg}")


def launch(dev, fn, *args):
    """Call a library entry point that enqueues work on a stream of ``dev`` (a torch.device) with ``dev`` as the calling
    thread's current device -- the library's per-device bookkeeping (side stream, kernel attributes) goes by it -- and map
    its return code like ``check``."""
    import torch
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx == torch.cuda.current_device():
        return check(fn(*args))
    with torch.cuda.device(idx):
        return check(fn(*args))


def make_desc(B, N, Z, H, D, C, O, dx, invariant_id, use_window, precision, d_true=0, h_true=0, variants=(0, 0),
              masks=None):
    """``variants``: (forward, backward) ENF_VARIANT_*; ``masks``: None or (int32 device tensor, "write" | "read", signals)."""
    d = EnfDesc()
    d.B, d.N, d.Z, d.H, d.D, d.C, d.O, d.dx = int(B), int(N), int(Z), int(H), int(D), int(C), int(O), int(dx)
    d.invariant_id, d.use_window, d.precision = int(invariant_id), int(bool(use_window)), int(precision)
    d.d_true, d.h_true = int(d_true), int(h_true)
    d.pair_fwd_variant, d.pair_bwd_variant = int(variants[0]), int(variants[1])
    if masks is not None:
        buf, mode, signals = masks
        d.relu_masks, d.mask_mode, d.mask_signals = buf.data_ptr(), MASK_MODE[mode], int(signals)
    return d
