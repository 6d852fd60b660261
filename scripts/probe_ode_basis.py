"""Timing of the fused kernel-basis kernels alone (csrc/enf_ode_basis.hip) at the bench shape, for a list of builds:
    python scripts/probe_ode_basis.py [lib.so ...]        (default: the product library)
Builds with -DOB_SKIP_* leave out a phase of the backward kernel (wrong results, timing only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from enf_pde_amd import _lib

dev = torch.device("cuda:0")
B, Z, I, H1, J = 16, 64, 4, 128, 64
P, F = B * Z * Z, 340
g = torch.Generator().manual_seed(0)
mk = lambda *s: torch.randn(*s, generator=g).to(dev)
x, W1, b1, W3, b3, dkb = mk(P, I) * 0.7, mk(F, H1) / F ** 0.5, mk(H1) * 0.1, mk(H1, J) / H1 ** 0.5, mk(J) * 0.1, mk(P, J)
kb, dx, dW1, db1, dW3, db3 = torch.empty(P, J, device=dev), torch.empty_like(x), torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(W3), torch.empty_like(b3)
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
vp, ci, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
for path in (sys.argv[1:] or [_lib.LIB_PATH]):
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.enf_ode_basis_scratch_bytes.restype = sz
    lib.enf_ode_basis_scratch_bytes.argtypes = [i64, ci, ci, ci, ci]
    lib.enf_ode_basis_forward.argtypes = [i64, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.enf_ode_basis_backward.argtypes = [i64, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    n = lib.enf_ode_basis_scratch_bytes(P, I, H1, J, 1)
    sc = torch.empty(n // 4, device=dev)
    st = vp(torch.cuda.current_stream().cuda_stream)
    fwd = lambda: lib.enf_ode_basis_forward(P, I, 3, H1, J, ptr(x), ptr(W1), ptr(b1), ptr(W3), ptr(b3), ptr(kb), ptr(sc), n, st)
    bwd = lambda: lib.enf_ode_basis_backward(P, I, 3, H1, J, ptr(x), ptr(W1), ptr(b1), ptr(W3), ptr(b3), ptr(dkb), ptr(dx), ptr(dW1),
                                             ptr(db1), ptr(dW3), ptr(db3), ptr(sc), n, st)
    out = []
    for fn in (fwd, bwd):
        for _ in range(3):
            assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
    flops_b = 2.0 * P * ((340 * H1 + H1 * J) + H1 * J + 352 * H1 + 352 * H1 + J * H1)   # recompute, d h1, d features, d W1, d W3
    print(f"{os.path.basename(path):28s} forward {out[0]:7.1f} us   backward (pack + kernel + reduce) {out[1]:7.1f} us   "
          f"[backward {flops_b / out[1] / 1e6:5.1f} TFLOP/s of 157]", flush=True)
