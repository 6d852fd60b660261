"""Weight-gradient parity: the training path (HIP pair kernels, enf_backward_weights for every per-pair dW, differentiable
prologue / tail) against fp64 autograd of the torch oracle over every tensor of the parameter
tree -- what the reference gets from jax.value_and_grad over params['nef']
(pde_trainer.py:255, nonmaml_pde_trainer.py:304-339)."""
import numpy as np
import pytest
import torch

from oracle import enf_ref_np as R
from oracle import enf_ref_torch as T
from tests.helpers import make_cfg, make_inputs, build_nef
from enf_pde_amd.enf.models import TENSOR_PATHS
from types import SimpleNamespace as NS_

pytestmark = pytest.mark.gpu

# relative L2 error per tensor; tensors whose reference gradient is ~0 are compared absolutely.
# bf16: typical tensors sit at ~1 %.  The two relu layers are the outliers (3 % value branch, 6-13 %
# query branch): a pre-activation within bf16 noise of zero flips its relu mask, an O(1) change of
# that element's delta, and in the query branch the deltas additionally cancel over z
# (sum_z dlogit = 0).  Measured per tensor with scripts/wgrad_err.py; f32 mode has no such effect.
TOL = {"f32": 5e-4, "bf16": 2e-1}


def _get(tree, path):
    for k in path:
        tree = tree[k]
    return tree


def ref(prm, cfg, x, p, a, s, w):
    tp = T.to_torch(prm, torch.float64, requires_grad=True)
    tpp, ta, ts = (torch.tensor(v, requires_grad=True) for v in (p, a, s))
    out = T.nef_apply(tp, cfg, torch.tensor(x), tpp, ta, ts)
    (out * torch.tensor(w)).sum().backward()
    g = lambda t: np.zeros(tuple(t.shape)) if t.grad is None else t.grad.numpy()
    return out.detach().numpy(), [g(_get(tp["params"], path)) for path in TENSOR_PATHS], g(tpp), g(ta), g(ts)


def hip(cuda, nef, prm, x, p, a, s, w):
    params = nef.load_params(prm, device=cuda)
    ts_ = nef.param_tensors(params)
    for t in ts_:
        t.requires_grad_(True)
    t = lambda v, g=False: torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=g)
    tp, ta, ts = t(p, True), t(a, True), t(s, True)
    out = nef.apply(params, t(x), tp, ta, ts)
    (out * t(w)).sum().backward()
    torch.cuda.synchronize()
    g = lambda v: np.zeros(tuple(v.shape)) if v.grad is None else v.grad.cpu().numpy().astype(np.float64)
    return out.detach().cpu().numpy(), [g(v) for v in ts_], g(tp), g(ta), g(ts)


def check(cuda, cfg, B, N, Z, precision, seed=0):
    prm = R.init_params(seed, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, B, N, Z, seed + 1)
    w = np.random.default_rng(seed + 2).standard_normal((B, N, cfg["num_out"]))
    ro, rg, rp, ra, rs = ref(prm, cfg, x, p, a, s, w)
    nef = build_nef(cfg, precision)
    ho, hg, hp, ha, hs = hip(cuda, nef, prm, x, p, a, s, w)
    tol = TOL[precision]
    assert np.abs(ho - ro).max() / np.abs(ro).max() < (2e-5 if precision == "f32" else 3e-2)
    gmax = max(np.linalg.norm(g) for g in rg)
    bad = []
    for path, g, r in zip(TENSOR_PATHS, hg, rg):
        nr = np.linalg.norm(r)
        if path[-1] == "coefficients":                       # frozen: stop_gradient (rff.py:87-90)
            assert np.all(g == 0) and nr == 0
            continue
        err = np.linalg.norm(g - r) / nr if nr > 1e-6 * gmax else np.linalg.norm(g - r) / gmax
        if not (np.isfinite(err) and err < tol):
            bad.append(("/".join(path), err))
    assert not bad, (precision, bad)
    for name, g, r in (("p", hp, rp), ("a", ha, ra), ("sigma", hs, rs)):
        if np.linalg.norm(r) > 0:
            e = np.linalg.norm(g - r) / np.linalg.norm(r)
            assert e < tol, (name, e)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("invariant", ["rel_pos_periodic", "ponita", "polar_periodic"])
def test_weight_grads_invariants(cuda, invariant, precision):
    cfg = make_cfg(invariant, D=128, H=2, C=16, O=3, freq=(0.5, 1.0))
    check(cuda, cfg, B=2, N=70, Z=9, precision=precision)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("D,H,C,O,Z,N", [(128, 2, 16, 1, 64, 256), (64, 2, 16, 1, 16, 100), (128, 1, 32, 3, 18, 33),
                                         (64, 1, 8, 2, 4, 32)])
def test_weight_grads_shapes(cuda, D, H, C, O, Z, N, precision):
    cfg = make_cfg("rel_pos_periodic", D=D, H=H, C=C, O=O)
    check(cuda, cfg, B=3, N=N, Z=Z, precision=precision, seed=D + Z)


def test_weight_grads_chunked_store(cuda, monkeypatch):
    """The backward materialises activations in batch chunks; a tiny budget forces several chunks."""
    from enf_pde_amd.enf.models import _train
    cfg = make_cfg("rel_pos_periodic", D=64, H=2, C=16, O=1)
    per_b = 8 * 48 * 64 * (7 + 8) * 4
    monkeypatch.setattr(_train, "STORE_BUDGET_BYTES", 2 * per_b + (1 << 20))      # two signals' store + the partial sums
    check(cuda, cfg, B=5, N=48, Z=8, precision="f32", seed=3)


def _pair_problem(cuda, D, H, precision, B, N, Z, seed):
    """Random pair-level inputs of enf_backward_weights / enf_pair_backward: (nef, desc pieces) through the training path's own
    latent table and effective tensors, so that the activations are in their working range."""
    import ctypes
    from enf_pde_amd import _lib
    from enf_pde_amd.enf.models import _train
    cfg = make_cfg("rel_pos_periodic", D=D, H=H, C=16, O=1)
    nef = build_nef(cfg, precision)
    nef.pair_variants = ("latent_split", "latent_split")
    prm = R.init_params(seed, cfg, jitter=0.1)
    params = nef.load_params(prm, device=cuda)
    x, p, a, s = make_inputs(cfg, B, N, Z, seed + 1)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    W = dict(zip(_train.W_NAMES, nef.param_tensors(params)))
    desc = nef._desc(B, N, Z)
    lt = _train.latent_table(nef, W, t(p), t(a), t(s), _train.lt_layout(desc)).contiguous()
    eff = [e.float().contiguous() for e in _train.effective_pair_params(nef, W)]
    lib = _lib.load()
    blob = torch.empty(int(lib.enf_packed_weight_bytes(ctypes.byref(desc))), device=cuda, dtype=torch.uint8)
    st = ctypes.c_void_p(torch.cuda.current_stream(cuda).cuda_stream)
    P = lambda v: ctypes.c_void_p(v.data_ptr()) if v is not None else ctypes.c_void_p(0)
    _lib.check(lib.enf_pack_pair(ctypes.byref(desc), (ctypes.c_void_p * 12)(*[e.data_ptr() for e in eff]), P(blob), st))
    xs = t(x).contiguous()
    ybar, lse = torch.empty(B, N, H * D, device=cuda), torch.empty(B, N, H, device=cuda)
    _lib.check(lib.enf_pair_forward(ctypes.byref(desc), P(xs), N * 2, P(lt), P(blob), P(ybar), P(lse), None, 0, st))
    g = torch.Generator().manual_seed(seed)
    dybar = (torch.randn(B, N, H * D, generator=g) / N).to(cuda)
    delta = (dybar * ybar).view(B, N, H, D).sum(-1).contiguous()
    return NS_(lib=lib, _lib=_lib, nef=nef, desc=desc, xs=xs, lt=lt, blob=blob, lse=lse, dybar=dybar, delta=delta, st=st, P=P,
               B=B, N=N, Z=Z, D=D, H=H, bf16=precision == "bf16", keep=(eff, ybar))


def _run_backward_weights(q, chunk):
    import ctypes
    D, HD = q.D, q.H * q.D
    shapes = [(D, D), (D,), (D, D), (D,), (D, D), (D,), (D, 2 * HD), (2 * HD,), (D, D), (D,)]
    grads = [torch.full(sh, float("nan"), device=q.lt.device) for sh in shapes]
    arr = (ctypes.c_void_p * 12)(*([g.data_ptr() for g in grads] + [None, None]))
    nbytes = int(q.lib.enf_backward_weights_scratch_bytes(ctypes.byref(q.desc), chunk))
    assert nbytes > 0
    scratch = torch.empty(nbytes, device=q.lt.device, dtype=torch.uint8)
    dlt = torch.empty_like(q.lt)
    q._lib.check(q.lib.enf_backward_weights(ctypes.byref(q.desc), q.P(q.xs), q.N * 2, q.P(q.lt), q.P(q.blob), q.P(q.lse),
                                            q.P(q.dybar), q.P(q.delta), q.P(dlt), arr, None, q.P(scratch), nbytes, q.st))
    torch.cuda.synchronize()
    q.last_scratch = scratch                             # (the activation store of this call: read by the blame report below)
    return grads, dlt


def _blame(q, store_ref, ns, sdt):
    """When enf_backward_weights disagrees with the products of a second K3 run's store (the open run-to-run item, DESIGN.md
    "K3 run-to-run deviations"): which rows of which ENF_S_* buffer differ between the two K3 runs -- none means K4 is at fault."""
    P, D = store_ref.shape[1], store_ref.shape[2]
    sb = (P * D * store_ref.element_size() + 255) // 256 * 256
    out = []
    for i in range(ns):
        own = q.last_scratch[i * sb:i * sb + P * D * store_ref.element_size()].view(sdt).view(P, D)
        bad = (own != store_ref[i]).any(1).nonzero().flatten().tolist()
        if bad:
            out.append(f"ENF_S[{i}]: {len(bad)} rows differ, first {bad[:8]}")
    return "two K3 runs stored different activations: " + "; ".join(out) if out else "the two K3 stores are identical: K4 differs"


@pytest.mark.parametrize("D,H", [(128, 1), (64, 2)])
def test_k3_store_is_bitwise_reproducible_behind_foreign_kernels(cuda, D, H):
    """Regression test of the round-1/2 "K3 run-to-run deviations" (DESIGN.md): with a foreign kernel (the fused ODE kernel basis on
    NaN inputs: all LDS, all registers, every CU) in front of every second call, the activations K3 stores and the weight
    gradients are bitwise those of the first call, 800 times.  The build before the fix (LayerNorm apply as SLP-packed
    v_pk_add_f32 with an op_sel broadcast) deviates in 1.3 % of such calls for <128, 1, bf16> -- one feature of one 16-query
    tile keeps its mean -- (scripts/k3_race/store_probe.py), so it fails this test with near certainty."""
    import ctypes
    from enf_pde_amd.fitting.ode_models.ponita_ode_g import kernel_basis
    B, N, Z = 5, 77, 6
    q = _pair_problem(cuda, D, H, "bf16", B, N, Z, seed=D + H)
    nanx = torch.full((65536, 4), float("nan"), device=cuda, requires_grad=True)
    K1 = {"kernel": torch.full((340, 128), float("nan"), device=cuda), "bias": torch.full((128,), float("nan"), device=cuda)}
    K3 = {"kernel": torch.full((128, 64), float("nan"), device=cuda), "bias": torch.full((64,), float("nan"), device=cuda)}
    first = None
    for it in range(800):
        if it % 2 == 1:
            kernel_basis(nanx, 3, K1, K3).sum().backward()
        grads, _ = _run_backward_weights(q, B)
        store = q.last_scratch.clone()
        assert all(torch.isfinite(g).all() for g in grads)
        if first is None:
            first = (store, grads)
        else:
            part0 = (7 + 4 * H) * ((B * Z * N * D * 2 + 255) // 256 * 256)          # the ENF_S_* buffers (the partials behind them are scratch)
            assert torch.equal(store[:part0], first[0][:part0]), f"call {it}: K3 stored different activations"
            assert all(torch.equal(a, b) for a, b in zip(grads, first[1])), f"call {it}: different weight gradients"


@pytest.mark.parametrize("D,H,precision", [(128, 2, "bf16"), (128, 2, "f32"), (64, 2, "bf16"), (64, 1, "f32"), (128, 1, "bf16"), (64, 4, "bf16")])
def test_backward_weights_kernel(cuda, D, H, precision):
    """enf_backward_weights (K3 store + K4) against fp64 X^T delta / column sums of the very activations K3 stores
    (enf_pair_backward with an explicit store): K4 itself adds nothing but fp32 accumulation order.  Chunked passes give the
    one-pass result, and the same call twice gives the same bits (fixed reduction order)."""
    import ctypes
    B, N, Z = 5, 77, 6                                  # P = 2310 rows: not a multiple of the 32-row tile
    q = _pair_problem(cuda, D, H, precision, B, N, Z, seed=D + H)
    ns = 7 + 4 * H
    sdt = torch.bfloat16 if q.bf16 else torch.float32
    store = torch.zeros(ns, B * Z * N, D, device=cuda, dtype=sdt)
    dlt0 = torch.empty_like(q.lt)
    q._lib.check(q.lib.enf_pair_backward(ctypes.byref(q.desc), q.P(q.xs), N * 2, q.P(q.lt), q.P(q.blob), q.P(q.lse), q.P(q.dybar),
                                         q.P(q.delta), q.P(dlt0), (ctypes.c_void_p * ns)(*[store[i].data_ptr() for i in range(ns)]), q.st))
    torch.cuda.synchronize()
    S = store.double()
    if q.bf16:                                           # stored column -> true feature (include/enf_hip.h, ENF_S_*)
        c = torch.arange(D, device=cuda)
        j = c % 8
        true = 32 * (c // 32) + torch.where(j < 4, 4 * ((c % 32) // 8) + j, 16 + 4 * ((c % 32) // 8) + j - 4)
        S = torch.empty_like(S).index_copy_(2, true, S)
    HD = H * D
    xtd = lambda i, k: S[i].t() @ S[k]
    want = [xtd(0, 4), S[4].sum(0), xtd(1, 5), S[5].sum(0), xtd(2, 6), S[6].sum(0),
            torch.cat([xtd(3, 7 + 4 * h + 2) for h in range(H)] + [xtd(3, 7 + 4 * h + 3) for h in range(H)], 1),
            torch.cat([S[7 + 4 * h + 2].sum(0) for h in range(H)] + [S[7 + 4 * h + 3].sum(0) for h in range(H)]),
            sum(xtd(7 + 4 * h, 7 + 4 * h + 1) for h in range(H)), sum(S[7 + 4 * h + 1].sum(0) for h in range(H))]
    one, dlt1 = _run_backward_weights(q, B)
    for i, (g, w) in enumerate(zip(one, want)):
        assert torch.isfinite(g).all()
        if float((g.double() - w).abs().max()) > 2e-5 * float(w.abs().max()) + 1e-30:
            pytest.fail(f"tensor {i}: off by {float((g.double() - w).abs().max() / w.abs().max()):.2e}; " + _blame(q, store, ns, sdt))
    assert float((dlt1 - dlt0).abs().max()) <= 2e-5 * float(dlt0.abs().max())
    again, _ = _run_backward_weights(q, B)
    assert all(torch.equal(g, h) for g, h in zip(one, again))
    for chunk in (2, 1):
        part, dltc = _run_backward_weights(q, chunk)
        for g, h in zip(part, one):
            assert float((g - h).abs().max()) <= 2e-5 * float(h.abs().max()) + 1e-30
        assert float((dltc - dlt0).abs().max()) <= 2e-5 * float(dlt0.abs().max())
    # too little scratch for even one signal is refused
    grads = [torch.empty_like(g) for g in one]
    arr = (ctypes.c_void_p * 12)(*([g.data_ptr() for g in grads] + [None, None]))
    small = torch.empty(1024, device=cuda, dtype=torch.uint8)
    assert q.lib.enf_backward_weights(ctypes.byref(q.desc), q.P(q.xs), N * 2, q.P(q.lt), q.P(q.blob), q.P(q.lse), q.P(q.dybar),
                                      q.P(q.delta), q.P(dlt0), arr, None, q.P(small), 1024, q.st) == -4


def test_training_path_matches_inference_path(cuda):
    """Same outputs and latent gradients whichever path apply() takes."""
    cfg = make_cfg("rel_pos_periodic", D=128, H=2, C=16, O=1)
    prm = R.init_params(5, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 2, 96, 16, 6)
    nef = build_nef(cfg, "f32")
    t = lambda v, g=False: torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=g)
    res = []
    for train in (False, True):
        params = nef.load_params(prm, device=cuda)
        if train:
            for w in nef.param_tensors(params):
                w.requires_grad_(True)
        tp, ta, ts = t(p, True), t(a, True), t(s, True)
        out = nef.apply(params, t(x), tp, ta, ts)
        out.square().sum().backward()
        res.append((out.detach(), tp.grad, ta.grad, ts.grad))
    for u, v in zip(*res):
        assert torch.allclose(u, v, rtol=2e-3, atol=2e-4 * float(v.abs().max()))


def test_relu_masks_replay(cuda):
    """EnfDesc.mask_mode: masks written at a point and replayed AT THE SAME POINT change nothing (the linearised relu
    equals the relu there), forward and every gradient; replayed at a perturbed point they differ from the free relu."""
    cfg = make_cfg("rel_pos_periodic", D=64, H=2, C=8, O=1, freq=(0.5, 1.0))
    prm = R.init_params(5, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 2, 40, 9, 6)
    nef = build_nef(cfg, "f32")
    P = nef.load_params(prm, device=cuda)
    t = lambda v, g=False: torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=g)
    w = t(np.random.default_rng(7).standard_normal((2, 40, 1)))

    def run(pp, masks=None):
        ws = [v.detach().clone().requires_grad_(True) for v in nef.param_tensors(P)]
        from enf_pde_amd.fitting.trainers.pde_trainer import _tree_from_tensors
        import contextlib
        pa, aa = t(pp, True), t(a, True)
        with (nef.relu_masks(masks, "read", 2) if masks is not None else contextlib.nullcontext()):
            out = nef.apply(_tree_from_tensors(ws), t(x), pa, aa, t(s))
            g = torch.autograd.grad((out * w).sum(), ws + [pa, aa], allow_unused=True)
        return out.detach(), [gi for gi in g if gi is not None]
    buf = nef.relu_mask_buffer(2, 40, 9, cuda)
    buf.zero_()
    with torch.no_grad(), nef.relu_masks(buf, "write", 2):
        nef.apply(P, t(x), t(p), t(a), t(s))
    assert int((buf != 0).sum()) > 0
    o0, g0 = run(p)
    o1, g1 = run(p, buf)
    assert torch.equal(o0, o1) and all(torch.equal(u, v) for u, v in zip(g0, g1))
    p2 = p + 0.05
    o2, g2 = run(p2)
    o3, g3 = run(p2, buf)                       # relu linearised at p, evaluated at p2
    assert not torch.equal(o2, o3)
    o4, _ = run(p)                              # the setting is consumed: the next pass is a plain one again
    assert torch.equal(o4, o0)


def test_relu_masks_replay_in_chunked_passes(cuda, monkeypatch):
    """The outer step's difference pass runs 2B signals against masks taken for B (signal b replays b % B): when the
    activation store is chunked, a chunk starts at a signal offset and must still pick its own masks (EnfDims.mask_b0)."""
    from enf_pde_amd.enf.models import _train
    from enf_pde_amd.fitting.trainers.pde_trainer import _tree_from_tensors
    cfg = make_cfg("rel_pos_periodic", D=64, H=2, C=8, O=1, freq=(0.5, 1.0))
    prm = R.init_params(5, cfg, jitter=0.1)
    B, N, Z = 3, 40, 9
    x, p, a, s = make_inputs(cfg, 2 * B, N, Z, 6)
    nef = build_nef(cfg, "f32")
    P = nef.load_params(prm, device=cuda)
    t = lambda v, g=False: torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=g)
    w = t(np.random.default_rng(7).standard_normal((2 * B, N, 1)))
    buf = nef.relu_mask_buffer(B, N, Z, cuda)
    with torch.no_grad(), nef.relu_masks(buf, "write", B):
        nef.apply(P, t(x[:B]), t(p[:B] + 0.03), t(a[:B]), t(s[:B]))         # masks at a nearby point: replaying them matters

    def run():
        ws = [v.detach().clone().requires_grad_(True) for v in nef.param_tensors(P)]
        pa, aa = t(p, True), t(a, True)
        with nef.relu_masks(buf, "read", B):
            out = nef.apply(_tree_from_tensors(ws), t(x), pa, aa, t(s))
            g = torch.autograd.grad((out * w).sum(), ws + [pa, aa], allow_unused=True)
        return [gi for gi in g if gi is not None]
    one = run()
    per_b = Z * N * 64 * (7 + 8) * 4
    monkeypatch.setattr(_train, "STORE_BUDGET_BYTES", B * per_b + (1 << 20))      # B of the 2B signals per chunk
    two = run()
    for u, v in zip(one, two):
        assert float((u - v).abs().max()) <= 2e-5 * float(u.abs().max()) + 1e-30


@pytest.mark.parametrize("case", range(8))
def test_weight_grads_random_shape_sweep(cuda, case):
    """Seeded random shapes for the training path (ragged N / Z, invariants, widths, heads; fp32 kernels and, for the
    odd cases, bf16 kernels with their permuted activation store)."""
    rng = np.random.default_rng(3000 + case)
    inv = ["rel_pos_periodic", "latitude_periodic", "ponita", "rel_pos", "norm_rel_pos", "ball", "polar_periodic", "abs_pos"][case]
    D, H = [(64, 1), (64, 2), (128, 1), (128, 2), (64, 4)][int(rng.integers(5))]
    if inv == "ball":
        D, H = 64, min(H, 2)
    B, N, Z = int(rng.integers(1, 4)), int(rng.integers(2, 120)), int(rng.integers(2, 30))
    cfg = make_cfg(inv, D=D, H=H, C=int(rng.integers(2, 20)), O=int(rng.integers(1, 4)), freq=(0.3, 0.6))
    check(cuda, cfg, B=B, N=N, Z=Z, precision="f32" if case % 2 == 0 else "bf16", seed=4000 + case)


def test_weight_grads_are_reproducible(cuda):
    """The training path's kernel (unfolded chain with the activation store) at D = 64, H = 2, bf16: the same inputs give the
    same weight and latent gradients run after run (see tests/test_gpu_backward.py::test_backward_is_reproducible)."""
    cfg = make_cfg("rel_pos", D=64, H=2, C=7, O=2, freq=(0.3, 0.6))
    prm = R.init_params(5, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 2, 87, 11, 6)
    w = np.random.default_rng(7).standard_normal((2, 87, cfg["num_out"]))
    first = None
    for it in range(8):
        junk = [torch.randn(int(n), device=cuda) * 10 for n in np.random.default_rng(it).integers(1 << 10, 1 << 21, 8)]
        del junk
        res = hip(cuda, build_nef(cfg, "bf16"), prm, x, p, a, s, w)
        flat = np.concatenate([np.asarray(v, dtype=np.float64).ravel() for v in _leaves_of(res)])
        if first is None:
            first = flat
            continue
        d = np.linalg.norm(flat - first) / np.linalg.norm(first)
        assert d < 1e-5, (it, d)


def _leaves_of(obj):
    if isinstance(obj, dict):
        for k in sorted(obj):
            yield from _leaves_of(obj[k])
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _leaves_of(v)
    else:
        yield obj.detach().cpu().numpy() if hasattr(obj, "detach") else obj


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_native_backward_matches_the_composed_path(cuda, precision, monkeypatch):
    """enf_backward_all (tail, prologue and fold gradients inside the library) against the older composition of differentiable
    device ops around the pair kernels: same 46 weight gradients and latent gradients (f32: rounding of different summation
    orders; bf16: the tail's layer inputs are kept in fp32 by the library and rounded to bf16 by neither path's products)."""
    from enf_pde_amd.enf.models import _train
    cfg = make_cfg("rel_pos_periodic", D=128, H=2, C=16, O=3, freq=(0.5, 1.0))
    prm = R.init_params(9, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 3, 90, 11, 10)
    w = np.random.default_rng(11).standard_normal((3, 90, 3))
    res = {}
    for native in (True, False):
        monkeypatch.setattr(_train, "NATIVE_BACKWARD", native)
        res[native] = hip(cuda, build_nef(cfg, precision), prm, x, p, a, s, w)
    tol = 2e-5 if precision == "f32" else 2e-2
    assert np.abs(res[True][0] - res[False][0]).max() <= tol * np.abs(res[False][0]).max()
    gmax = max(np.linalg.norm(g) for g in res[False][1])
    for path, g1, g0 in zip(TENSOR_PATHS, res[True][1], res[False][1]):
        n0 = np.linalg.norm(g0)
        err = np.linalg.norm(g1 - g0) / (n0 if n0 > 1e-6 * gmax else gmax)
        assert err < 10 * tol, ("/".join(path), err)
    for k in (2, 3, 4):
        assert np.linalg.norm(res[True][k] - res[False][k]) <= 10 * tol * max(np.linalg.norm(res[False][k]), 1e-12)
