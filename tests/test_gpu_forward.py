"""Forward parity of the HIP path (through the C-ABI) against the fp64 numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import enf_ref_np as R
from tests.helpers import make_cfg, make_inputs, build_nef

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pair_variant")]

# tolerances on max|err| / max|ref| : fp32 mode = exact-fp32 MFMA chain; bf16 mode = bf16 operands
TOL = {"f32": 2e-5, "bf16": 3e-2}


def run_case(cuda, cfg, B, N, Z, precision, seed=0, jitter=0.1, broadcast_x=False):
    prm = R.init_params(seed, cfg, jitter=jitter)
    x, p, a, s = make_inputs(cfg, B, N, Z, seed + 1)
    if broadcast_x:
        x = np.broadcast_to(x[:1], x.shape).copy()
    ref = R.nef_apply(prm, cfg, x, p, a, s)
    nef = build_nef(cfg, precision)
    params = nef.load_params(prm, device=cuda)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    xt = t(x[0])[None].expand(B, -1, -1) if broadcast_x else t(x)
    out = nef.apply(params, xt, t(p), t(a), t(s))
    torch.cuda.synchronize()
    out = out.cpu().numpy().astype(np.float64)
    assert np.isfinite(out).all()
    err = np.abs(out - ref).max() / max(np.abs(ref).max(), 1e-6)
    mse = ((out - ref) ** 2).mean()
    return err, mse


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("invariant", ["rel_pos_periodic", "latitude_periodic", "polar_periodic", "ponita", "abs_pos",
                                       "rel_pos", "norm_rel_pos"])
def test_forward_invariants(cuda, invariant, precision):
    cfg = make_cfg(invariant, D=128, H=2, C=16, O=3, freq=(0.5, 1.0))
    err, mse = run_case(cuda, cfg, B=2, N=70, Z=9, precision=precision)
    assert err < TOL[precision], (invariant, precision, err)
    assert mse < 1e-5


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("D,H,C,O,Z,N", [(128, 2, 16, 1, 64, 512), (64, 2, 16, 1, 16, 100), (128, 1, 32, 3, 18, 33),
                                         (64, 1, 8, 2, 4, 32), (128, 2, 16, 1, 3, 40)])
def test_forward_shapes(cuda, D, H, C, O, Z, N, precision):
    cfg = make_cfg("rel_pos_periodic", D=D, H=H, C=C, O=O)
    err, mse = run_case(cuda, cfg, B=3, N=N, Z=Z, precision=precision, seed=D + Z)
    assert err < TOL[precision], err
    assert mse < 1e-5


def test_forward_broadcast_grid(cuda):
    cfg = make_cfg("rel_pos_periodic")
    err, _ = run_case(cuda, cfg, B=4, N=200, Z=16, precision="f32", broadcast_x=True)
    assert err < TOL["f32"]


def test_forward_no_window(cuda):
    cfg = make_cfg("rel_pos", use_window=False, freq=(0.5, 0.5))
    err, _ = run_case(cuda, cfg, B=2, N=64, Z=8, precision="f32")
    assert err < TOL["f32"]


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_variants_agree_at_full_size(cuda, precision):
    """BASELINE.json's full decode shape (16 signals x 64^2 queries x 64 latents) is too large for the oracle:
    the two independent forward variants (latent-split, z-fold with per-latent folded matrices) must agree."""
    from enf_pde_amd import _lib
    cfg = make_cfg("rel_pos_periodic", D=128, H=2, C=16, O=1)
    prm = R.init_params(11, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 16, 4096, 64, 12)
    nef = build_nef(cfg, precision)
    params = nef.load_params(prm, device=cuda)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    outs = []
    for mode in ("latent_split", "z_fold", "z_fold_zsplit"):
        nef.pair_variants = (mode, "auto")
        outs.append(nef.apply(params, t(x), t(p), t(a), t(s)))
    ref = outs[0]
    for got in outs[1:]:
        err = float((got - ref).abs().max() / ref.abs().max())
        assert torch.isfinite(got).all() and err < (2e-5 if precision == "f32" else 3e-2), err


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_config3_decode_shape_picks_the_split_z_fold_and_agrees(cuda, precision):
    """BASELINE config 3's decode (4 signals x 96 x 48 sphere grid, 128 latents: 144 workgroups of 128 queries on 256 CUs): AUTO now
    resolves to the z-fold kernel with the latents split over two workgroups per query tile (ENF_VARIANT_ZFOLD_ZSPLIT, 288
    workgroups); it must agree with the latent-split kernel there, and the latent gradients through it with the ones through
    the latent-split forward (the backward reads the forward's lse)."""
    import ctypes
    from enf_pde_amd import _lib
    cfg = make_cfg("latitude_periodic", D=128, H=2, C=32, O=3, freq=(0.05, 0.2))
    prm = R.init_params(31, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 4, 96 * 48, 128, 32)
    nef = build_nef(cfg, precision)
    nef.pair_variants = ("auto", "auto")
    assert _lib.load().enf_pair_variant(ctypes.byref(nef._desc(4, 96 * 48, 128)), 0) == 3
    params = nef.load_params(prm, device=cuda)
    t = lambda v, g=False: torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=g)
    res = []
    for mode in ("latent_split", "auto"):
        nef.pair_variants = (mode, "auto")
        tp, ta = t(p, True), t(a, True)
        out = nef.apply(params, t(x), tp, ta, t(s))
        out.square().sum().backward()
        res.append((out.detach(), tp.grad, ta.grad))
    tol = 2e-5 if precision == "f32" else 3e-2
    for r, g in zip(res[0], res[1]):
        assert torch.isfinite(g).all() and float((g - r).abs().max() / r.abs().max()) < (tol if r is res[0][0] else 20 * tol)


def test_full_size_invariances(cuda):
    """Size-independent properties at BASELINE.json's full shape (oracle too slow there): the output is invariant to
    a permutation of the latent set (softmax over a set, ECA:141) and, for rel_pos_periodic, to shifting any latent
    or query coordinate by the period 2 (INV/rel_pos_periodic:35-60); both are what the reference eyeballs in
    _base_pde_trainer.py:731-757."""
    cfg = make_cfg("rel_pos_periodic", D=128, H=2, C=16, O=1)
    prm = R.init_params(21, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 16, 4096, 64, 22)
    nef = build_nef(cfg, "f32")
    params = nef.load_params(prm, device=cuda)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    tx, tp, ta, ts = t(x), t(p), t(a), t(s)
    base = nef.apply(params, tx, tp, ta, ts)
    scale = float(base.abs().max())
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(0)).to(cuda)
    out = nef.apply(params, tx, tp[:, perm], ta[:, perm], ts[:, perm])
    assert float((out - base).abs().max()) < 2e-5 * scale          # summation order only
    shift_p = tp.clone(); shift_p[:, ::2, 0] += 2.0; shift_p[:, 1::3, 1] -= 2.0
    shift_x = tx.clone(); shift_x[:, ::5, 1] += 2.0
    out = nef.apply(params, shift_x, shift_p, ta, ts)
    assert float((out - base).abs().max()) < 5e-4 * scale          # cos/sin(pi (D +- 2)) in fp32


@pytest.mark.parametrize("B,N,Z", [(1, 1, 1), (1, 17, 1), (2, 1, 5), (1, 129, 2)])
def test_forward_edge_shapes(cuda, B, N, Z):
    """Degenerate sizes: a single query, a single latent (softmax weight 1), one more query than a tile / a group."""
    cfg = make_cfg("rel_pos_periodic", D=64, H=2, C=8, O=2)
    err, _ = run_case(cuda, cfg, B=B, N=N, Z=Z, precision="f32", seed=B + N + Z)
    assert err < TOL["f32"]


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("invariant", ["polar_periodic", "latitude_periodic"])
def test_forward_config3_sphere(cuda, invariant, precision):
    """SURVEY.md 8d config 3 (shallow water / diffusion on the sphere): Z=128 latents, C=32, O=3."""
    cfg = make_cfg(invariant, D=128, H=2, C=32, O=3, freq=(0.2, 0.4))
    err, mse = run_case(cuda, cfg, B=2, N=700, Z=128, precision=precision, seed=31)
    assert err < TOL[precision] and mse < 1e-5


def test_latent_table_is_reused_only_for_the_same_latents(cuda, pair_variant):
    """A forward on the same latent tensors and weights as the last call on the workspace (a decode right after the fit's
    final-loss forward) skips the prologue kernel; any change of the latents -- another tensor, or the same one modified in
    place -- or of the weights must not.  (A query's value does not depend on how many other queries the call carries, bit for bit --
    except in the split z-fold variant, whose runs of latent steps, and with them the order of a tile's partial sums, follow from the shape.)"""
    import ctypes
    from enf_pde_amd import _lib
    cfg = make_cfg("rel_pos_periodic", D=128, H=2, C=16, O=1)
    prm = R.init_params(21, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 3, 300, 16, 22)
    nef = build_nef(cfg, "f32")
    params = nef.load_params(prm, device=cuda)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    xs, xl = t(x[:, :40]), t(x)
    tp, ta, ts = t(p), t(a), t(s)
    calls = []
    lib = _lib.load()
    orig = lib.enf_forward_stages

    class Spy:                                   # records the `stages` argument of every forward
        def __call__(self, *args):
            calls.append(int(args[-2]))
            return orig(*args)
    spy = Spy()
    try:
        lib.enf_forward_stages = spy
        with torch.no_grad():
            nef.apply(params, xl, tp, ta, ts)                 # sizes the cached workspace for the larger call
            calls.clear()
            small = nef.apply(params, xs, tp, ta, ts)         # "final-loss forward" on few points
            full = nef.apply(params, xl, tp, ta, ts)          # "decode": same latents, same weights
            assert [c & 1 for c in calls] == [0, 0]           # both found the table of the first call
            fresh = build_nef(cfg, "f32")
            ref = fresh.apply(fresh.load_params(prm, device=cuda), xl, tp, ta, ts)
            assert torch.equal(full, ref)
            assert torch.equal(small, ref[:, :40]) if pair_variant != "z_fold_zsplit" else torch.allclose(small, ref[:, :40], rtol=0, atol=5e-6)
            calls.clear()
            ta.add_(0.05)                                     # same tensor, new contents
            moved = nef.apply(params, xl, tp, ta, ts)
            assert calls[-1] & 1 == 1 and not torch.equal(moved, full)
            assert torch.equal(moved, fresh.apply(fresh.load_params(prm, device=cuda), xl, tp, ta, ts))
            calls.clear()
            nef.apply(params, xl, tp.clone(), ta, ts)         # another tensor with the same values: no reuse either
            assert calls[-1] & 1 == 1
            prm2 = R.init_params(22, cfg, jitter=0.1)
            calls.clear()
            other = nef.apply(nef.load_params(prm2, device=cuda), xl, tp.clone(), ta, ts)     # other weights
            assert calls[-1] & 1 == 1 and not torch.equal(other, moved)
    finally:
        lib.enf_forward_stages = orig


@pytest.mark.parametrize("B,N,Z", [(6, 4096, 64), (3, 300, 16)])
def test_half_precision_hand_off_to_the_tail_changes_no_bit(cuda, B, N, Z, pair_variant):
    """ENF_STAGE_YBAR_HALF (include/enf_hip.h): a forward that no backward follows hands ybar to the tail as bf16 through the workspace.
    The tail rounds ybar to bf16 for its first matrix product anyway, so `out` must equal the fp32 hand-off's bit for bit (bf16 mode; both
    kernels that write ybar themselves: the z-fold with >= 192 query tiles and the latent-split one; the split z-fold ignores the flag)."""
    import ctypes
    from enf_pde_amd import _lib
    cfg = make_cfg("rel_pos_periodic", D=128, H=2, C=16, O=1)
    prm = R.init_params(31, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, B, N, Z, 32)
    nef = build_nef(cfg, "bf16")
    params = nef.load_params(prm, device=cuda)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    tx, tp, ta, ts = t(x), t(p), t(a), t(s)
    calls = []
    lib = _lib.load()
    orig = lib.enf_forward_stages

    class Spy:
        def __call__(self, *args):
            calls.append(int(args[-2]))
            return orig(*args)
    try:
        lib.enf_forward_stages = Spy()
        with torch.no_grad():
            half = nef.apply(params, tx, tp, ta, ts)
        full = nef.apply(params, tx, tp.clone().requires_grad_(True), ta, ts).detach()
    finally:
        lib.enf_forward_stages = orig
    assert calls[0] & 64 and not calls[1] & 64
    assert torch.isfinite(half).all() and torch.equal(half, full)
    ref = R.nef_apply(prm, cfg, x[:1, :64], p[:1], a[:1], s[:1])
    assert np.abs(half[:1, :64].cpu().numpy() - ref).max() < 3e-2 * np.abs(ref).max()
