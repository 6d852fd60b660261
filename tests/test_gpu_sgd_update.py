"""enf_meta_sgd_update: the inner loop's update of all latent components in one launch (pde_trainer.py:206-219)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("with_ori,window", [(False, False), (True, False), (True, True), (False, True)])
def test_update_matches_formula(cuda, with_ori, window):
    from enf_pde_amd.fitting.inner_loop import meta_sgd_update, default_meta_sgd_lrs
    g = torch.Generator().manual_seed(3)
    B, Z, C = 5, 7, 12
    n_pos, n_ori = 2, (1 if with_ori else 0)
    r = lambda *s: torch.randn(*s, generator=g).to(cuda)
    lat = {"p_pos": r(B, Z, n_pos), "a": r(B, Z, C), "gaussian_window": r(B, Z, 1)}
    lrs = default_meta_sgd_lrs(C, lr_p=0.7, lr_a=3.0, lr_window=0.2, with_ori=with_ori, device=cuda)
    lrs["a"] = lrs["a"] * (1 + 0.1 * r(C))                      # per-channel rates
    dp, da, dsig = r(B, Z, n_pos + n_ori), r(B, Z, C), r(B, Z, 1)
    grads = {"p_pos": dp[..., :n_pos], "a": da}
    if with_ori:
        lat["p_ori"] = r(B, Z, 1)
        grads["p_ori"] = dp[..., n_pos:]
    if window:
        grads["gaussian_window"] = dsig
    new = meta_sgd_update(lat, grads, lrs, B)
    assert set(new) == set(lat)
    for k in lat:
        if k in grads:
            want = lat[k] - lrs[k] * (grads[k] * B)
            assert new[k] is not lat[k]
            torch.testing.assert_close(new[k], want, rtol=1e-6, atol=1e-6)
        else:
            assert new[k] is lat[k]                             # zeroed update (pde_trainer.py:209-212): passed through


def test_update_rejects_bad_segments(cuda):
    from enf_pde_amd import _lib
    lib = _lib.load()
    x = torch.zeros(4, 6, device=cuda)
    lr = torch.ones(3, device=cuda)
    seg = (_lib.EnfSgdSegment * 4)()
    seg[0] = _lib.EnfSgdSegment(x.data_ptr(), x.data_ptr(), lr.data_ptr(), x.data_ptr(), x.numel(), 6, 6, 3, 0)
    assert lib.enf_meta_sgd_update(1, seg, 1.0, None) == -6       # ENF_EDIM: lr_len is neither 1 nor width
    assert lib.enf_meta_sgd_update(0, seg, 1.0, None) == -1
    assert lib.enf_meta_sgd_update(5, seg, 1.0, None) == -1
    seg[0].lr_len, seg[0].g_stride = 1, 4
    assert lib.enf_meta_sgd_update(1, seg, 1.0, None) == -1       # rows of g overlap
    seg[0].g_stride, seg[0].x = 6, None
    assert lib.enf_meta_sgd_update(1, seg, 1.0, None) == -1


@pytest.mark.parametrize("B,Z,N,Ns,S,dx,O,ori", [(16, 64, 4096, 512, 3, 2, 1, False), (3, 7, 100, 37, 2, 3, 3, True), (1, 1, 5, 5, 0, 2, 1, False)])
def test_fit_inputs_in_one_launch_equal_the_framework_ops(cuda, B, Z, N, Ns, S, dx, O, ori):
    """enf_fit_inputs (inner_loop's setup: the signals' copies of the latent initialisation, the gathered coordinates / targets of all
    S + 1 point sets, zeroed loss accumulators; pde_trainer.py:157-159, 193-197) against the torch operations it replaces: bit for bit."""
    from enf_pde_amd.fitting.inner_loop import _fit_inputs
    g = torch.Generator().manual_seed(B * 1000 + N)
    lat0 = {"p_pos": torch.randn(1, Z, dx, generator=g), "a": torch.randn(1, Z, 9, generator=g), "gaussian_window": torch.rand(1, Z, 1, generator=g)}
    if ori:
        lat0["p_ori"] = torch.randn(1, Z, 1, generator=g)
    lat0 = {k: v.to(cuda) for k, v in lat0.items()}
    coords = torch.randn(N, dx, generator=g).to(cuda)
    img = torch.randn(B, N, O, generator=g).to(cuda)
    masks = torch.stack([torch.randperm(N, generator=g)[:Ns] for _ in range(S + 1)], dim=1).to(cuda)
    got = _fit_inputs(lat0, coords, img, masks)
    assert got is not None
    lat, xs, ys, losses = got
    mt = masks.t().contiguous()
    for k, v in lat0.items():
        assert torch.equal(lat[k], v.repeat_interleave(B, dim=0))
    assert torch.equal(xs, coords[mt]) and torch.equal(ys, img[:, mt].transpose(0, 1).contiguous())
    assert losses.shape == (S + 1,) and float(losses.abs().sum()) == 0.0
    # arguments the kernel does not take fall back to the framework path
    assert _fit_inputs(lat0, coords.double(), img, masks) is None and _fit_inputs(lat0, coords, img, masks.int()) is None

