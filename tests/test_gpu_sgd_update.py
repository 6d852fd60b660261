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
