"""The C-ABI keeps no settings and no cross-call state beyond what its arguments name ("re-entrant across
streams; no global state"): two models on two streams, their forwards and backwards interleaved, give what each gives alone.

The library's only host-side bookkeeping is the side-stream work a forward leaves pending for its backward
(ENF_STAGE_PREPARE_BWD -> ENF_BWD_REUSE_PREPARED); it is recorded against the workspace it was prepared in."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import enf_ref_np as R
from tests.helpers import make_cfg, make_inputs, build_nef

pytestmark = pytest.mark.gpu

FWD_ALL, TAIL_SAVE, PREPARE = 15, 16, 32
REUSE_PROLOGUE, REUSE_TAIL, REUSE_PREPARED = 1, 2, 4


class _Job:
    """One model + inputs + its own stream and workspace; forward / backward as two separate C-ABI calls."""

    def __init__(self, cuda, cfg, B, N, Z, seed, precision="f32"):
        from enf_pde_amd import _lib
        self.lib, self._lib = _lib.load(), _lib
        self.nef = build_nef(cfg, precision)
        self.nef.pair_variants = ("auto", "z_fold")           # the backward whose per-latent matrices run on the side stream
        prm = R.init_params(seed, cfg, jitter=0.1)
        x, p, a, s = make_inputs(cfg, B, N, Z, seed + 1)
        t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda).contiguous()
        self.x, self.p, self.a, self.s = t(x), t(p), t(a), t(s)
        self.dout = t(np.random.default_rng(seed + 2).standard_normal((B, N, cfg["num_out"])))
        self.stream = torch.cuda.Stream(device=cuda)
        self.packed = self.nef.pack(self.nef.load_params(prm, device=cuda))
        torch.cuda.synchronize()
        self.desc = self.nef._desc(B, N, Z)
        HD = self.nef._Hp * self.nef._Dp
        self.out = torch.empty(B, N, cfg["num_out"], device=cuda)
        self.ybar, self.lse = torch.empty(B, N, HD, device=cuda), torch.empty(B, N, self.nef._Hp, device=cuda)
        self.ws = torch.empty(int(self.lib.enf_workspace_bytes(ctypes.byref(self.desc))), device=cuda, dtype=torch.uint8)
        self.grads = [torch.empty_like(self.p), torch.empty_like(self.a), torch.empty_like(self.s)]

    def _st(self):
        return ctypes.c_void_p(self.stream.cuda_stream)

    def forward(self, stages):
        P = lambda v: ctypes.c_void_p(v.data_ptr())
        self._lib.check(self.lib.enf_forward_stages(ctypes.byref(self.desc), P(self.x), self.x.shape[1] * self.x.shape[2], P(self.p),
                                                    P(self.a), P(self.s), P(self.packed), P(self.out), P(self.ybar), P(self.lse),
                                                    P(self.ws), self.ws.numel(), stages, self._st()))

    def backward(self, flags):
        P = lambda v: ctypes.c_void_p(v.data_ptr())
        self._lib.check(self.lib.enf_backward_latents_ex(ctypes.byref(self.desc), P(self.x), self.x.shape[1] * self.x.shape[2],
                                                         P(self.p), P(self.a), P(self.s), P(self.packed), P(self.ybar), P(self.lse),
                                                         P(self.dout), *[P(g) for g in self.grads], P(self.ws), self.ws.numel(),
                                                         flags, self._st()))

    def result(self):
        self.stream.synchronize()
        return [self.out.clone()] + [g.clone() for g in self.grads]


def _close(got, ref):
    for g, r in zip(got, ref):
        assert torch.isfinite(g).all()
        assert float((g - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-30      # float atomics reorder the last sums


@pytest.fixture()
def jobs(cuda):
    a = _Job(cuda, make_cfg("rel_pos_periodic", D=128, H=2, C=16, O=1), 4, 700, 64, 21)
    b = _Job(cuda, make_cfg("ponita", D=64, H=2, C=8, O=2, freq=(0.3, 0.6)), 3, 300, 70, 31)
    refs = []
    for j in (a, b):                   # each alone, nothing prepared
        j.forward(FWD_ALL)
        j.backward(0)
        refs.append(j.result())
    return a, b, refs


def test_two_models_two_streams_interleaved(jobs):
    a, b, (ra, rb) = jobs
    for order in ("abba", "abab"):
        a.forward(FWD_ALL | TAIL_SAVE | PREPARE)
        b.forward(FWD_ALL | TAIL_SAVE | PREPARE)
        for j in ((b, a) if order == "abba" else (a, b)):
            j.backward(REUSE_PROLOGUE | REUSE_TAIL | REUSE_PREPARED)
        _close(a.result(), ra)
        _close(b.result(), rb)


def test_prepared_work_belongs_to_its_workspace(jobs):
    """A forward of A that prepared its backward must not satisfy ENF_BWD_REUSE_PREPARED of B (whose forward prepared nothing)."""
    a, b, (ra, rb) = jobs
    for g in b.grads:
        g.fill_(float("nan"))
    a.forward(FWD_ALL | TAIL_SAVE | PREPARE)
    b.forward(FWD_ALL | TAIL_SAVE)                                        # no PREPARE
    b.backward(REUSE_PROLOGUE | REUSE_TAIL | REUSE_PREPARED)              # must do the latent-only work itself
    _close(b.result(), rb)
    a.backward(REUSE_PROLOGUE | REUSE_TAIL | REUSE_PREPARED)              # A's pending work is still A's
    _close(a.result(), ra)
    # pending work nobody consumed is joined by the next call on that workspace, whatever it is
    a.forward(FWD_ALL | TAIL_SAVE | PREPARE)
    a.forward(FWD_ALL)
    a.backward(0)
    _close(a.result(), ra)


def test_relu_masks_and_variants_are_per_call(cuda):
    """Masks and kernel variants travel in the descriptor: a model inside relu_masks() / with a forced variant does not change
    what another model's calls do."""
    cfg = make_cfg("rel_pos_periodic", D=64, H=2, C=8, O=1)
    prm = R.init_params(3, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 2, 200, 9, 4)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    m1, m2 = build_nef(cfg, "f32"), build_nef(cfg, "f32")
    P1, P2 = m1.load_params(prm, device=cuda), m2.load_params(prm, device=cuda)
    base = m2.apply(P2, t(x), t(p), t(a), t(s))
    buf = m1.relu_mask_buffer(2, 200, 9, cuda)
    buf.zero_()
    m1.pair_variants = ("z_fold", "auto")
    with torch.no_grad(), m1.relu_masks(buf, "write", 2):
        o1 = m1.apply(P1, t(x), t(p), t(a), t(s))
        o2 = m2.apply(P2, t(x), t(p), t(a), t(s))          # another model, inside the block: no masks, its own variant
    assert torch.equal(o2, base) and buf.abs().sum() > 0
    assert float((o1 - base).abs().max()) < 2e-5 * float(base.abs().max())
    from enf_pde_amd import _lib
    assert _lib.load().enf_pair_variant(ctypes.byref(m1._desc(2, 200, 9)), 0) == 2
    assert _lib.load().enf_pair_variant(ctypes.byref(m2._desc(2, 200, 9)), 0) == 1
    # replaying all-zero masks (every relu unit off) must change m1's output and nobody else's
    zero = torch.zeros_like(buf)
    with torch.no_grad(), m1.relu_masks(zero, "read", 2):
        z1 = m1.apply(P1, t(x), t(p), t(a), t(s))
        z2 = m2.apply(P2, t(x), t(p), t(a), t(s))
    assert torch.equal(z2, base) and float((z1 - base).abs().max()) > 1e-3 * float(base.abs().max())
    assert m1._masks is None
