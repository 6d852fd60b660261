"""Dev diagnostic (round 2): in-launch duplicate check with EPILOGUE-ONLY instrumentation (-DENF_DIAG_POST: the tile loop is
untouched).  B*Z = 2: waves 2..7 of every workgroup recompute latent 1, so their final per-wave sums must equal wave 1's."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tests.test_gpu_backward as T
from enf_pde_amd import _lib

lib = _lib.load()
cuda = torch.device("cuda:0")
lib.enf_set_zfold(0)
lib.enf_set_zfold_bwd(0)
its = int(sys.argv[1]) if len(sys.argv) > 1 else 10
D = int(os.environ.get("DUP_D", "64"))
H = int(os.environ.get("DUP_H", "2"))
prec = os.environ.get("DUP_PREC", "bf16")
cfg = T.make_cfg("ponita", D=D, H=H, C=7, O=2, freq=(0.3, 0.6))
prm = T.R.init_params(5, cfg, jitter=0.1)
Z = int(os.environ.get("DUP_Z", "2"))
N = int(os.environ.get("DUP_N", "1400"))
x, p, a, s = T.make_inputs(cfg, 1, N, Z, 6)
w = np.random.default_rng(7).standard_normal((1, N, cfg["num_out"]))
REFW = (Z - 1) % 8                      # last active wave of the last x-workgroup; the waves after it recompute its latent
NX = (Z + 7) // 8
W = 16 * 64 + 16
bad_sc = np.zeros(8, int)
bad_la = np.zeros(8, int)
cells = 0
names = ["dC0", "dC1", "dpose0", "dpose1", "dpose2", "dpose3", "dwc"]
shown = 0
for it in range(its):
    nef = T.build_nef(cfg, prec)
    T.hip_grads(cuda, nef, prm, x, p, a, s, w)
    buf = (ctypes.c_float * (64 * 8 * W))()
    assert lib.enf_debug_read_post(buf) == 0
    post = np.array(buf, dtype=np.float32).reshape(64, 8, W)
    wgs = [wg for wg in range(64) if wg % NX == NX - 1 and post[wg, 0, 16 * 64 + 10] > 0]      # last x column, launched
    post = post[wgs]
    ref = post[:, REFW:REFW + 1]
    d = np.zeros(post[:, 2:].shape, bool)
    d[:, REFW - 1:] = post[:, REFW + 1:] != ref
    cells += len(wgs) * (7 - REFW)
    nsc = H + 5
    dsc = d[:, :, 16 * 64:16 * 64 + nsc].any(-1)
    dla = d[:, :, :16 * 64].any(-1)
    bad_sc[2:] += dsc.sum(0)
    bad_la[2:] += (dla & ~dsc).sum(0)
    for wg, wv in zip(*np.nonzero(dsc | dla)):
        if shown < 8:
            shown += 1
            sl = d[wg, wv, :16 * 64].reshape(16, 64)
            print(f"  run {it} wg {wg} wave {wv + 2}: scalars differing {[names[i] if i < len(names) else i for i in np.nonzero(d[wg, wv, 16 * 64:16 * 64 + nsc])[0]]}; "
                  f"lacc slots (lanes): { {int(k): int(sl[k].sum()) for k in range(16) if sl[k].any()} }", flush=True)
print(f"[{os.environ.get('ENF_HIP_LIB', 'default')} D={D} H={H} {prec} Z={Z} N={N}] duplicate waves differing from wave 1 out of {cells}: scalars {bad_sc.sum()} per wave {bad_sc.tolist()}; "
      f"LDS accumulators only {bad_la.sum()} per wave {bad_la.tolist()}", flush=True)
