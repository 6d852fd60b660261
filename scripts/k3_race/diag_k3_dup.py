"""Dev diagnostic (round 2): in-launch duplicate check of enf_pair_bwd_kernel<64, 2, bf16, unfolded>.  With B*Z = 2 latents the
waves 2..7 of every workgroup are inactive and recompute latent 1 (the kernel keeps them in the barrier cadence), so in ONE
launch six waves must reproduce wave 1's intermediate values bit for bit.  Needs a -DENF_DIAG_TRACE build."""
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
SLOTS = ["F00_at_pack", "Fn7_at_pack", "F00_lazy", "Fn7_lazy", "x4", "nh00_prepack", "nhn3_prepack", "F00_after_gb", "x8", "opgf00", "Ev", "a2.0", "a2.n", "a3.0", "a3.n", "mu1", "r1",
         "v.0", "v.n", "a5.0", "a5.n", "opgf_n3", "x22", "upart00"]
cfg = T.make_cfg("ponita", D=64, H=2, C=7, O=2, freq=(0.3, 0.6))
prm = T.R.init_params(5, cfg, jitter=0.1)
x, p, a, s = T.make_inputs(cfg, 1, 1400, 2, 6)
w = np.random.default_rng(7).standard_normal((1, 1400, cfg["num_out"]))
tot = {}
waves = np.zeros(8, int)
cells = 0
for it in range(its):
    nef = T.build_nef(cfg, "bf16")
    T.hip_grads(cuda, nef, prm, x, p, a, s, w)
    buf = (ctypes.c_float * (64 * 8 * 2 * 24 * 64))()
    assert lib.enf_debug_read_trace(buf) == 0
    tr = np.array(buf, dtype=np.float32).reshape(64, 8, 2, 24, 64)
    ref = tr[:, 1:2]
    diff = (tr[:, 2:] != ref) & ~(np.isnan(tr[:, 2:]) & np.isnan(ref))          # (wg, wave-2, tile, slot, lane)
    cells += diff.shape[0] * diff.shape[1] * diff.shape[2]
    for wg, wv, ti in zip(*np.nonzero(diff.any(axis=(3, 4)))):
        d = diff[wg, wv, ti].any(axis=1)
        k = int(np.argmax(d))
        tot[SLOTS[k]] = tot.get(SLOTS[k], 0) + 1
        waves[wv + 2] += 1
        if sum(tot.values()) <= 12:
            lanes = np.nonzero(diff[wg, wv, ti, k])[0]
            print(f"  run {it} wg {wg} wave {wv + 2} tile {ti}: first {SLOTS[k]} ({len(lanes)} lanes; lane {lanes[0]}: ref {ref[wg, 0, ti, k, lanes[0]]:.7g} "
                  f"got {tr[wg, wv + 2, ti, k, lanes[0]]:.7g}); all differing: {[SLOTS[j] for j in np.nonzero(d)[0]]}", flush=True)
print(f"[{os.environ.get('ENF_HIP_LIB', 'default')}] {sum(tot.values())} of {cells} duplicate wave-tiles differ from wave 1; first differing slot: {tot}; per wave: {waves.tolist()}")
