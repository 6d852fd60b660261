"""Dev diagnostic (round 2): the start-up race of enf_pair_bwd_kernel<64, 2, bf16, *, unfolded>.
One process per library variant (ENF_HIP_LIB=variants/libenf_<name>.so): sweep case 3 of tests/test_gpu_backward.py
(ponita, D=64, H=2, B=1, N=87, Z=11, bf16, unfolded backward) repeated, every result compared with the first one; an arm
without and an arm with junk written into freed allocator blocks between the runs.  The read-back variant
(-DENF_DIAG_READBACK) also reports how many 16-byte words of ring slot 0 / the LDS constants differed from the blob right
after first_stage, and the start-up stamps."""
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
its = int(sys.argv[1]) if len(sys.argv) > 1 else 25
c = dict(inv="ponita", D=64, H=2, B=1, N=87, Z=11, prec="bf16", C=7, O=2)
cfg = T.make_cfg(c["inv"], D=c["D"], H=c["H"], C=c["C"], O=c["O"], freq=(0.3, 0.6))
prm = T.R.init_params(5, cfg, jitter=0.1)
x, p, a, s = T.make_inputs(cfg, c["B"], c["N"], c["Z"], 6)
w = np.random.default_rng(7).standard_normal((c["B"], c["N"], cfg["num_out"]))
has_rb = hasattr(lib, "enf_debug_read_diag")
has_tr = hasattr(lib, "enf_debug_read_trace")
has_post = hasattr(lib, "enf_debug_read_post")
first = None
first_tr = None
first_post = None


def read_post():
    buf = (ctypes.c_float * (8 * 8 * (16 * 64 + 16)))()
    assert lib.enf_debug_read_post(buf) == 0
    return np.array(buf, dtype=np.float32).reshape(8, 8, 16 * 64 + 16)



def read_trace():
    buf = (ctypes.c_float * (8 * 8 * 2 * 24 * 64))()
    assert lib.enf_debug_read_trace(buf) == 0
    return np.array(buf, dtype=np.float32).reshape(8, 8, 2, 24, 64)


SLOTS = ["inv0", "win", "pz0", "wcoef", "zvsum", "Eq", "a1.0", "a1.n", "logit", "att", "lacc0_before", "qflip.af0[0]", "dl00", "relu(af0[0])", "qflip.af0[3]", "dl03",
         "a5.0", "a5.n", "datt", "dlogit", "dnh.0", "upart00", "dinv0", "lacc0"]
for arm in ("plain", "junk"):
    worst, nbad = {}, 0
    for it in range(its):
        if arm == "junk":
            junk = [torch.randn(int(n), device=cuda) * 10 for n in np.random.default_rng(it).integers(1 << 10, 1 << 22, 12)]
            del junk
        nef = T.build_nef(cfg, c["prec"])
        res = T.hip_grads(cuda, nef, prm, x, p, a, s, w)
        tr = read_trace() if has_tr else None
        post = read_post() if has_post else None
        if has_post and (post[:, :, 16 * 64 + 14:] != 0).any():
            print(f"    run {it}: LDS constants / zv mismatches at kernel end [wg][wave] (consts, zv):",
                  [(wg, wv, post[wg, wv, -2], post[wg, wv, -1]) for wg in range(8) for wv in range(8) if post[wg, wv, -2:].any()], flush=True)
        if has_post:
            sc = post[:, :, 16 * 64:]
            print(f"    run {it}: dp[lat 4..7, 0] = {res[1][0, 4:8, 0]}", flush=True)
            print(f"      sum_y dpose0 (x=0) waves 0..7 = {sc[0::2, :, 2].sum(0)}", flush=True)
            print(f"      dC0 [wg][wave]:\n{np.array2string(sc[:, :, 0], precision=5, max_line_width=200)}", flush=True)
            if it == 0:
                print(f"      bz/bzc/tiles/pz0/wcoef of wg 1: {sc[1, :, 8:13].tolist()}", flush=True)
        if hasattr(lib, "enf_debug_read_phase"):
            ph = (ctypes.c_uint * (64 * 8 * 4))()
            assert lib.enf_debug_read_phase(ph) == 0
            ph = np.array(ph).reshape(64, 8, 4)
            if ph[:, :, 0].any():
                print(f"    run {it}: BARRIER PHASE VIOLATIONS [wg, wave, count, at stage k, offender wave, its phase]:",
                      [(wg, wv, *ph[wg, wv].tolist()) for wg in range(64) for wv in range(8) if ph[wg, wv, 0]][:12], flush=True)
        nan = [n for n, r in zip(("out", "dp", "da", "dsigma"), res) if not np.isfinite(r).all()]
        if nan:
            print(f"    run {it}: NaN/Inf in {nan}; dp rows with NaN: {np.nonzero(~np.isfinite(res[1]).all(-1))[1].tolist()}, "
                  f"da rows: {np.nonzero(~np.isfinite(res[2]).all(-1))[1].tolist()}", flush=True)
            continue
        if first is None:
            first, first_tr, first_post = res, tr, post
            continue
        if has_post and nbad < 6:
            d = post != first_post
            if d.any():
                print(f"    run {it}: epilogue state differs (wg = blockIdx.y*2 + blockIdx.x; x=1 waves 3..7 inactive):", flush=True)
                for wg in range(8):
                    for wv in range(8):
                        if d[wg, wv].any():
                            sc = np.nonzero(d[wg, wv, 16 * 64:16 * 64 + 7])[0]
                            la = d[wg, wv, :16 * 64].reshape(16, 64)
                            slots = {int(k): int(la[k].sum()) for k in range(16) if la[k].any()}
                            names = ["dC0", "dC1", "dpose0", "dpose1", "dpose2", "dpose3", "dwc"]
                            print(f"      wg {wg} (x={wg % 2}, y={wg // 2}) wave {wv}: scalars {[names[i] for i in sc]} "
                                  f"{[(float(first_post[wg, wv, 16 * 64 + i]), float(post[wg, wv, 16 * 64 + i])) for i in sc[:2]]}; "
                                  f"lacc slots(lanes) {slots}", flush=True)
        if has_tr and nbad < 4:
            diff = (tr != first_tr) & ~(np.isnan(tr) & np.isnan(first_tr))
            if diff.any():
                print(f"    run {it}: trace differs; first differing slot per (wg, wave, tile):", flush=True)
                for wg in range(8):
                    for wv in range(8):
                        for ti in range(2):
                            d = diff[wg, wv, ti].any(axis=1)
                            if d.any():
                                k = int(np.argmax(d))
                                lanes = np.nonzero(diff[wg, wv, ti, k])[0]
                                print(f"      wg {wg} wave {wv} tile {ti}: first {SLOTS[k]} ({len(lanes)} lanes, e.g. lane {lanes[0]}: "
                                      f"{first_tr[wg, wv, ti, k, lanes[0]]:.6g} -> {tr[wg, wv, ti, k, lanes[0]]:.6g}); all: "
                                      f"{[SLOTS[j] for j in np.nonzero(d)[0]]}", flush=True)
        dev = 0.0
        for name, r0, r1 in zip(("out", "dp", "da", "dsigma"), first, res):
            d = np.linalg.norm(r1 - r0) / max(np.linalg.norm(r0), 1e-30)
            worst[name] = max(worst.get(name, 0.0), d)
            dev = max(dev, d)
        nbad += dev > 1e-4
        if dev > 1e-4 and nbad <= 6:          # which latents (waves) deviate: |d da| per latent relative to the tensor's norm
            for name, r0, r1 in zip(("dp", "da", "dsigma"), first[1:], res[1:]):
                per = np.linalg.norm((r1 - r0).reshape(-1, r0.shape[-1]), axis=1) / max(np.linalg.norm(r0), 1e-30)
                print(f"    run {it} {name} per latent:", " ".join(f"{e:.0e}" for e in per), flush=True)
    print(f"[{os.environ.get('ENF_HIP_LIB', 'default')} sync={os.environ.get('ENF_DIAG_SYNC_K3', '0')}] {arm}: "
          f"{nbad}/{its} runs deviate > 1e-4; worst {({k: f'{e:.1e}' for k, e in worst.items()})}", flush=True)
    if hasattr(lib, "enf_debug_read_kcache"):
        kc = (ctypes.c_uint * 16)()
        assert lib.enf_debug_read_kcache(kc) == 0
        print(f"  scalar-cache check: mismatches per wave {list(kc)[:8]} of {kc[8]} wave starts (cumulative)", flush=True)
    if has_rb:
        bad = (ctypes.c_uint * (64 * 8 * 12))()
        t = (ctypes.c_ulonglong * (64 * 8 * 4))()
        assert lib.enf_debug_read_diag(bad, t) == 0
        bad = np.array(bad).reshape(64, 8, 12)
        t = np.array(t, dtype=np.uint64).reshape(64, 8, 4).astype(np.int64)
        print("  launches seen per (wg 0..7, wave 0):", bad[:8, 0, 11].tolist())
        print("  ring-slot-0 mismatching 16-B words [reader wg][reader wave] -> per piece:")
        for wg in range(8):
            for wv in range(8):
                if bad[wg, wv, :9].any():
                    print(f"    wg {wg} wave {wv}: pieces {bad[wg, wv, :8].tolist()} consts {bad[wg, wv, 8]}")
        print("  total mismatches:", int(bad[:, :, :9].sum()))
        # stamps of the LAST launch: per wave relative to the workgroup's first entry
        for wg in range(2):
            t0 = t[wg, :, 0].min()
            print(f"  wg {wg} stamps (entry, issued, vmcnt0, after barrier+reads) per wave, cycles from first entry:")
            for wv in range(8):
                print("    wave", wv, (t[wg, wv] - t0).tolist())
