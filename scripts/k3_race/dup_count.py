"""In-launch duplicate-wave check (tests/test_gpu_backward.py::test_duplicate_waves_agree) as a counter, for investigation builds:
   python scripts/k3_race/dup_count.py [ITERS] LIB [LIB ...]      (each LIB: a -DENF_TEST_HOOKS=1 variant from scripts/build_variant.sh)
For every library (a fresh process each) and case: launches, workgroups checked, waves that differ from wave 1, per wave index."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import ctypes, numpy as np, torch
    from oracle import enf_ref_np as R
    from tests.helpers import make_cfg, make_inputs, build_nef
    from tests.test_gpu_backward import hip_grads
    from enf_pde_amd import _lib
    iters = int(sys.argv[2])
    cuda = torch.device("cuda:0")
    tl = _lib.load_test()
    row = 16 * 64 + 16
    for (D, H, prec) in [(64, 2, "bf16"), (128, 2, "bf16"), (64, 1, "bf16"), (64, 2, "f32")]:
        cfg = make_cfg("ponita", D=D, H=H, C=7, O=2, freq=(0.3, 0.6))
        prm = R.init_params(5, cfg, jitter=0.1)
        N = 1400
        x, p, a, s = make_inputs(cfg, 1, N, 2, 6)
        w = np.random.default_rng(7).standard_normal((1, N, cfg["num_out"]))
        nef = build_nef(cfg, prec)
        nef.pair_variants = ("latent_split", "latent_split")
        per_wave = np.zeros(6, dtype=np.int64)
        wgs = 0
        with _lib.using(tl):
            for it in range(iters):
                junk = [torch.randn(int(n), device=cuda) * 10 for n in np.random.default_rng(it).integers(1 << 10, 1 << 22, 6)]
                del junk
                hip_grads(cuda, nef, prm, x, p, a, s, w)
                buf = (ctypes.c_float * (64 * 8 * row))()
                assert tl.enf_test_read_wave_sums(buf) == 0
                sums = np.array(buf, dtype=np.float32).reshape(64, 8, row)
                launched = sums[:, 0, 16 * 64 + 10] > 0
                ref = sums[launched][:, 1:2, :16 * 64 + H + 5]
                dup = sums[launched][:, 2:, :16 * 64 + H + 5]
                differ = (dup != ref).any(-1)
                per_wave += differ.sum(0)
                wgs += int(launched.sum())
        print(f"  D={D} H={H} {prec}: {iters} launches, {wgs} workgroups, differing waves 2..7: {per_wave.tolist()}", flush=True)
    sys.exit(0)
iters = sys.argv[1] if sys.argv[1].isdigit() else "20"
libs = [a for a in sys.argv[1:] if not a.isdigit()]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, ENF_HIP_LIB=os.path.abspath(lib), ENF_HIP_TEST_LIB=os.path.abspath(lib))
        print(f"== round {rnd} {lib}", flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", iters], env=env, capture_output=True, text=True)
        print(r.stdout + (r.stderr[-800:] if r.returncode else ""), flush=True)
