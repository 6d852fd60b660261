"""Same-box A/B of library builds on the bench's per-kernel legs:  python scripts/ab_kernels.py variants/libenf_A.so ...
('-' = the in-tree library), 2 rounds; prints the launch time of K2 (decode / fit shape) and K3 (fit shape) and the step time."""
import json, os, subprocess, sys
libs = sys.argv[1:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(int(os.environ.get("AB_ROUNDS", 2))):
    for lib in libs:
        env = dict(os.environ)
        if lib != "-":
            env["ENF_HIP_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-meta", "--no-ode", "--kernel-iters", "60", "--steps", "40", "--events-steps", "0", "--no-accuracy"], env=env, capture_output=True, text=True)
        try:
            d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            k = d["roofline_kernels"]
            print(f"{lib:36s} step {d['ms_per_step']:.4f}  fit {d['split']['ms_fit']:.4f}  K2dec {k['decode_fwd']['launch_ms']:.4f}  "
                  f"K2fit {k['fit_fwd']['launch_ms']:.4f}  K3fit {k['fit_bwd']['launch_ms']:.4f}  loss {d['final_fit_loss']}", flush=True)
        except Exception as e:
            print(lib, "FAILED", e, r.stderr[-500:], flush=True)
