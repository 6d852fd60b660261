"""Same-box A/B of bench.py between library builds: python scripts/ab_bench.py variants/libenf_A.so variants/libenf_B.so ...
('-' = the in-tree library), 2 rounds, prints ms_per_step / ms_fit / ms_decode."""
import json, os, subprocess, sys
libs = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != "-":
            env["ENF_HIP_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print(f"{lib:32s} {d['ms_per_step']:.4f}  fit {d['split']['ms_fit']:.4f}  decode {d['split']['ms_decode']:.4f}", flush=True)
