"""How long the HOST needs to enqueue one fit + decode step (python + ctypes + the runtime's launch path), against the GPU's step time:
the step is GPU-bound only while the host stays ahead.  python scripts/probe_host_step.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
c = bench.CONFIGS[bench.HEADLINE]
m = bench.build_config(c, dev, "bf16")
coords = bench.coords_of(c, c["grid"], dev)
img = bench.synth_targets(c, coords, c["B"], 100, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(5):
    bench.step(c, m, coords, coords, img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    bench.step(c, m, coords, coords, img)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms per step; total {1e3 * (t2 - t0) / n:.3f} ms per step; drain after the last enqueue {1e3 * (t2 - t1):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    bench.step(c, m, coords, coords, img)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(18)
