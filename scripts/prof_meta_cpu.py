"""Where the host time of an outer (meta) step goes: cProfile over a few MetaSGDPDETrainer.nef_train_step calls at the bench
shape (the step is launch-bound: ~16 ms of kernels in a ~36 ms step).  python scripts/prof_meta_cpu.py [steps]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from enf_pde_amd.fitting import MetaSGDPDETrainer
dev = torch.device("cuda:0")
c = bench.CONFIGS[2]
m = bench.build_config(c, dev, "bf16")
coords = bench.coords_of(c, c["grid"], dev)
img = bench.synth_targets(c, coords, c["B"], 100, dev)
tr = MetaSGDPDETrainer(m.cfg, m.nef, m.ad, coords, seed=0)
state = tr.init_train_state(nef_params=m.params)
batch = img.reshape(img.shape[0], *c["grid"][::-1], c["O"])
for _ in range(3):
    loss, state = tr.nef_train_step(state, batch)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    loss, state = tr.nef_train_step(state, batch)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
pr.disable()
t_all = time.perf_counter() - t0
print(f"{n} steps: host returns after {t_host / n * 1e3:.1f} ms per step, GPU done after {t_all / n * 1e3:.1f} ms per step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
