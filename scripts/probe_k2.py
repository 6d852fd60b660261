import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
nef, params, lat0, lrs, masks = bench.build(dev, "bf16")
coords, img = bench.synth_fields(bench.B_PER_GPU, 100, dev)
r = bench.roofline_leg(nef, params, coords, dev, iters=40)
print(sys.argv[1] if len(sys.argv) > 1 else "", "K2 launch_ms", r["launch_ms"], "frac", r["frac"])
