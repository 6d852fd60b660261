import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
nef, params, lat0, lrs, masks = bench.build(dev, "bf16")
coords, img = bench.synth_fields(bench.B_PER_GPU, 100, dev)
B = bench.B_PER_GPU
lat = {k: v.repeat_interleave(B, 0).clone() for k, v in lat0.items()}
xs = coords[masks[:, 0]][None].expand(B, -1, -1)
def fb():
    l = {k: v.detach().requires_grad_(True) for k, v in lat.items()}
    nef.apply(params, xs, l["p_pos"], l["a"], l["gaussian_window"]).sum().backward()
for _ in range(5): fb()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): fb()
e1.record(); torch.cuda.synchronize()
print(sys.argv[1] if len(sys.argv) > 1 else "", "fit fwd+bwd ms", round(e0.elapsed_time(e1) / 40, 4))
