"""Dev probe for rocprofv3 --pmc: a few launches of the pair kernels at the bench shapes."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
nef, params, lat0, lrs, masks = bench.build(dev, "bf16")
coords, img = bench.synth_fields(bench.B_PER_GPU, 100, dev)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if which == "fwd":
    print(bench.roofline_leg(nef, params, coords, dev, iters=3))
else:
    B = bench.B_PER_GPU
    lat = {k: v.repeat_interleave(B, 0).clone() for k, v in lat0.items()}
    xs = coords[masks[:, 0]][None].expand(B, -1, -1)
    for _ in range(3):
        l = {k: v.detach().requires_grad_(True) for k, v in lat.items()}
        nef.apply(params, xs, l["p_pos"], l["a"], l["gaussian_window"]).sum().backward()
    torch.cuda.synchronize()
