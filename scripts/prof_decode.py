"""Dev probe for rocprofv3 --pmc: three decodes (nef.apply under no_grad at the bench's decode shape), i.e. the forward pair kernel and
the tail with the hand-off a decode really uses (ENF_STAGE_YBAR_HALF).  YBAR_FULL=1: the same with a gradient-requiring input, so that
the fp32 hand-off runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
nef, params, lat0, lrs, masks = bench.build(dev, "bf16")
coords, img = bench.synth_fields(bench.B_PER_GPU, 100, dev)
B = bench.B_PER_GPU
lat = {k: v.repeat_interleave(B, 0).clone() for k, v in lat0.items()}
x = coords[None].expand(B, -1, -1)
full = os.environ.get("YBAR_FULL") == "1"
for _ in range(3):
    if full:
        out = nef.apply(params, x, lat["p_pos"].clone().requires_grad_(True), lat["a"], lat["gaussian_window"])
    else:
        with torch.no_grad():
            out = nef.apply(params, x, lat["p_pos"], lat["a"], lat["gaussian_window"])
torch.cuda.synchronize()
print(float(out.float().abs().mean()))
