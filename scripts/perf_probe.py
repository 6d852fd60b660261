"""Dev probe (GPU box): per-call timings of forward / backward at the bench shapes."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nef, params, lat0, lrs, masks = bench.build(dev, prec)
coords, img = bench.synth_fields(bench.B_PER_GPU, 100, dev)
B = bench.B_PER_GPU

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

lat = {k: v.repeat_interleave(B, 0).clone() for k, v in lat0.items()}
xs = coords[masks[:, 0]][None].expand(B, -1, -1)
xf = coords[None].expand(B, -1, -1)
with torch.no_grad():
    print("fwd decode N=4096: %.3f ms" % timeit(lambda: nef.apply(params, xf, lat["p_pos"], lat["a"], lat["gaussian_window"])))
    print("fwd fit    N=512 : %.3f ms" % timeit(lambda: nef.apply(params, xs, lat["p_pos"], lat["a"], lat["gaussian_window"])))
def fb():
    l = {k: v.detach().requires_grad_(True) for k, v in lat.items()}
    out = nef.apply(params, xs, l["p_pos"], l["a"], l["gaussian_window"])
    out.sum().backward()
print("fwd+bwd fit N=512 : %.3f ms" % timeit(fb))
print("full step         : %.3f ms" % timeit(lambda: bench.one_step(nef, params, lat0, lrs, coords, img, masks), iters=5))
print(bench.roofline_leg(nef, params, coords, dev))
