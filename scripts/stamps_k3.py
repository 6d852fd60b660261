"""Per-phase cycle stamps of K3 (needs a -DENF_STAMPS build selected with ENF_HIP_LIB)."""
import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from enf_pde_amd import _lib
dev = torch.device("cuda:0")
nef, params, lat0, lrs, masks = bench.build(dev, "bf16")
coords, img = bench.synth_fields(bench.B_PER_GPU, 100, dev)
B = bench.B_PER_GPU
lat = {k: v.repeat_interleave(B, 0).clone() for k, v in lat0.items()}
xs = coords[masks[:, 0]][None].expand(B, -1, -1)
for _ in range(3):
    l = {k: v.detach().requires_grad_(True) for k, v in lat.items()}
    nef.apply(params, xs, l["p_pos"], l["a"], l["gaussian_window"]).sum().backward()
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 4 * 24))()
assert lib.enf_debug_read_stamps_bwd(buf) == 0
a = np.array(buf, dtype=np.int64).reshape(8, 4, 24)
# the z-fold kernel (what the fit shape runs) stamps 4 + 6h / 5 + 6h / 6 + 6h per head; the unfolded one 3 .. 7 + 6h
zf = a[0, 1, 3] == 0
if zf:
    names = {1: "q-fwd", 2: "v-fwd", 4: "a5-gemm0", 5: "gelu/LN/softmax-bwd0", 6: "dn+flips0", 10: "a5-gemm1", 11: "gelu/LN/softmax-bwd1",
             12: "dn+flips1", 15: "LN-bwd", 16: "v-bwd", 17: "q-bwd"}
    order = [1, 2, 4, 5, 6, 10, 11, 12, 15, 16, 17]
else:
    names = {1: "q-fwd", 2: "v-fwd", 3: "gb0", 4: "mixer0", 5: "gelu/LN/softmax-bwd0", 6: "gM0", 7: "film-bwd0", 9: "gb1", 10: "mixer1",
             11: "gelu/LN/softmax-bwd1", 12: "gM1", 13: "film-bwd1", 15: "gGB1", 16: "v-bwd", 17: "q-bwd"}
    order = [1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 15, 16, 17]
for w in (0, 4):
    for ti in (1, 2):
        t = a[w, ti]
        prev = 0; d = {}
        for k in order:
            d[names[k]] = int(t[k] - t[prev]); prev = k
        print(f"wave {w} tile {ti} total {int(t[17]-t[0])} gap_to_next {int(a[w,ti+1,0]-t[17])}")
        print("   ", d)

# workgroup-level: entry / loop start / loop end / exit of one workgroup per round (the four that land on the same CU slot)
try:
    wg = (ctypes.c_ulonglong * (4 * 8 * 4))()
    assert lib.enf_debug_read_stamps_bwd_wg(wg) == 0
    w = np.array(wg, dtype=np.int64).reshape(4, 8, 4)
    t0 = w[0, :, 0].min()
    for r in range(4):
        e, ls, le, x = (w[r, :, k] for k in range(4))
        print(f"round {r}: entry {int(e.min() - t0)}..{int(e.max() - t0)}  prologue {int((ls - e).mean())}  loop {int((le - ls).mean())}  "
              f"epilogue {int((x - le).mean())}  exit {int(x.max() - t0)}")
except AttributeError:
    pass
