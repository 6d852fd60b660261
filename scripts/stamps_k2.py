"""Per-phase cycle stamps of K2 (needs a -DENF_STAMPS build selected with ENF_HIP_LIB)."""
import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from enf_pde_amd import _lib
dev = torch.device("cuda:0")
nef, params, lat0, lrs, masks = bench.build(dev, "bf16")
coords, img = bench.synth_fields(bench.B_PER_GPU, 100, dev)
import os as _os
if _os.environ.get("ZF"): nef.pair_variants = (("latent_split", "z_fold")[int(_os.environ["ZF"])], "auto")
r = bench.roofline_leg(nef, params, coords, dev, iters=3)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 4 * 24))()
assert lib.enf_debug_read_stamps(buf) == 0
a = np.array(buf, dtype=np.int64).reshape(8, 4, 24)
names = {1: "zv+inv+rffq", 2: "gemmQ1", 3: "logits", 4: "rffv", 5: "gemmV1", 6: "relu", 7: "gemmF", 8: "gelu+LN",
         9: "gb0", 10: "frag0", 11: "gemmM0", 12: "gelu/LN/acc0", 13: "gb1", 14: "frag1", 15: "gemmM1", 16: "gelu/LN/acc1"}
print("launch_ms", r["launch_ms"])
for w in (0, 4, 1):
    for it in (1, 2):
        t = a[w, it]
        d = {names[k]: int(t[k] - t[k - 1]) for k in range(1, 17)}
        print(f"wave {w} it {it} total {int(t[16]-t[0])} next_it_gap {int(a[w,it+1,0]-t[16]) if it<3 else 0}")
        print("   ", d)
print("start offsets it1:", [int(a[w,1,0]-a[0,1,0]) for w in range(8)])
