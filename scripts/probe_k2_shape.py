"""Time the forward pair kernel alone at an arbitrary shape, both variants: python scripts/probe_k2_shape.py B N Z"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from enf_pde_amd import _lib
B, N, Z = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
nef, params, lat0, lrs, masks = bench.build(dev, "bf16")
lib = _lib.load()
g = torch.Generator().manual_seed(3)
x = (torch.rand(B, N, 2, generator=g) * 2 - 1).to(dev)
p = (torch.rand(B, Z, 2, generator=g) * 2 - 1).to(dev)
a = (1 + 0.1 * torch.randn(B, Z, bench.C, generator=g)).to(dev)
sg = torch.full((B, Z, 1), 0.25, device=dev)
packed = nef.pack(params)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for mode in (0, 1):
    nef.pair_variants = (("latent_split", "z_fold")[mode], "auto")
    desc = nef._desc(B, N, Z)
    ws = torch.empty(int(lib.enf_workspace_bytes(ctypes.byref(desc))), device=dev, dtype=torch.uint8)
    out = torch.empty(B, N, bench.O, device=dev)
    ybar = torch.empty(B, N, bench.H * bench.D, device=dev)
    lse = torch.empty(B, N, bench.H, device=dev)
    run = lambda stages: _lib.check(lib.enf_forward_stages(ctypes.byref(desc), P(x), N * 2, P(p), P(a), P(sg), P(packed), P(out),
                                                           P(ybar), P(lse), P(ws), ws.numel(), stages, st))
    run(1 | 8)
    res = {}
    for name, stages in (("pair", 2), ("fold", 8)):
        for _ in range(3):
            run(stages)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            run(stages)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 30 * 1e3
    print(f"B={B} N={N} Z={Z} zfold={mode}: pair {res['pair']:.1f} us  fold(wz) {res['fold']:.1f} us")
