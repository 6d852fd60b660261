"""Dev probe: enf_wz_kernel (forward orientation, decode shape) against the number of latents -- fixed cost vs per-latent cost.
python scripts/probe_wz.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from enf_pde_amd import _lib
from enf_pde_amd.enf.models import _ptr
dev = torch.device("cuda:0")
nef, params, lat0, lrs, masks = bench.build(dev, "bf16")
coords, _ = bench.synth_fields(1, 100, dev)
lib = _lib.load()
packed = nef.pack(params)
for B in (1, 2, 4, 8, 16, 32, 64):
    lat = {k: v.repeat_interleave(B, 0).clone() for k, v in lat0.items()}
    p_, a_, s_ = lat["p_pos"].float().contiguous(), lat["a"].float().contiguous(), lat["gaussian_window"].float().contiguous()
    N, Z = coords.shape[0], p_.shape[1]
    desc = nef._desc(B, N, Z)
    ws = nef._workspace(desc, dev)
    out = torch.empty((B, N, nef.num_out), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    x = coords.contiguous()
    def run(stages):
        _lib.check(lib.enf_forward_stages(ctypes.byref(desc), _ptr(x), 0, _ptr(p_), _ptr(a_), _ptr(s_), _ptr(packed), _ptr(out),
                                          None, None, _ptr(ws), ws.numel(), stages, st))
    run(1)
    for _ in range(5): run(8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): run(8)
    e1.record(); torch.cuda.synchronize()
    print(f"B={B:3d} latents={B * Z:5d}  wz {e0.elapsed_time(e1) * 10:.1f} us per launch", flush=True)
