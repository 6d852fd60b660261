"""enf_backward_weights (K3 with the activation store + K4 enf_xtd_kernel + reduction) timed at the bench shape:
python scripts/probe_wgrad.py [B ...]   (default 16 32; N_s = 512, Z = 64, D = 128, H = 2, bf16).  Under rocprofv3 --kernel-trace
--stats the per-kernel split is in the summary."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_weight_grads import _pair_problem
cuda = torch.device("cuda:0")
for B in [int(v) for v in sys.argv[1:]] or [16, 32]:
    q = _pair_problem(cuda, 128, 2, "bf16", B, 512, 64, seed=1)
    D, HD = q.D, q.H * q.D
    shapes = [(D, D), (D,), (D, D), (D,), (D, D), (D,), (D, 2 * HD), (2 * HD,), (D, D), (D,)]
    grads = [torch.empty(sh, device=cuda) for sh in shapes]
    arr = (ctypes.c_void_p * 12)(*([g.data_ptr() for g in grads] + [None, None]))
    nbytes = int(q.lib.enf_backward_weights_scratch_bytes(ctypes.byref(q.desc), B))
    scratch = torch.empty(nbytes, device=cuda, dtype=torch.uint8)
    dlt = torch.empty_like(q.lt)
    run = lambda: q._lib.check(q.lib.enf_backward_weights(ctypes.byref(q.desc), q.P(q.xs), q.N * 2, q.P(q.lt), q.P(q.blob), q.P(q.lse),
                                                          q.P(q.dybar), q.P(q.delta), q.P(dlt), arr, None, q.P(scratch), nbytes, q.st))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    store = (7 + 4 * q.H) * B * 64 * 512 * D * 2
    print(f"B={B}: enf_backward_weights {ms:.3f} ms per call; store {store / 1e9:.2f} GB written + read once "
          f"-> {2 * store / (ms * 1e-3) / 1e12:.2f} TB/s over the whole call", flush=True)
