"""Follow-up to the one failure of tests/test_gpu_weight_grads.py::test_backward_weights_kernel[128-1-bf16] in a full-suite run
("K3 run-to-run deviations"): which of the three explanations holds?
  H1  enf_backward_weights reads scratch it did not write (the test's scratch is torch.empty: stale finite data in a long process)
  H2  K3's STORE instantiation stores different activations from run to run
  H3  K4 (enf_xtd_kernel + reduce) differs from run to run on identical stores
Per iteration: the scratch is POISONED with NaN (H1 -> NaN in a gradient), the call runs, the ENF_S_* stores inside the scratch
and the ten gradient tensors are compared bitwise with the first iteration's (H2 / H3).
    python scripts/k3_race/store_probe.py [iterations] [D] [H] [precision]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_gpu_weight_grads import _pair_problem

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 300
D, H, prec = (int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]) if len(sys.argv) > 4 else (128, 1, "bf16")
cuda = torch.device("cuda:0")
B, N, Z = (int(v) for v in os.environ.get("SHAPE", "5,77,6").split(","))
q = _pair_problem(cuda, D, H, prec, B, N, Z, seed=D + H)
HD, ns = H * D, 7 + 4 * H
shapes = [(D, D), (D,), (D, D), (D,), (D, D), (D,), (D, 2 * HD), (2 * HD,), (D, D), (D,)]
nbytes = int(q.lib.enf_backward_weights_scratch_bytes(ctypes.byref(q.desc), B))
es = 2 if q.bf16 else 4
P = B * Z * N
sb = (P * D * es + 255) // 256 * 256
first = None
bad = {"nan": 0, "store": 0, "grad": 0, "dlt": 0}
filler = torch.randn(64 << 20, device=cuda)                       # churn for the allocator / caches between iterations
# H4: K3 / K4 read LDS or registers they did not initialise -- then the result depends on the kernel that ran on the CU before.
# Polluter (POLLUTE=1): the fused ODE kernel-basis backward (133 KB of LDS, 512 registers per lane, every CU) on all-NaN inputs
# in front of every second iteration.
pollute = os.environ.get("POLLUTE", "0") == "1"
if pollute:
    from enf_pde_amd.fitting.ode_models.ponita_ode_g import kernel_basis
    nanx = torch.full((65536, 4), float("nan"), device=cuda, requires_grad=True)
    K1 = {"kernel": torch.full((340, 128), float("nan"), device=cuda), "bias": torch.full((128,), float("nan"), device=cuda)}
    K3 = {"kernel": torch.full((128, 64), float("nan"), device=cuda), "bias": torch.full((64,), float("nan"), device=cuda)}
for it in range(n_it):
    if pollute and it % 2 == 1:
        kernel_basis(nanx, 3, K1, K3).sum().backward()
    scratch = torch.full((nbytes // 4,), float("nan"), device=cuda).view(torch.uint8)
    grads = [torch.full(sh, float("nan"), device=cuda) for sh in shapes]
    arr = (ctypes.c_void_p * 12)(*([g.data_ptr() for g in grads] + [None, None]))
    dlt = torch.empty_like(q.lt)
    q._lib.check(q.lib.enf_backward_weights(ctypes.byref(q.desc), q.P(q.xs), q.N * 2, q.P(q.lt), q.P(q.blob), q.P(q.lse),
                                            q.P(q.dybar), q.P(q.delta), q.P(dlt), arr, None, q.P(scratch), nbytes, q.st))
    torch.cuda.synchronize()
    stores = [scratch[i * sb:i * sb + P * D * es].clone() for i in range(ns)]
    if any(not torch.isfinite(g).all() for g in grads):
        bad["nan"] += 1
    if first is None:
        first = (stores, [g.clone() for g in grads], dlt.clone())
    else:
        ds = [i for i in range(ns) if not torch.equal(stores[i], first[0][i])]
        dg = [i for i in range(10) if not torch.equal(grads[i], first[1][i])]
        if ds:
            bad["store"] += 1
            i = ds[0]
            rows = (stores[i].view(P, D * es) != first[0][i].view(P, D * es)).any(1).nonzero().flatten().tolist()
            print(f"it {it}: ENF_S buffers {ds} differ; buffer {i}: {len(rows)} rows, first {rows[:12]}", flush=True)
            if i == 3 and q.bf16:                                  # signature: (out-tile, quad, r) of every differing stored column
                dt0 = torch.bfloat16
                A0 = stores[3].view(dt0).view(P, D)[rows]
                B0 = first[0][3].view(dt0).view(P, D)[rows]
                sig = []
                for cs_ in (A0 != B0).any(0).nonzero().flatten().tolist():
                    jj = cs_ % 8
                    tf = 32 * (cs_ // 32) + (4 * ((cs_ % 32) // 8) + jj if jj < 4 else 16 + 4 * ((cs_ % 32) // 8) + jj - 4)
                    sig.append((tf // 16, tf % 16 // 4, tf % 4))
                print(f"   signature (out-tile, quad, r): {sig}; wave {rows[0] // N % 8}, query tile {rows[0] % N // 16}", flush=True)
            if bad["store"] <= 2 and os.environ.get("ANATOMY", "0") == "1":   # anatomy of the first few: which columns, how far off
                dt = torch.bfloat16 if q.bf16 else torch.float32
                A_ = stores[i].view(dt).view(P, D).float()[rows]
                B_ = first[0][i].view(dt).view(P, D).float()[rows]
                cols = (A_ != B_).any(0).nonzero().flatten().tolist()
                print(f"   buffer {i}: {len(cols)} of {D} stored columns differ: {cols[:40]}")
                print(f"   per row: differing columns {(A_ != B_).sum(1).tolist()}")
                print(f"   max |diff| {float((A_ - B_).abs().max()):.4g}  max |ref| {float(B_.abs().max()):.4g};  row 0 got/ref (first 8 differing): "
                      f"{[(round(float(A_[0, c]), 4), round(float(B_[0, c]), 4)) for c in cols[:8]]}")
                if i == 3 and q.bf16 and len(cols) == 1:
                    # which error of the pre-activation a3 = G1 AF + bF explains it?  (f = gelu(a3), NH = LayerNorm(f) without affine)
                    c = torch.arange(D, device=cuda)
                    j = c % 8
                    true = 32 * (c // 32) + torch.where(j < 4, 4 * ((c % 32) // 8) + j, 16 + 4 * ((c % 32) // 8) + j - 4)
                    unperm = lambda t: torch.empty_like(t).index_copy_(1, true, t)
                    G1 = unperm(first[0][2].view(dt).view(P, D).double()[rows])
                    NHr, NHg = unperm(B_.double()), unperm(A_.double())
                    AF, bF = q.keep[0][4].double(), q.keep[0][5].double()
                    AFb = AF.to(torch.bfloat16).double()
                    a3 = G1 @ AFb + bF
                    gelu = lambda t: torch.nn.functional.gelu(t, approximate="tanh")
                    f = gelu(a3)
                    mu, var = f.mean(1, keepdim=True), f.var(1, unbiased=False, keepdim=True)
                    rstd = (var + 1e-6).rsqrt()
                    ct = int(true[cols[0]])
                    print(f"   true feature {ct} (out-tile {ct // 16}, row {ct % 16}: quad {ct % 16 // 4}, r {ct % 4}); recomputed NH vs stored ref: "
                          f"{float(((f - mu) * rstd - NHr).abs().max()):.3g}")
                    target = NHg[:, ct]
                    cands = {"bias missing": a3[:, ct] - bF[ct], "bias twice": a3[:, ct] + bF[ct]}
                    for kb in range(D // 32):
                        ck = G1[:, 32 * kb:32 * kb + 32] @ AFb[32 * kb:32 * kb + 32, ct]
                        cands[f"k-block {kb} missing"] = a3[:, ct] - ck
                        cands[f"k-block {kb} twice"] = a3[:, ct] + ck
                    for name, a in cands.items():
                        err = float((((gelu(a)[:, None] - mu) * rstd)[:, 0] - target).abs().max())
                        if err < 0.05:
                            print(f"   explained by: {name} (max err {err:.3g})")
                    fp = target / rstd[:, 0] + mu[:, 0]                       # the value that was normalised in place of f[:, ct]
                    xx = a3
                    sg = torch.sigmoid(2 * 0.7978845608028654 * (xx + 0.044715 * xx ** 3))
                    dgel = sg + xx * sg * (1 - sg) * 2 * 0.7978845608028654 * (1 + 3 * 0.044715 * xx ** 2)
                    for name, M_ in (("f = gelu(a3)", f), ("a3", a3), ("gelu'(a3)", dgel), ("sigmoid", sg), ("relu(a3)", a3.clamp_min(0)), ("G1", G1),
                                     ("NH ref", NHr)):
                        e = (M_ - fp[:, None]).abs().max(0).values
                        best = int(e.argmin())
                        print(f"   closest column of {name:14s}: feature {best:3d} (max err over the 16 rows {float(e[best]):.3g}; same feature: {float(e[ct]):.3g})")
                    r3 = lambda t: [round(float(v), 3) for v in t]
                    print(f"   a3[:, {ct}] = {r3(a3[:, ct])}")
                    print(f"   f [:, {ct}] = {r3(f[:, ct])}")
                    print(f"   f'        = {r3(fp)}")
                    # inverse gelu of f' by bisection -> the a3' that would give it
                    lo, hi = torch.full_like(fp, -0.7), torch.full_like(fp, 8.0)
                    for _ in range(60):
                        mid = (lo + hi) / 2
                        big = gelu(mid) < fp
                        lo, hi = torch.where(big, mid, lo), torch.where(big, hi, mid)
                    print(f"   a3' - a3  = {r3((lo + hi) / 2 - a3[:, ct])}   (if the pre-activation was off)")
                    print(f"   implied f' - f over the 16 rows: {[round(float(v), 3) for v in (target / rstd[:, 0] + mu[:, 0] - f[:, ct])]};  bF[{ct}] = {float(bF[ct]):.3f}")
                print(f"   row-wise mean got/ref {[(round(float(a), 4), round(float(b), 4)) for a, b in zip(A_.mean(1)[:4], B_.mean(1)[:4])]}")
        if dg:
            bad["grad"] += 1
            if not ds:
                print(f"it {it}: gradients {dg} differ on identical stores (K4)", flush=True)
        if not torch.equal(dlt, first[2]):
            bad["dlt"] += 1
    filler.mul_(1.0001)
print(f"{n_it} iterations of enf_backward_weights<{D},{H},{prec}> (B={B}, N={N}, Z={Z}): NaN from poisoned scratch {bad['nan']}, "
      f"runs whose stores differ from the first {bad['store']}, runs whose weight gradients differ {bad['grad']}, "
      f"latent gradients (atomic sums) differ {bad['dlt']}")
