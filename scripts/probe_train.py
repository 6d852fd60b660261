"""Wall time of one outer (meta) step and one auto-decoder step at the bench shape (16 signals, N_s=512, Z=64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import torch
import bench
from enf_pde_amd.fitting.trainers import MetaSGDPDETrainer, NonMetaPDETrainer
from enf_pde_amd.enf.latents.autodecoder_meta import PositionOrientationFeatureAutodecoderMeta
from enf_pde_amd.enf.latents.autodecoder import PositionOrientationFeatureAutodecoder
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nef, params, lat0, lrs, masks = bench.build(dev, prec)
coords, img = bench.synth_fields(bench.B_PER_GPU, 100, dev)
conf = NS(optimizer=NS(learning_rate_enf=1e-4, learning_rate_codes=0.0), meta=NS(learning_rate_meta_sgd=1e-3, num_inner_steps=3,
          inner_learning_rate_p=1.0, inner_learning_rate_a=5.0, inner_learning_rate_window=0.0, noise_pos_inner_loop=0.0),
          nef=NS(optimize_gaussian_window=False), training=NS(max_num_sampled_points=512))
ad = PositionOrientationFeatureAutodecoderMeta(1, bench.Z, bench.C, 2, 0, gaussian_window_size=-1)
batch = img.reshape(bench.B_PER_GPU, bench.GRID, bench.GRID, bench.O)
for mode in ("fd", "none"):
    tr = MetaSGDPDETrainer(conf, nef, ad, coords, seed=0, second_order=mode)
    st = tr.init_train_state(params)
    for _ in range(2):
        loss, st = tr.nef_train_step(st, batch)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        loss, st = tr.nef_train_step(st, batch)
    torch.cuda.synchronize()
    print(f"{prec} meta step second_order={mode}: {(time.time() - t0) / 5 * 1e3:.1f} ms  loss {float(loss):.4f}")
conf2 = NS(optimizer=NS(learning_rate_enf=1e-4, learning_rate_codes=1e-3), training=NS(max_num_sampled_points=512))
ad2 = PositionOrientationFeatureAutodecoder(64, bench.Z, bench.C, 2, 0, gaussian_window_size=-1)
tr2 = NonMetaPDETrainer(conf2, nef, ad2, coords, seed=0)
st2 = tr2.init_train_state(params)
idx = torch.arange(bench.B_PER_GPU, device=dev)
for _ in range(2):
    loss, st2 = tr2.nef_train_step(st2, (batch, idx))
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5):
    loss, st2 = tr2.nef_train_step(st2, (batch, idx))
torch.cuda.synchronize()
print(f"{prec} auto-decoder step: {(time.time() - t0) / 5 * 1e3:.1f} ms  loss {float(loss):.4f}")
