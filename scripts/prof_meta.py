"""rocprofv3 target: a few outer (meta) steps at the bench shape (second_order='fd')."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import torch
import bench
from enf_pde_amd.fitting.trainers import MetaSGDPDETrainer
from enf_pde_amd.enf.latents.autodecoder_meta import PositionOrientationFeatureAutodecoderMeta
dev = torch.device("cuda:0")
nef, params, lat0, lrs, masks = bench.build(dev, "bf16")
coords, img = bench.synth_fields(bench.B_PER_GPU, 100, dev)
conf = NS(optimizer=NS(learning_rate_enf=1e-4, learning_rate_codes=0.0), meta=NS(learning_rate_meta_sgd=1e-3, num_inner_steps=3,
          inner_learning_rate_p=1.0, inner_learning_rate_a=5.0, inner_learning_rate_window=0.0, noise_pos_inner_loop=0.0),
          nef=NS(optimize_gaussian_window=False), training=NS(max_num_sampled_points=512))
ad = PositionOrientationFeatureAutodecoderMeta(1, bench.Z, bench.C, 2, 0, gaussian_window_size=-1)
batch = img.reshape(bench.B_PER_GPU, bench.GRID, bench.GRID, bench.O)
tr = MetaSGDPDETrainer(conf, nef, ad, coords, seed=0, second_order="fd")
st = tr.init_train_state(params)
for _ in range(5):
    loss, st = tr.nef_train_step(st, batch)
torch.cuda.synchronize()
print("loss", float(loss))
