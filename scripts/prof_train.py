import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import torch
import bench
from enf_pde_amd.fitting.trainers import NonMetaPDETrainer
from enf_pde_amd.enf.latents.autodecoder import PositionOrientationFeatureAutodecoder
dev = torch.device("cuda:0")
nef, params, lat0, lrs, masks = bench.build(dev, "bf16")
coords, img = bench.synth_fields(bench.B_PER_GPU, 100, dev)
batch = img.reshape(bench.B_PER_GPU, bench.GRID, bench.GRID, bench.O)
conf2 = NS(optimizer=NS(learning_rate_enf=1e-4, learning_rate_codes=1e-3), training=NS(max_num_sampled_points=512))
ad2 = PositionOrientationFeatureAutodecoder(64, bench.Z, bench.C, 2, 0, gaussian_window_size=-1)
tr2 = NonMetaPDETrainer(conf2, nef, ad2, coords, seed=0)
st2 = tr2.init_train_state(params)
idx = torch.arange(bench.B_PER_GPU, device=dev)
for _ in range(6):
    loss, st2 = tr2.nef_train_step(st2, (batch, idx))
torch.cuda.synchronize()
