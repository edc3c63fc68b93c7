"""wall-clock of the default bench step without the per-kernel timer (bench.py always installs it)"""
import os, sys, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch, torch.nn as nn
import climate_learn as cl
from climate_learn.metrics import Bayesian_TV
from climate_learn.models.hub import Res_Slim_ViT
from climate_learn.models.hub.components.vit_blocks import Block
from climate_learn.trainer import training_step
import bench as bm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
m = bm.MODELS["interm_1b"]; in_vars = bm.ERA5_VARS; V, C = len(in_vars), len(bm.OUT_VARS); h, w = 128, 256
torch.manual_seed(0); cl.manual_seed(0, 0)
with torch.device(dev):
    model = Res_Slim_ViT(in_vars, (h, w), V, C, 1, superres_mag=4, cnn_ratio=4, patch_size=2, drop_path=0.1, drop_rate=0.1,
                         learn_pos_emb=True, embed_dim=m["embed_dim"], depth=m["depth"], decoder_depth=4,
                         num_heads=m["num_heads"], mlp_ratio=4)
model.data_config(156.0, (h, w), V, C)
eng = cl.HipDataParallel(model, unit_types=(Block, nn.Sequential))
opt = cl.load_optimizer(eng, "adamw", {"lr": 5e-4, "betas": (0.9, 0.99), "weight_decay": 1e-5})
scaler = cl.HipGradScaler(init_scale=8192.0)
lossf = Bayesian_TV(aggregate_only=True); eng.train()
g = torch.Generator().manual_seed(1)
x = torch.randn(B, V, h, w, generator=g).to(dev); y = torch.randn(B, C, 721, 1440, generator=g).to(dev)
batch = (x, y, in_vars, bm.OUT_VARS)
def step(i):
    loss = training_step(batch, i, eng, dev, bm.VAR_WEIGHTS, lossf); opt.zero_grad(); scaler.scale(loss).backward(); scaler.step(opt); scaler.update(); return loss
for i in range(2): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for i in range(n): l = step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("PLAIN_GEMM=%s  B=%d  %.1f ms/step  %.3f samples/s  loss %.4f" % (os.environ.get("ORBIT2_PLAIN_GEMM", "own"), B, dt * 1e3, B / dt, float(l)))
