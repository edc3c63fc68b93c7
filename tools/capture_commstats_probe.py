"""ONE run, in a child process: does hipGraph capture of the NO_SHARD engine's step die when CommStats (timing-enabled HIP events
recorded on the capturing streams + collective waits on the communication stream) is active?  Round 3 saw a segfault in
hipStreamEndCapture while CommStats was being added (gpurun_out/r03_gputest_e.log); GraphedTrainStep.capture now refuses the
combination, this probe switches the guard off to record what the runtime does.  Prints the child's exit status."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path[:0] = [%r, %r]
os.environ.update(ORBIT2_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29671")
import torch, torch.nn as nn, torch.distributed as dist
import climate_learn as cl
from climate_learn.graphs import GraphedTrainStep
from climate_learn.metrics import Bayesian_TV
from climate_learn.models.hub.components.vit_blocks import Block
from oracle.harness import build_pair
mode = sys.argv[1]
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
model, sd, cfg, O, x, y, in_vars, out_vars = build_pair(D=128, depth=2, heads=2, grid=(16, 32), B=2, seed=23)
eng = cl.HipDataParallel(model.cuda().train(), unit_types=(Block, nn.Sequential))
if mode == "stats":
    eng.comm_stats = cl.CommStats()
step = GraphedTrainStep(eng, Bayesian_TV(aggregate_only=True), (x, y, in_vars, out_vars), {"total_precipitation_24hr": 1.0})
step._allow_comm_stats = True
print("capturing", mode, flush=True)
l = float(step())
torch.cuda.synchronize()
print("captured and replayed, loss", l, flush=True)
if mode == "stats":
    try:
        print("summary", eng.comm_stats.summary(1), flush=True)
    except Exception as e:
        print("summary raised", type(e).__name__, str(e)[:200], flush=True)
''' % (ROOT, os.path.join(ROOT, "orbit-2_amd"))
for mode in ("plain", "stats"):
    r = subprocess.run([sys.executable, "-c", CHILD, mode], capture_output=True, text=True, timeout=240)
    print("== mode %s: exit code %d" % (mode, r.returncode))
    print(r.stdout[-1500:])
    print(r.stderr[-1500:])
