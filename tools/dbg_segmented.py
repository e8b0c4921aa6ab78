"""Stage-by-stage run of SegmentedGraphStep under a one-rank RCCL group, with progress lines (debug aid)."""
import os, sys, socket, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch
def say(*a): print(*a, file=sys.stderr, flush=True)
import torch.distributed as dist
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
torch.cuda.set_device(0)
if os.environ.get("NO_PG") is None:
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    say("process group up")
import models, utils
from be_hip import dp, synth, train_local
DEV = "cuda:0"
args = utils.get_args("local_train", argv=[])
helper = utils.PostProcessLocalBase(args, DEV)
data = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synthetic_training_patches(64 * 4, seed=77).items()}
mm = models.LocalStage().to(DEV)
mm.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
mm.train()
opt = torch.optim.AdamW(mm.parameters(), lr=1e-3, capturable=True, fused=dp.fused_adamw())
sync = dp.GradSync(1, always=True) if os.environ.get("NO_PG") is None else None
seg = train_local.SegmentedGraphStep(mm, helper, opt, sync, world=1)
for it in range(8):
    b = {k: v[(it % 4) * 64:(it % 4 + 1) * 64] for k, v in data.items()}
    say("step", it, "begin")
    l = seg(b, args.beta_bndry_loc, args.beta_smthns)
    torch.cuda.synchronize()
    say("step", it, "loss", float(l))
t0 = time.perf_counter()
for it in range(50):
    b = {k: v[(it % 4) * 64:(it % 4 + 1) * 64] for k, v in data.items()}
    seg(b, args.beta_bndry_loc, args.beta_smthns)
torch.cuda.synchronize()
say("segmented step ms", (time.perf_counter() - t0) / 50 * 1e3)
if dist.is_initialized(): dist.destroy_process_group()
