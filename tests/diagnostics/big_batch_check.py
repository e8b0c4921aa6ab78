import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch, time
from be_hip import synth
import models
m = models.LocalStage(); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()}); m = m.to("cuda:0").eval()
x = torch.from_numpy(synth.uniform_patches(4096, name="bigbatch")).to("cuda:0")
xb = x.repeat(10, 1, 1, 1)[:40001].contiguous()          # 40001 patches: ragged halves, several sub-batches per half
with torch.no_grad():
    y2 = m(xb).clone(); m.streams = 1; y1 = m(xb).clone()
    ref = m(x)
print("big ragged batch identical across schedules:", torch.equal(y1, y2), " periodic:", torch.equal(y1[:4096], ref), torch.equal(y1[4096:8192], ref), torch.equal(y2[36864:40001], ref[:3137]))
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
