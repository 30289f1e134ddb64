"""Full IGGT forward incl. the part branch (H, W multiples of 28) on the synthetic checkpoint, for profiling:
python probes/run_full.py S H W [forwards]   (bash probes/profile_cmd.sh OUT.txt probes/run_full.py 32 532 532)"""
import json, os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iggt.models.vggt import IGGT
from iggt_official_amd import synthetic
a = sys.argv[1:]
S, H, W = (int(x) for x in (a[:3] if len(a) >= 3 else (8, 504, 504)))
n = int(a[3]) if len(a) > 3 else 3
with torch.device("cuda"):
    model = IGGT().eval()
with open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")) as f:
    sd = synthetic.fill_state_dict(json.load(f), seed=0, mode="stress", device="cuda")
model.load_state_dict(sd, strict=False)
del sd
img = synthetic.make_images(S, H, W, seed=11, device="cuda")
for i in range(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = model(img)
    torch.cuda.synchronize(); print(f"forward {i}: {(time.perf_counter()-t)*1e3:.1f} ms", {k: tuple(v.shape) for k, v in out.items() if torch.is_tensor(v)} if i == 0 else "")
print(f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
