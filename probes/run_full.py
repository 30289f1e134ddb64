"""Full IGGT forward incl. the part branch (H, W multiples of 28) for profiling: python probes/run_full.py S H W"""
import os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt.models.vggt import IGGT
S, H, W = (int(x) for x in (sys.argv[1:4] or (8, 504, 504)))
torch.manual_seed(0)
with torch.device("cuda"):
    model = IGGT().eval()
img = torch.rand(S, 3, H, W, device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = model(img)
    torch.cuda.synchronize(); print(f"forward {i}: {(time.perf_counter()-t)*1e3:.1f} ms", {k: tuple(v.shape) for k, v in out.items() if torch.is_tensor(v)} if i == 0 else "")
