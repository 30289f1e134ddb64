"""Emulate one rank of an N-GPU run on a single GPU: the K/V all-gather is replaced by a local repeat (same bytes
land in the gathered buffer, no transport), so this measures per-rank compute + host launch overhead at N ranks."""
import os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt.models.vggt import IGGT

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
SIZE = int(sys.argv[3]) if len(sys.argv) > 3 else 518


from probes.emulated_shard import EmulatedShard  # noqa: E402

FakeShard = lambda: EmulatedShard(N, int(os.environ["IGGT_EMU_RANK"]) if "IGGT_EMU_RANK" in os.environ else None)  # noqa: E731


torch.manual_seed(0)
with torch.device("cuda"):
    model = IGGT(part_on_invalid_grid="skip").eval()
model.set_view_shard(FakeShard())
GRAPHS = os.environ.get("IGGT_GRAPHS", "0") == "1"
if GRAPHS:
    model.enable_graphs(True)
img = torch.rand(S // N, 3, SIZE, SIZE, device="cuda")
for i in range(3 if SIZE > 518 else 6):
    torch.cuda.synchronize(); t = time.perf_counter()
    model(img)
    t_host = time.perf_counter() - t
    torch.cuda.synchronize(); t_all = time.perf_counter() - t
    print(f"graphs={int(GRAPHS)} N={N} local views={S//N} @{SIZE}: peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, host-side launch time {t_host*1e3:.1f} ms, forward {t_all*1e3:.1f} ms "
          f"-> {S/t_all:.1f} views/s if comm were free")
