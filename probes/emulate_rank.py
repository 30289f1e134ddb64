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


class FakeShard:
    # a MIDDLE rank by default: rank 0 owns view 0 and (before round 3) had the cheapest key-segment layout
    rank, world, active = int(os.environ.get("IGGT_EMU_RANK", str(N // 2))), N, True
    kv_groups = int(os.environ.get("IGGT_KV_GROUPS", "1"))
    force = False
    _streams, _events = [], []

    def gather_kv_groups(self, kv_local):
        return [(None, kv_local[g].repeat(N, 1)) for g in range(kv_local.shape[0])]

    def side_stream(self, g):
        while len(self._streams) <= g:
            self._streams.append(torch.cuda.Stream())
        return self._streams[g]

    def event(self, i):
        while len(self._events) <= i:
            self._events.append(torch.cuda.Event())
        return self._events[i]

    def all_gather_kv(self, kv):
        return kv.repeat(N, 1)

    def all_gather_kv_begin(self, kv):     # overlap path (own keys first): same bytes land, no transport to hide here
        return kv.repeat(N, 1), (lambda: None)   # (the copy runs on the compute stream: it is counted, RCCL's would overlap)

    def all_gather_rows(self, x):
        return x.repeat(N, *([1] * (x.dim() - 1)))


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
