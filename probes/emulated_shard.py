"""DEVELOPER TOOL, not product code (moved out of iggt_official_amd/dist.py in round 4: nothing in the package may return
outputs that are not the model's).  Used by `bench.py --emulate-world W` and probes/emulate_rank.py."""
from typing import Optional, Tuple

from iggt_official_amd.dist import view_partition


class EmulatedShard:
    """DEVELOPER TOOL (bench.py --emulate-world, probes/emulate_rank.py): stands in for ViewShard on ONE GPU to measure what a
    single rank of a `world`-GPU run computes.  The gathers are local copies -- this rank's rows repeated `world` times land in
    the gathered buffers, so every kernel runs at the per-rank shapes with the right byte counts, but nothing is transported
    and the other ranks' keys are copies of this rank's: the OUTPUTS ARE NOT THE MODEL'S (parity of the sharded path is what
    tests/test_shard_gpu.py and tests/test_headline_gpu.py check).  The copies run on the compute stream and are counted
    (RCCL's transport would overlap with the own-key attention)."""
    active, force, kv_groups = True, False, 1

    def __init__(self, world: int, rank: Optional[int] = None):
        self.world = int(world)
        self.rank = self.world // 2 if rank is None else int(rank)     # a middle rank: rank 0 is not the typical one
        self.ctl = None

    def local_views(self, S: int) -> Tuple[int, int]:
        return view_partition(S, self.world, self.rank)

    def all_gather_kv(self, kv, stats=None):
        out = kv.repeat(self.world, 1)
        return out if stats is None else (out, stats.reshape(1, 32).repeat(self.world, 1))

    def all_gather_kv_begin(self, kv, stats=None):
        out = kv.repeat(self.world, 1)
        if stats is None:
            return out, (lambda: None)
        return out, stats.reshape(1, 32).repeat(self.world, 1), (lambda: None)

    def agree_any(self, flags, device):
        return [1 if f else 0 for f in flags]

    def all_gather_rows(self, x):
        return x.repeat(self.world, *([1] * (x.dim() - 1)))
