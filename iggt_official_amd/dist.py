"""View sharding across GPUs (one process per GPU, torch.distributed over RCCL/xGMI).

The reference has no inference parallelism; this is the new design of
section 8e: views are independent everywhere except (a) global attention, which needs every view's
K and V, and (b) the camera head, which attends across the S camera tokens.

* rank r owns views [r*S/G, (r+1)*S/G); rank 0 owns view 0 (the only view that uses slot 0 of
  camera_token/register_token, aggregator.py:338-361);
* before each of the 24 global attentions every rank contributes its post-RoPE K rows and its V
  rows as ONE contiguous [T_local, 2*C] bf16 message (token-major layout makes the gathered
  buffer directly consumable by the attention kernel: keys of rank r are rows [r*T_l, (r+1)*T_l));
  softmax is permutation-invariant over keys so no re-ordering is needed.  32 views x 518^2 on 8
  GPUs: 22.5 MB per rank per block; on the xGMI full mesh RCCL moves it as direct peer writes.
* the S camera tokens ([S, 2048] fp32, 256 KB at S=32) are all-gathered once for the camera head.

Works with any initialised process group: "nccl" (= RCCL) on GPUs, "gloo" on CPU for the
host-logic tests (tests/test_dist_gloo.py).
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def view_partition(S: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, equal slices (all_gather_into_tensor needs equal message sizes)."""
    if S % world != 0:
        raise ValueError(f"number of views ({S}) must be divisible by the number of ranks ({world})")
    per = S // world
    return rank * per, (rank + 1) * per


class ViewShard:
    def __init__(self, group: Optional["dist.ProcessGroup"] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._bufs = {}

    def local_views(self, S: int) -> Tuple[int, int]:
        return view_partition(S, self.world, self.rank)

    def _buf(self, name, shape, like):
        n = 1
        for s in shape:
            n *= s
        cur = self._bufs.get(name)
        if cur is None or cur.numel() < n or cur.dtype != like.dtype or cur.device != like.device:
            cur = torch.empty(n, dtype=like.dtype, device=like.device)
            self._bufs[name] = cur
        return cur[:n].view(*shape)

    def all_gather_kv(self, kv_local: torch.Tensor) -> torch.Tensor:
        """kv_local [T_l, 2C] (contiguous) -> [world*T_l, 2C], rank-major = view-major token order."""
        assert kv_local.is_contiguous()
        out = self._buf("kv_all", (self.world * kv_local.shape[0], kv_local.shape[1]), kv_local)
        self._gather(out, kv_local)
        return out

    def _gather(self, out, x):
        """all_gather_into_tensor (RCCL: one direct all-gather); list-based fallback for backends that lack the
        flat variant for device tensors (gloo, used by the single-GPU two-rank test)."""
        try:
            dist.all_gather_into_tensor(out, x, group=self.group)
        except (RuntimeError, NotImplementedError):
            parts = list(out.view(self.world, *x.shape).unbind(0))
            dist.all_gather(parts, x, group=self.group)

    def all_gather_rows(self, x_local: torch.Tensor) -> torch.Tensor:
        """[n_l, ...] -> [world*n_l, ...] (camera tokens, small outputs)."""
        x_local = x_local.contiguous()
        out = torch.empty((self.world * x_local.shape[0],) + tuple(x_local.shape[1:]), dtype=x_local.dtype,
                          device=x_local.device)
        self._gather(out, x_local)
        return out
