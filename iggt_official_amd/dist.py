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
* optional (kv_groups = G > 1, IGGT_KV_GROUPS): the gather PIPELINED over head groups: q/k-norm+RoPE writes K and V
  in head-group layout [G][T_local][2 * (16/G) * 64] (csrc/elementwise.hip), the G all-gathers are issued back to
  back as async collectives, and the attention of head group g (its own launch, on a side stream) starts as soon as
  collective g has landed while g+1.. are still on the wire -- heads are independent, so no partial-softmax merge
  is needed and every collective still uses all xGMI links.  Correct (tests/test_shard_gpu.py, test_dist_gloo.py) but
  OFF by default: on one MI355X the four concurrent 4-head launches of an 8-GPU rank take 1.84 ms against 1.26 ms
  for the single 16-head launch (probes/attn_groups.py) -- more than the ~0.45 ms of transport they can hide.
* the S camera tokens ([S, 2048] fp32, 256 KB at S=32) are all-gathered once for the camera head.

Works with any initialised process group: "nccl" (= RCCL) on GPUs, "gloo" on CPU for the
host-logic tests (tests/test_dist_gloo.py).
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def view_partition(S: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, equal slices (all_gather_into_tensor needs equal message sizes)."""
    if S % world != 0:
        raise ValueError(f"number of views ({S}) must be divisible by the number of ranks ({world})")
    per = S // world
    return rank * per, (rank + 1) * per


class ViewShard:
    def __init__(self, group: Optional["dist.ProcessGroup"] = None, kv_groups: Optional[int] = None,
                 force: Optional[bool] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        # force (IGGT_FORCE_COLLECTIVES=1): issue every collective of the sharded path even in a world of ONE rank.  A
        # one-GPU box can then run the real RCCL calls (communicator set-up, 16-bit all_gather_into_tensor, asynchronous
        # work handles, the eager steps between hipGraph segments) and time their per-call overhead -- everything but the
        # transport itself (tests/test_shard_gpu.py::test_rccl_world_of_one, bench.py).
        self.force = (os.environ.get("IGGT_FORCE_COLLECTIVES", "0") == "1") if force is None else bool(force)
        self._bufs = {}
        g = int(os.environ.get("IGGT_KV_GROUPS", "1")) if kv_groups is None else int(kv_groups)
        if g not in (1, 2, 4, 8, 16):
            raise ValueError("kv_groups must divide the 16 heads")
        self.kv_groups = g if self.active else 1
        self._streams: List = []
        self._events: List = []
        self.ctl = None   # graphs.SegmentedGraph while a forward is being captured: collectives become eager steps

    @property
    def active(self) -> bool:
        """True when the collectives of the sharded path have to be issued."""
        return self.world > 1 or self.force

    def local_views(self, S: int) -> Tuple[int, int]:
        return view_partition(S, self.world, self.rank)

    def _buf(self, name, shape, like):
        n = 1
        for s in shape:
            n *= s
        # keyed by (name, dtype): a block that fell to bf16 operands beside fp16 neighbours (layers/blocks.py) must not make
        # the shared buffer flip its dtype -- and with it reallocate and invalidate every captured graph -- twice per forward
        name = (name, like.dtype)
        cur = self._bufs.get(name)
        if cur is None or cur.numel() < n or cur.dtype != like.dtype or cur.device != like.device:
            if cur is not None:
                from . import graphs

                graphs.buffers_changed()    # captured graph segments (and their eager steps) point into the old buffer
            cur = torch.empty(n, dtype=like.dtype, device=like.device)
            self._bufs[name] = cur
        return cur[:n].view(*shape)

    def all_gather_kv(self, kv_local: torch.Tensor, stats: Optional[torch.Tensor] = None):
        """kv_local [T_l, 2C] (contiguous) -> [world*T_l, 2C], rank-major = view-major token order.  With `stats` (this
        rank's 32 q/k norm maxima, fp32 [32]) returns (kv_all, stats_all [world, 32]): the key bounds travel with the keys."""
        assert kv_local.is_contiguous()
        out = self._buf("kv_all", (self.world * kv_local.shape[0], kv_local.shape[1]), kv_local)
        if stats is None:
            self._step(lambda: self._gather(out, kv_local))
            return out
        src, sout = self._stats_bufs(stats)

        def both():
            self._gather(out, kv_local)
            self._gather(sout, src)

        self._step(both)
        return out, sout

    def _stats_bufs(self, stats):
        assert stats.dtype == torch.float32 and stats.numel() == 32
        src = self._buf("kv_stats_in", (32,), stats)
        src.copy_(stats.reshape(32))                 # own copy: the caller's buffer is rewritten by the next block
        return src, self._buf("kv_stats_all", (self.world, 32), stats)

    def all_gather_kv_begin(self, kv_local: torch.Tensor, stats: Optional[torch.Tensor] = None):
        """Start the K/V all-gather without making the compute stream wait for it: returns (kv_all, finish) -- or, with
        `stats` (fp32 [32], see all_gather_kv), (kv_all, stats_all, finish) -- where `finish()` must be called before the
        gathered buffers are read.  Between the two calls the caller runs the part of the global attention that only needs
        this rank's own keys (layers/blocks.py), hiding that much of the transport.  Under graph capture both halves are
        eager steps between graph segments."""
        assert kv_local.is_contiguous()
        out = self._buf("kv_all", (self.world * kv_local.shape[0], kv_local.shape[1]), kv_local)
        src, sout = self._stats_bufs(stats) if stats is not None else (None, None)
        state = {}

        def gather_async(dst, x):
            if self._flat():
                return dist.all_gather_into_tensor(dst, x, group=self.group, async_op=True)
            return dist.all_gather(list(dst.view(self.world, *x.shape).unbind(0)), x, group=self.group, async_op=True)

        def start():
            state["work"] = [gather_async(out, kv_local)] + ([gather_async(sout, src)] if src is not None else [])

        def finish():
            for w in state.pop("work"):
                w.wait()                      # RCCL: the current stream waits for the collective; gloo: the host does

        self._step(start)
        if stats is None:
            return out, (lambda: self._step(finish))
        return out, sout, (lambda: self._step(finish))

    def _step(self, fn):
        """Run a collective now; under graph capture it is recorded as an eager step between two graph segments."""
        return fn() if self.ctl is None else self.ctl.eager(fn)

    def _flat(self) -> bool:
        """RCCL ("nccl"): one flat all_gather_into_tensor.  gloo (host-logic tests, the single-GPU two-rank test) lacks the
        flat variant for some tensors, so it always takes the list form.  The variant is a property of the BACKEND, chosen
        once: a failing collective must surface as an error, never be re-issued as a different collective on one rank."""
        return dist.get_backend(self.group) == "nccl"

    def _gather(self, out, x):
        if self._flat():
            dist.all_gather_into_tensor(out, x, group=self.group)
        else:
            dist.all_gather(list(out.view(self.world, *x.shape).unbind(0)), x, group=self.group)

    def gather_kv_groups(self, kv_local: torch.Tensor):
        """kv_local [G, T_l, D] (contiguous, head-group layout) -> list of G (work, kv_all_g [world*T_l, D]).
        The G collectives are issued asynchronously, in order; `work.wait()` makes the CURRENT stream wait for that
        collective only (RCCL) -- the caller overlaps the attention of group g with the transport of g+1.."""
        assert kv_local.dim() == 3 and kv_local.is_contiguous()
        G, Tl, D = kv_local.shape
        out = self._buf("kv_groups", (G, self.world * Tl, D), kv_local)
        handles = []
        for g in range(G):
            if self._flat():
                work = dist.all_gather_into_tensor(out[g], kv_local[g], group=self.group, async_op=True)
            else:
                work = dist.all_gather(list(out[g].view(self.world, Tl, D).unbind(0)), kv_local[g], group=self.group,
                                       async_op=True)
            handles.append((work, out[g]))
        return handles

    def side_stream(self, g: int):
        """Per-head-group CUDA stream (created lazily on the current device)."""
        while len(self._streams) <= g:
            self._streams.append(torch.cuda.Stream())
        return self._streams[g]

    def event(self, i: int):
        while len(self._events) <= i:
            self._events.append(torch.cuda.Event())
        return self._events[i]

    def agree_any(self, flags: List[int], device) -> List[int]:
        """Element-wise OR of a small list of 0 / 1 host flags over all ranks (one int32 all-reduce MAX on `device`, read back:
        synchronises).  Host decisions that change WHICH launches / how many warm-up forwards a rank issues must be identical
        on every rank -- every forward issues collectives (models/aggregator.py _apply_guard_snapshot)."""
        t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32, device=device)
        if self.world > 1 or self.force:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t.tolist()

    def all_gather_rows(self, x_local: torch.Tensor) -> torch.Tensor:
        """[n_l, ...] -> [world*n_l, ...] (camera tokens, small outputs)."""
        src = self._buf("rows_in", tuple(x_local.shape), x_local)
        src.copy_(x_local)
        out = self._buf("rows_out", (self.world * x_local.shape[0],) + tuple(x_local.shape[1:]), x_local)
        self._step(lambda: self._gather(out, src))
        return out
