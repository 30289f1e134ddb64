"""hipGraph replay of the forward pass (one process per GPU; matters for the per-rank forward of a multi-GPU run).

A 32-view forward is ~2 500 kernel launches.  On one GPU they are hidden behind 380 ms of kernels, but the per-rank
forward of an 8-GPU run (4 views, ~65 ms of kernels) has 19-75 ms of Python + ctypes launch time next to it depending on
the host (probes/emulate_rank.py: 0.6 ms with graphs).  Every entry point of libiggt_hip.so only enqueues work on the
stream it is given (include/iggt_hip.h), so the whole forward can be captured once per input shape and replayed with a
handful of host calls.

Collectives are NOT captured: RCCL all-gathers stay ordinary eager calls between graph segments (`SegmentedGraph.eager`),
so the capture never depends on the collective library's graph support and a communicator error surfaces as a normal
exception.  A sharded 24-block forward becomes 26 graph segments around 24 K/V all-gathers and the camera-token gather
(50 when the gather overlaps the own-key attention: its start and its completion are separate eager steps).

Contract of a graphed forward (same as any CUDA/HIP-graph runtime): inputs are copied into a static buffer, outputs live
in static buffers that the NEXT call overwrites -- clone what must survive.  Graphs are keyed by input shape / operand
format and dropped when parameters are reloaded (`load_state_dict`) or `reset()` is called.
"""
from typing import Callable, List, Tuple

import torch

# A captured graph holds RAW POINTERS into buffers it does not own: the block engine's workspaces (layers/blocks.py
# Workspace), the K/V gather buffers (dist.py ViewShard), the 16-bit weight packs of every module.  Each of those owners bumps
# this counter whenever it frees / replaces a buffer (`buffers_changed()`), every cache entry remembers the value at which it
# was captured, and `GraphCache.run` re-captures an entry whose value is stale instead of replaying into memory the allocator
# may have handed to someone else (sequence that used to corrupt silently: capture shape A, capture the larger shape B -- the
# workspaces grow --, replay A; or operand format f16 -> bf16 -> f16, which re-packs every weight).
_ALLOC_GENERATION = [0]


def buffers_changed() -> None:
    """Called by every owner of device buffers that graphs may point into, when it reallocates or drops one."""
    _ALLOC_GENERATION[0] += 1


def alloc_generation() -> int:
    return _ALLOC_GENERATION[0]


class SegmentedGraph:
    """Capture `fn(ctl)` as a sequence of hipGraph segments separated by the eager steps `fn` requests via ctl.eager()."""

    def __init__(self):
        self.steps: List[Tuple[str, object]] = []
        self._pool = None
        self._cur = None
        self._stream = None
        self.result = None
        self.capturing = False

    # -- capture -------------------------------------------------------------------------------------------------------
    def _begin(self):
        self._cur = torch.cuda.CUDAGraph()
        # thread_local: an asynchronous collective started by an eager step may still be progressing on the transport's own
        # thread / stream while the next segment is captured (gloo copies through the host; RCCL only launches kernels);
        # this thread itself issues nothing but kernel launches and memsets between begin and end
        self._cur.capture_begin(pool=self._pool, capture_error_mode="thread_local")

    def _end(self):
        self._cur.capture_end()
        self.steps.append(("graph", self._cur))
        self._cur = None

    def capture(self, fn: Callable[["SegmentedGraph"], object]):
        assert not self.steps, "already captured"
        self._pool = torch.cuda.graph_pool_handle()
        self._stream = torch.cuda.Stream()
        self._stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
        with torch.cuda.stream(self._stream):
            self.capturing = True
            self._begin()
            try:
                self.result = fn(self)
            except BaseException:
                self.capturing = False
                try:
                    self._cur.capture_end()
                except Exception:  # noqa: BLE001
                    pass
                self.steps.clear()
                raise
            self._end()
            self.capturing = False
        torch.cuda.current_stream().wait_stream(self._stream)
        torch.cuda.synchronize()
        return self.result

    def eager(self, step: Callable[[], object]):
        """Called by the captured function at a point that must stay an ordinary call (a collective): closes the
        current segment, runs `step` now (on the capture stream, not captured) and opens the next segment.  `step` must
        only touch buffers that exist before the capture (it is re-executed verbatim by every replay)."""
        if not self.capturing:
            return step()
        self._end()
        out = step()
        self.steps.append(("eager", step))
        self._begin()
        return out

    # -- replay --------------------------------------------------------------------------------------------------------
    def replay(self):
        for kind, obj in self.steps:
            if kind == "graph":
                obj.replay()
            else:
                obj()
        return self.result

    @property
    def num_segments(self) -> int:
        return sum(1 for k, _ in self.steps if k == "graph")


class GraphCache:
    """Per-model cache: input signature -> (static input buffer, SegmentedGraph, allocation generation at capture)."""

    def __init__(self):
        self._graphs = {}
        self.captures = 0    # number of captures so far (tests / reports)

    def reset(self):
        self._graphs.clear()

    def run(self, key, images: torch.Tensor, forward: Callable[[torch.Tensor, SegmentedGraph], object],
            fixed_warmups: int = 0):
        """fixed_warmups > 0: exactly that many eager warm-up forwards in front of a capture.  A view-sharded run needs it: every
        forward issues collectives, so all ranks must run the SAME number of them, while "did a decision change" is rank-local."""
        entry = self._graphs.get(key)
        if entry is not None and entry[2] != alloc_generation():
            # some buffer this graph may point into was reallocated since the capture: every entry of that age is unsafe
            for k in [k for k, e in self._graphs.items() if e[2] != alloc_generation()]:
                del self._graphs[k]
            entry = None
        if entry is None:
            static_in = images.clone()
            # eager warm-ups: weight packs, workspaces, RCCL set-up -- and the host-side decisions a replay can never revisit
            # (models/aggregator.py _apply_guard_snapshot: which attention call sites issue the estimated-shift launches).  Each
            # warm-up is followed by a device synchronisation so that the guard snapshot it ends with HAS landed when the next
            # one looks at it (back to back the second warm-up started while the device was still running the first, the
            # snapshot was not there yet, and the captured graph stayed on the round-3 sequence: ADVICE r4); a warm-up that
            # changed a decision or a buffer (allocation generation moved) is followed by another, up to four.
            for i in range(fixed_warmups or 4):
                gen = alloc_generation()
                forward(static_in, None)
                torch.cuda.synchronize()
                if not fixed_warmups and i >= 1 and gen == alloc_generation():
                    break
            g = SegmentedGraph()
            g.capture(lambda ctl: forward(static_in, ctl))
            self.captures += 1
            # the warm-up (and only the warm-up: a capture that reallocated would have pointed into freed memory already)
            # may have grown workspaces; entries captured earlier notice through their older generation
            entry = self._graphs[key] = (static_in, g, alloc_generation())
        static_in, g, _ = entry
        static_in.copy_(images)
        return g.replay()
