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
    """Per-model cache: input signature -> (static input buffer, SegmentedGraph)."""

    def __init__(self):
        self._graphs = {}

    def reset(self):
        self._graphs.clear()

    def run(self, key, images: torch.Tensor, forward: Callable[[torch.Tensor, SegmentedGraph], object]):
        entry = self._graphs.get(key)
        if entry is None:
            static_in = images.clone()
            forward(static_in, None)                      # eager warm-up: weight packs, workspaces, RCCL set-up
            forward(static_in, None)
            g = SegmentedGraph()
            g.capture(lambda ctl: forward(static_in, ctl))
            entry = self._graphs[key] = (static_in, g)
        static_in, g = entry
        static_in.copy_(images)
        return g.replay()
