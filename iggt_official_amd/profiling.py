"""Live kernel timing with HIP events on the launch stream (used by bench.py for the roofline leg).

`torch.cuda.Event` records on torch's *current* stream, which is the stream every libiggt_hip kernel
is launched on (_C._stream()), so the pair brackets exactly that kernel."""
import torch

_ACTIVE = {}
_TAG = [None]     # which part of the model the launches belong to ("part": the instance-feature branch), see `tagged`


def enable(name):
    _ACTIVE[name] = []


def disable(name):
    return _ACTIVE.pop(name, None)


def active(name):
    return name in _ACTIVE


class tagged:
    """with tagged("part"): ...  -- every region recorded inside carries the tag (bench.py separates the part branch's
    convolutions from the DPT heads' with it)."""

    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        self.prev, _TAG[0] = _TAG[0], self.tag
        return self

    def __exit__(self, *a):
        _TAG[0] = self.prev


class region:
    def __init__(self, name, meta=None):
        self.rec = _ACTIVE.get(name)
        self.meta = meta

    def __enter__(self):
        if self.rec is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if self.rec is not None:
            self.e.record()
            self.rec.append((self.s, self.e, self.meta, _TAG[0]))


def summarize(records):
    """[(start, end, meta, tag)] -> list of (ms, meta, tag); call after torch.cuda.synchronize()."""
    return [(s.elapsed_time(e), m, t) for s, e, m, t in records]
