#!/usr/bin/env python3
"""Would two half-batches on two streams fill each other's launch tails?  The 24 frame blocks (and the 24 DINOv2 blocks) act per view,
so the 32-view token matrix can run as two independent 16-view chains on two HIP streams; every GEMM / attention launch of one chain
ends in a partial round of workgroups that the other chain's launches could fill.  Times 24 frame blocks at 32 views @ 518^2:
  A  one chain over all rows (what the aggregator does),
  B  two chains of 16 views each on two streams (own workspaces),
  C  the two half chains back to back on ONE stream (what the split costs by itself: smaller grids).
Usage: python probes/two_stream_halves.py [views]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_gpu_model  # noqa: E402

from iggt_official_amd.layers.blocks import Workspace  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
gh = gw = 37
P, C = 5 + gh * gw, 1024
T = S * P
model = build_gpu_model("stress", 0)
agg = model.aggregator
dev = torch.device("cuda")
cos, sin = agg.rope.tables(64, max(gh, gw), dev)
geom = dict(P=P, gw=gw, patch_start=5, cos=cos, sin=sin)
x0 = torch.randn(T, C, device=dev)
ws = [Workspace(), Workspace(), Workspace()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
half = (S // 2) * P


def chain(x, w, views, blocks):
    for b in blocks:
        b.forward_inplace(x, w, batch=views, tokens=P, rope_geom=geom if b.attn.qk_norm else None)


def run_a(x, blocks):
    chain(x, ws[0], S, blocks)


def run_b(x, blocks):
    cur = torch.cuda.current_stream()
    for i, st in enumerate(streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            chain(x[i * half:(i + 1) * half] if i == 0 else x[half:], ws[1 + i], S // 2 if i == 0 else S - S // 2, blocks)
    for st in streams:
        cur.wait_stream(st)


def run_c(x, blocks):
    chain(x[:half], ws[1], S // 2, blocks)
    chain(x[half:], ws[2], S - S // 2, blocks)


def timed(fn, blocks, reps=5):
    x = x0.clone()
    fn(x, blocks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(x, blocks)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, blocks in (("24 frame blocks", list(agg.frame_blocks)), ("24 DINOv2 blocks", list(agg.patch_embed.blocks))):
    a, b, c = timed(run_a, blocks), timed(run_b, blocks), timed(run_c, blocks)
    a2, b2 = timed(run_a, blocks), timed(run_b, blocks)
    print(f"{name}, {S} views: one chain {a:.2f} / {a2:.2f} ms | two streams x {S // 2} views {b:.2f} / {b2:.2f} ms | two halves on one stream {c:.2f} ms")
