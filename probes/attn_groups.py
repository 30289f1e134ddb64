"""Per-rank global attention of an 8-GPU run (Nq=5496, Nk=43968): one 16-head launch vs 4 head-group launches
(serial on one stream / concurrent on 4 streams), flat vs head-group K/V layout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iggt_official_amd import _C

Nq, Nk, H, C = 5496, 43968, 16, 1024
dt = torch.float16
qkv = (torch.randn(Nq, 3 * C, device="cuda") * 0.5).to(dt)
kv = (torch.randn(Nk, 2 * C, device="cuda") * 0.5).to(dt)
ao = torch.empty(Nq, C, dtype=dt, device="cuda")
G, hg = 4, 4
D = 2 * hg * 64
kvg = (torch.randn(G, Nk, D, device="cuda") * 0.5).to(dt)
streams = [torch.cuda.Stream() for _ in range(G)]


def flat(tile=0):
    _C.flash_attn_d64(qkv, kv, kv[:, C:], ao, 1, H, Nq, Nk, 0, 3 * C, 0, 2 * C, 0, 2 * C, 0, C, 0.125, tile)


def group(g, tile=0, src=None):
    k = kvg[g] if src is None else src
    _C.flash_attn_d64(qkv[:, g * hg * 64:], k, k[:, hg * 64:], ao[:, g * hg * 64:], 1, hg, Nq, Nk, 0, 3 * C, 0, D, 0, D,
                      0, C, 0.125, tile)


def serial(tile=0):
    for g in range(G):
        group(g, tile)


def conc(tile=0):
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    dones = []
    for g in range(G):
        streams[g].wait_event(ev)
        with torch.cuda.stream(streams[g]):
            group(g, tile)
            e = torch.cuda.Event(); e.record(streams[g]); dones.append(e)
    for e in dones:
        main.wait_event(e)


def bench(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


fl = 4.0 * Nq * Nk * C
for name, fn in (("one launch auto", flat), ("one launch 6256", lambda: flat(6256)), ("one launch 5128", lambda: flat(5128)),
                 ("4 groups serial auto", serial), ("4 groups serial 6256", lambda: serial(6256)),
                 ("4 groups concurrent auto", conc), ("4 groups concurrent 6256", lambda: conc(6256)),
                 ("4 groups concurrent 5256", lambda: conc(5256))):
    ms = bench(fn)
    print(f"{name:28s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s", flush=True)
