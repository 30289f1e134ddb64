"""Timing of the track head at BASELINE-size feature maps (259 x 259 x 128 per view, random data):
python probes/track_bench.py [S N] ...  -> per-phase milliseconds of one refinement iteration and of the whole tracker."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from iggt.heads.track_head import TrackHead  # noqa: E402
from iggt_official_amd import _C  # noqa: E402


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n, out


def main():
    pairs = [(8, 1024), (32, 1024), (8, 4096)]
    if len(sys.argv) > 2:
        pairs = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    torch.manual_seed(0)
    with torch.device("cuda"):
        th = TrackHead(dim_in=2048).eval()
    tr = th.tracker
    for S, N in pairs:
        fm = torch.randn(S, 259, 259, 128, device="cuda")
        q = torch.rand(1, N, 2, device="cuda") * 517
        with torch.no_grad():
            t_all, _ = timed(lambda: tr(query_points=q, fmaps_nhwc=fm, iters=4), 2)
            t_prep, st = timed(lambda: tr.prepare(q, fm), 2)
            feats, coords = st["feats"], st["coords"]
            t_corr, fc = timed(lambda: st["corr"].corr_sample(feats, coords))
            t_mlp, fcm = timed(lambda: tr.corr_mlp.forward_rows(fc))
            t_tok, x = timed(lambda: _C.track_tokens(coords, fcm, feats.view(N * S, 128), st["pos"], st["ref"], 64, 518.0))
            t_uf, _ = timed(lambda: tr.updateformer.forward_rows(x, N, S))
            blk = tr.updateformer.time_blocks[0]
            tok = torch.randn((N + 64) * S, 384, device="cuda")
            t_time, _ = timed(lambda: blk.forward_rows(tok, N + 64, S, S, 1))
            p2v = tr.updateformer.space_point2virtual_blocks[0]
            t_p2v, _ = timed(lambda: p2v.forward_rows(tok[:N * S], tok[N * S:], S, N, 64, (1, S), (1, S)))
        print(f"S={S} N={N}: tracker(4 iters) {t_all:.2f} ms | prepare {t_prep:.2f} | per iteration: corr {t_corr:.3f} "
              f"corr_mlp {t_mlp:.3f} tokens {t_tok:.3f} update_former {t_uf:.2f} (time block {t_time:.3f}, "
              f"point<-virtual block {t_p2v:.3f})", flush=True)


if __name__ == "__main__":
    main()
