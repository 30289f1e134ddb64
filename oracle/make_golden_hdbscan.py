#!/usr/bin/env python3
"""HDBSCAN parity fixture at the demo's scale, on MODEL features (build container only; TEST INFRASTRUCTURE).

    python oracle/make_golden_hdbscan.py   ->   tests/golden/hdbscan_demo7_part_feat.pt

What the reference's demo clusters (demo.py:365-400, iggt/utils/misc.py:81-170): the L2-normalised 8-channel `part_feat` of all
views, every pixel one sample, with min_samples = 100, min_cluster_size = 500, cluster_selection_epsilon = 0.06
(demo.py:78-83).  The reference calls a library estimator there (cuml's or the `hdbscan` package's, misc.py:19-22); neither is
installed here, so the labels of this fixture come from scikit-learn's HDBSCAN (same algorithm, Campello et al. 2013 + the
epsilon rule of Malzer & Baum 2020), `algorithm="kd_tree"`, on

    part_feat of the REFERENCE model (oracle/ref_shim.py, stress weights) on the demo7 photographs at the demo's 504 x 336,
    all 4 views, every second pixel in both directions: 4 x 168 x 252 = 169 344 samples x 8 channels,

stored as fp16 (both estimators are then fed the SAME fp32 values, the widened fp16 ones).  The GPU test
(tests/test_post_gpu.py) runs iggt_official_amd/utils/hdbscan.py on these samples and compares the partitions."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_golden, ref_shim, weights  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "hdbscan_demo7_part_feat.pt")
PARAMS = dict(min_cluster_size=500, min_samples=100, cluster_selection_epsilon=0.06)


def main():
    assert ref_shim.available()
    torch.manual_seed(0)
    model = ref_shim.build_reference_iggt(fast_init=True)
    schema = make_golden.schema_of(model)
    sd = weights.fill_state_dict(schema, seed=0, mode="stress")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    paths = make_golden.stage_demo_images("demo7")
    imgs = make_golden.reference_loader()(paths, mode="resize", resize_target_size=(504, 336))
    t0 = time.time()
    with torch.no_grad():
        images = imgs[None]
        tokens, psi = model.aggregator(images)
        _pts, _conf, point_feat = model.point_head(tokens, images=images, patch_start_idx=psi, frames_chunk_size=None)
        ada, _pos = model.part_adaptor(tokens, images=images, patch_start_idx=psi)
        part = model.part_head(list(ada.values()), point_feature=point_feat, images=images, patch_start_idx=psi,
                               frames_chunk_size=None)                      # [1, S, 8, H, W]
    print(f"reference part_feat {tuple(part.shape)} in {time.time() - t0:.0f}s", flush=True)
    feat = torch.nn.functional.normalize(part[0].permute(0, 2, 3, 1), dim=-1)          # demo.py:368-369: [S, H, W, 8], unit rows
    sub = feat[:, ::2, ::2].reshape(-1, feat.shape[-1]).contiguous()
    x16 = sub.to(torch.float16)
    x = x16.float().numpy()
    from sklearn.cluster import HDBSCAN

    t0 = time.time()
    est = HDBSCAN(algorithm="kd_tree", allow_single_cluster=False, **PARAMS).fit(x)
    labels = est.labels_.astype(np.int32)
    dt = time.time() - t0
    ncl = int(labels.max()) + 1
    print(f"scikit-learn HDBSCAN on {x.shape}: {dt:.0f}s, {ncl} clusters, {int((labels < 0).sum())} noise samples, sizes "
          f"{np.bincount(labels[labels >= 0]).tolist() if ncl else []}", flush=True)
    import sklearn

    torch.save(dict(features=x16, labels=torch.from_numpy(labels), params=PARAMS, views=int(feat.shape[0]),
                    grid=(int(feat.shape[1] // 2 + feat.shape[1] % 2), int(feat.shape[2] // 2 + feat.shape[2] % 2)),
                    sklearn=sklearn.__version__, seconds_sklearn=dt, scene="demo7", size=(336, 504)), OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
