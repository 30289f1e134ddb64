"""TrackHead -- reference iggt/heads/track_head.py:12-109: DPT feature extractor (128 channels at half resolution, no
position embedding) + BaseTrackerPredictor.  Runs only when `query_points` is given (iggt/models/vggt.py:220-227).

Everything on HIP kernels: the feature extractor is heads/dpt_head.py (split-bf16 MFMA convolutions, NHWC), the tracker
is heads/track_modules/ (fp32).  The feature maps stay NHWC between the two (the reference's [B, S, C, H, W] view is
available through `feature_extractor(...)` for API parity).  The reference's default `frames_chunk_size=12` path raises
for S > 12 (dpt_head.py:170-188 unpacks a single tensor into three); here any S works."""
import torch.nn as nn

from .dpt_head import DPTHead
from .track_modules.base_track_predictor import BaseTrackerPredictor


class TrackHead(nn.Module):
    def __init__(self, dim_in, patch_size=14, features=128, iters=4, predict_conf=True, stride=2, corr_levels=7,
                 corr_radius=4, hidden_size=384):
        super().__init__()
        self.patch_size = patch_size
        self.feature_extractor = DPTHead(dim_in=dim_in, patch_size=patch_size, features=features, use_point_feat=True,
                                         for_tracker=True, down_ratio=2, pos_embed=False)
        self.tracker = BaseTrackerPredictor(latent_dim=features, predict_conf=predict_conf, stride=stride,
                                            corr_levels=corr_levels, corr_radius=corr_radius, hidden_size=hidden_size)
        self.iters = iters

    def forward(self, aggregated_tokens_list, images, patch_start_idx, query_points=None, iters=None, gather=None):
        """-> (list of iters x [1, S, N, 2] pixel coordinates, vis [1, S, N], conf [1, S, N]).
        gather: optional callable [S_local, ...] -> [S, ...] (view-sharded runs: every rank tracks over all views)."""
        fm = self.feature_extractor.features_nhwc(aggregated_tokens_list, images, patch_start_idx)   # [S, H/2, W/2, C]
        if gather is not None:
            fm = gather(fm)
        if iters is None:
            iters = self.iters
        return self.tracker(query_points=query_points, fmaps_nhwc=fm, iters=iters)
