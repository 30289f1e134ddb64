"""BaseTrackerPredictor -- reference iggt/heads/track_modules/base_track_predictor.py:17-209, on HIP kernels.

Iterative point tracker (CoTracker / VGGSfM lineage): query features sampled in frame 0, then `iters` rounds of
(correlation pyramid sampling -> corr MLP -> update transformer -> coordinate and feature update).  State is kept
track-major ([N, S, .], row = track * S + frame), which is the layout the reference permutes to before every block.
All arithmetic is fp32 on HIP kernels (csrc/track.hip, csrc/smallops.hip); torch only allocates and copies."""
import torch
import torch.nn as nn

from ... import _C
from .blocks import CorrBlock, EfficientUpdateFormer
from .modules import Mlp, _p
from .utils import sample_features4d, sincos_tables


class BaseTrackerPredictor(nn.Module):
    def __init__(self, stride=1, corr_levels=5, corr_radius=4, latent_dim=128, hidden_size=384, use_spaceatt=True,
                 depth=6, max_scale=518, predict_conf=True):
        super().__init__()
        self.stride = stride
        self.latent_dim = latent_dim
        self.corr_levels = corr_levels
        self.corr_radius = corr_radius
        self.hidden_size = hidden_size
        self.max_scale = max_scale
        self.predict_conf = predict_conf
        self.flows_emb_dim = latent_dim // 2
        self.corr_mlp = Mlp(in_features=corr_levels * (corr_radius * 2 + 1) ** 2, hidden_features=hidden_size,
                            out_features=latent_dim)
        self.transformer_dim = latent_dim * 3 + 4
        self.query_ref_token = nn.Parameter(torch.randn(1, 2, self.transformer_dim))
        self.updateformer = EfficientUpdateFormer(space_depth=depth if use_spaceatt else 0, time_depth=depth,
                                                  input_dim=self.transformer_dim, hidden_size=hidden_size,
                                                  output_dim=latent_dim + 2, mlp_ratio=4.0, add_space_attn=use_spaceatt)
        self.fmap_norm = nn.LayerNorm(latent_dim)
        self.ffeat_norm = nn.GroupNorm(1, latent_dim)
        self.ffeat_updater = nn.Sequential(nn.Linear(latent_dim, latent_dim), nn.GELU())
        self.vis_predictor = nn.Sequential(nn.Linear(latent_dim, 1))
        if predict_conf:
            self.conf_predictor = nn.Sequential(nn.Linear(latent_dim, 1))

    # ------------------------------------------------------------------------------------------------------------
    # The forward of the reference (base_track_predictor.py:87-209) in three pieces, so that a single refinement
    # iteration can be checked against the oracle from a given state (tests/test_track_gpu.py).
    def prepare(self, query_points, fmaps_nhwc, down_ratio=1):
        """:100-126.  query_points [1, N, 2] (x, y); fmaps_nhwc [S, HH, WW, C] -> state dict (track-major tensors)."""
        if not fmaps_nhwc.is_cuda:
            raise _C.HipExtensionError("track head runs on HIP kernels only (no CPU fallback)")
        B, N, D = query_points.shape
        assert D == 2, "Input points must be 2D coordinates"
        if B != 1:
            raise NotImplementedError("one scene at a time (IGGT.forward loops over B)")
        S, HH, WW, C = fmaps_nhwc.shape
        dev = fmaps_nhwc.device
        fm = fmaps_nhwc.float().contiguous()
        normed = torch.empty_like(fm)
        _C.layernorm(fm.view(-1, C), _p(self.fmap_norm.weight), _p(self.fmap_norm.bias), normed.view(-1, C),
                     self.fmap_norm.eps)                                                       # :100-101
        q = query_points[0].to(dev).float()
        if down_ratio > 1:
            q = q / float(down_ratio)                                                          # :107-108
        q = (q / float(self.stride)).contiguous()                                              # :110, [N, 2]
        qfeat = sample_features4d(normed[0], q)                                                # :117, [N, C]
        tabx, taby = sincos_tables(self.transformer_dim, (HH, WW), dev)
        return dict(N=N, S=S, C=C,
                    coords=q[:, None, :].expand(N, S, 2).contiguous(),                         # :114, [N, S, 2]
                    feats=qfeat[:, None, :].expand(N, S, C).contiguous(),                      # :120, [N, S, C]
                    qfeat=qfeat,
                    corr=CorrBlock(normed, num_levels=self.corr_levels, radius=self.corr_radius),   # :124
                    pos=_C.track_posemb(tabx, taby, q),                                        # :152-154 (coords[:, 0] never moves)
                    ref=self.query_ref_token.detach().float()[0].contiguous(),                 # [2, D]: frame 0 | others
                    out_mul=float(self.stride * (down_ratio if down_ratio > 1 else 1)), fc_buf=None)

    def refine(self, st):
        """One iteration, :129-195: updates st["coords"] / st["feats"] in place, returns the prediction [1, S, N, 2]."""
        N, S, C = st["N"], st["S"], st["C"]
        coords, feats = st["coords"], st["feats"]
        feats2 = feats.view(N * S, C)
        st["fc_buf"] = st["corr"].corr_sample(feats, coords, out=st["fc_buf"])                 # :134
        fc = self.corr_mlp.forward_rows(st["fc_buf"])                                          # :138, [N*S, C]
        x = _C.track_tokens(coords, fc, feats2, st["pos"], st["ref"], self.flows_emb_dim, self.max_scale)   # :141-163
        delta = self.updateformer.forward_rows(x, N, S)                                        # :166, [N*S, C + 2]
        gn = _C.layernorm_rows(delta[:, 2:], _p(self.ffeat_norm.weight), _p(self.ffeat_norm.bias),
                               self.ffeat_norm.eps)                                            # GroupNorm(1, C) of a row
        lin = self.ffeat_updater[0]
        _C.linear_f32(gn, _p(lin.weight), _p(lin.bias), act="gelu", res=feats2, out=feats2)    # :183
        pred = torch.empty(S, N, 2, dtype=torch.float32, device=coords.device)
        _C.track_update(coords, delta, pred, st["out_mul"])                                    # :187-195
        st["delta"] = delta
        return pred[None]

    def heads(self, st, apply_sigmoid=True):
        """:197-207 -> vis [1, S, N], conf [1, S, N] | None."""
        feats2 = st["feats"].view(st["N"] * st["S"], st["C"])

        def head(seq):
            lin = seq[0]
            v = _C.linear_f32(feats2, _p(lin.weight), _p(lin.bias), act="sigmoid" if apply_sigmoid else None)
            return v.view(st["N"], st["S"]).t().contiguous()[None]

        return head(self.vis_predictor), (head(self.conf_predictor) if self.predict_conf else None)

    def forward(self, query_points, fmaps=None, iters=6, return_feat=False, down_ratio=1, apply_sigmoid=True,
                fmaps_nhwc=None):
        """query_points [1, N, 2] (x, y); fmaps [1, S, C, HH, WW] (reference layout) or fmaps_nhwc [S, HH, WW, C]
        -> (list of iters x [1, S, N, 2], vis [1, S, N], conf [1, S, N] | None)."""
        if fmaps_nhwc is None:
            if fmaps.shape[0] != 1:
                raise NotImplementedError("one scene at a time (IGGT.forward loops over B)")
            fmaps_nhwc = fmaps[0].permute(0, 2, 3, 1)
        st = self.prepare(query_points, fmaps_nhwc, down_ratio)
        coord_preds = [self.refine(st) for _ in range(iters)]
        vis_e, conf_e = self.heads(st, apply_sigmoid)
        if return_feat:
            return coord_preds, vis_e, st["feats"].permute(1, 0, 2)[None], st["qfeat"][None], conf_e
        return coord_preds, vis_e, conf_e
