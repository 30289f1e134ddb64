"""EfficientUpdateFormer and CorrBlock of the tracker -- reference iggt/heads/track_modules/blocks.py.

EfficientUpdateFormer: one fp32 token matrix [(N + 64) * S, 384] (N tracks + 64 virtual tracks, row = token * S + frame)
lives through the 6 x (time attention, virtual <- points, virtual self-attention, points <- virtual) blocks and is
updated in place by the HIP kernels of modules.py; "time" and "space" attention differ only in the strides handed to the
attention kernel.

CorrBlock: the reference materialises, per pyramid level, the full correlation volume [S, N, H_l * W_l] with a matmul and
then grid_samples (2r+1)^2 values out of each map (blocks.py:189-241).  Sampling is linear, so the same numbers are
bilinear mixes of the dot products of the track feature with the (2r+2)^2 feature vectors under the window:
`iggt_track_corr_f32` reads 100 x 512 B per (track, frame, level) and never builds the volume."""
import torch
import torch.nn as nn

from ... import _C
from .modules import AttnBlock, CrossAttnBlock, _ln, _p


class EfficientUpdateFormer(nn.Module):
    def __init__(self, space_depth=6, time_depth=6, input_dim=320, hidden_size=384, num_heads=8, output_dim=130,
                 mlp_ratio=4.0, add_space_attn=True, num_virtual_tracks=64):
        super().__init__()
        self.out_channels = 2
        self.num_heads = num_heads
        self.hidden_size = hidden_size
        self.add_space_attn = add_space_attn
        self.input_norm = nn.LayerNorm(input_dim)
        self.input_transform = nn.Linear(input_dim, hidden_size, bias=True)
        self.output_norm = nn.LayerNorm(hidden_size)
        self.flow_head = nn.Linear(hidden_size, output_dim, bias=True)
        self.num_virtual_tracks = num_virtual_tracks
        self.virual_tracks = nn.Parameter(torch.randn(1, num_virtual_tracks, 1, hidden_size)) if add_space_attn else None
        self.time_blocks = nn.ModuleList([AttnBlock(hidden_size, num_heads, mlp_ratio=mlp_ratio,
                                                    attn_class=nn.MultiheadAttention) for _ in range(time_depth)])
        if add_space_attn:
            self.space_virtual_blocks = nn.ModuleList([AttnBlock(hidden_size, num_heads, mlp_ratio=mlp_ratio,
                                                                 attn_class=nn.MultiheadAttention)
                                                       for _ in range(space_depth)])
            self.space_point2virtual_blocks = nn.ModuleList([CrossAttnBlock(hidden_size, hidden_size, num_heads,
                                                                            mlp_ratio=mlp_ratio)
                                                             for _ in range(space_depth)])
            self.space_virtual2point_blocks = nn.ModuleList([CrossAttnBlock(hidden_size, hidden_size, num_heads,
                                                                            mlp_ratio=mlp_ratio)
                                                             for _ in range(space_depth)])
            assert len(self.time_blocks) >= len(self.space_virtual2point_blocks)
        self.initialize_weights()

    def initialize_weights(self):
        def _basic_init(module):
            if isinstance(module, nn.Linear):
                nn.init.xavier_uniform_(module.weight)
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)

        self.apply(_basic_init)
        nn.init.trunc_normal_(self.flow_head.weight, std=0.001)

    def forward_rows(self, x, N, S):
        """x [N * S, input_dim] fp32 (row = track * S + frame) -> flow [N * S, output_dim] (blocks.py:100-143, B = 1)."""
        if not x.is_cuda:
            raise _C.HipExtensionError("track head runs on HIP kernels only (no CPU fallback)")
        C, V = self.hidden_size, self.num_virtual_tracks if self.add_space_attn else 0
        init = _C.linear_f32(_ln(self.input_norm, x), _p(self.input_transform.weight), _p(self.input_transform.bias))
        tok = torch.empty((N + V) * S, C, dtype=torch.float32, device=x.device)
        tok[:N * S].copy_(init)
        if V:
            tok[N * S:].view(V, S, C).copy_(self.virual_tracks.detach().float().view(V, 1, C).expand(V, S, C))
        pts, vts = tok[:N * S], tok[N * S:]
        every = len(self.time_blocks) // len(self.space_virtual_blocks) if V else 0
        j = 0
        for i, blk in enumerate(self.time_blocks):
            blk.forward_rows(tok, N + V, S, S, 1)                            # along the frames of every (virtual) track
            if V and i % every == 0:
                # along the tracks of every frame: batch = frame (one row apart), token stride = S rows
                self.space_virtual2point_blocks[j].forward_rows(vts, pts, S, V, N, (1, S), (1, S))
                self.space_virtual_blocks[j].forward_rows(vts, S, V, 1, S)
                self.space_point2virtual_blocks[j].forward_rows(pts, vts, S, N, V, (1, S), (1, S))
                j += 1
        out = _ln(self.output_norm, pts, add=init)                            # output_norm(tokens + init_tokens)
        return _C.linear_f32(out, _p(self.flow_head.weight), _p(self.flow_head.bias))

    def forward(self, input_tensor, mask=None):
        """input_tensor [1, N, T, input_dim] -> (flow [1, N, T, output_dim], None)."""
        if mask is not None:
            raise NotImplementedError("attention masks are unused by IGGT's tracker")
        B, N, T, D = input_tensor.shape
        if B != 1:
            raise NotImplementedError("one scene at a time (IGGT.forward loops over B)")
        flow = self.forward_rows(input_tensor.reshape(N * T, D).float().contiguous(), N, T)
        return flow.view(1, N, T, -1), None


class CorrBlock:
    """Average-pooled feature pyramid + windowed correlation sampling (blocks.py:146-241) on NHWC maps."""

    def __init__(self, fmaps_nhwc, num_levels=4, radius=4, multiple_track_feats=False, padding_mode="zeros"):
        """fmaps_nhwc [S, H, W, C] fp32 contiguous (one scene)."""
        if multiple_track_feats or padding_mode != "zeros":
            raise NotImplementedError("IGGT's tracker uses one track feature and zero padding")
        if not fmaps_nhwc.is_cuda:
            raise _C.HipExtensionError("track head runs on HIP kernels only (no CPU fallback)")
        self.S, self.H, self.W, self.C = fmaps_nhwc.shape
        self.num_levels, self.radius = num_levels, radius
        self.fmaps_pyramid = [fmaps_nhwc.contiguous()]
        for _ in range(num_levels - 1):
            cur = self.fmaps_pyramid[-1]
            if cur.shape[1] < 2 or cur.shape[2] < 2:
                # F.avg_pool2d raises here too ("Output size is too small", blocks.py:176)
                raise RuntimeError(f"correlation pyramid of {num_levels} levels needs feature maps of at least "
                                   f"{2 ** (num_levels - 1)} pixels a side, got {self.H} x {self.W}")
            self.fmaps_pyramid.append(_C.avgpool2_nhwc(cur))
        self.width = num_levels * (2 * radius + 1) ** 2
        self.ld = (self.width + 31) // 32 * 32         # zero-padded rows: aligned loads, K % 32 for the MFMA GEMM path

    def corr_sample(self, targets, coords, out=None):
        """targets [N, S, C], coords [N, S, 2] (track-major, level-0 pixels) -> [N * S, ld] (columns >= width are 0)."""
        N = targets.shape[0]
        if out is None:
            out = torch.empty(N * self.S, self.ld, dtype=torch.float32, device=targets.device)
        return _C.track_corr(self.fmaps_pyramid, targets, coords, self.radius, out)
