"""IGGT / VGGT model API -- drop-in for `from iggt.models.vggt import IGGT, VGGT` (reference
iggt/models/vggt.py:14-230; caller: demo.py:35,102-121,195).

Same constructor, sub-module attribute names, state-dict keys (all 2 053 tensors) and
output dict (keys, shapes, fp32) as the reference; the forward runs on MI355X HIP kernels (aggregator, DPT / part /
adaptor / camera heads: csrc/*.hip behind include/iggt_hip.h).  There is no CPU path: inputs and parameters must live on
the GPU and libiggt_hip.so must be built, otherwise forward raises.  Like the reference (vggt.py:66,189) the forward runs
with autocast DISABLED whatever the caller's context (demo.py:193-195 calls the model under autocast(bf16)): the 16-bit
operand format of the trunk is the kernels' own (iggt_official_amd/precision.py), outputs are always fp32.

Deliberate differences, all documented in DESIGN.md:
  * S > 12 views works (the reference's chunked head path is broken there, appendix D.1): heads
    behave as the reference does unchunked, which is bit-identical to chunking where both exist;
  * `part_feat` needs H, W in 28N exactly like the reference (appendix D.2); for other sizes the
    reference raises inside the part head after having computed everything else -- here
    `IGGT(..., part_on_invalid_grid="skip")` (default "raise") returns the geometry outputs only;
  * `track_head` (only run when `query_points` is given, vggt.py:220-227): any number of views works (the reference's
    default chunked feature extractor raises for S > 12); a forward with query points runs eagerly even when
    `enable_graphs()` is on (the number of tracks is not part of a captured shape);
  * multi-GPU: `set_view_shard(ViewShard())` makes `forward` take this rank's slice of the views.
"""
from typing import Optional

import torch
import torch.nn as nn

try:  # keeps from_pretrained / save_pretrained of the reference class (vggt.py:4,132)
    from huggingface_hub import PyTorchModelHubMixin
except Exception:  # pragma: no cover
    class PyTorchModelHubMixin:  # type: ignore
        pass

import os

from .. import _C, precision, profiling
from ..dist import ViewShard
from ..graphs import GraphCache
from ..heads import convops
from ..heads.adaptor import SamProjector
from ..heads.camera_head import CameraHead
from ..heads.dpt_head import DPTHead
from ..heads.part_head import PartHead
from ..heads.track_head import TrackHead
from .aggregator import Aggregator


_CAMERA_STREAM = os.environ.get("IGGT_CAMERA_STREAM", "1") != "0"
# depth head beside the point head on a second side stream (see _Base._fork_head): "auto" = from 8 local views @ 518^2 worth of
# pixels on (measured, profiles/r03_head_streams_ab.txt: 8 views 64.7 -> 63.7 ms, 16 views 141.1 -> 139.4, 32 views 344.6 -> 343.0;
# 4 local views of an 8-GPU run 60.4 -> 61.3 ms, so not there), 0 / 1 = never / always
_HEAD_STREAMS = os.environ.get("IGGT_HEAD_STREAMS", "auto")
_HEAD_STREAMS_MIN_PIXELS = 8 * 500 * 500
# Frames per pass of the part head.  The reference's signature default is 8 (part_head.py:108; its chunked branch does not even run,
# SURVEY appendix D.1); every operator of the head is per frame, so the chunk size is purely a memory / launch-size choice: one
# pass over all frames up to this many pixels (32 views @ 532^2 fit: its 304^2 x 128-channel maps are 1.5 GB each), chunks beyond
# (64 views @ 1036^2: 9 frames per pass).  Measured at 32 x 532^2: 4 passes of 8 frames -> one pass of 32 (profiles/r06_*).
_PART_CHUNK_PIXELS = int(os.environ.get("IGGT_PART_CHUNK_PIXELS", str(32 * 560 * 560)))


class _Base(nn.Module, PyTorchModelHubMixin):
    def _init_runtime(self):
        self._cam_stream = None
        self._cam_pending = False
        self._head_stream = None
        self._head_pending = False
        self._graphs_on = False
        self._gcache = GraphCache()
        # packed weights are rebuilt when parameters change; captured graphs hold pointers to the old packs
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._gcache.reset())

    def set_view_shard(self, shard: Optional[ViewShard]):
        self.aggregator.shard = shard
        self._gcache.reset()
        return self

    def enable_graphs(self, on: bool = True):
        """Replay the forward as hipGraph segments (iggt_official_amd/graphs.py): one capture per input shape, then a
        handful of host calls per forward instead of ~2 500 launches -- what the per-rank forward of a multi-GPU run
        needs.  Outputs then live in static buffers that the next call overwrites (clone what must survive); call
        `reset_graphs()` after changing parameters in place."""
        self._graphs_on = bool(on)
        if not on:
            self._gcache.reset()
        return self

    def reset_graphs(self):
        self._gcache.reset()

    def _dispatch(self, images, run, query_points=None):
        """run(images[1,S,3,H,W]) -> dict, eagerly or through the graph cache."""
        if query_points is not None:
            return run(images, query_points)
        if not self._graphs_on:
            return run(images)
        shard = self.aggregator.shard
        # everything that selects WHICH kernels / collectives get captured is part of the key
        key = (tuple(images.shape), precision.operand_name(), precision.static_softmax(),
               precision.mean_compensation_sites(), precision.gather_overlap(), precision.debug_saturation(), precision.static_guard(),
               precision.attn_estimated_shift(), precision.escalation(),
               convops.PREC, convops.DPT_PREC, convops.PART_PREC,
               None if shard is None else (shard.rank, shard.world, shard.kv_groups, shard.force))

        def fwd(static_in, ctl):
            if shard is not None:
                shard.ctl = ctl
            try:
                return run(static_in)
            finally:
                if shard is not None:
                    shard.ctl = None

        # sharded: a rank-uniform number of warm-up forwards (each issues collectives; graphs.GraphCache.run)
        return self._gcache.run(key, images, fwd, fixed_warmups=3 if (shard is not None and shard.active) else 0)

    def _common(self, images, query_points):
        if images.dim() == 4:
            images = images.unsqueeze(0)
        if images.dim() != 5:
            raise ValueError(f"images must be [S,3,H,W] or [B,S,3,H,W], got {tuple(images.shape)}")
        if not images.is_cuda:
            raise _C.HipExtensionError("IGGT forward runs on the MI355X only: move model and images to 'cuda' "
                                       "(no CPU fallback; the CPU restatement lives in oracle/ for tests)")
        _C.load()
        return images.float().contiguous()

    @staticmethod
    def _query(query_points, batch):
        """reference vggt.py:59-60 / 192-193: [N, 2] -> [1, N, 2]; one set of points per scene."""
        if query_points is None:
            return None
        if query_points.dim() == 2:
            query_points = query_points.unsqueeze(0)
        if query_points.dim() != 3 or query_points.shape[-1] != 2 or query_points.shape[0] != batch:
            raise ValueError(f"query_points must be [N,2] or [B,N,2] with B = {batch}, got {tuple(query_points.shape)}")
        return query_points

    def _track(self, pred, tokens, images, psi, query_points):
        """reference vggt.py:220-227: track of the last iteration, visibility, confidence."""
        if query_points is None:
            return
        if self.track_head is None:
            raise ValueError("this model was built without a track head")
        shard = self.aggregator.shard
        gather = shard.all_gather_rows if (shard is not None and shard.active) else None
        track_list, vis, conf = self.track_head(tokens, images=images, patch_start_idx=psi, query_points=query_points,
                                                gather=gather)
        pred["track"] = track_list[-1]
        pred["vis"] = vis
        pred["conf"] = conf

    def _scenes(self, images, query_points):
        """B > 1 (reference vggt.py:149: images [B,S,3,H,W]): scenes are independent, so they run one after the other
        through the single-scene path and the outputs are concatenated along the batch dimension."""
        outs = []
        for b in range(images.shape[0]):
            o = self.forward(images[b], None if query_points is None else query_points[b:b + 1])
            if self._graphs_on:
                # a graphed forward returns its STATIC output buffers, which the next scene's replay overwrites
                o = {k: ([t.clone() for t in v] if isinstance(v, (list, tuple)) else v.clone()) for k, v in o.items()}
            outs.append(o)
        pred = {}
        for k, v in outs[0].items():
            if k == "pose_enc":
                pred[k] = [torch.cat([o[k][i] for o in outs], 0) for i in range(len(v))]
            else:
                pred[k] = torch.cat([o[k] for o in outs], 0)
        return pred

    def _camera(self, tokens_list):
        """Camera head on a SIDE stream: 16 block evaluations over S tokens against 1.6 GB of fp32 weights are bound by weight
        bandwidth on a few dozen workgroups, the dense heads that follow by the matrix pipe -- run side by side they overlap
        almost completely (-1.5 ms per rank-forward of an 8-GPU run, -2.7 ms at 32 views on one GPU).  The camera-token gather
        of a sharded run stays on the main stream (under graph capture it is an eager step between segments); `_join_camera`
        makes the main stream wait before the outputs are handed out.  IGGT_CAMERA_STREAM=0: in line."""
        shard = self.aggregator.shard
        cam = None
        if shard is not None and shard.active:
            local = tokens_list[-1][0, :, 0]                      # [S_local, 2C]
            # a private copy, taken on the main stream before the fork: the gathered rows live in the shard's reusable
            # `rows_out` buffer, which the track head's feature-map gather (main stream) rewrites -- and on its first call
            # reallocates -- while the camera head may still be reading it on the side stream
            cam = shard.all_gather_rows(local)[None].clone()       # [1, S, 2C]
        if not _CAMERA_STREAM:
            return self.camera_head(tokens_list, camera_tokens=cam)
        main = torch.cuda.current_stream()
        if self._cam_stream is None or self._cam_stream.device != main.device:
            self._cam_stream = torch.cuda.Stream(device=main.device)
        self._cam_stream.wait_stream(main)
        with torch.cuda.stream(self._cam_stream):
            pose = self.camera_head(tokens_list, camera_tokens=cam)
        self._cam_pending = True
        for t in pose:
            t.record_stream(main)                                  # allocated on the side stream, consumed on the caller's
        return pose

    def _join_camera(self):
        if self._cam_pending:
            torch.cuda.current_stream().wait_stream(self._cam_stream)
            self._cam_pending = False

    def _fork_head(self, head, tokens_list, images, psi):
        """Run `head` (the depth head) on a SECOND side stream while the caller goes on with the point head on its own.  The two
        DPT heads share nothing but their read-only inputs (per-head weight packs and workspaces; split-K / compensation scratch
        is per stream; the position tables are complete when they enter their cache), and a large part of a head is small maps --
        19^2 ... 74^2 at 4 local views is a few dozen workgroups per launch -- that leave most of the chip idle on their own.
        `_join_heads` makes the caller's stream wait before the outputs are handed out.  Captured like the camera head's stream
        (fork / join inside the graph).  Worth 1.5 % at 8 views, 1.2 % at 16, 0.5 % at 32; at the 4 local views of an 8-GPU run it LOSES 1.5 %
        (the launches are too short to gain from sharing the chip), hence the size rule.  IGGT_HEAD_STREAMS=0 / 1: never / always."""
        pixels = images.shape[-4] * images.shape[-2] * images.shape[-1]
        if _HEAD_STREAMS == "0" or (_HEAD_STREAMS != "1" and pixels < _HEAD_STREAMS_MIN_PIXELS):
            return head(tokens_list, images=images, patch_start_idx=psi)
        main = torch.cuda.current_stream()
        if self._head_stream is None or self._head_stream.device != main.device:
            self._head_stream = torch.cuda.Stream(device=main.device)
        self._head_stream.wait_stream(main)
        with torch.cuda.stream(self._head_stream):
            out = head(tokens_list, images=images, patch_start_idx=psi)
        self._head_pending = True
        for t in out:
            if torch.is_tensor(t):
                t.record_stream(main)                              # allocated on the side stream, consumed on the caller's
        return out

    def _join_heads(self):
        if self._head_pending:
            torch.cuda.current_stream().wait_stream(self._head_stream)
            self._head_pending = False


class VGGT(_Base):
    def __init__(self, img_size=518, patch_size=14, embed_dim=1024, only_train_adaptor=False):
        super().__init__()
        self.aggregator = Aggregator(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim)
        self.camera_head = CameraHead(dim_in=2 * embed_dim)
        self.point_head = DPTHead(dim_in=2 * embed_dim, output_dim=4, activation="inv_log", conf_activation="expp1",
                                  use_point_feat=False)
        self.depth_head = DPTHead(dim_in=2 * embed_dim, output_dim=2, activation="exp", conf_activation="expp1",
                                  use_point_feat=False)
        self.track_head = TrackHead(dim_in=2 * embed_dim, patch_size=patch_size)
        self._init_runtime()

    def _run(self, images, query_points=None):
        tokens, psi = self.aggregator(images)
        pred = {"pose_enc": self._camera(tokens)}
        pred["depth"], pred["depth_conf"] = self._fork_head(self.depth_head, tokens, images, psi)
        pred["world_points"], pred["world_points_conf"] = self.point_head(tokens, images=images, patch_start_idx=psi)
        self._track(pred, tokens, images, psi, query_points)
        self._join_heads()
        self._join_camera()
        pred["images"] = images
        return pred

    @torch.no_grad()
    def forward(self, images, query_points=None):
        with torch.amp.autocast("cuda", enabled=False):   # reference vggt.py:66
            images = self._common(images, query_points)
            query_points = self._query(query_points, images.shape[0])
            if images.shape[0] != 1:
                return self._scenes(images, query_points)
            return self._dispatch(images, self._run, query_points)


class IGGT(_Base):
    def __init__(self, img_size=518, patch_size=14, embed_dim=1024, only_train_adaptor=False,
                 part_on_invalid_grid: str = "raise"):
        super().__init__()
        self.aggregator = Aggregator(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim)
        self.camera_head = CameraHead(dim_in=2 * embed_dim)
        self.point_head = DPTHead(dim_in=2 * embed_dim, output_dim=4, activation="inv_log", conf_activation="expp1",
                                  use_point_feat=True)
        self.depth_head = DPTHead(dim_in=2 * embed_dim, output_dim=2, activation="exp", conf_activation="expp1",
                                  use_point_feat=False)
        self.track_head = TrackHead(dim_in=2 * embed_dim, patch_size=patch_size)
        self.part_adaptor = SamProjector(dim_in=2 * embed_dim, out_channels=[256, 256, 256, 256], pos_embed=False)
        self.part_head = PartHead(dim_in=2 * embed_dim, output_dim=8, activation="norm")
        assert part_on_invalid_grid in ("raise", "skip")
        self.part_on_invalid_grid = part_on_invalid_grid
        self._init_runtime()

    def _run(self, images, query_points=None):
        H, W = images.shape[-2:]
        part_ok = (H % 28 == 0) and (W % 28 == 0)
        tokens, psi = self.aggregator(images)
        pred = {"pose_enc": self._camera(tokens)}
        pred["depth"], pred["depth_conf"] = self._fork_head(self.depth_head, tokens, images, psi)
        pts, conf, point_feat = self.point_head(tokens, images=images, patch_start_idx=psi)
        pred["world_points"], pred["world_points_conf"] = pts, conf
        if part_ok:
            # the instance-feature branch (reference vggt.py:204-218); bench.py times it as a whole and per kernel family
            with profiling.tagged("part"), profiling.region("part_branch", None):
                pyramid, _ = self.part_adaptor(tokens, images=images, patch_start_idx=psi)
                pred["part_feat"] = self.part_head(list(pyramid.values()), point_feature=point_feat, images=images,
                                                   patch_start_idx=psi,
                                                   frames_chunk_size=max(1, _PART_CHUNK_PIXELS // (H * W)))
        self._track(pred, tokens, images, psi, query_points)
        self._join_heads()
        self._join_camera()
        pred["images"] = images
        return pred

    @torch.no_grad()
    def forward(self, images, query_points=None):
        """images [S,3,H,W] or [1,S,3,H,W] in [0,1] (this rank's views when sharded) -> dict with
        pose_enc (list of 4 x [1,S_all,9]), depth [1,S,H,W,1], depth_conf [1,S,H,W],
        world_points [1,S,H,W,3], world_points_conf [1,S,H,W], part_feat [1,S,8,H,W], images; with query_points
        ([N,2] or [1,N,2] pixel (x, y) in view 0) also track [1,S,N,2], vis [1,S,N], conf [1,S,N]."""
        with torch.amp.autocast("cuda", enabled=False):   # reference vggt.py:189: the caller's autocast never reaches the kernels
            images = self._common(images, query_points)
            query_points = self._query(query_points, images.shape[0])
            if images.shape[0] != 1:
                return self._scenes(images, query_points)
            H, W = images.shape[-2:]
            part_ok = (H % 28 == 0) and (W % 28 == 0)
            if not part_ok and self.part_on_invalid_grid == "raise":
                raise ValueError(f"IGGT part head needs H, W multiples of 28, got {H}x{W} (the reference fails in "
                                 "window_sa.py:73); construct IGGT(part_on_invalid_grid='skip') for geometry only")
            return self._dispatch(images, self._run, query_points)
