"""Aggregator: DINOv2 patch tokens -> 24 x (frame attention, global attention) on HIP kernels.

Mirrors reference iggt/models/aggregator.py:19-361 (ctor signature, parameter names
`camera_token`, `register_token`, `patch_embed.*`, `frame_blocks.*`, `global_blocks.*`, return value
`(list, patch_start_idx)`), re-designed for the MI355X:

* one flat fp32 token matrix x[S*P, C] lives in HBM for the whole trunk; every block updates it in
  place (LayerNorm -> bf16, MFMA GEMMs with fused bias/GELU/LayerScale/residual epilogues, fused
  q/k-norm + RoPE, flash attention).  "frame" and "global" attention differ only in the
  (batch, tokens) factorisation handed to the attention kernel -- no reshapes, no copies;
* of the 24 [frame|global] concatenations the reference materialises (aggregator.py:267-270) only
  layers 4, 11, 17, 23 are ever read (dpt_head.py:52, adaptor.py:146, camera_head.py:96;
  SURVEY.md section 0 fact 7): by default only those are kept (`keep_layers`), the other list
  entries are None.  `keep_layers="all"` restores the reference's full list;
* multi-GPU: views are sharded over ranks (rank r owns a contiguous slice, rank 0 owns view 0);
  the only exchange is an RCCL all-gather of the post-RoPE K and V rows in front of each global
  attention (`ViewShard`), everything else is rank-local.
"""
from typing import List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from .. import _C
from ..layers.blocks import Block, Workspace
from ..layers.rope import PositionGetter, RotaryPositionEmbedding2D
from ..layers.vision_transformer import vit_base, vit_giant2, vit_large, vit_small
from ..dist import ViewShard

_RESNET_MEAN = [0.485, 0.456, 0.406]
_RESNET_STD = [0.229, 0.224, 0.225]
DEFAULT_KEEP = (4, 11, 17, 23)


class Aggregator(nn.Module):
    def __init__(self, img_size=518, patch_size=14, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0,
                 num_register_tokens=4, block_fn=Block, qkv_bias=True, proj_bias=True, ffn_bias=True,
                 patch_embed="dinov2_vitl14_reg", aa_order=["frame", "global"], aa_block_size=1, qk_norm=True,
                 rope_freq=100, init_values=0.01, enable_checkpoint=True,
                 keep_layers: Union[str, Sequence[int]] = DEFAULT_KEEP):
        super().__init__()
        self.__build_patch_embed__(patch_embed, img_size, patch_size, num_register_tokens, embed_dim=embed_dim)
        self.use_checkpoint = enable_checkpoint  # training-only in the reference; unused here
        self.rope = RotaryPositionEmbedding2D(frequency=rope_freq) if rope_freq > 0 else None
        self.position_getter = PositionGetter() if self.rope is not None else None
        mk = lambda: block_fn(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,  # noqa: E731
                              proj_bias=proj_bias, ffn_bias=ffn_bias, init_values=init_values,
                              qk_norm=qk_norm, rope=self.rope)
        self.frame_blocks = nn.ModuleList([mk() for _ in range(depth)])
        self.global_blocks = nn.ModuleList([mk() for _ in range(depth)])
        self.depth = depth
        self.aa_order = aa_order
        self.patch_size = patch_size
        self.aa_block_size = aa_block_size
        if self.depth % self.aa_block_size != 0:
            raise ValueError(f"depth ({depth}) must be divisible by aa_block_size ({aa_block_size})")
        self.aa_block_num = self.depth // self.aa_block_size
        self.camera_token = nn.Parameter(torch.randn(1, 2, 1, embed_dim))
        self.register_token = nn.Parameter(torch.randn(1, 2, num_register_tokens, embed_dim))
        self.patch_start_idx = 1 + num_register_tokens
        nn.init.normal_(self.camera_token, std=1e-6)
        nn.init.normal_(self.register_token, std=1e-6)
        for name, value in (("_resnet_mean", _RESNET_MEAN), ("_resnet_std", _RESNET_STD)):
            self.register_buffer(name, torch.FloatTensor(value).view(1, 1, 3, 1, 1), persistent=False)
        self.keep_layers = keep_layers
        self.shard: Optional[ViewShard] = None  # set by IGGT.set_view_shard for multi-GPU runs
        self._ws = Workspace()
        self._guard_snap = None   # (pinned host copy of the 48 guard words, event) of the last eager forward

    def __build_patch_embed__(self, patch_embed, img_size, patch_size, num_register_tokens,
                              interpolate_antialias=True, interpolate_offset=0.0, block_chunks=0,
                              init_values=1.0, embed_dim=1024):
        if "conv" in patch_embed:
            raise NotImplementedError("IGGT uses the DINOv2 ViT-L patch embed (vggt.py:136); 'conv' is not built")
        vit_models = {"dinov2_vitl14_reg": vit_large, "dinov2_vitb14_reg": vit_base,
                      "dinov2_vits14_reg": vit_small, "dinov2_vitg2_reg": vit_giant2}
        self.patch_embed = vit_models[patch_embed](
            img_size=img_size, patch_size=patch_size, num_register_tokens=num_register_tokens,
            interpolate_antialias=interpolate_antialias, interpolate_offset=interpolate_offset,
            block_chunks=block_chunks, init_values=init_values)
        if hasattr(self.patch_embed, "mask_token"):
            self.patch_embed.mask_token.requires_grad_(False)

    # ------------------------------------------------------------------------------------------
    def _keep(self):
        if self.keep_layers == "all":
            return tuple(range(self.depth))
        return tuple(sorted(set(int(i) % self.depth for i in self.keep_layers) | {self.depth - 1}))

    def forward(self, images: torch.Tensor) -> Tuple[List[Optional[torch.Tensor]], int]:
        """images [B=1, S_local, 3, H, W] in [0,1] -> (list[depth] of [1, S_local, P, 2C] fp32 or None,
        patch_start_idx).  With a ViewShard attached, S_local is this rank's slice of the views."""
        if not images.is_cuda:
            raise _C.HipExtensionError("Aggregator runs on HIP kernels only: move the model and images to "
                                       "the GPU (there is no CPU fallback; the CPU restatement is oracle/)")
        B, S, C_in, H, W = images.shape
        if C_in != 3:
            raise ValueError(f"Expected 3 input channels, got {C_in}")
        if B != 1:
            # the reference loops nothing over B either (demo uses B=1); batches are independent scenes
            outs = [self.forward(images[b:b + 1]) for b in range(B)]
            merged = [None if outs[0][0][i] is None else torch.cat([o[0][i] for o in outs], 0)
                      for i in range(self.depth)]
            return merged, self.patch_start_idx
        if set(self.aa_order) != {"frame", "global"} or len(self.aa_order) != 2 or self.aa_block_size != 1:
            raise NotImplementedError("only the IGGT alternation ['frame','global'] with block size 1 is built")
        dev = images.device
        ps = self.patch_size
        gh, gw = H // ps, W // ps
        psi = self.patch_start_idx
        P = psi + gh * gw
        C = self.camera_token.shape[-1]
        T = S * P

        self.plan_escalation()
        # tokens[s] = [camera(1), register(4), normed DINOv2 patch tokens]   (aggregator.py:209-234)
        tokens = torch.empty(S, P, C, dtype=torch.float32, device=dev)
        self.patch_embed.patch_tokens_into(images[0], tokens, psi)
        sp = torch.cat([self.camera_token.detach().float()[0], self.register_token.detach().float()[0]], 1)
        owns_view0 = self.shard is None or self.shard.rank == 0
        _C.write_special_tokens(tokens, sp[0].contiguous(), sp[1].contiguous(), S, psi, 0, owns_view0)

        rope_geom = None
        if self.rope is not None:
            cos, sin = self.rope.tables(self.frame_blocks[0].attn.head_dim, max(gh, gw), dev)
            rope_geom = dict(P=P, gw=gw, patch_start=psi, cos=cos, sin=sin)
        kv_gather = self.shard if (self.shard is not None and self.shard.active) else None
        self._apply_guard_snapshot()

        x2d = tokens.view(T, C)
        keep = self._keep()
        out: List[Optional[torch.Tensor]] = [None] * self.depth
        order = list(self.aa_order)
        for i in range(self.depth):
            cat = torch.empty(1, S, P, 2 * C, dtype=torch.float32, device=dev) if i in keep else None
            for j, kind in enumerate(order):
                if kind == "frame":
                    self.frame_blocks[i].forward_inplace(x2d, self._ws, batch=S, tokens=P, rope_geom=rope_geom,
                                                         guard_prev=self.frame_blocks[i - 1].attn_guard() if i else None)
                else:
                    self.global_blocks[i].forward_inplace(x2d, self._ws, batch=1, tokens=T,
                                                          rope_geom=rope_geom, kv_gather=kv_gather,
                                                          guard_prev=self.global_blocks[i - 1].attn_guard() if i else None)
                if cat is not None:  # [frame_out | global_out] (aggregator.py:267-270)
                    half = 0 if kind == "frame" else 1
                    cat[0, :, :, half * C:(half + 1) * C].copy_(tokens)
            out[i] = cat
        self._begin_guard_snapshot()
        return out, psi

    # ------------------------------------------------------------------------------------------
    # Which blocks need the estimated-shift launches is decided on the device (the guard words), but WHETHER the launches are
    # issued at all is a host decision -- and the host must not wait for the device.  So every eager forward ends with an
    # asynchronous copy of the 48 guard words into pinned memory, and the next forward, if that copy has landed (event query,
    # no wait), turns the launches on for the blocks whose norm-bound kernel flagged tiles or was skipped, and re-arms their
    # guard in estimated mode.  Blocks on LayerNorm-of-noise statistics (every fixture, the bench) never pay for the machinery;
    # a block that needs it runs one forward on the round-3 fallback first.  Under hipGraph replay nothing is re-decided: the
    # two eager warm-up forwards in front of every capture (graphs.GraphCache.run) take the decision for the captured graph.
    def _guards(self):
        blocks = list(self.frame_blocks) + list(self.global_blocks)
        return [(b, b.attn_guard()) for b in blocks if b.attn_guard() is not None]

    def _begin_guard_snapshot(self):
        from .. import precision

        if not precision.attn_estimated_shift() or torch.cuda.is_current_stream_capturing():
            return
        gs = self._guards()
        if not gs or all(b._est_on for b, _ in gs) or self._guard_snap is not None:   # (an older snapshot is still on its way)
            return
        host = getattr(self, "_guard_host", None)
        if host is None or host.shape[0] != len(gs):
            host = self._guard_host = torch.empty(len(gs), _C.GUARD_WORDS, dtype=torch.int32, pin_memory=True)
        host.copy_(torch.stack([g for _, g in gs]), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._guard_snap = (host, ev, [b for b, _ in gs], [g for _, g in gs])

    def _apply_guard_snapshot(self):
        from .. import graphs, precision

        snap, self._guard_snap = self._guard_snap, None
        if torch.cuda.is_current_stream_capturing():
            self._guard_snap = snap
            return
        sharded = self.shard is not None and self.shard.active
        every = list(self.frame_blocks) + list(self.global_blocks)
        want = [False] * len(every)          # per block (positional: the same index on every rank): turn the launches on
        if snap is not None:
            host, ev, blocks, guards = snap
            if ev.query():
                pos = {id(b): i for i, b in enumerate(every)}
                for i, (b, g) in enumerate(zip(blocks, guards)):
                    if int(host[i, 1]) != 0 and not b._est_on and b.attn_guard() is g:     # flagged tiles, or skipped
                        want[pos[id(b)]] = True
            else:                   # not there yet (the host runs ahead of the device): keep it for the next forward
                self._guard_snap = snap
        if sharded and precision.attn_estimated_shift():
            # View-sharded: a rank-LOCAL decision here would invalidate the captured graphs of that rank alone (buffers_changed
            # below) -- it would then re-run its warm-up forwards and a capture, each issuing collectives, while its peers replay
            # (ADVICE r5).  So the ranks agree: a block takes the estimated-shift launches on every rank as soon as any rank's
            # norm-bound kernel flagged tiles there.  One 48-word all-reduce per EAGER forward (replays never come here); it
            # reads the result back, which an eager sharded forward -- warm-up or debugging -- can afford.
            want = [bool(w) for w in self.shard.agree_any(want, self.camera_token.device)]
        changed = False
        for b, w in zip(every, want):
            g = b.attn_guard()
            if w and not b._est_on and g is not None:
                b._est_on = True
                g.copy_(torch.tensor([0, 0, 0, 0, 1, 0, 0, 0], dtype=torch.int32), non_blocking=True)   # estimated mode, armed
                changed = True
        if changed:
            graphs.buffers_changed()    # captured graphs do not hold the estimated-shift launches of these blocks: re-capture

    def reset_guards(self):
        """Forget everything the adaptive attention switch has learnt (guard words back to "never measured", estimated-shift
        launches off, pending snapshot dropped): outputs of the next forwards then do not depend on the inputs this model has seen
        before -- what golden / determinism fixtures want (INTEGRATION.md "call history")."""
        from .. import graphs

        self._guard_snap = None
        fresh = None
        for b in list(self.frame_blocks) + list(self.global_blocks):
            g = b.attn_guard()
            if g is not None:
                if fresh is None:
                    fresh = torch.tensor([-1, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=g.device)
                g.copy_(fresh)
            b._est_on = False
        graphs.buffers_changed()     # captured graphs hold the launch sequence of the old decisions

    def execution_order(self):
        """The 24 + 48 transformer blocks in the order the forward runs them."""
        order = list(self.patch_embed.blocks)
        for i in range(self.depth):
            for kind in self.aa_order:
                order.append(self.frame_blocks[i] if kind == "frame" else self.global_blocks[i])
        return order

    def plan_escalation(self):
        """Per-block precision rung (precision.py "x3"): every block that is ill-conditioned by its own LayerNorm / q-k-norm
        scales, and every block in front of it, runs on fp16 hi + lo operand pairs.  Returns the plan (72 booleans)."""
        from .. import precision

        return precision.plan_escalation(self.execution_order())

    def escalation_report(self) -> dict:
        """Which blocks run on the x3 rung / fell to bf16 operands (after a forward or `plan_escalation()`), and the worst
        conditioning figures seen (precision.block_condition)."""
        order = self.execution_order()
        names = ([f"patch_embed.blocks.{i}" for i in range(len(self.patch_embed.blocks))]
                 + [f"{k}_blocks.{i}" for i in range(self.depth) for k in self.aa_order])
        conds = [b.own_condition() for b in order]
        packs = [b._packed for b in order]
        return dict(blocks=len(order),
                    x3=[n for n, b, p in zip(names, order, packs) if (p["x3"] if p is not None else bool(b._x3_request))],
                    own_verdict=[n for n, b in zip(names, order) if b.own_escalation()],
                    bf16_fallback=[n for n, p in zip(names, packs) if p is not None and p.get("bf16_fallback")],
                    min_participation_ratio=min(min(c["pr_norm1"], c["pr_norm2"]) for c in conds),
                    min_participation_ratio_untrimmed=min(min(c["pr_norm1_raw"], c["pr_norm2_raw"]) for c in conds),
                    max_logit_rms=max(c["logit_rms"] for c in conds))

    def static_softmax_stats(self) -> dict:
        """Adaptive-switch words of the 48 aggregator blocks after a forward (synchronises): per kind, the number of query
        tiles the static-bound kernel flagged for the online-max pass in the LAST launch of each block, the tile count, and the
        blocks that skipped the static kernel altogether (include/iggt_hip.h `guard`)."""
        out = {}
        for kind, blocks in (("frame", self.frame_blocks), ("global", self.global_blocks)):
            gs = [b.attn_guard() for b in blocks]
            gs = [g.tolist() for g in gs if g is not None]
            out[kind] = dict(blocks=len(gs), flagged_tiles=sum(max(g[1], 0) for g in gs), tiles=sum(g[2] for g in gs if g[1] >= 0),
                             skipped_static=sum(1 for g in gs if g[1] < 0), online_only_next=sum(1 for g in gs if g[0] > 0))
        return out


def slice_expand_and_flatten(token_tensor, B, S):
    """Reference aggregator.py:338-361: slot 0 for view 0, slot 1 for views 1..S-1 -> [B*S, X, C]."""
    query = token_tensor[:, 0:1, ...].expand(B, 1, *token_tensor.shape[2:])
    others = token_tensor[:, 1:, ...].expand(B, S - 1, *token_tensor.shape[2:])
    combined = torch.cat([query, others], dim=1)
    return combined.view(B * S, *combined.shape[2:])
