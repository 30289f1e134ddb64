"""Camera head -- reference iggt/heads/camera_head.py:19-162, on HIP kernels.

4 refinement iterations of a 4-block transformer trunk (dim 2048, 16 heads x 128, LayerScale 0.01) over the S camera
tokens of the last aggregator layer, with adaLN modulation from the previous pose estimate.  The reference runs it in
fp32 (autocast off, vggt.py:189) and so does this: every Linear is the exact-fp32 MFMA kernel `iggt_linear_f32` (the
problem is S rows against 1.6 GB of fp32 weights per iteration -- weight-bandwidth bound, a 16-bit operand format would
buy nothing but error), the attention over the S views is `iggt_attn_f32` (head dim 128), LayerNorms are
`iggt_layernorm_f32` with fp32 output, the adaLN modulation and the pose accumulation + activation are one small kernel
each (csrc/smallops.hip).  State-dict names match the reference (trunk blocks reuse layers.blocks.Block as parameter
holders).  With sharded views the camera tokens of all ranks are all-gathered first (dist.ViewShard).
"""
import torch
import torch.nn as nn

from .. import _C
from ..layers.blocks import Block, LayerScale, Mlp


def modulate(x, shift, scale):
    return x * (1 + scale) + shift


def _p(t):
    return None if t is None else t.detach()


def _ln(norm: nn.LayerNorm, x):
    out = torch.empty_like(x)
    _C.layernorm(x, _p(norm.weight), _p(norm.bias), out, norm.eps)
    return out


def _block_hip(blk: Block, x):
    """x [N, C] fp32 -> x + g1 * proj(attn(qkv(norm1 x))), then + g2 * fc2(gelu(fc1(norm2 .)))  (block.py:105-106);
    updates x in place."""
    a = blk.attn
    N, C = x.shape
    H, d = a.num_heads, a.head_dim
    qkv = _C.linear_f32(_ln(blk.norm1, x), _p(a.qkv.weight), _p(a.qkv.bias))
    ao = torch.empty(N, C, dtype=torch.float32, device=x.device)
    _C.attn_f32(qkv, qkv[:, C:], qkv[:, 2 * C:], ao, 1, H, N, N, d, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C, a.scale)
    g1 = _p(blk.ls1.gamma) if isinstance(blk.ls1, LayerScale) else None
    g2 = _p(blk.ls2.gamma) if isinstance(blk.ls2, LayerScale) else None
    _C.linear_f32(ao, _p(a.proj.weight), _p(a.proj.bias), gamma=g1, res=x, out=x)
    hid = _C.linear_f32(_ln(blk.norm2, x), _p(blk.mlp.fc1.weight), _p(blk.mlp.fc1.bias), act="gelu")
    _C.linear_f32(hid, _p(blk.mlp.fc2.weight), _p(blk.mlp.fc2.bias), gamma=g2, res=x, out=x)
    return x


class CameraHead(nn.Module):
    def __init__(self, dim_in=2048, trunk_depth=4, pose_encoding_type="absT_quaR_FoV", num_heads=16, mlp_ratio=4,
                 init_values=0.01, trans_act="linear", quat_act="linear", fl_act="relu"):
        super().__init__()
        if pose_encoding_type != "absT_quaR_FoV":
            raise ValueError(f"Unsupported camera encoding type: {pose_encoding_type}")
        if (trans_act, quat_act, fl_act) != ("linear", "linear", "relu"):
            raise NotImplementedError("the HIP pose kernel implements IGGT's activations (linear, linear, relu)")
        self.target_dim = 9
        self.trans_act, self.quat_act, self.fl_act = trans_act, quat_act, fl_act
        self.trunk_depth = trunk_depth
        self.trunk = nn.Sequential(*[Block(dim=dim_in, num_heads=num_heads, mlp_ratio=mlp_ratio,
                                           init_values=init_values) for _ in range(trunk_depth)])
        self.token_norm = nn.LayerNorm(dim_in)
        self.trunk_norm = nn.LayerNorm(dim_in)
        self.empty_pose_tokens = nn.Parameter(torch.zeros(1, 1, self.target_dim))
        self.embed_pose = nn.Linear(self.target_dim, dim_in)
        self.poseLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(dim_in, 3 * dim_in, bias=True))
        self.adaln_norm = nn.LayerNorm(dim_in, elementwise_affine=False, eps=1e-6)
        self.pose_branch = Mlp(in_features=dim_in, hidden_features=dim_in // 2, out_features=self.target_dim, drop=0)

    def forward(self, aggregated_tokens_list, num_iterations=4, camera_tokens=None):
        """camera_tokens [B, S, 2048] overrides the slice of the last layer (multi-GPU: gathered tokens)."""
        tokens = aggregated_tokens_list[-1][:, :, 0] if camera_tokens is None else camera_tokens
        if not tokens.is_cuda:
            raise _C.HipExtensionError("CameraHead runs on HIP kernels only (no CPU fallback)")
        B, S, C = tokens.shape
        per_scene = [self.trunk_fn(_ln(self.token_norm, tokens[b].float().contiguous()), num_iterations)
                     for b in range(B)]
        return [torch.stack([ps[i] for ps in per_scene], 0) for i in range(num_iterations)]

    def trunk_fn(self, pose_tokens, num_iterations):
        """pose_tokens [S, C] fp32 (already token_norm-ed) -> list of num_iterations activated encodings [S, 9]."""
        S, C = pose_tokens.shape
        dev = pose_tokens.device
        pred = torch.empty(S, 9, dtype=torch.float32, device=dev)
        mod_lin = self.poseLN_modulation[1]
        outs = []
        for it in range(num_iterations):
            inp = self.empty_pose_tokens.detach().float().view(1, 9).expand(S, 9).contiguous() if it == 0 else pred
            # embed_pose, then the SiLU that opens poseLN_modulation fused into its epilogue
            emb = _C.linear_f32(inp, _p(self.embed_pose.weight), _p(self.embed_pose.bias), act="silu")
            mod = _C.linear_f32(emb, _p(mod_lin.weight), _p(mod_lin.bias))                       # [S, 3C] = shift | scale | gate
            x = _C.adaln_modulate(pose_tokens, mod[:, :C], mod[:, C:2 * C], mod[:, 2 * C:], self.adaln_norm.eps)
            for blk in self.trunk:
                x = _block_hip(blk, x)
            hid = _C.linear_f32(_ln(self.trunk_norm, x), _p(self.pose_branch.fc1.weight), _p(self.pose_branch.fc1.bias),
                                act="gelu")
            delta = _C.linear_f32(hid, _p(self.pose_branch.fc2.weight), _p(self.pose_branch.fc2.bias))
            out = torch.empty(S, 9, dtype=torch.float32, device=dev)
            _C.pose_update(delta, pred, out, first=(it == 0))
            outs.append(out)
        return outs
