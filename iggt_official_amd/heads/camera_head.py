"""Camera head -- reference iggt/heads/camera_head.py:19-162.

4 refinement iterations of a 4-block transformer trunk (dim 2048, 16 heads x 128, LayerScale 0.01)
over the S camera tokens of the last aggregator layer, with adaLN modulation from the previous pose
estimate.  < 0.1 % of the forward FLOPs (S tokens!), so round 1 keeps it on plain PyTorch-ROCm fp32
ops ("keep in PyTorch first"); its state-dict names match
the reference (trunk blocks reuse iggt_official_amd.layers.blocks.Block as parameter holders).
With sharded views the camera tokens of all ranks are all-gathered first (dist.ViewShard).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..layers.blocks import Block, Mlp
from .head_act import activate_pose


def modulate(x, shift, scale):
    return x * (1 + scale) + shift


def _block_torch(blk: Block, x):
    """x += g1*proj(sdpa(qkv(norm1 x))); x += g2*fc2(gelu(fc1(norm2 x)))  -- fp32 eager (block.py:105-106)."""
    a = blk.attn
    B, N, C = x.shape
    qkv = a.qkv(blk.norm1(x)).view(B, N, 3, a.num_heads, a.head_dim).permute(2, 0, 3, 1, 4)
    o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, N, C)
    x = x + blk.ls1(a.proj(o))
    return x + blk.ls2(blk.mlp.fc2(F.gelu(blk.mlp.fc1(blk.norm2(x)))))


class CameraHead(nn.Module):
    def __init__(self, dim_in=2048, trunk_depth=4, pose_encoding_type="absT_quaR_FoV", num_heads=16, mlp_ratio=4,
                 init_values=0.01, trans_act="linear", quat_act="linear", fl_act="relu"):
        super().__init__()
        if pose_encoding_type != "absT_quaR_FoV":
            raise ValueError(f"Unsupported camera encoding type: {pose_encoding_type}")
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
        return self.trunk_fn(self.token_norm(tokens.float()), num_iterations)

    def trunk_fn(self, pose_tokens, num_iterations):
        B, S, C = pose_tokens.shape
        pred, outs = None, []
        for _ in range(num_iterations):
            inp = self.empty_pose_tokens.expand(B, S, -1) if pred is None else pred.detach()
            shift, scale, gate = self.poseLN_modulation(self.embed_pose(inp)).chunk(3, dim=-1)
            x = gate * modulate(self.adaln_norm(pose_tokens), shift, scale) + pose_tokens
            for blk in self.trunk:
                x = _block_torch(blk, x)
            delta = self.pose_branch.fc2(F.gelu(self.pose_branch.fc1(self.trunk_norm(x))))
            pred = delta if pred is None else pred + delta
            outs.append(activate_pose(pred, trans_act=self.trans_act, quat_act=self.quat_act, fl_act=self.fl_act))
        return outs
