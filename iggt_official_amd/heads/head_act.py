"""Output activations (reference iggt/heads/head_act.py:11-125): the module-level API of the reference, kept for callers that
import it.  NOT on the product's forward path -- DPTHead / CameraHead apply their activations inside HIP kernels
(csrc/dpt_tail.hip, csrc/elementwise.hip head_tail_kernel, csrc/smallops.hip pose_update_kernel)."""
import torch
import torch.nn.functional as F


def inverse_log_transform(y):
    return torch.sign(y) * torch.expm1(torch.abs(y))


def _pose_act(x, kind):
    if kind == "linear":
        return x
    if kind == "inv_log":
        return inverse_log_transform(x)
    if kind == "exp":
        return torch.exp(x)
    if kind == "relu":
        return F.relu(x)
    raise ValueError(f"Unknown act_type: {kind}")


def activate_pose(pred_pose_enc, trans_act="linear", quat_act="linear", fl_act="linear"):
    T, quat, fl = pred_pose_enc[..., :3], pred_pose_enc[..., 3:7], pred_pose_enc[..., 7:]
    return torch.cat([_pose_act(T, trans_act), _pose_act(quat, quat_act), _pose_act(fl, fl_act)], dim=-1)


def activate_head(out, activation="norm_exp", conf_activation="expp1"):
    """out [B, C, H, W] -> (pts [B,H,W,C-1], conf [B,H,W])."""
    fmap = out.permute(0, 2, 3, 1)
    xyz, conf = fmap[..., :-1], fmap[..., -1]
    if activation == "norm_exp":
        d = xyz.norm(dim=-1, keepdim=True).clamp(min=1e-8)
        pts = xyz / d * torch.expm1(d)
    elif activation == "norm":
        pts = xyz / xyz.norm(dim=-1, keepdim=True)
    elif activation == "exp":
        pts = torch.exp(xyz)
    elif activation == "relu":
        pts = F.relu(xyz)
    elif activation == "inv_log":
        pts = inverse_log_transform(xyz)
    elif activation == "xy_inv_log":
        xy, z = xyz.split([2, 1], dim=-1)
        z = inverse_log_transform(z)
        pts = torch.cat([xy * z, z], dim=-1)
    elif activation == "sigmoid":
        pts = torch.sigmoid(xyz)
    elif activation == "linear":
        pts = xyz
    else:
        raise ValueError(f"Unknown activation: {activation}")
    if conf_activation == "expp1":
        c = 1 + conf.exp()
    elif conf_activation == "expp0":
        c = conf.exp()
    elif conf_activation == "sigmoid":
        c = torch.sigmoid(conf)
    else:
        raise ValueError(f"Unknown conf_activation: {conf_activation}")
    return pts, c
