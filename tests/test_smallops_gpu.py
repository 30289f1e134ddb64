"""fp32 small operators of the heads (csrc/smallops.hip), the GPU image loader (csrc/preprocess.hip) and the pose / geometry
utilities, each against an fp64 evaluation (or, for integer work, bit for bit against PIL) and against golden vectors
produced by the reference functions."""
import numpy as np
import pytest
import torch

from conftest import load_golden, report
from test_kernels_gpu import _rand, _relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def C():
    from iggt_official_amd import _C

    _C.load()
    return _C


ACTS = {None: lambda x: x, "gelu": torch.nn.functional.gelu, "relu": torch.relu, "silu": torch.nn.functional.silu,
        "sigmoid": torch.sigmoid}


@pytest.mark.parametrize("M,N,K,act,extras", [(32, 6144, 2048, None, ""), (32, 2048, 2048, None, "gr"), (32, 8192, 2048, "gelu", ""),
                                              (32, 2048, 8192, None, "gr"), (32, 2048, 9, "silu", ""), (32, 9, 1024, None, ""),
                                              (5, 1024, 2048, "gelu", ""), (70, 333, 200, "relu", "g"), (8, 4, 128, "relu", ""),
                                              (8, 128, 4, "sigmoid", ""), (1, 6144, 2048, None, "")])
def test_linear_f32(C, M, N, K, act, extras):
    x, w, b = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3)
    gamma = _rand((N,), 4) if "g" in extras else None
    res = _rand((M, N), 5) if "r" in extras else None
    ref = ACTS[act](x.double() @ w.double().t() + b.double())
    if gamma is not None:
        ref = ref * gamma.double()
    if res is not None:
        ref = ref + res.double()
    out = res.clone() if res is not None else None           # in-place residual (res aliases out), as the camera trunk uses it
    out = C.linear_f32(x, w, b, act=act, gamma=gamma, res=out, out=out)
    mx, l2 = _relerr(out, ref)
    report(f"linear_f32_{M}x{N}x{K}_{act}", dict(max=mx, l2=l2))
    assert mx < 1e-5, (mx, l2)
    # split-K partial sums are added in a fixed order: bit-identical from launch to launch
    a1 = C.linear_f32(x, w, b, act=act, gamma=gamma)
    a2 = C.linear_f32(x, w, b, act=act, gamma=gamma)
    assert torch.equal(a1, a2)
    # strided input rows (a column slice of a wider matrix)
    big = _rand((M, K + 8), 6)
    out2 = C.linear_f32(big[:, 4:4 + K] if K % 4 == 0 else big[:, :K], w, b)
    ref2 = (big[:, 4:4 + K] if K % 4 == 0 else big[:, :K]).double() @ w.double().t() + b.double()
    assert _relerr(out2, ref2)[0] < 1e-5


@pytest.mark.parametrize("B,H,Nq,Nk,d", [(1, 16, 32, 32, 128), (1, 16, 3, 3, 128), (2, 8, 1296, 1296, 32), (3, 8, 100, 77, 32),
                                          (2, 4, 65, 130, 64), (1, 1, 1, 1, 32)])
def test_attn_f32(C, B, H, Nq, Nk, d):
    Cd = H * d
    q, k, v = _rand((B, Nq, Cd), 11), _rand((B, Nk, Cd), 12), _rand((B, Nk, Cd), 13)
    o = torch.full((B, Nq, Cd), float("nan"), device="cuda")
    scale = d ** -0.5
    C.attn_f32(q, k, v, o, B, H, Nq, Nk, d, Nq * Cd, Cd, Nk * Cd, Cd, Nk * Cd, Cd, Nq * Cd, Cd, scale)
    qh, kh, vh = (t.double().view(B, -1, H, d).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).transpose(1, 2).reshape(B, Nq, Cd)
    mx, l2 = _relerr(o, ref)
    report(f"attn_f32_B{B}_H{H}_{Nq}x{Nk}_d{d}", dict(max=mx, l2=l2))
    assert not torch.isnan(o).any() and mx < 2e-5, (mx, l2)


@pytest.mark.parametrize("B,L", [(700, 8), (1088, 3), (600, 16), (64, 32), (300, 32)])
def test_attn_f32_many_short_sequences(C, B, L):
    """B x 8 sequences of L <= 32 tokens, head slot 64 (the tracker's attention along time, packed [B*L, 3*512] qkv with a
    peaked row): B * H >= 512 with L <= 16 takes the one-thread-per-query-row kernel, the rest the MFMA kernel -- same contract."""
    H, d = 8, 64
    HS = H * d
    qkv = _rand((B * L, 3 * HS), 41)
    qkv.view(B, L, 3, H, d)[5, 1, 1, 2] = qkv.view(B, L, 3, H, d)[5, 0, 0, 2] * 4.0
    o = torch.full((B * L, HS), float("nan"), device="cuda")
    ld = 3 * HS
    C.attn_f32(qkv, qkv[:, HS:], qkv[:, 2 * HS:], o, B, H, L, L, d, L * ld, ld, L * ld, ld, L * ld, ld, L * HS, HS, 48 ** -0.5)
    x = qkv.double().view(B, L, 3, H, d)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 48 ** -0.5, -1) @ v).transpose(1, 2).reshape(B * L, HS)
    assert not torch.isnan(o).any() and _relerr(o, ref)[0] < 2e-5


def test_attn_f32_packed_qkv_and_peaked(C):
    """q/k/v as column slices of one [N, 3C] matrix (camera trunk layout); one key 30 nats above the rest."""
    N, H, d = 32, 16, 128
    Cd = H * d
    qkv = _rand((N, 3 * Cd), 21)
    x = qkv.view(N, 3, H, d)
    x[7, 1, 3] = x[20, 0, 3] * 3.0
    o = torch.empty(N, Cd, device="cuda")
    C.attn_f32(qkv, qkv[:, Cd:], qkv[:, 2 * Cd:], o, 1, H, N, N, d, 0, 3 * Cd, 0, 3 * Cd, 0, 3 * Cd, 0, Cd, d ** -0.5)
    q, k, v = (x[:, i].double().transpose(0, 1) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, -1) @ v).transpose(0, 1).reshape(N, Cd)
    assert _relerr(o, ref)[0] < 2e-5


def test_adaln_modulate_and_pose_update(C):
    S, Cd = 7, 2048
    x, mod = _rand((S, Cd), 31, 2.0) + 0.3, _rand((S, 3 * Cd), 32)
    out = C.adaln_modulate(x, mod[:, :Cd], mod[:, Cd:2 * Cd], mod[:, 2 * Cd:], 1e-6)
    xd, md = x.double(), mod.double()
    ln = torch.nn.functional.layer_norm(xd, (Cd,), eps=1e-6)
    ref = md[:, 2 * Cd:] * (ln * (1 + md[:, Cd:2 * Cd]) + md[:, :Cd]) + xd
    assert _relerr(out, ref)[0] < 1e-5
    delta, pred, act = _rand((S, 9), 33), torch.zeros(S, 9, device="cuda"), torch.empty(S, 9, device="cuda")
    C.pose_update(delta, pred, act, first=True)
    assert torch.equal(pred, delta)
    C.pose_update(delta, pred, act, first=False)
    assert torch.equal(pred, delta + delta)
    want = pred.clone()
    want[:, 7:] = want[:, 7:].relu()
    assert torch.equal(act, want)


def test_conv1x1_c32_nchw(C):
    x = _rand((3, 20, 28, 32), 41)
    w, b = _rand((8, 32, 1, 1), 42, 0.2), _rand((8,), 43)
    y = C.conv1x1_c32_nchw(x, w, b)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double())
    assert y.shape == (3, 8, 20, 28) and _relerr(y, ref)[0] < 1e-5


def test_camera_head_matches_fp64_restatement(C):
    """The whole HIP camera head (32 views) against an fp64 evaluation of the reference's equations
    (camera_head.py:83-154) with the same parameters."""
    from helpers import build_gpu_model

    model = build_gpu_model("stress", 0)
    head = model.camera_head
    S, Cd = 32, 2048
    tokens = _rand((1, S, Cd), 51, 1.5)
    got = head([None], camera_tokens=tokens)
    F = torch.nn.functional
    P = {k: v.detach().double() for k, v in head.named_parameters()}

    def lin(name, x):
        return F.linear(x, P[name + ".weight"], P[name + ".bias"])

    t = F.layer_norm(tokens.double(), (Cd,), P["token_norm.weight"], P["token_norm.bias"], 1e-5)
    pred, outs = None, []
    for _ in range(4):
        inp = P["empty_pose_tokens"].expand(1, S, -1) if pred is None else pred
        mod = lin("poseLN_modulation.1", F.silu(lin("embed_pose", inp)))
        shift, scale, gate = mod.chunk(3, dim=-1)
        x = gate * (F.layer_norm(t, (Cd,), eps=1e-6) * (1 + scale) + shift) + t
        for i in range(4):
            p = f"trunk.{i}."
            h = F.layer_norm(x, (Cd,), P[p + "norm1.weight"], P[p + "norm1.bias"], 1e-5)
            qkv = lin(p + "attn.qkv", h).view(1, S, 3, 16, 128).permute(2, 0, 3, 1, 4)
            a = torch.softmax(qkv[0] @ qkv[1].transpose(-1, -2) * 128 ** -0.5, -1) @ qkv[2]
            x = x + P[p + "ls1.gamma"] * lin(p + "attn.proj", a.transpose(1, 2).reshape(1, S, Cd))
            h = F.layer_norm(x, (Cd,), P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-5)
            x = x + P[p + "ls2.gamma"] * lin(p + "mlp.fc2", F.gelu(lin(p + "mlp.fc1", h)))
        h = F.layer_norm(x, (Cd,), P["trunk_norm.weight"], P["trunk_norm.bias"], 1e-5)
        delta = lin("pose_branch.fc2", F.gelu(lin("pose_branch.fc1", h)))
        pred = delta if pred is None else pred + delta
        o = pred.clone()
        o[..., 7:] = o[..., 7:].relu()
        outs.append(o)
    for i in range(4):
        mx, l2 = _relerr(got[i], outs[i])
        report(f"camera_head_iter{i}", dict(max=mx, l2=l2))
        assert got[i].shape == (1, S, 9) and mx < 2e-5, (i, mx, l2)


def test_pose_decode_and_unprojection_match_reference_golden(C):
    from iggt.utils.geometry import unproject_depth_map_to_point_map
    from iggt.utils.pose_enc import pose_encoding_to_extri_intri

    g = load_golden("utils_pose_geometry")
    pose = g["pose"].cuda()
    extri, intri = pose_encoding_to_extri_intri(pose, (g["H"], g["W"]))
    assert extri.shape == (1, 5, 3, 4) and intri.shape == (1, 5, 3, 3)
    assert _relerr(extri, g["extri"].cuda())[0] < 1e-6 and _relerr(intri, g["intri"].cuda())[0] < 1e-6
    e2, none = pose_encoding_to_extri_intri(pose, None, build_intrinsics=False)
    assert none is None and torch.equal(e2, extri)
    # depth unprojection with the REFERENCE's camera parameters: numpy out, like the reference
    world = unproject_depth_map_to_point_map(g["depth"], g["extri"][0], g["intri"][0])
    assert isinstance(world, np.ndarray) and world.shape == (5, 28, 42, 3)
    ref = g["world"].numpy()
    err = np.abs(world - ref).max() / np.abs(ref).max()
    report("unproject_depth_vs_reference", dict(max=float(err)))
    assert err < 2e-7                                           # fp32 rounding of an fp64 evaluation
    t = unproject_depth_map_to_point_map(g["depth"].cuda(), g["extri"][0].cuda(), g["intri"][0].cuda(), as_tensor=True)
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), world)


@pytest.mark.parametrize("mode", ["crop", "pad", "resize"])
def test_load_and_preprocess_images_bit_exact(C, tmp_path, mode):
    """GPU loader (Pillow-exact integer resampler + ToTensor / crop / pad kernels) == PIL + ToTensor restatement of the
    reference, bit for bit, for landscape / portrait / tiny / RGBA inputs and mixed output shapes."""
    from PIL import Image

    from iggt.utils.load_fn import load_and_preprocess_images
    from oracle import restate_utils as ru

    rng = np.random.default_rng(3)
    paths = []
    for i, (h, w) in enumerate(((341, 512), (900, 600), (120, 100), (518, 518), (1500, 2000))):
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        arr[: h // 2] = (arr[: h // 2].astype(np.int32) * 3 // 4 + 40).astype(np.uint8)      # some structure
        p = str(tmp_path / f"img{i}.png")
        Image.fromarray(arr).save(p)
        paths.append(p)
    rgba = rng.integers(0, 256, (200, 300, 4), dtype=np.uint8)
    p = str(tmp_path / "rgba.png")
    Image.fromarray(rgba, "RGBA").save(p)
    paths.append(p)
    kw = dict(resize_target_size=(392, 294)) if mode == "resize" else {}
    got = load_and_preprocess_images(paths, mode=mode, **kw)
    ref = ru.load_and_preprocess_images(paths, mode=mode, **kw)
    assert got.is_cuda and got.dtype == torch.float32 and got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.equal(got.cpu(), ref)
    one = load_and_preprocess_images(paths[:1], mode=mode, **kw)
    assert one.shape[0] == 1 and torch.equal(one.cpu(), ru.load_and_preprocess_images(paths[:1], mode=mode, **kw))
    with pytest.raises(ValueError):
        load_and_preprocess_images([], mode=mode)
