"""Implicit-GEMM convolution / resize kernels (GPU) vs PyTorch fp64 references.

prec=3 (split-bf16 hi+lo, what the heads use) must be fp32-grade: 2e-5 of the output range.
prec=1 (plain bf16 operands) is gated at the bf16 rounding budget (1e-2 max, 4e-3 l2)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import report

pytestmark = pytest.mark.gpu


def _err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("cin,cout,k,stride,pad,hw", [(256, 256, 3, 1, 1, (37, 41)), (1024, 256, 3, 1, 1, (19, 19)),
                                                      (1024, 1024, 3, 2, 1, (37, 37)), (256, 256, 1, 1, 0, (30, 50)),
                                                      (256, 128, 3, 1, 1, (24, 40)), (128, 32, 3, 1, 1, (56, 70)),
                                                      (128, 64, 3, 1, 1, (16, 24)), (64, 128, 3, 1, 1, (16, 24))])
@pytest.mark.parametrize("prec", [3, 1])
def test_conv2d_vs_torch(cin, cout, k, stride, pad, hw, prec):
    from iggt_official_amd.heads import convops as co

    conv = nn.Conv2d(cin, cout, k, stride, pad).cuda()
    with torch.no_grad():
        conv.weight.copy_(_mk(conv.weight.shape, 1, (cin * k * k) ** -0.5))
        conv.bias.copy_(_mk(conv.bias.shape, 2, 0.1))
    x = _mk((3, hw[0], hw[1], cin), 3)
    y = co.run(co.pack_conv2d(conv), x, prec=prec)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), conv.weight.double(), conv.bias.double(), stride, pad)
    mx, l2 = _err(y.permute(0, 3, 1, 2), ref)
    report(f"conv_{cin}_{cout}_k{k}s{stride}_p{prec}", dict(max=mx, l2=l2))
    assert (mx < 2e-5) if prec == 3 else (mx < 1e-2 and l2 < 4e-3), (mx, l2)


def test_conv_fused_relu_residuals_and_acts():
    from iggt_official_amd.heads import convops as co

    conv = nn.Conv2d(256, 256, 3, 1, 1).cuda()
    x, r1, r2 = _mk((2, 20, 28, 256), 4), _mk((2, 20, 28, 256), 5), _mk((2, 20, 28, 256), 6)
    pc = co.pack_conv2d(conv)
    xd = x.permute(0, 3, 1, 2).double()
    base = F.conv2d(F.relu(xd), conv.weight.double(), conv.bias.double(), 1, 1)
    y = co.run(pc, x, relu_in=True, res=r1, relu_res=True, res2=r2)
    ref = base + F.relu(r1.permute(0, 3, 1, 2).double()) + r2.permute(0, 3, 1, 2).double()
    assert _err(y.permute(0, 3, 1, 2), ref)[0] < 2e-5
    for act, fn in [(1, F.relu), (2, lambda t: F.leaky_relu(t, 0.01)), (3, F.gelu)]:
        y = co.run(pc, x, relu_in=True, act=act)
        assert _err(y.permute(0, 3, 1, 2), fn(base))[0] < 2e-5, act


@pytest.mark.parametrize("cin,cout,k,stride,n,hw", [(1024, 256, 3, 1, 4, (19, 19)), (1024, 1024, 3, 2, 4, (37, 37)),
                                                    (512, 320, 3, 1, 1, (13, 21)), (1536, 384, 1, 1, 1, (1, 4800)),
                                                    (256, 256, 3, 1, 2, (12, 18))])
def test_conv_split_k_small_maps(cin, cout, k, stride, n, hw):
    """Few output tiles, long K: the launcher splits the K loop over grid.y (csrc/conv_igemm.hip launch_splitk) and a
    finalize pass applies bias / activation / residuals.  Ragged M and Cout, K chunks not divisible by the slice count,
    every epilogue feature; the last shape is the tracker's fc2 as a 1 x 1 convolution over 4 800 token rows."""
    from iggt_official_amd.heads import convops as co

    pad = k // 2
    conv = nn.Conv2d(cin, cout, k, stride, pad).cuda()
    with torch.no_grad():
        conv.weight.copy_(_mk(conv.weight.shape, 31, (cin * k * k) ** -0.5))
        conv.bias.copy_(_mk(conv.bias.shape, 32, 0.1))
    pc = co.pack_conv2d(conv)
    x = _mk((n, hw[0], hw[1], cin), 33)
    xd = x.permute(0, 3, 1, 2).double()
    base = F.conv2d(F.relu(xd), conv.weight.double(), conv.bias.double(), stride, pad)
    ho, wo = base.shape[-2:]
    r1, r2 = _mk((n, ho, wo, cout), 34), _mk((n, ho, wo, cout), 35)
    y = co.run(pc, x, relu_in=True, res=r1, relu_res=True, res2=r2)
    ref = base + F.relu(r1.permute(0, 3, 1, 2).double()) + r2.permute(0, 3, 1, 2).double()
    assert _err(y.permute(0, 3, 1, 2), ref)[0] < 2e-5
    y = co.run(pc, x, relu_in=True, act=3)
    assert _err(y.permute(0, 3, 1, 2), F.gelu(base))[0] < 2e-5
    out = r1.clone()                                        # residual and output in the same buffer
    co.run(pc, x, relu_in=True, res=out, out=out)
    assert _err(out.permute(0, 3, 1, 2), base + r1.permute(0, 3, 1, 2).double())[0] < 2e-5


@pytest.mark.parametrize("cin,cout,n,hw", [(256, 256, 2, (74, 74)), (256, 128, 1, (70, 100)), (64, 128, 3, (16, 50)),
                                           (512, 256, 1, (37, 74)), (128, 512, 2, (24, 48)), (32, 256, 1, (148, 148))])
def test_conv3x3_halo_tile_kernel(cin, cout, n, hw):
    """Shapes that take the spatial-halo kernel (csrc/conv3x3_halo.hip: 3x3 / s1 / p1, Cout in 128N, maps >= 16 x 48): ragged
    tiles in both directions, one to sixteen 32-channel slices, both column-tile widths (256 / 128), every fused epilogue
    feature -- against fp64 and against the GEMM-shaped kernel on a sub-threshold crop of the same problem."""
    from iggt_official_amd.heads import convops as co

    conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda()
    with torch.no_grad():
        conv.weight.copy_(_mk(conv.weight.shape, 21, (cin * 9) ** -0.5))
        conv.bias.copy_(_mk(conv.bias.shape, 22, 0.1))
    pc = co.pack_conv2d(conv)
    x = _mk((n, hw[0], hw[1], cin), 23)
    xd = x.permute(0, 3, 1, 2).double()
    y = co.run(pc, x)
    ref = F.conv2d(xd, conv.weight.double(), conv.bias.double(), 1, 1)
    mx, l2 = _err(y.permute(0, 3, 1, 2), ref)
    report(f"conv_halo_{cin}_{cout}_{hw[0]}x{hw[1]}", dict(max=mx, l2=l2))
    assert mx < 2e-5, (mx, l2)
    r1, r2 = _mk((n, hw[0], hw[1], cout), 24), _mk((n, hw[0], hw[1], cout), 25)
    base = F.conv2d(F.relu(xd), conv.weight.double(), conv.bias.double(), 1, 1)
    y = co.run(pc, x, relu_in=True, res=r1, relu_res=True, res2=r2)
    want = base + F.relu(r1.permute(0, 3, 1, 2).double()) + r2.permute(0, 3, 1, 2).double()
    assert _err(y.permute(0, 3, 1, 2), want)[0] < 2e-5
    for act, fn in [(1, F.relu), (2, lambda t: F.leaky_relu(t, 0.01)), (3, F.gelu)]:
        y = co.run(pc, x, relu_in=True, act=act)
        assert _err(y.permute(0, 3, 1, 2), fn(base))[0] < 2e-5, act
    # interior of the map == the GEMM-shaped kernel on a crop too narrow for the halo kernel (W < 48)
    crop = x[:1, :, :40].contiguous()
    yc = co.run(pc, crop)
    yf = co.run(pc, x)
    assert _err(yc[:, :, :38], yf[:1, :, :38])[0] < 2e-6


def test_conv_bn_fold_and_channel_padding():
    from iggt_official_amd.heads import convops as co

    conv, bn = nn.Conv2d(128, 42, 3, 1, 1, bias=False).cuda(), nn.BatchNorm2d(42).cuda().eval()
    with torch.no_grad():
        bn.running_mean.copy_(_mk((42,), 7, 0.3)); bn.running_var.copy_(_mk((42,), 8).abs() + 0.5)
        bn.weight.copy_(_mk((42,), 9) * 0.1 + 1); bn.bias.copy_(_mk((42,), 10, 0.1))
    x = _mk((2, 16, 24, 128), 11)
    y = co.run(co.pack_conv2d(conv, bn), x, ldy=64)
    ref = bn(conv(x.permute(0, 3, 1, 2))).double()
    assert y.shape[-1] == 64 and torch.all(y[..., 42:] == 0)
    assert _err(y[..., :42].permute(0, 3, 1, 2), ref)[0] < 5e-5
    conv2 = nn.Conv2d(42, 128, 3, 1, 1).cuda()
    y2 = co.run(co.pack_conv2d(conv2, cin_pad=64), y)
    ref2 = conv2(ref.float()).double()
    assert _err(y2.permute(0, 3, 1, 2), ref2)[0] < 5e-5


@pytest.mark.parametrize("s", [2, 4])
def test_conv_transpose_kernel_eq_stride(s):
    from iggt_official_amd.heads import convops as co

    ct = nn.ConvTranspose2d(256, 256, s, s, 0).cuda()
    x = _mk((2, 9, 13, 256), 12)
    y = co.run(co.pack_convT_kernel_eq_stride(ct), x)
    ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), ct.weight.double(), ct.bias.double(), s, 0)
    assert y.shape == (2, 9 * s, 13 * s, 256)
    assert _err(y.permute(0, 3, 1, 2), ref)[0] < 2e-5


def test_conv_transpose_k4s2p1():
    from iggt_official_amd.heads import convops as co

    ct = nn.ConvTranspose2d(256, 256, 4, 2, 1).cuda()
    x = _mk((2, 7, 10, 256), 13)
    y = co.run_convT_k4s2p1(co.pack_convT_k4s2p1(ct), x)
    ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), ct.weight.double(), ct.bias.double(), 2, 1)
    assert y.shape == (2, 14, 20, 256)
    assert _err(y.permute(0, 3, 1, 2), ref)[0] < 2e-5


@pytest.mark.parametrize("hi,ho", [((19, 19), (37, 37)), ((37, 24), (74, 48)), ((296, 296), (518, 518)),
                                   ((8, 12), (8, 12))])
def test_bilinear_align_corners(hi, ho):
    from iggt_official_amd.heads import convops as co

    x = _mk((2, hi[0], hi[1], 128), 14)
    y = co.resize(x, ho)
    # fp64 reference: the kernel takes the fraction of the source coordinate from the exact product index * fp32 scale (one fused
    # multiply-subtract), 1.3e-6 of a pixel from the real-number coordinate at 296 -> 518; an implementation that rounds the
    # product to fp32 first is 1.6e-5 of a pixel off there (3e-5 of the range on white noise)
    ref = F.interpolate(x.permute(0, 3, 1, 2).double(), size=ho, mode="bilinear", align_corners=True)
    tol = 1e-5
    assert _err(y.permute(0, 3, 1, 2), ref)[0] < tol
    xr, yr = _mk((ho[1], 64), 15), _mk((ho[0], 64), 16)
    y2 = co.resize(x, ho, xr, yr)
    add = torch.cat([xr.t()[None, :, None, :].expand(1, 64, ho[0], ho[1]),
                     yr.t()[None, :, :, None].expand(1, 64, ho[0], ho[1])], 1)
    assert _err(y2.permute(0, 3, 1, 2), ref + add)[0] < tol


@pytest.mark.parametrize("cin,cout,k,stride", [(256, 256, 3, 1), (64, 512, 1, 1), (128, 256, 3, 2)])
def test_conv_big_tile_path(cin, cout, k, stride):
    """Shapes large enough for the 8-wave 256x256 tile (Cout % 256 == 0, >= 2 x CUs tiles): ragged last row tile,
    fused ReLU-on-load / residuals / activation, two-pass LDS epilogue.  fp64 reference on a pixel sample (the full
    fp64 convolution of 2e5 pixels is slow); every pixel is checked for finiteness."""
    from iggt_official_amd.heads import convops as co

    N, H, W = 6, 150, 151            # 135 900 pixels at stride 1: 531 row tiles (the last one ragged)
    pad = k // 2
    conv = nn.Conv2d(cin, cout, k, stride, pad).cuda()
    with torch.no_grad():
        conv.weight.copy_(_mk(conv.weight.shape, 21, (cin * k * k) ** -0.5))
        conv.bias.copy_(_mk(conv.bias.shape, 22, 0.1))
    x = _mk((N, H, W, cin), 23)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = _mk((N, Ho, Wo, cout), 24)
    y = co.run(co.pack_conv2d(conv), x, relu_in=True, res=res, relu_res=True, act=0)
    assert torch.isfinite(y).all() and y.shape == (N, Ho, Wo, cout)
    # reference on image 0's top rows and the last image's bottom rows (covers the first and the ragged last tile)
    for img, rows in ((0, slice(0, 12)), (N - 1, slice(Ho - 12, Ho))):
        r0 = rows.start * stride
        xin = x[img:img + 1, max(r0 - pad, 0):min((rows.stop - 1) * stride + k - pad, H)]
        top_pad = pad if r0 - pad < 0 else 0
        bot_pad = pad if (rows.stop - 1) * stride + k - pad > H else 0
        xd = F.pad(F.relu(xin.permute(0, 3, 1, 2).double()), (pad, pad, top_pad, bot_pad))
        ref = F.conv2d(xd, conv.weight.double(), conv.bias.double(), stride, 0)
        ref = ref + F.relu(res[img:img + 1, rows].permute(0, 3, 1, 2).double())
        got = y[img:img + 1, rows].permute(0, 3, 1, 2)
        assert got.shape == ref.shape, (got.shape, ref.shape)
        assert _err(got, ref)[0] < 2e-5


def test_conv_transpose_big_tile_path():
    """ConvTranspose2d(k = s = 4) of the DPT resize stage at the size of a 32-view pass: the GEMM has 4096 output columns
    (pixel-shuffle scatter epilogue) and enough row tiles for the 256x256 tile."""
    from iggt_official_amd.heads import convops as co

    s = 4
    ct = nn.ConvTranspose2d(256, 256, s, s, 0).cuda()
    x = _mk((32, 37, 37, 256), 31)
    y = co.run(co.pack_convT_kernel_eq_stride(ct), x)
    assert y.shape == (32, 37 * s, 37 * s, 256) and torch.isfinite(y).all()
    for img in (0, 31):
        ref = F.conv_transpose2d(x[img:img + 1].permute(0, 3, 1, 2).double(), ct.weight.double(), ct.bias.double(), s, 0)
        assert _err(y[img:img + 1].permute(0, 3, 1, 2), ref)[0] < 2e-5


@pytest.mark.parametrize("hw_in,size,cout,act,pos", [((37, 45), (64, 78), 4, "inv_log", True),
                                                     ((19, 19), (37, 37), 2, "exp", True),
                                                     ((30, 41), (56, 70), 4, "linear", False)])
def test_dpt_tail_fused_vs_separate_kernels(hw_in, size, cout, act, pos):
    """Fused upsample + position map + conv3x3 + ReLU + conv1x1 + activate_head kernel against the chain of separately
    tested kernels (bilinear resize, implicit-GEMM conv, head tail) and against an fp64 PyTorch evaluation."""
    from iggt_official_amd import _C
    from iggt_official_amd.heads import convops as co

    N = 3
    x = _mk((N, hw_in[0], hw_in[1], 128), 41)
    conv1 = nn.Conv2d(128, 32, 3, 1, 1).cuda()
    with torch.no_grad():
        conv1.weight.copy_(_mk(conv1.weight.shape, 42, (128 * 9) ** -0.5))
        conv1.bias.copy_(_mk((32,), 43, 0.1))
    w2, b2 = _mk((cout, 32), 44, 0.2), _mk((cout,), 45, 0.2)
    xr = _mk((size[1], 64), 46, 0.1) if pos else None
    yr = _mk((size[0], 64), 47, 0.1) if pos else None
    pc = co.pack_conv2d(conv1)
    pts, conf = _C.dpt_tail(x, size, xr, yr, pc.w_hi, pc.w_lo, pc.bias, w2, b2, act, "expp1")
    up = co.resize(x, size, xr, yr)
    mid = co.run(pc, up, act=1)
    pts_s, conf_s = _C.head_tail(mid, w2, b2, act, "expp1")
    assert pts.shape == pts_s.shape and conf.shape == conf_s.shape
    assert _err(pts, pts_s)[0] < 1e-5 and _err(conf, conf_s)[0] < 1e-5
    # fp64 reference
    xd = x.permute(0, 3, 1, 2).double()
    upd = F.interpolate(xd, size=size, mode="bilinear", align_corners=True)
    if pos:
        upd = upd + torch.cat([xr.t()[None, :, None, :].expand(1, 64, size[0], size[1]),
                               yr.t()[None, :, :, None].expand(1, 64, size[0], size[1])], 1).double()
    lin = F.conv2d(F.relu(F.conv2d(upd, conv1.weight.double(), conv1.bias.double(), 1, 1)), w2.double()[:, :, None, None],
                   b2.double()).permute(0, 2, 3, 1)
    xyz = lin[..., :-1]
    ref = {"inv_log": torch.sign(xyz) * torch.expm1(xyz.abs()), "exp": xyz.exp(), "linear": xyz}[act]
    assert _err(pts, ref)[0] < 5e-5 and _err(conf, 1 + lin[..., -1].exp())[0] < 5e-5


def test_dpt_tail_part_head_mode_nchw_no_activation():
    """Round 6: the same fused kernel as the part head's tail (reference part_head.py:228-243: bilinear upsample, conv3x3 128 ->
    32, ReLU, conv1x1 32 -> 8, NO position map, NO activation), all 8 channels as NCHW planes -- against the separate kernels it
    replaces (resize, implicit-GEMM conv, conv1x1_c32_nchw) and an fp64 PyTorch evaluation; ragged tile edges."""
    from iggt_official_amd import _C
    from iggt_official_amd.heads import convops as co

    N, hw_in, size, cout = 3, (38, 46), (67, 83), 8
    x = _mk((N, hw_in[0], hw_in[1], 128), 141)
    conv1 = nn.Conv2d(128, 32, 3, 1, 1).cuda()
    with torch.no_grad():
        conv1.weight.copy_(_mk(conv1.weight.shape, 142, (128 * 9) ** -0.5))
        conv1.bias.copy_(_mk((32,), 143, 0.1))
    w2, b2 = _mk((cout, 32), 144, 0.2), _mk((cout,), 145, 0.2)
    pc = co.pack_conv2d(conv1)
    out = _C.dpt_tail(x, size, None, None, pc.w_hi, pc.w_lo, pc.bias, w2, b2, "linear", "expp1", nchw=True)
    assert out.shape == (N, cout, size[0], size[1]) and out.is_contiguous()
    mid = co.run(pc, co.resize(x, size), act=1)
    sep = _C.conv1x1_c32_nchw(mid, w2.view(cout, 32, 1, 1), b2)
    assert _err(out, sep)[0] < 1e-5
    upd = F.interpolate(x.permute(0, 3, 1, 2).double(), size=size, mode="bilinear", align_corners=True)
    ref = F.conv2d(F.relu(F.conv2d(upd, conv1.weight.double(), conv1.bias.double(), 1, 1)), w2.double()[:, :, None, None], b2.double())
    assert _err(out, ref)[0] < 5e-5


# ---------------------------------------------------------------------------------------------------------------------------
# prec = 2: fp16 activations hi + lo x weights rounded once to fp16 + mean-input compensation per border class
# (csrc/conv_meancomp.hip).  Three checks per geometry, the first two EXACT (fp32-grade) by construction:
#   (a) weights that are fp16 numbers: nothing is rounded, the correction is the bias -> the MFMA path alone
#   (b) generic weights, an input that is constant per channel: x == mu everywhere, so the correction restores the weight
#       rounding completely -- including the first / last rows and columns, where the padded taps drop out (all nine classes)
#   (c) generic weights, rectified noise with a large mean: the statistical gate (weight rounding 2^-12, mean removed)
@pytest.fixture
def force_prec2():
    """convops.run demotes small prec-2 requests to prec 3 (PREC2_MIN_FLOPS); the kernel tests want the prec-2 kernels."""
    from iggt_official_amd.heads import convops as co

    old, co.PREC2_MIN_FLOPS = co.PREC2_MIN_FLOPS, 0.0
    yield
    co.PREC2_MIN_FLOPS = old


_P2 = [  # cin, cout, k, stride, pad, n, (h, w)                       kernel the launcher picks
    (256, 256, 3, 1, 1, 2, (74, 74)),      # halo 256 x (16 x 16 tile)
    (256, 128, 3, 1, 1, 1, (70, 100)),     # halo 128 x (8 x 32)
    (64, 128, 3, 1, 1, 3, (16, 50)),       # halo, two 32-channel slices
    (256, 256, 3, 1, 1, 3, (37, 41)),      # GEMM-shaped 128 x 128 (map below the halo threshold)
    (1024, 256, 3, 1, 1, 4, (19, 19)),     # split K
    (1024, 1024, 3, 2, 1, 4, (37, 37)),    # stride 2 (odd map: the last row sees the padding), split K
    (256, 256, 3, 2, 1, 2, (24, 36)),      # stride 2, even map: the last row / column sees no padding
    (256, 256, 1, 1, 0, 3, (30, 50)),      # 1 x 1: one correction vector through the bias path
    (128, 32, 3, 1, 1, 2, (56, 70)),       # narrow column tile
    (128, 64, 3, 1, 1, 1, (16, 24)),
    (256, 256, 3, 1, 1, 6, (150, 151)),    # halo, ragged tiles both ways, sampled mean (step > 1)
    (64, 512, 1, 1, 0, 6, (150, 151)),     # 256 x 256 GEMM tile
    (128, 256, 3, 2, 1, 6, (150, 151)),    # 256 x 256 GEMM tile with border classes
]


def _p2_conv(cin, cout, k, stride, pad, seed, fp16_weights):
    conv = nn.Conv2d(cin, cout, k, stride, pad).cuda()
    with torch.no_grad():
        w = _mk(conv.weight.shape, seed, (cin * k * k) ** -0.5)
        conv.weight.copy_(w.half().float() if fp16_weights else w)
        conv.bias.copy_(_mk(conv.bias.shape, seed + 1, 0.1))
    return conv


def _ref64(conv, x_nhwc, relu_in=False):
    xd = x_nhwc.permute(0, 3, 1, 2).double()
    return F.conv2d(F.relu(xd) if relu_in else xd, conv.weight.double(), conv.bias.double(), conv.stride, conv.padding)


@pytest.mark.parametrize("cin,cout,k,stride,pad,n,hw", _P2)
def test_conv_prec2_exact_cases_and_statistical_gate(cin, cout, k, stride, pad, n, hw, force_prec2):
    from iggt_official_amd.heads import convops as co

    big = n * hw[0] * hw[1] > 50000                      # fp64 reference on a crop of the first and last image only
    def check(conv, x, gate_max, gate_l2=None, relu_in=False, tag=""):
        y = co.run(co.pack_conv2d(conv), x, prec=2, relu_in=relu_in)
        assert torch.isfinite(y).all()
        worst = (0.0, 0.0)
        for img in ((0, n - 1) if big else range(n)):
            ref = _ref64(conv, x[img:img + 1], relu_in)
            got = y[img:img + 1].permute(0, 3, 1, 2)
            worst = max(worst, _err(got, ref))
        report(f"conv_p2_{cin}_{cout}_k{k}s{stride}_{hw[0]}x{hw[1]}{tag}", dict(max=worst[0], l2=worst[1]))
        assert worst[0] < gate_max and (gate_l2 is None or worst[1] < gate_l2), (tag, worst)

    # (a) fp16-representable weights
    check(_p2_conv(cin, cout, k, stride, pad, 51, True), _mk((n, hw[0], hw[1], cin), 53), 2e-5, tag="_a")
    # (b) per-channel constant input (positive and negative constants; no ReLU on load)
    const = _mk((1, 1, 1, cin), 54) + 0.5
    check(_p2_conv(cin, cout, k, stride, pad, 55, False), const.expand(n, hw[0], hw[1], cin).contiguous(), 2e-5, tag="_b")
    # (c) rectified noise on load, mean ~0.9 sigma: weight rounding minus its mean response
    check(_p2_conv(cin, cout, k, stride, pad, 57, False), _mk((n, hw[0], hw[1], cin), 59) + 0.5, 1.5e-3, 3e-4, relu_in=True,
          tag="_c")


def test_conv_prec2_fused_epilogue_and_residuals(force_prec2):
    """prec 2 through every epilogue feature (residuals, rectified residual, activations, in-place residual) on the halo kernel
    and on the GEMM-shaped kernel, with fp16-representable weights so that the comparison is fp32-grade."""
    from iggt_official_amd.heads import convops as co

    for hw in ((40, 64), (20, 28)):                      # halo kernel / GEMM-shaped kernel
        conv = _p2_conv(256, 256, 3, 1, 1, 61, True)
        pc = co.pack_conv2d(conv)
        x, r1, r2 = _mk((2, hw[0], hw[1], 256), 63), _mk((2, hw[0], hw[1], 256), 64), _mk((2, hw[0], hw[1], 256), 65)
        base = _ref64(conv, x, relu_in=True)
        y = co.run(pc, x, relu_in=True, res=r1, relu_res=True, res2=r2, prec=2)
        want = base + F.relu(r1.permute(0, 3, 1, 2).double()) + r2.permute(0, 3, 1, 2).double()
        assert _err(y.permute(0, 3, 1, 2), want)[0] < 2e-5
        for act, fn in [(1, F.relu), (2, lambda t: F.leaky_relu(t, 0.01)), (3, F.gelu)]:
            y = co.run(pc, x, relu_in=True, act=act, prec=2)
            assert _err(y.permute(0, 3, 1, 2), fn(base))[0] < 2e-5, act
        out = r1.clone()
        co.run(pc, x, relu_in=True, res=out, out=out, prec=2)
        assert _err(out.permute(0, 3, 1, 2), base + r1.permute(0, 3, 1, 2).double())[0] < 2e-5
        # the same pack serves prec 3 (its own planes, built from the fp32 weights) and agrees
        y3 = co.run(pc, x, relu_in=True, prec=3)
        assert _err(y3.permute(0, 3, 1, 2), base)[0] < 2e-5


@pytest.mark.parametrize("s", [2, 4])
def test_conv_prec2_transpose_kernel_eq_stride(s, force_prec2):
    from iggt_official_amd.heads import convops as co

    ct = nn.ConvTranspose2d(256, 256, s, s, 0).cuda()
    x = (_mk((2, 9, 13, 256), 12) + 0.5).relu()
    y = co.run(co.pack_convT_kernel_eq_stride(ct), x, prec=2)
    ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), ct.weight.double(), ct.bias.double(), s, 0)
    mx, l2 = _err(y.permute(0, 3, 1, 2), ref)
    assert mx < 1.5e-3 and l2 < 3e-4, (mx, l2)
    xc = (_mk((1, 1, 1, 256), 13) + 0.5).expand(2, 9, 13, 256).contiguous()          # constant per channel: exact
    y = co.run(co.pack_convT_kernel_eq_stride(ct), xc, prec=2)
    ref = F.conv_transpose2d(xc.permute(0, 3, 1, 2).double(), ct.weight.double(), ct.bias.double(), s, 0)
    assert _err(y.permute(0, 3, 1, 2), ref)[0] < 2e-5


def test_conv_prec2_transpose_k4s2p1_border_classes(force_prec2):
    """The four 2 x 2 parity convolutions of ConvTranspose2d(k4, s2, p1) pad one side only: per-axis border masks differ between
    the first and the last placement.  Constant-per-channel input -> exact."""
    from iggt_official_amd.heads import convops as co

    ct = nn.ConvTranspose2d(256, 256, 4, 2, 1).cuda()
    xc = (_mk((1, 1, 1, 256), 14) + 0.5).expand(2, 7, 10, 256).contiguous()
    y = co.run_convT_k4s2p1(co.pack_convT_k4s2p1(ct), xc, prec=2)
    ref = F.conv_transpose2d(xc.permute(0, 3, 1, 2).double(), ct.weight.double(), ct.bias.double(), 2, 1)
    assert _err(y.permute(0, 3, 1, 2), ref)[0] < 2e-5


def test_conv_prec2_small_problems_run_at_prec3():
    """Below PREC2_MIN_FLOPS a prec-2 request runs the split-bf16 kernels: fp32-grade on generic weights."""
    from iggt_official_amd.heads import convops as co

    conv = _p2_conv(256, 256, 3, 1, 1, 71, False)
    x = _mk((2, 20, 28, 256), 73)
    y = co.run(co.pack_conv2d(conv), x, prec=2)
    assert _err(y.permute(0, 3, 1, 2), _ref64(conv, x))[0] < 2e-5


def test_fusion_block_out_conv_before_resize_is_the_same_map():
    """FeatureFusionBlock: the 1 x 1 out_conv applied before the bilinear upsampling (dpt_head.OUT_CONV_FIRST) against the
    reference's order (interpolate, then out_conv: dpt_head.py:471-479) and against fp64 PyTorch."""
    from iggt_official_amd.heads import dpt_head as dh

    blk = dh._make_fusion_block(256).cuda()
    with torch.no_grad():
        for i, prm in enumerate(blk.parameters()):
            prm.copy_(_mk(prm.shape, 80 + i, 0.02 if prm.dim() == 4 else 0.1))
    x0, x1 = _mk((2, 20, 28, 256), 90), _mk((2, 20, 28, 256), 91)
    old = dh.OUT_CONV_FIRST
    try:
        dh.OUT_CONV_FIRST = True
        first = blk.forward_nhwc(x0, x1, size=(37, 53))
        dh.OUT_CONV_FIRST = False
        after = blk.forward_nhwc(x0, x1, size=(37, 53))
    finally:
        dh.OUT_CONV_FIRST = old
    assert first.shape == after.shape == (2, 37, 53, 256)
    assert _err(first, after)[0] < 2e-5             # both orders run split-bf16 products (2^-16 each)
    xd0, xd1 = x0.permute(0, 3, 1, 2).double(), x1.permute(0, 3, 1, 2).double()
    b = blk.double()
    def rcu(m, t):
        r = F.relu(t)
        return m.conv2(F.relu(m.conv1(r))) + r
    y = rcu(b.resConfUnit2, xd0 + rcu(b.resConfUnit1, xd1))
    ref = b.out_conv(F.interpolate(y, size=(37, 53), mode="bilinear", align_corners=True))
    assert _err(first.permute(0, 3, 1, 2), ref)[0] < 2e-5


@pytest.mark.parametrize("C", [5, 8, 13])
def test_custom_interpolate_any_channel_count(C):
    """Reference dpt_head.py:484-509: a general bilinear / align_corners wrapper.  The HIP resize kernel moves 8 channels per
    thread; other channel counts are zero-padded for the call, and the result is a contiguous fp32 NCHW tensor."""
    from iggt_official_amd.heads.dpt_head import custom_interpolate

    g = torch.Generator(device="cuda").manual_seed(C)
    x = torch.randn(2, C, 19, 23, generator=g, device="cuda")
    y = custom_interpolate(x, size=(37, 52))
    ref = torch.nn.functional.interpolate(x.double(), size=(37, 52), mode="bilinear", align_corners=True)
    assert y.shape == (2, C, 37, 52) and y.is_contiguous() and y.dtype == torch.float32
    assert float((y.double() - ref).abs().max()) < 1e-5
    y2 = custom_interpolate(x, scale_factor=2)
    assert y2.shape == (2, C, 38, 46)
