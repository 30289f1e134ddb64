"""Host-side decisions of the instance-feature branch that need no GPU: which MLPs of the window-attention stages take the trunk's
16-bit GEMM path (heads/tokenops.py, round 6), the precision switches of the head convolutions (heads/convops.py), and that the
token ops refuse to run without the HIP extension's device (no CPU fallback: reference window_sa.py:83-99,225,317 on torch ops)."""
import pytest
import torch
import torch.nn as nn


def test_mlp_h16_applicability_table(monkeypatch):
    from iggt_official_amd import precision
    from iggt_official_amd.heads import convops as co
    from iggt_official_amd.heads import tokenops as tk

    hab = (nn.Linear(128, 512), nn.Linear(512, 128))      # HAB at the 8g stage: dim 128, mlp_ratio 4
    ocab = (nn.Linear(256, 512), nn.Linear(512, 256))     # OCAB at the 4g stage: dim 256, mlp_ratio 2
    big = 32 * 304 * 304
    old = precision.operand_dtype()
    try:
        precision.set_operand_dtype("fp16")
        monkeypatch.setattr(co, "PART_PREC", 2)
        assert tk.mlp_h16_applicable(*hab, big) and tk.mlp_h16_applicable(*ocab, big)
        assert not tk.mlp_h16_applicable(*hab, tk._H16_MIN_TOKENS - 1)            # tiny maps: the fp32-grade Linears
        assert tk.mlp_h16_applicable(*hab, tk._H16_MIN_TOKENS)
        assert not tk.mlp_h16_applicable(nn.Linear(96, 384), nn.Linear(384, 96), big)     # K not a multiple of 64
        assert not tk.mlp_h16_applicable(nn.Linear(128, 320), nn.Linear(320, 128), big)   # hidden width not a multiple of 256
        assert not tk.mlp_h16_applicable(nn.Linear(128, 512), nn.Linear(512, 64), big)    # not a residual MLP (out != in)
        monkeypatch.setattr(co, "PART_PREC", 3)                                           # IGGT_PART_CONV_PREC=3: everything fp32-grade
        assert not tk.mlp_h16_applicable(*hab, big)
        monkeypatch.setattr(co, "PART_PREC", 2)
        precision.set_operand_dtype("bf16")                                               # bf16 operands: the 6e-3 mode stays out of the heads
        assert not tk.mlp_h16_applicable(*hab, big)
    finally:
        precision.set_operand_dtype(old)


def test_conv_precision_switches(monkeypatch):
    from iggt_official_amd.heads import convops as co

    monkeypatch.delenv("IGGT_X_PREC", raising=False)
    assert co._env_prec("IGGT_X_PREC", 2) == 2
    monkeypatch.setenv("IGGT_X_PREC", "3")
    assert co._env_prec("IGGT_X_PREC", 2) == 3
    monkeypatch.setenv("IGGT_X_PREC", " 2 ")
    assert co._env_prec("IGGT_X_PREC", 3) == 2
    monkeypatch.setenv("IGGT_X_PREC", "1")        # one pass is not a supported precision (5-7e-4 of head-only error)
    with pytest.raises(ValueError):
        co._env_prec("IGGT_X_PREC", 2)
    assert co.PART_PREC in (2, 3) and co.DPT_PREC in (2, 3) and co.PREC2_MIN_FLOPS == 1.0e11


def test_token_ops_refuse_cpu_tensors():
    from iggt_official_amd import _C
    from iggt_official_amd.heads import tokenops as tk

    x = torch.zeros(4, 128)
    with pytest.raises(_C.HipExtensionError):
        tk.layer_norm(nn.LayerNorm(128), x)
    with pytest.raises(_C.HipExtensionError):
        tk.linear(nn.Linear(128, 128), x)
    with pytest.raises(_C.HipExtensionError):
        tk.mlp_h16_(nn.LayerNorm(128), nn.Linear(128, 512), nn.Linear(512, 128), x)
