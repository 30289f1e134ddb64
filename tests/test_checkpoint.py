"""Checkpoint ingestion, CPU side: `utils.model.load_checkpoint` -- the demo.py:112-119 sequence (read the
file, strip the DDP "module." prefix, align with the model's schema through `align_and_update_state_dicts`,
`load_state_dict(strict=False)`) plus the operand-range validation and the escalation plan -- on a CPU-RESIDENT model, where
nothing may be packed (packing happens on the device).  The GPU side (bit-equal forward after a load from file, save_pretrained /
from_pretrained) is tests/test_checkpoint_gpu.py."""
import logging
from unittest import mock

import pytest
import torch


def _empty_model():
    """IGGT() on the CPU without paying for 1.3 B random draws: the expensive initialisers are no-ops (the tests below fill what
    they read); LayerNorm scales keep their ones."""
    noop = lambda t, *a, **k: t  # noqa: E731
    with mock.patch("torch.nn.init.kaiming_uniform_", noop), mock.patch("torch.nn.init.uniform_", noop), \
            mock.patch("torch.nn.init.normal_", noop), mock.patch("torch.nn.init.trunc_normal_", noop):
        from iggt.models.vggt import IGGT

        return IGGT().eval()


@pytest.fixture(scope="module")
def model():
    return _empty_model()


def test_load_checkpoint_from_a_wrapped_ddp_file_into_a_cpu_model(model, tmp_path, caplog):
    from iggt_official_amd import precision
    from utils.model import load_checkpoint          # the reference's import path (demo.py:39), alias of iggt_official_amd.utils.model

    g = torch.Generator().manual_seed(0)
    ck = {}
    for k, v in model.state_dict().items():
        if k.startswith("aggregator.") and (".norm1." in k or ".norm2." in k or "q_norm" in k or "k_norm" in k):
            ck[k] = torch.ones_like(v) if k.endswith("weight") else torch.zeros_like(v)
    ck["aggregator.global_blocks.5.norm2.weight"] = torch.exp(1.5 * torch.randn(1024, generator=g))   # participation ratio << 0.15
    ck["aggregator.camera_token"] = torch.zeros(1, 2, 1, 1024)
    big = torch.zeros(1024, 1024)
    big[3, 7] = 1.0e5                                                                                  # beyond fp16's 65 504
    ck["aggregator.frame_blocks.0.attn.proj.weight"] = big
    ck["aggregator.register_token"] = torch.zeros(1, 2, 3, 1024)                                       # wrong shape: *UNMATCHED*
    ck["some.key.of.another.model"] = torch.zeros(3)                                                   # $UNUSED$
    n_match = len(ck) - 2
    path = str(tmp_path / "ckpt.pt")
    torch.save({"model": {"module." + k: v for k, v in ck.items()}, "epoch": 3}, path)     # trainer wrapper + DDP prefix

    old_dt = precision.operand_dtype()
    try:
        with caplog.at_level(logging.WARNING):
            rep = load_checkpoint(model, path, logger=logging.getLogger("ckpt-test"))
        assert precision.operand_dtype() == old_dt == torch.float16        # range folding on: no global switch to bf16
    finally:
        precision.set_operand_dtype(old_dt)
    assert rep["loaded"] == n_match
    assert "aggregator.register_token" in rep["missing"] and "aggregator.patch_embed.cls_token" in rep["missing"]
    assert rep["unexpected"] == []
    assert rep["beyond_fp16"] == ["aggregator.frame_blocks.0.attn.proj.weight"] and rep["max_abs_weight"] == pytest.approx(1.0e5)
    msgs = "\n".join(r.getMessage() for r in caplog.records)
    assert "*UNMATCHED* aggregator.register_token" in msgs and "$UNUSED$ some.key.of.another.model" in msgs
    assert "*UNLOADED* aggregator.patch_embed.cls_token" in msgs and "exceed the fp16 range" in msgs
    # the loaded values are in the model
    assert torch.equal(model.aggregator.global_blocks[5].norm2.weight, ck["aggregator.global_blocks.5.norm2.weight"])
    assert float(model.aggregator.frame_blocks[0].attn.proj.weight.detach()[3, 7]) == 1.0e5
    # escalation plan from the loaded LayerNorm scales: the ill-conditioned block and everything upstream of it
    assert rep["ill_conditioned_blocks"] == ["global_blocks.5"]
    want = [f"patch_embed.blocks.{i}" for i in range(24)] + [f"{k}_blocks.{i}" for i in range(6) for k in ("frame", "global")]
    assert rep["escalated_blocks"] == want
    assert rep["min_participation_ratio"] < precision.ESC_PR_MIN and rep["bf16_blocks"] == []
    assert "x3 precision rung" in msgs
    # CPU-resident: only the plan is made, nothing is packed (packs are device buffers)
    assert all(b._packed is None for b in model.aggregator.execution_order())


def test_load_checkpoint_accepts_a_state_dict_and_switches_to_bf16_without_range_folding(model):
    from iggt_official_amd import precision
    from iggt_official_amd.utils.model import load_checkpoint

    sd = {"aggregator.frame_blocks.1.mlp.fc1.weight": torch.full((4096, 1024), 7.0e4),
          "aggregator.global_blocks.5.norm2.weight": torch.ones(1024)}
    old_dt, old_fold = precision.operand_dtype(), precision.range_folding()
    try:
        precision.set_range_folding(False)
        rep = load_checkpoint(model, sd)
        assert rep["loaded"] == 2 and rep["beyond_fp16"] == ["aggregator.frame_blocks.1.mlp.fc1.weight"]
        assert precision.operand_dtype() == torch.bfloat16          # round-3 behaviour: the whole trunk, loudly
        assert rep["escalated_blocks"] == []                        # bf16 operands are never escalated
    finally:
        precision.set_operand_dtype(old_dt)
        precision.set_range_folding(old_fold)
        with torch.no_grad():
            model.aggregator.frame_blocks[1].mlp.fc1.weight.zero_()
