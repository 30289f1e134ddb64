"""Checkpoint ingestion on the device (GPU): the entry points real weights will come through.

  * `utils.model.load_checkpoint` (reference demo.py:112-119 + utils/model.py:27-55): the synthetic checkpoint is written to a
    FILE in the two forms checkpoints come in -- a bare state dict with the DDP "module." prefix, and the trainer's
    {"model": ...} wrapper -- and loaded into a fresh IGGT; the report fields are checked and the forward must equal the
    directly-filled model BIT FOR BIT (same parameters -> same packs -> same kernels);
  * `IGGT.save_pretrained` / `IGGT.from_pretrained` (the reference class is a PyTorchModelHubMixin, vggt.py:4,132): round trip
    through a safetensors directory, same bit-equality."""
import logging
import os

import pytest
import torch

from helpers import build_gpu_model, schema

pytestmark = pytest.mark.gpu


def _outputs_equal(a, b):
    for k, v in a.items():
        if k == "pose_enc":
            assert all(torch.equal(x, y) for x, y in zip(v, b[k])), k
        else:
            assert torch.equal(v, b[k]), k


def _fresh():
    from iggt.models.vggt import IGGT

    with torch.device("cuda"):
        return IGGT().eval()


def test_load_checkpoint_from_file_equals_the_directly_filled_model(tmp_path):
    from iggt_official_amd import precision
    from oracle import weights
    from utils.model import load_checkpoint               # reference import path (demo.py:39)

    ref_model = build_gpu_model("stress", 0)
    images = weights.make_images(2, 56, 84, seed=21, device="cuda")
    want = ref_model(images)
    want = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in want.items()}
    sd = {k: v.cpu() for k, v in weights.fill_state_dict(schema(), seed=0, mode="stress", device="cuda").items()}
    n_float = len(sd)
    path = str(tmp_path / "iggt.pt")
    for form in ("ddp_prefix", "trainer_wrapper"):
        if form == "ddp_prefix":
            torch.save({"module." + k: v for k, v in sd.items()}, path)
        else:
            torch.save({"model": dict(sd), "optimizer": {"state": {}}, "epoch": 7}, path)
        model = _fresh()
        rep = load_checkpoint(model, path, logger=logging.getLogger("ckpt-gpu-test"))
        os.remove(path)
        assert rep["loaded"] == n_float and rep["unexpected"] == []
        assert all(m.startswith("track_head.") or "relative_position_index" in m or "num_batches_tracked" in m
                   for m in rep["missing"]), rep["missing"]
        assert rep["beyond_fp16"] == [] and 0.0 < rep["max_abs_weight"] < 65504.0
        assert rep["escalated_blocks"] == [] and rep["ill_conditioned_blocks"] == [] and rep["bf16_blocks"] == []
        assert rep["min_participation_ratio"] > precision.ESC_PR_MIN and rep["max_logit_rms"] < precision.ESC_LOGIT_RMS_MAX
        # GPU-resident: the packs were built by load_checkpoint, so the lists above are final
        assert all(b._packed is not None and not b._packed["x3"] for b in model.aggregator.execution_order())
        _outputs_equal(want, model(images))
        del model
        torch.cuda.empty_cache()


def test_load_checkpoint_reports_the_rung_of_a_heavy_tailed_checkpoint():
    """The report is where a user learns what a checkpoint costs: the dose fixture sigma 0 / 1 escalates every block."""
    from iggt_official_amd import precision
    from oracle import weights
    from utils.model import load_checkpoint

    sd = weights.fill_state_dict(schema(), seed=0, mode="trained_like(qk=0,norm=1)", device="cuda")
    model = _fresh()
    rep = load_checkpoint(model, sd)
    assert len(rep["escalated_blocks"]) == 72 and rep["min_participation_ratio"] < precision.ESC_PR_MIN
    assert all(b._packed["x3"] for b in model.aggregator.execution_order())


def test_save_pretrained_from_pretrained_round_trip(tmp_path):
    from iggt.models.vggt import IGGT
    from oracle import weights

    ref_model = build_gpu_model("stress", 0)
    images = weights.make_images(2, 56, 56, seed=22, device="cuda")
    want = ref_model(images)
    want = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in want.items()}
    d = str(tmp_path / "hub")
    ref_model.save_pretrained(d)
    assert os.path.exists(os.path.join(d, "model.safetensors")) and os.path.exists(os.path.join(d, "config.json"))
    model = IGGT.from_pretrained(d)
    assert isinstance(model, IGGT) and not model.training and model.part_on_invalid_grid == ref_model.part_on_invalid_grid
    with pytest.raises(Exception):
        model(images.cpu())                       # CPU-resident after from_pretrained, as in the reference: no CPU path here
    model = model.to("cuda")
    _outputs_equal(want, model(images))
