import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("MIOPEN_FIND_MODE", "2")  # heuristic conv algo pick: no per-shape benchmarking on a fresh box


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a MI355X (run with `pytest -m gpu` via gpurun)")


@pytest.fixture(scope="session")
def schema():
    with open(os.path.join(GOLDEN, "state_dict_schema.json")) as f:
        return json.load(f)


def load_golden(name):
    import torch

    p = os.path.join(GOLDEN, name + ".pt")
    if not os.path.exists(p):
        pytest.skip(f"golden fixture {name} missing")
    return torch.load(p, map_location="cpu", weights_only=False)


_REPORT = {}


def report(key, value):
    """Collect parity numbers; written to gpurun_out/parity_report.json at session end."""
    _REPORT[key] = value


def pytest_sessionfinish(session, exitstatus):
    """One report per run.  Under pytest-xdist every worker writes its own part and the controller (whose session ends after
    the workers') merges them."""
    import glob

    out = os.path.join(ROOT, "gpurun_out")
    worker = os.environ.get("PYTEST_XDIST_WORKER")
    if worker:
        if _REPORT:
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, f"parity_report_part_{worker}.json"), "w") as f:
                json.dump(_REPORT, f)
        return
    merged = dict(_REPORT)
    for part in sorted(glob.glob(os.path.join(out, "parity_report_part_*.json"))):
        try:
            merged.update(json.load(open(part)))
        except Exception:
            pass
        os.remove(part)
    if merged:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_report.json")
        old = {}
        if os.path.exists(path):
            try:
                old = json.load(open(path))
            except Exception:
                old = {}
        old.update(merged)
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
