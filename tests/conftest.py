import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LOG_DIR = os.path.join(ROOT, "gpurun_out")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def log_metric(name: str, **kv):
    """Append one parity measurement to gpurun_out/parity_log.jsonl (merged back from the GPU box)."""
    os.makedirs(LOG_DIR, exist_ok=True)
    rec = {"test": name}
    rec.update({k: (float(v) if hasattr(v, "__float__") else v) for k, v in kv.items()})
    with open(os.path.join(LOG_DIR, "parity_log.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")


@pytest.fixture
def metric_log():
    return log_metric
