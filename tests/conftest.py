import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def observed():
    """record(name, **numbers): appends one JSON line per call to $BIFOLD_OBSERVED (when set) and prints it --
    how the observed parity errors of a GPU run are carried back to DESIGN.md and to the test tolerances."""
    import json

    def record(name, **values):
        line = json.dumps({"test": name, **{k: (float(f"{v:.4e}") if isinstance(v, float) else v)
                                            for k, v in values.items()}})
        print("OBSERVED", line)
        path = os.environ.get("BIFOLD_OBSERVED")
        if path:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "a") as f:
                f.write(line + "\n")
    return record
