import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def parity_report(line: str) -> None:
    """print a parity report line (flip counts, margin histograms, error ratios) AND append it to the file
    the round keeps: $DT_PARITY_REPORT, default gpurun_out/parity_report.txt (copied to profiles/ after the GPU run)"""
    print(line)
    path = os.environ.get("DT_PARITY_REPORT") or os.path.join(ROOT, "gpurun_out", "parity_report.txt")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
