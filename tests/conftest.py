import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cross-resolution-face-recognition_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _three_plane_fp32_by_default():
    """The fp32 matrix mode is process-wide state (ops.set_compute_dtype("fp32x2") selects two planes): a test that switched to two
    planes must not leak it into op tests that enter fp32 tensors directly."""
    yield
    ops = sys.modules.get("xrface.ops")
    if ops is not None:
        ops._cfg["f32_planes"] = 3


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_terminal_summary(terminalreporter):
    """Print (and save) the measured parity errors so DESIGN.md can quote them."""
    try:
        from tests.helpers import ERRORS
    except Exception:
        return
    if not ERRORS:
        return
    worst = {}
    for what, e, tol in ERRORS:
        if what not in worst or not (e <= worst[what][0]):
            worst[what] = (e, tol)
    rows = sorted(worst.items(), key=lambda kv: -(kv[1][0] / kv[1][1] if kv[1][0] == kv[1][0] else 9e9))
    terminalreporter.write_line("parity errors (measured / tolerance), worst first:")
    for what, (e, tol) in rows[:25]:
        terminalreporter.write_line(f"  {what:60s} {e:.3e} / {tol:.1e}")
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump({k: {"err": v[0], "tol": v[1]} for k, v in worst.items()}, f, indent=1)
