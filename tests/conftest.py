"""pytest configuration: registers the `gpu` marker and puts the oracle + drop-in package on sys.path.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks, gloo world_size-2 tests (CPU only).
`-m gpu`     : HIP path vs oracle / golden, called through the C-ABI (needs an MI355X).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "glow-tts-train_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The C oracle is tiny; build it on demand so a fresh checkout can run the CPU suite directly."""
    import subprocess

    so = os.path.join(ROOT, "oracle", "libmas_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    yield


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The parity margins the full-size tests of THIS run measured (tests/test_full_size_parity.py -> gpurun_out/parity_margins.json)
    become part of the test log: error / tolerance per case and arithmetic, whoever runs the suite and wherever its files go."""
    import json

    path = os.path.join(ROOT, "gpurun_out", "parity_margins.json")
    started = getattr(config, "_glowtts_session_start", None)
    try:
        if started is None or os.path.getmtime(path) < started:
            return
        data = json.load(open(path))
    except (OSError, ValueError):
        return
    tr = terminalreporter
    tr.section("parity margins (error / tolerance; source digest %s)" % data.get("source_digest"))
    for test in sorted(k for k in data if k != "source_digest"):
        for arith, f in sorted(data[test].items()):
            nums = ", ".join(f"{k}={v:.3g}" for k, v in sorted(f.items()) if isinstance(v, float) and k.endswith(("over_tol", "z", "dx", "logdet")))
            extra = f.get("worst_grad_key", "")
            frames = f" frames differing {f['n_alignment_frames_differing']}/{f['n_frames']}" if "n_frames" in f else ""
            tr.write_line(f"{test} [{arith}]: {nums} ({extra}){frames}")


def pytest_sessionstart(session):
    import time

    session.config._glowtts_session_start = time.time() - 1.0
