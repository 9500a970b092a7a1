"""pytest configuration: ``gpu`` marker, import paths, seeding (mirrors the reference's
tests/conftest.py:8-29 -- session ``device`` fixtures + autouse ``torch.manual_seed(42)``)."""

from __future__ import annotations

import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "whisper-sae_amd"
for p in (str(ROOT), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def device():
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def set_seed():
    torch.manual_seed(42)


@pytest.fixture(scope="session")
def parity_note():
    """``parity_note(key, measured, bound)``: append a measured parity gap to ``gpurun_out/parity_notes.jsonl`` (scratch;
    the GPU run's copy is committed as ``profiles/rNN_parity_notes.jsonl``) so that the slack of every multi-step
    tolerance is on record next to the bound the test enforces."""
    import json
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    path = out / "parity_notes.jsonl"

    def note(key: str, measured: float, bound: float) -> None:
        with path.open("a") as fh:
            fh.write(json.dumps({"key": key, "measured": float(measured), "bound": float(bound)}) + "\n")

    return note
