"""The C-ABI library loads and exports every symbol include/wsae.h declares (no compute calls: CPU only)."""

from __future__ import annotations

import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "wsae.h"


def declared_functions() -> set:
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return set(re.findall(r"\b(wsae_[a-z0-9_]+)\s*\(", text))


def test_header_declares_the_documented_entry_points():
    names = declared_functions()
    for must in ("wsae_ctx_create", "wsae_encode_topk", "wsae_decode_loss", "wsae_weight_grads", "wsae_adamw_step",
                 "wsae_resample_dead", "wsae_ring_sample", "wsae_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from whisper_sae import _native as N
    lib = N.lib()
    assert lib.wsae_version() == 1
    declared = declared_functions()
    assert declared == set(N.SIGNATURES), (declared ^ set(N.SIGNATURES))
    out = subprocess.run(["nm", "-D", "--defined-only", str(N.library_path())], capture_output=True, text=True,
                         check=True).stdout
    exported = set(re.findall(r"\bT (wsae_[a-z0-9_]+)\b", out))
    assert declared <= exported, declared - exported
    for name in declared:
        assert getattr(lib, name) is not None


def test_pack_layout_matches_the_library():
    from whisper_sae import _native as N
    for d, h in ((384, 3072), (768, 12288), (1280, 40960), (64, 256), (32, 128)):
        total, off = N.pack_layout(d, h)
        assert N.param_count(d, h) == total == 2 * d * h + h + 2 * d
        assert N.param_offsets(d, h) == off


def test_kernel_names_and_struct_sizes():
    import ctypes as C

    from whisper_sae import _native as N
    assert C.sizeof(N.Stats) == 32 and C.sizeof(N.Config) == 24
    names = [N.lib().wsae_kernel_name(i).decode() for i in range(N.KERNEL_COUNT)]
    assert len(set(names)) == N.KERNEL_COUNT and "wgrad" in names and "encode_gemm" in names
    assert N.lib().wsae_kernel_name(N.KERNEL_COUNT) == b"?"


def test_argument_errors_do_not_need_a_gpu():
    from whisper_sae import _native as N
    rc = N.lib().wsae_param_offsets(0, 0, None)
    assert rc == -1 and "wsae_param_offsets" in N.last_error()
    with pytest.raises(N.WsaeError):
        N.check(rc, "wsae_param_offsets")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from whisper_sae import _native as N
    monkeypatch.setenv("WSAE_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(N, "_lib", None)
    with pytest.raises(N.WsaeError, match="no CPU path"):
        N.lib()
    monkeypatch.undo()
    N._lib = None
    assert N.lib().wsae_version() == 1
