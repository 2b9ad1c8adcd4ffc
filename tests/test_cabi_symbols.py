"""CPU: the C-ABI library loads and exports every symbol include/m3vit_hip.h declares
(no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "m3vit_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(m3_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_bound_and_exported():
    from m3vit_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    names = _declared()
    assert len(names) >= 25
    assert sorted(_lib.SIGNATURES) == names, set(names) ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), n
    assert L.m3_version() >= 100
    assert L.m3_gate_num_blocks(25216) == 394
    assert L.m3_route_ws_elems(100864, 16) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from m3vit_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.M3Error):
        _lib.lib()
