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


def test_wgrad_plan_covers_every_group_layout():
    """host logic of the balanced grouped weight gradient: for any split of M rows over G groups the slab slots the
    host reserves (M // chunk + G) cover the units the kernel deals out (sum of ceil(rows_g / chunk)), the chunk is a
    multiple of the 32-row step, and groups at the mean size keep `splits` units"""
    import random
    from m3vit_amd import ops
    rng = random.Random(7)
    for _ in range(300):
        G = rng.choice([2, 4, 16, 64])
        M = rng.randint(1, 200000)
        N, K = rng.choice([(384, 384), (1536, 384), (768, 3072)])
        splits = ops.default_wgrad_splits(M, N, K, G)
        chunk, units = ops.wgrad_plan(M, G, splits, grouped=True)
        assert chunk % 32 == 0 and chunk >= 32
        cuts = sorted(rng.randint(0, M) for _ in range(G - 1))
        rows = [b - a for a, b in zip([0] + cuts, cuts + [M])]
        if rng.random() < 0.3:                      # everything on one group
            rows = [M] + [0] * (G - 1)
        need = sum(-(-r // chunk) for r in rows)
        assert need <= units, (M, G, splits, chunk, rows)
        assert -(-(M // G) // chunk) <= splits
        assert ops.wgrad_ws_elems(M, N, K, G, grouped=True) == units * N * (K + 1)
    assert ops.wgrad_plan(1000, 1, 4, grouped=True) == (0, 4)          # one group: equal parts
    assert ops.wgrad_plan(1000, 128, 2, grouped=True) == (0, 256)      # more groups than lanes: equal parts


def test_wide_wgrad_tile_rule_and_splits_on_the_host():
    """m3_wgrad_tile / m3_wgrad_set_wide are host code: the tile rule (128 x 384 for K = 384, 384 x 128 for N = 384, fp16
    only, off by default) and the split heuristic that goes with it (256 one-per-CU slots; grouped calls with many tiles
    prefer one unit per expert) - and the slab workspace the planner reserves always covers what the kernel indexes"""
    import pytest
    import torch
    from m3vit_amd import _lib, ops
    h = torch.float16
    assert ops.wgrad_tile(1536, 384, h) == (128, 128)
    # the default split rule follows the kernel that takes the launch (m3_wgrad_set_dma's rule): 1024 slots where the LDS-DMA
    # kernel runs (fp32; 16-bit weights of >= 1.5 M elements or launches whose tiles fill the chip), 512 for the register-staged one
    assert ops.default_wgrad_splits(25216, 1536, 384, 1, h) == 14 and ops.default_wgrad_splits(25216, 1536, 384, 1, torch.float32) == 28
    assert ops.wgrad_tile(3072, 768, h) == (256, 256) and ops.default_wgrad_splits(25216, 3072, 768, 1, h) == 7   # ViT-Base fc1: 36 big tiles on 256 slots
    ops.wgrad_set_big(0)
    try:
        assert ops.wgrad_tile(3072, 768, h) == (128, 128)
        assert ops.default_wgrad_splits(25216, 3072, 768, 1, h) == 7                # 144 tiles on 1024 slots
        ops.wgrad_set_dma(0)
        try:
            assert ops.default_wgrad_splits(25216, 3072, 768, 1, h) == 3 and ops.default_wgrad_splits(25216, 1536, 384, 1, torch.float32) == 14
        finally:
            ops.wgrad_set_dma(-1)
        assert ops.default_wgrad_splits(38432, 3072, 768, 16, h) == 1              # the ViT-Base experts: tiles fill the chip -> direct mode
    finally:
        ops.wgrad_set_big(-1)
    assert ops.default_wgrad_splits(38432, 3072, 768, 16, h) == 1                  # (and with the big tile: 576 tiles on 256 slots)
    # dense row parts come in whole multiples of the 8 XCDs where that costs at most an eighth of them (ops._xcd_aligned)
    assert [ops._xcd_aligned(n) for n in (1, 7, 8, 9, 14, 18, 28, 32, 37)] == [1, 7, 8, 8, 14, 16, 28, 32, 37]
    assert ops.default_wgrad_splits(25216, 1152, 384, 1, h) == 16 and ops.default_wgrad_splits(25216, 2304, 768, 1, h) == 8
    if not _lib.lib().m3_experimental():
        with pytest.raises(_lib.M3Error):
            ops.wgrad_set_wide(1)                                                   # the wide kernel is not in a default build
        return
    try:
        ops.wgrad_set_wide(0)
        ops.wgrad_set_wide(1)
        assert ops.wgrad_tile(1536, 384, h) == (128, 384) and ops.wgrad_tile(384, 1536, h) == (384, 128)
        assert ops.wgrad_tile(384, 384, h) == (128, 128) and ops.wgrad_tile(1536, 384, torch.float32) == (128, 128)
        assert ops.wgrad_tile(1536, 384, torch.bfloat16) == (128, 128)            # the wide kernel is fp16 only
        assert ops.default_wgrad_splits(25216, 1536, 384, 1, h) == 21              # 12 tiles x 21 = 252 of 256 slots
        assert ops.default_wgrad_splits(25216, 1152, 384, 1, h) == 28
        assert ops.default_wgrad_splits(100864, 1536, 384, 16, h) == 1             # 192 tile-experts: one unit each
        assert ops.default_wgrad_splits(100864, 1536, 384, 4, h) >= 2              # 48 tile-experts: split the rows
        for (M, N, K, G) in ((25216, 1536, 384, 1), (100864, 384, 1536, 16), (3000, 1152, 384, 1)):
            s = ops.default_wgrad_splits(M, N, K, G, h)
            _, units = ops.wgrad_plan(M, G, s, grouped=G > 1)
            assert ops.wgrad_ws_elems(M, N, K, G, grouped=G > 1, dtype=h) == units * N * (K + 1)
    finally:
        ops.wgrad_set_wide(0)
