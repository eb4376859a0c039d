"""CPU-side checks of the drop-in boundary: libazg_hip.so loads and exports every symbol include/azg.h declares
(no compute calls without a GPU), and the product fails loudly when no HIP device is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    from alphazero_general_amd import build
    return build.build()


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'azg.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(azg_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_exported(built):
    L = C.CDLL(built)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), 'libazg_hip.so does not export %s' % s


def test_binding_covers_header(built):
    from alphazero_general_amd import _abi
    assert sorted(_abi.SYMBOLS) == header_symbols()


def test_game_info_and_tape_host_side(built):
    from alphazero_general_amd import _abi
    import oracle_lib as ol
    gi = _abi.game_info(_abi.GAME_CONNECT4)
    og = ol.game_info(ol.GAME_CONNECT4)
    for f in ('action_size', 'obs_c', 'obs_h', 'obs_w', 'num_players', 'has_draw', 'max_turns', 'num_symmetries', 'cells'):
        assert getattr(gi, f) == getattr(og, f)
    # the product's tape implementation is independent of the oracle's: same spec, same numbers
    L, O = _abi.lib(), ol.lib()
    for seed, stream, ctr in [(0, 0, 0), (1, 2, 3), (2 ** 63 + 5, 0x4000000000000000, 12345), (99, 1000, 7)]:
        assert L.azg_tape_u64(seed, stream, ctr) == O.azo_tape_u64(seed, stream, ctr)
        assert L.azg_tape_uniform(seed, stream, ctr) == O.azo_tape_uniform(seed, stream, ctr)
    import numpy as np
    for k in (1, 2, 7, 40, 64, 100):
        a = np.zeros(k, np.int32); b = np.zeros(k, np.int32)
        L.azg_tape_shuffle_pos(5, 6, 7, k, a.ctypes.data_as(C.POINTER(C.c_int32)))
        O.azo_tape_shuffle_pos(5, 6, 7, k, b)
        assert (a == b).all() and sorted(a) == list(range(k))


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from alphazero_general_amd import _abi
    from alphazero_general_amd.engine import DeviceEngine
    with pytest.raises(RuntimeError):
        DeviceEngine(_abi.GAME_CONNECT4, 4)
    cfg = _abi.Config()
    cfg.abi_version, cfg.game, cfg.num_slots = _abi.ABI_VERSION, 0, 4
    h = C.c_void_p()
    assert _abi.lib().azg_engine_create(C.byref(cfg), C.byref(h)) == _abi.E_HIP
    assert b'no CPU fallback' in _abi.lib().azg_last_error()


@pytest.mark.parametrize('game,bt,ch,H,W', [(0, 1, 128, 6, 7), (0, 2, 128, 6, 7), (0, 4, 128, 6, 7), (0, 4, 64, 6, 7),
                                            (1, 1, 64, 7, 7), (1, 2, 64, 7, 7), (1, 2, 128, 7, 7), (2, 2, 32, 5, 5), (2, 5, 32, 5, 5)])
def test_tower_lds_layout_invariants(game, bt, ch, H, W):
    """The LDS image of the MFMA tower (csrc/azg_conv.h TowerGeom / tower_pixmap), checked on the host for every instantiated
    shape: every pixel sits in exactly one (subtile, lane); all nine taps of a pixel are in-bounds rows; pad rows never
    coincide with pixel rows; and the bank-conflict rule DESIGN.md states -- the 8 lanes {0-3,12-15} and the 8 lanes {4-11} of
    a fragment read rows of pairwise different residue mod 8 (row stride = 2 (mod 4) 16-byte slots) -- holds wherever the
    residue classes allow it (at most one doubled residue per 8-lane set)."""
    import ctypes as C
    import numpy as np
    from alphazero_general_amd import _abi
    L = _abi.lib()
    info = (C.c_int32 * 8)()
    assert L.azg_tower_layout(game, bt, ch, None, None, info) == 0
    nsub, rows, rstride, trows, tile, pw, lead, bstride = list(info)
    assert rows == bt * H * W and nsub == (rows + 15) // 16 and tile == trows * rstride and tile <= 160 * 1024 // (2 if rows > 64 else 1)
    assert rstride == 2 * ch + 32 and (rstride // 16) % 4 == 2 and pw == W + 2 and bstride % 8 == 2
    pm = np.zeros(nsub * 16, np.int16); q = np.zeros(rows, np.int32)
    assert L.azg_tower_layout(game, bt, ch, pm.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), info) == 0
    live = pm[pm >= 0]
    assert sorted(live.tolist()) == list(range(rows))                        # a bijection pixels <-> live lanes
    assert len(set(q.tolist())) == rows                                      # distinct rows
    for p in range(rows):                                                    # every tap row is inside the image
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                r = q[p] + dy * pw + dx
                assert 0 <= r < trows
    pixel_rows = set(q.tolist())
    for p in range(rows):                                                    # off-board taps land on pad rows, never on a pixel
        b, pos = divmod(p, H * W); y, x = divmod(pos, W)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                inside = 0 <= y + dy < H and 0 <= x + dx < W
                r = q[p] + dy * pw + dx
                if inside:
                    assert r == q[b * H * W + (y + dy) * W + (x + dx)]
                else:
                    assert r not in pixel_rows
    setA, setB = (0, 1, 2, 3, 12, 13, 14, 15), (4, 5, 6, 7, 8, 9, 10, 11)
    doubled = 0
    for s in range(nsub):
        for lanes in (setA, setB):
            res = [q[pm[s * 16 + l]] % 8 for l in lanes if pm[s * 16 + l] >= 0]
            doubled += len(res) - len(set(res))
    assert doubled <= nsub                                                   # conflict-free up to the leftovers of unequal classes
    # border classes (TowerGeom::CLASSES): where the class-wise subtile count equals nsub, subtiles hold pixels of ONE class
    # -- [interior | top row | bottom row | left column | right column] or [middle rows | top row | bottom row] -- which is
    # what lets the main loop drop the taps that would only read the zero padding
    sub = lambda n: (n + 15) // 16
    five = [sub(bt * (H - 2) * (W - 2)), sub(bt * W), sub(bt * W), sub(bt * (H - 2)), sub(bt * (H - 2))]
    three = [sub(bt * (H - 2) * W), sub(bt * W), sub(bt * W)]
    counts = five if sum(five) == nsub else three if sum(three) == nsub else None
    assert (counts is not None) == ((bt, H, W) in {(4, 6, 7), (2, 6, 7), (2, 7, 7), (2, 5, 5)})
    if counts is not None:
        def pclass(p):
            y, x = divmod(p % (H * W), W)
            if y == 0: return 1
            if y == H - 1: return 2
            if len(counts) == 3: return 0
            return 3 if x == 0 else 4 if x == W - 1 else 0
        first = np.cumsum([0] + counts)
        for s in range(nsub):
            c = int(np.searchsorted(first, s, side='right') - 1)
            assert all(pclass(int(p)) == c for p in pm[s * 16:(s + 1) * 16] if p >= 0), (s, c)
        assert doubled == 0 or (bt, H, W) != (4, 6, 7)                       # the headline tile stays conflict-free
    assert L.azg_tower_layout(game, 3, ch, None, None, info) == _abi.E_UNSUPPORTED


def test_source_stamp_covers_everything_the_binary_is_made_from(tmp_path, monkeypatch):
    """build.source_sha -- compiled into the library as azg_source_sha and compared with the committed counters' stamp by bench.py --
    changes when ANY input of the binary does: a kernel header, the host translation unit (azg_engine.hip: tile choice, launch bounds,
    LDS sizing, the cost model), the public header, or a compile flag; and the library in the tree carries the stamp of the tree."""
    import shutil
    from alphazero_general_amd import build as b
    base = b.source_sha()
    assert len(base) == 16 and base == b.source_sha()
    assert len({b.source_sha(v) for v in ('product', 'debug', 'tuning', 'timing-tree', 'timing-tower')}) == 5      # flags are part of the stamp
    names = {os.path.basename(d) for d in b.DEPS}
    assert {'azg_engine.hip', 'azg_kernels.h', 'azg_conv.h', 'azg_games.h', 'azg_device.h', 'azg.h'} <= names
    for name in ('azg_engine.hip', 'azg_kernels.h', 'azg.h'):
        deps = []
        for d in b.DEPS:
            if os.path.basename(d) == name:
                c = str(tmp_path / name)
                shutil.copy(d, c)
                with open(c, 'a') as f:
                    f.write('\n// touched\n')
                d = c
            deps.append(d)
        monkeypatch.setattr(b, 'DEPS', deps)
        assert b.source_sha() != base, name
        monkeypatch.undo()
    assert b.source_sha() == base
    if os.path.exists(b.OUT) and not b.needs_build():                         # the in-tree library is the tree's build
        L = C.CDLL(b.OUT)
        L.azg_source_sha.restype = C.c_char_p
        assert L.azg_source_sha().decode() == base
