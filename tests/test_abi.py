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
