"""ctypes binding of include/azg.h (libazg_hip.so).  There is no fallback: if the library is missing or no GPU is
visible, every compute entry point raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AZG_LIB_PATH') or os.path.join(HERE, 'lib', 'libazg_hip.so')   # (override: measurement builds)
ABI_VERSION = 6

GAME_CONNECT4, GAME_BRANDUBH, GAME_TRIMOK = 0, 1, 2
E_INVALID_ARG, E_HIP, E_INVALID_ACTION, E_TREE_FULL, E_EXAMPLES_FULL, E_UNSUPPORTED, E_INTERNAL, E_FLOATING_POINT = -1, -2, -3, -4, -5, -6, -7, -8
ERROR_NAMES = {-1: 'AZG_E_INVALID_ARG', -2: 'AZG_E_HIP', -3: 'AZG_E_INVALID_ACTION', -4: 'AZG_E_TREE_FULL',
               -5: 'AZG_E_EXAMPLES_FULL', -6: 'AZG_E_UNSUPPORTED', -7: 'AZG_E_INTERNAL', -8: 'AZG_E_FLOATING_POINT'}


class State(C.Structure):
    _fields_ = [('cells', C.c_int8 * 64), ('player', C.c_int32), ('turns', C.c_int32), ('aux', C.c_int32 * 2)]


class GameInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('action_size', 'obs_c', 'obs_h', 'obs_w', 'num_players', 'has_draw',
                                          'max_turns', 'num_symmetries', 'cells', 'max_children')]


class Config(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('game', C.c_int32), ('device', C.c_int32), ('num_slots', C.c_int32),
                ('arena', C.c_int32), ('nodes_per_tree', C.c_int32), ('example_capacity', C.c_int32),
                ('result_capacity', C.c_int32),
                ('cpuct', C.c_float), ('fpu_reduction', C.c_float), ('root_noise_frac', C.c_float),
                ('root_policy_temp', C.c_float), ('min_discount', C.c_float),
                ('add_root_noise', C.c_int32), ('add_root_temp', C.c_int32), ('symmetric_samples', C.c_int32),
                ('mcts_reset_threshold', C.c_int32), ('games_per_iteration', C.c_int32),
                ('start_temp', C.c_float), ('arena_temp', C.c_float), ('temp_table_len', C.c_int32),
                ('sims_per_move', C.c_int32), ('temp_table', C.POINTER(C.c_float)), ('tape_seed', C.c_uint64), ('slot_base', C.c_uint64)]


class Counters(C.Structure):
    _fields_ = [('sims', C.c_int64), ('expansions', C.c_int64), ('games_played', C.c_int32), ('num_results', C.c_int32),
                ('num_examples', C.c_int32), ('error', C.c_int32), ('max_nodes_used', C.c_int32), ('max_nodes_kept', C.c_int32)]


# every symbol include/azg.h declares: name -> (restype, argtypes)
_vp, _i, _u64, _f, _d = C.c_void_p, C.c_int, C.c_uint64, C.c_float, C.c_double
_i32p, _f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
SYMBOLS = {
    'azg_abi_version': (_i, []),
    'azg_source_sha': (C.c_char_p, []),
    'azg_last_error': (C.c_char_p, []),
    'azg_game_info_get': (_i, [_i, C.POINTER(GameInfo)]),
    'azg_device_count': (_i, []),
    'azg_engine_create': (_i, [C.POINTER(Config), C.POINTER(_vp)]),
    'azg_engine_destroy': (_i, [_vp]),
    'azg_engine_reset': (_i, [_vp, _vp]),
    'azg_engine_info': (_i, [_vp, _i32p]),
    'azg_set_root_flags': (_i, [_vp, _i]),
    'azg_set_states': (_i, [_vp, _vp, _i, _i, C.POINTER(State), _i]),
    'azg_get_states': (_i, [_vp, _vp, _i, _i, C.POINTER(State)]),
    'azg_get_leaf_states': (_i, [_vp, _vp, _i, _i, C.POINTER(State)]),
    'azg_set_tape_counters': (_i, [_vp, _vp, _i, _i, C.POINTER(_u64)]),
    'azg_get_tape_counters': (_i, [_vp, _vp, _i, _i, C.POINTER(_u64)]),
    'azg_select': (_i, [_vp, _vp, _vp, _i, _vp]),
    'azg_arena_rows': (_i, [_vp, _vp, _i32p, _vp, _vp]),
    'azg_arena_rows_seats': (_i, [_vp, _vp, _vp, _vp, _vp]),
    'azg_backup': (_i, [_vp, _vp, _vp, _vp, _vp, _i]),
    'azg_backup_select': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i]),
    'azg_backup_select_logits': (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i]),
    'azg_backup_select_features': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i]),
    'azg_leaf_heads_sparse_f16': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i]),
    'azg_heads_softmax': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'azg_advance': (_i, [_vp, _vp, _i]),
    'azg_advance_begin': (_i, [_vp, _vp, _i, _i32p]),
    'azg_advance_commit': (_i, [_vp, _vp, _i32p]),
    'azg_root_counts': (_i, [_vp, _vp, _vp]),
    'azg_root_probs': (_i, [_vp, _vp, _f, _vp]),
    'azg_root_value': (_i, [_vp, _vp, _i, _vp]),
    'azg_update_root': (_i, [_vp, _vp, _i, _i]),
    'azg_compact': (_i, [_vp, _vp, _i, _i]),
    'azg_slot_export': (C.c_int64, [_vp, _vp, _i, _vp, C.c_int64]),
    'azg_slot_import': (_i, [_vp, _vp, _i, _vp, C.c_int64]),
    'azg_root_children': (_i, [_vp, _vp, _i, _i, _i, _i32p, _i32p, _f32p, _f32p, _f32p]),
    'azg_node_children': (_i, [_vp, _vp, _i, _i, _i, _i, _i32p, _i32p, _i32p, _f32p, _f32p, _f32p, _i32p, _i32p]),
    'azg_reset_max_depth': (_i, [_vp, _vp]),
    'azg_tree_info': (_i, [_vp, _vp, _i, _i, _i32p]),
    'azg_last_path': (_i, [_vp, _vp, _i, _i, _i, _i32p]),
    'azg_read_counters': (_i, [_vp, _vp, C.POINTER(Counters)]),
    'azg_examples_dev': (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    'azg_copy_examples': (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    'azg_read_results': (_i, [_vp, _vp, _i, _i, C.POINTER(C.c_uint8), _i32p, _i32p]),
    'azg_clear_outputs': (_i, [_vp, _vp]),
    'azg_last_actions_dev': (_i, [_vp, C.POINTER(_vp)]),
    'azg_tower_weights_size': (C.c_int64, [_i, _i]),
    'azg_resnet_tower_f16': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i]),
    'azg_resnet_policy_value_f16': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    'azg_resnet_policy_value_multi_f16': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    'azg_search_f16': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i]),
    'azg_policy_value_heads_f16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'azg_tower_layout': (_i, [_i, _i, _i, _vp, _vp, _vp]),
    'azg_resnet_tower_features_f16': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i]),
    'azg_policy_value_heads_fact_f16': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'azg_search_arena_f16': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i]),
    'azg_obs_to_nhwc8_f16': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'azg_set_shuffle_tape': (_i, [_vp, _vp, _vp, _i]),
    'azg_set_random_tape': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    'azg_search_wide_f16': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i]),
    'azg_search_wide_exact_f16': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    'azg_search_wide_tile_info': (_i, [_vp, _i, _i, _i, _i32p]),
    'azg_debug_bounds_site': (_i, [_vp, _vp, _i32p, _i32p]),
    'azg_profile_net_enable': (_i, [_i]),
    'azg_profile_net_read': (_i, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'azg_profile_enable': (_i, [_vp, _i]),
    'azg_profile_read': (_i, [_vp, C.POINTER(_d), C.POINTER(C.c_int64)]),
    'azg_tape_u64': (_u64, [_u64, _u64, _u64]),
    'azg_tape_uniform': (_d, [_u64, _u64, _u64]),
    'azg_tape_shuffle_pos': (None, [_u64, _u64, _u64, _i, _i32p]),
}

_LIB = None


class AzgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('%s: %s' % (ERROR_NAMES.get(code, code), msg))
        self.code = code


def lib():
    """Load libazg_hip.so (after torch, so that both share torch's HIP runtime)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError('libazg_hip.so is not built (%s). Run `python -m alphazero_general_amd.build`; '
                          'there is no CPU fallback.' % LIB_PATH)
    try:
        import torch  # noqa: F401  (loads libamdhip64 first: one HIP runtime per process)
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    if L.azg_abi_version() != ABI_VERSION:
        raise ImportError('libazg_hip.so ABI version mismatch')
    _LIB = L
    return L


def check(rc):
    if rc < 0:
        msg = lib().azg_last_error().decode()
        if rc == E_INVALID_ACTION:
            raise ValueError(msg)
        raise AzgError(rc, msg)
    return rc


def game_info(game):
    gi = GameInfo()
    check(lib().azg_game_info_get(game, C.byref(gi)))
    return gi


def states_array(n):
    return (State * n)()


def state_to_np(s, cells):
    return np.frombuffer(bytes(s.cells), dtype=np.int8)[:cells].copy()
