"""Host-side helpers mirroring alphazero/utils.py of the reference (dotdict, temperature schedules)."""
import numpy as np


class dotdict(dict):
    """alphazero/utils.py:1-12"""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError
        return self[name]

    def __setattr__(self, key, value):
        self[key] = value

    def copy(self):
        return self.__class__(super().copy())


def scale_temp(scale_factor, min_temp, cur_temp, turns, const_max_turns):
    """alphazero/utils.py:19-23: halve every int(scale_factor * max_turns) turns down to min_temp."""
    if const_max_turns and (turns + 1) % int(scale_factor * const_max_turns) == 0:
        return max(min_temp, cur_temp / 2)
    return cur_temp


def default_temp_scaling(*args, **kwargs):
    """alphazero/utils.py:26-27"""
    return scale_temp(0.15, 0.2, *args, **kwargs)


def const_temp_scaling(temp, *args, **kwargs):
    """alphazero/utils.py:30-31"""
    return temp


def temp_table(temp_fn, start_temp, max_turns):
    """temp_by_turn[t]: the temperature SelfPlayAgent.playMoves uses for the move made at turn t, i.e.
    args.temp_scaling_fn iterated from args.startTemp (alphazero/SelfPlayAgent.pyx:156-157).  The callable is
    evaluated on the host once; the device indexes the table."""
    out, t = [], float(start_temp)
    for turn in range(max(int(max_turns or 0), 1) + 2):
        t = temp_fn(t, turn, max_turns)
        out.append(t)
    return np.asarray(out, dtype=np.float32)


AGENT_STREAM = 0x4000000000000000   # tape stream of agent-level draws (fast coin, seat shuffle); DESIGN.md
