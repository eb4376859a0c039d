"""Import the REFERENCE (kevaday/alphazero-general, /root/reference) in the build container and drive it under the
random tape.  Used ONLY by the golden-vector generators in this directory (make_goldens*.py); never imported by a
test, by smoke() or by bench.py -- /root/reference does not exist on the GPU box.

Recipe: pyximport-compile the reference's .pyx files into /tmp/pyxbld (nothing is written under
/root/reference), stub the missing tensorboardX dependency, then monkeypatch np.random.{shuffle,choice,dirichlet,
random_sample} with the counter-based tape (DESIGN.md "Random tape") so that "identical seeds" is well defined.
"""
import os
import sys
import types
import zlib
from collections import defaultdict

import numpy as np

sys.dont_write_bytecode = True
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))      # tests/
import oracle_lib as ol  # noqa: E402

AGENT_STREAM = 0x4000000000000000


def import_reference():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    tbx = types.ModuleType('tensorboardX')

    class _W:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, n):
            return lambda *a, **k: None
    tbx.SummaryWriter = _W
    sys.modules.setdefault('tensorboardX', tbx)
    import pyximport
    os.makedirs('/tmp/pyxbld', exist_ok=True)
    pyximport.install(setup_args={'include_dirs': np.get_include()}, build_dir='/tmp/pyxbld', language_level=3)
    import alphazero.MCTS  # noqa: F401
    import alphazero.SelfPlayAgent  # noqa: F401


class Tape:
    """Counter-based replacement for the global legacy np.random stream (per-stream counters)."""

    def __init__(self, seed):
        self.seed = int(seed)
        self.ctr = defaultdict(int)
        self.stream = 0
        self.choice_log = []

    def install(self):
        self._saved = (np.random.shuffle, np.random.choice, np.random.dirichlet, np.random.random_sample)
        np.random.shuffle, np.random.choice = self.shuffle, self.choice
        np.random.dirichlet, np.random.random_sample = self.dirichlet, self.random_sample

    def uninstall(self):
        np.random.shuffle, np.random.choice, np.random.dirichlet, np.random.random_sample = self._saved

    def shuffle(self, x):
        k = len(x)
        pos = np.zeros(max(k, 1), np.int32)
        ol.lib().azo_tape_shuffle_pos(self.seed, self.stream, self.ctr[self.stream], k, pos)
        self.ctr[self.stream] += k
        old = list(x)
        for i, e in enumerate(old):
            x[pos[i]] = e

    def choice(self, a, p=None):
        n = a if isinstance(a, (int, np.integer)) else len(a)
        p32 = np.ascontiguousarray(p, np.float32)
        assert p32.shape == (n,)
        idx = ol.lib().azo_tape_choice(self.seed, self.stream, self.ctr[self.stream], p32, n)
        self.ctr[self.stream] += 1
        self.choice_log.append((self.stream, idx))
        return idx

    def dirichlet(self, alpha):
        k = len(alpha)
        out = np.zeros(k, np.float64)
        ol.lib().azo_tape_dirichlet(self.seed, self.stream, self.ctr[self.stream], k, float(alpha[0]), out)
        self.ctr[self.stream] += 1
        return out

    def random_sample(self):
        u = ol.lib().azo_tape_uniform(self.seed, self.stream, self.ctr[self.stream])
        self.ctr[self.stream] += 1
        return u


class ObservedRng:
    """The MT19937 tier: numpy's global legacy stream UNTOUCHED -- np.random.{shuffle, dirichlet, choice, random_sample} are observed (the real
    function is called and draws from the real stream), not replaced; what each call produced is recorded with the game slot it was made
    for (`stream`, set by TapedAgent._mcts) in global call order: ('shuffle', slot, ranks), ('dirichlet', slot, float64 vector),
    ('choice', slot, u, index), ('coin', -1, u).  For np.random.choice the uniform it drew is recovered from a clone of the stream's state
    (the legacy choice draws exactly ONE random_sample and searches the cdf), and the clone is checked to end where the real stream does."""

    def __init__(self):
        self.calls = []
        self.stream = -1

    def install(self):
        self._saved = (np.random.shuffle, np.random.choice, np.random.dirichlet, np.random.random_sample)
        np.random.shuffle, np.random.choice = self.shuffle, self.choice
        np.random.dirichlet, np.random.random_sample = self.dirichlet, self.random_sample

    def uninstall(self):
        np.random.shuffle, np.random.choice, np.random.dirichlet, np.random.random_sample = self._saved

    def shuffle(self, lst):
        before = list(lst)
        self._saved[0](lst)
        pos = [-1] * len(before)
        for new_i, obj in enumerate(lst):
            for old_i, b in enumerate(before):
                if b is obj:
                    pos[old_i] = new_i
        assert sorted(pos) == list(range(len(before)))
        self.calls.append(('shuffle', self.stream, pos))

    def dirichlet(self, alpha):
        out = self._saved[2](alpha)
        self.calls.append(('dirichlet', self.stream, np.array(out, np.float64)))
        return out

    def choice(self, a, p=None):
        st = np.random.get_state()
        idx = self._saved[1](a, p=p)
        rs = np.random.RandomState(); rs.set_state(st)
        u = rs.random_sample()
        after, mine = np.random.get_state(), rs.get_state()
        assert after[2] == mine[2] and (after[1] == mine[1]).all(), 'np.random.choice drew something else than one random_sample'
        cdf = np.asarray(p, np.float64).cumsum(); cdf /= cdf[-1]
        assert int(cdf.searchsorted(u, side='right')) == int(idx)
        self.calls.append(('choice', self.stream, float(u), int(idx)))
        return idx

    def random_sample(self):
        u = self._saved[3]()
        self.calls.append(('coin', -1, float(u)))
        return u


class _Ev:
    def __init__(self):
        self._s = False

    def is_set(self):
        return self._s

    def set(self):
        self._s = True

    def clear(self):
        self._s = False

    def wait(self, timeout=None):
        return True


class _Q:
    def __init__(self):
        self.items = []

    def put(self, x):
        self.items.append(x)

    def qsize(self):
        return len(self.items)

    def close(self):
        pass

    def join_thread(self):
        pass


def ref_args(game_cls, **kw):
    from alphazero.utils import dotdict, default_temp_scaling
    a = dotdict(root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1, fpu_reduction=0.2, cpuct=1.25,
                _num_players=game_cls.num_players() + game_cls.has_draw(), startTemp=1,
                temp_scaling_fn=default_temp_scaling, numMCTSSims=25, numFastSims=20, numWarmupSims=5, probFastSim=0.0,
                gamesPerIteration=32, add_root_noise=False, add_root_temp=False, arenaTemp=0.25,
                mctsResetThreshold=None, symmetricSamples=True)
    a.update(kw)
    return a


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def make_ref_agent(game_cls, game_id, B, args, tape, is_arena=False, is_warmup=False, slot_base=0):
    """Construct the reference SelfPlayAgent in-process with stub queues/events; _mcts() sets the tape stream."""
    import torch
    import torch.multiprocessing as mp
    from alphazero.SelfPlayAgent import SelfPlayAgent

    class TapedAgent(SelfPlayAgent):
        def _mcts(self, index):
            tape.stream = slot_base + index
            return super()._mcts(index)

    gi = ol.game_info(game_id)
    batch = torch.zeros([B, gi.obs_c, gi.obs_h, gi.obs_w]) if not is_arena else [[] for _ in range(gi.num_players)]
    pol = torch.zeros([B, gi.action_size])
    val = torch.zeros([B, gi.num_players + 1])
    tape.stream = AGENT_STREAM + slot_base
    ag = TapedAgent(0, game_cls, _Q(), _Ev(), batch, pol, val, _Q(), _Q(), mp.Value('i', 0), mp.Value('i', 0),
                    _Ev(), _Ev(), args, _is_arena=is_arena, _is_warmup=is_warmup)
    return ag
