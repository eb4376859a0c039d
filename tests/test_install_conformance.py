"""Row (b) of SURVEY.md section 8, pinned against drift: after `alphazero_general_amd.install()` the reference's own callers resolve
to this package's classes, and those classes keep the reference's call signatures.

Runs only where the reference is present (the build container; /root/reference does not exist on the GPU box): a subprocess
imports the real `alphazero` package (pyximport, tensorboardX stubbed -- the recipe of tests/golden/refharness.py) AFTER install()
and reports what `alphazero.Coach.SelfPlayAgent`, `alphazero.Arena.SelfPlayAgent` and `alphazero.GenericPlayers.MCTS` are bound to
(import sites Coach.py:5, Arena.pyx:4, GenericPlayers.py:1).  The signatures are read from the reference's SOURCE TEXT
(SelfPlayAgent.pyx:14-16, MCTS.pyx:133-344: `cpdef` methods of a `cdef class` carry no introspectable signature)."""
import inspect
import json
import os
import re
import subprocess
import sys

import pytest

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'alphazero')), reason='needs the reference checkout (build container only)')

C_TYPES = {'object', 'int', 'bint', 'float', 'double', 'Py_ssize_t', 'float[:]', 'int[:]', 'np.ndarray'}


def _ref_signatures(path, cls):
    """{method: [(name, default or None), ...]} of class `cls` from .pyx source: def / cpdef lines, C types and annotations dropped."""
    src = open(path).read()
    m = re.search(r'^(?:cdef )?class %s\b.*?:\n(.*?)(?=^\S|\Z)' % cls, src, re.S | re.M)
    assert m, cls
    out = {}
    for fm in re.finditer(r'^    (?:def|cpdef)\s+(?:[\w\.\[\]:]+\s+)?(\w+)\((.*?)\)\s*(?:->.*?)?:', m.group(1), re.S | re.M):
        params = []
        for raw in fm.group(2).replace('\n', ' ').split(','):
            raw = raw.strip()
            if not raw:
                continue
            default = None
            if '=' in raw:
                raw, default = [x.strip() for x in raw.split('=', 1)]
            raw = raw.split(':')[0].strip() if not any(raw.startswith(t + ' ') for t in ('float[:]', 'int[:]')) else raw
            toks = raw.split()
            params.append((toks[-1], default))
        out[fm.group(1)] = params
    return out


def _our_signature(fn):
    out = []
    for name, p in inspect.signature(fn).parameters.items():
        assert p.kind in (p.POSITIONAL_OR_KEYWORD,), (fn, name)
        out.append((name, None if p.default is p.empty else repr(p.default)))
    return out


def test_signatures_match_the_reference_source():
    from alphazero_general_amd.MCTS import MCTS
    from alphazero_general_amd.SelfPlayAgent import SelfPlayAgent
    ref_agent = _ref_signatures(os.path.join(REF, 'alphazero', 'SelfPlayAgent.pyx'), 'SelfPlayAgent')
    assert _our_signature(SelfPlayAgent.__init__) == ref_agent['__init__']                      # SelfPlayAgent.pyx:14-16
    for meth in ('run', 'generateBatch', 'processBatch', 'playMoves'):                          # :79,103,137,153
        assert _our_signature(getattr(SelfPlayAgent, meth)) == ref_agent[meth], meth
    ref_mcts = _ref_signatures(os.path.join(REF, 'alphazero', 'MCTS.pyx'), 'MCTS')
    public = ['__init__', 'reset', 'search', 'raw_search', 'update_root', 'find_leaf', 'process_results', 'counts', 'best_action',
              'probs', 'value']                                                                  # MCTS.pyx:133-344
    for meth in public:
        ours = _our_signature(getattr(MCTS, meth))
        ref = [(n, None if d is None else repr(eval(d))) for n, d in ref_mcts[meth]]
        assert ours == ref, (meth, ours, ref)
    for attr in ('_root', 'max_depth', 'depth'):                                                 # utils.py:57-83, SelfPlayAgent.pyx, GUI
        assert hasattr(MCTS, attr) or attr in MCTS(_mcts_args()).__dict__ or hasattr(MCTS(_mcts_args()), attr), attr


def _mcts_args():
    from alphazero_general_amd.utils import dotdict
    return dotdict(root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1, fpu_reduction=0.2, cpuct=1.25, _num_players=3, numMCTSSims=10)


_PROBE = r'''
import os, sys, types, json
sys.dont_write_bytecode = True
sys.path.insert(0, %(root)r); sys.path.insert(1, %(ref)r)
import numpy as np
tbx = types.ModuleType('tensorboardX')
class _W:
    def __init__(self, *a, **k): pass
    def __getattr__(self, n): return lambda *a, **k: None
tbx.SummaryWriter = _W
sys.modules.setdefault('tensorboardX', tbx)
import pyximport
os.makedirs('/tmp/pyxbld', exist_ok=True)
pyximport.install(setup_args={'include_dirs': np.get_include()}, build_dir='/tmp/pyxbld', language_level=3)
import alphazero_general_amd as azg
azg.install()                                   # BEFORE the reference's callers are imported (INTEGRATION.md)
import alphazero.Coach, alphazero.Arena, alphazero.GenericPlayers
from alphazero_general_amd.MCTS import MCTS
from alphazero_general_amd.SelfPlayAgent import SelfPlayAgent
print(json.dumps(dict(
    coach=alphazero.Coach.SelfPlayAgent is SelfPlayAgent,
    arena=alphazero.Arena.SelfPlayAgent is SelfPlayAgent,
    players=alphazero.GenericPlayers.MCTS is MCTS,
    coach_file=os.path.realpath(alphazero.Coach.__file__), arena_mod=alphazero.Arena.__name__)))
'''


def test_install_rebinds_the_reference_callers():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, '-c', _PROBE % dict(root=ROOT, ref=REF)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d['coach'] and d['arena'] and d['players'], d
    assert d['coach_file'].startswith(REF), d                   # it really is the reference's Coach that was rebound


_FALLTHROUGH = r"""
import os, sys, types, json, pickle
sys.dont_write_bytecode = True
sys.path.insert(0, %(root)r); sys.path.insert(1, %(ref)r)
import numpy as np
tbx = types.ModuleType('tensorboardX')
class _W:
    def __init__(self, *a, **k): pass
    def __getattr__(self, n): return lambda *a, **k: None
tbx.SummaryWriter = _W
sys.modules.setdefault('tensorboardX', tbx)
import pyximport
os.makedirs('/tmp/pyxbld', exist_ok=True)
pyximport.install(setup_args={'include_dirs': np.get_include()}, build_dir='/tmp/pyxbld', language_level=3)
import alphazero_general_amd as azg
azg.install()
import alphazero.Coach, alphazero.Arena, alphazero.GenericPlayers
from alphazero.MCTS import MCTS                       # this package's (dispatching) class
from alphazero.SelfPlayAgent import SelfPlayAgent
from alphazero.utils import dotdict
from alphazero.envs.tictactoe.tictactoe import Game as TicTacToe          # an env WITHOUT device rule kernels
from alphazero.envs.connect4.connect4 import Game as Connect4             # an env WITH them (recognised by module name)
import torch, torch.multiprocessing as mp
args = dotdict(root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1, fpu_reduction=0.2, cpuct=1.25, _num_players=3,
               numMCTSSims=10, gamesPerIteration=4, probFastSim=0, numFastSims=2, startTemp=1.0, add_root_noise=False,
               add_root_temp=False, temp_scaling_fn=alphazero.utils.default_temp_scaling)
out = {}
g = TicTacToe()
m = MCTS(args)
np.random.seed(5); m.raw_search(g, 40, False, False)
ref_cls = azg.reference_class('MCTS')
out['ref_module'] = ref_cls.__module__
out['delegated'] = type(m._ref) is ref_cls and m._engine is None
r = ref_cls(args)
np.random.seed(5); r.raw_search(g, 40, False, False)
out['same_counts'] = bool((np.asarray(m.counts(g)) == np.asarray(r.counts(g))).all()) and int(np.asarray(m.counts(g)).sum()) == 39
out['same_probs'] = bool((np.asarray(m.probs(g)) == np.asarray(r.probs(g))).all())
out['depth'] = [m.depth, m.max_depth, r.depth, r.max_depth]
a = m.best_action(g); m.update_root(g, a); g.play_action(a)
m2 = pickle.loads(pickle.dumps(m))                    # an MCTSPlayer crossing a process boundary
out['pickled'] = type(m2) is MCTS and bool((np.asarray(m2.counts(g)) == np.asarray(m.counts(g))).all())
out['still_ours'] = sys.modules['alphazero.MCTS'].MCTS is MCTS and alphazero.GenericPlayers.MCTS is MCTS \
    and getattr(sys.modules['alphazero'], 'MCTS', sys.modules['alphazero.MCTS']) is sys.modules['alphazero.MCTS']
# a whole player of the reference on the fall-through game
p = alphazero.GenericPlayers.RawMCTSPlayer(TicTacToe, args)
g2 = TicTacToe(); np.random.seed(1)
out['player_move_legal'] = bool(g2.valid_moves()[p.play(g2)])
def agent(game_cls):
    bt = torch.zeros((4,) + tuple(game_cls.observation_size()))
    return SelfPlayAgent(0, game_cls, mp.Queue(), mp.Event(), bt, torch.zeros(4, game_cls.action_size()), torch.zeros(4, 3),
                         mp.Queue(), mp.Queue(), mp.Value('i', 0), mp.Value('i', 0), mp.Event(), mp.Event(), args)
ref_agent = azg.reference_class('SelfPlayAgent')
out['agent_ttt_is_reference'] = type(agent(TicTacToe)) is ref_agent and ref_agent.__module__ == 'alphazero.SelfPlayAgent' and ref_agent is not SelfPlayAgent
out['agent_c4_is_ours'] = type(agent(Connect4)) is SelfPlayAgent
out['coach_still_ours'] = alphazero.Coach.SelfPlayAgent is SelfPlayAgent and sys.modules['alphazero.SelfPlayAgent'].SelfPlayAgent is SelfPlayAgent
print(json.dumps(out))
"""


def test_games_without_device_rules_fall_through_to_the_reference():
    """SURVEY.md 8b "Game plugin": after install() an env without device rule kernels (tictactoe) keeps working -- the MCTS /
    SelfPlayAgent classes registered under the reference's names hand it to the REFERENCE'S OWN classes (same trees as calling the
    reference directly under the same numpy seed), the names keep resolving to this package, an MCTS object on the fall-through
    path pickles (MCTS.pyx:8), and connect4 -- the reference's own Game class -- is still routed to the device engine."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, '-c', _FALLTHROUGH % dict(root=ROOT, ref=REF)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d['ref_module'] == 'alphazero.MCTS', d
    for k in ('delegated', 'same_counts', 'same_probs', 'pickled', 'still_ours', 'player_move_legal', 'agent_ttt_is_reference',
              'agent_c4_is_ours', 'coach_still_ours'):
        assert d[k], (k, d)
    assert d['depth'][:2] == d['depth'][2:], d


_REF_STATES = r"""
import os, sys, types, json
sys.dont_write_bytecode = True
sys.path.insert(0, %(root)r); sys.path.insert(1, %(ref)r)
import numpy as np
tbx = types.ModuleType('tensorboardX')
class _W:
    def __init__(self, *a, **k): pass
    def __getattr__(self, n): return lambda *a, **k: None
tbx.SummaryWriter = _W
sys.modules.setdefault('tensorboardX', tbx)
import pyximport
os.makedirs('/tmp/pyxbld', exist_ok=True)
pyximport.install(setup_args={'include_dirs': np.get_include()}, build_dir='/tmp/pyxbld', language_level=3)
from alphazero.envs.brandubh.fastafl import Game as RefBR
from alphazero.envs.connect4.connect4 import Game as RefC4
from alphazero_general_amd.MCTS import encode_state, decode_state
from alphazero_general_amd.Game import azg_game_id
from alphazero_general_amd.envs.brandubh import Game as OurBR
from alphazero_general_amd.envs.connect4 import Game as OurC4
rng = np.random.RandomState(3)
out = {'br': 0, 'c4': 0, 'ok': True}
for Ref, Our, key in ((RefBR, OurBR, 'br'), (RefC4, OurC4, 'c4')):
    assert azg_game_id(Ref()) == Our.AZG_GAME_ID
    for game in range(12):
        r, o = Ref(), Our()
        for ply in range(60):
            if np.asarray(r.win_state()).any():
                break
            er, eo = encode_state(r), encode_state(o)
            same = (np.asarray(er[0]) == np.asarray(eo[0])).all() and tuple(int(x) for x in er[1:]) == tuple(int(x) for x in eo[1:])
            d = decode_state(r, *er)                    # back into the REFERENCE's class
            same = same and type(d) is Ref and (np.asarray(d.valid_moves()) == np.asarray(r.valid_moves())).all() \
                and (np.asarray(d.win_state()) == np.asarray(r.win_state())).all() \
                and (np.asarray(d.observation()) == np.asarray(r.observation())).all() and d.player == r.player and d.turns == r.turns
            out['ok'] = out['ok'] and bool(same)
            out[key] += 1
            a = int(rng.choice(np.flatnonzero(np.asarray(r.valid_moves()))))
            r.play_action(a); o.play_action(a)
print(json.dumps(out))
"""


def test_reference_game_objects_encode_and_decode():
    """GenericPlayers hands `MCTS.search` / `find_leaf` the REFERENCE'S own Game objects (connect4.pyx, brandubh/fastafl.pyx) and expects
    leaves of the same class back (MCTS.pyx:208-228): their state must cross the ABI both ways -- same azg_state as this package's
    env after the same moves; decoded objects indistinguishable from the originals (valid moves, win state, observation)."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, '-c', _REF_STATES % dict(root=ROOT, ref=REF)], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d['ok'] and d['br'] > 200 and d['c4'] > 100, d
