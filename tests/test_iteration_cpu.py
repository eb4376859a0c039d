"""The iteration component's host logic on CPU (alphazero_general_amd.iteration / coach): the exchange step + rank 0's sample files over
a world-2 gloo group on real self-play shards (produced by the CPU oracle -- the engine itself needs a GPU), the weight broadcast a
Coach on rank 0 uses to hand its live net to the other ranks, NNetWrapper.adopt, the reference's winrate rule; and, where the
reference checkout is present (build container), the adapter against the REAL alphazero.Coach: the five method names / signatures
and learn()'s call order (Coach.py:225-288,291,326,364,389,401) and the files consumed by the real Coach.train loader (:442-456)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference'


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _play(B, slot_base, seed, games):
    import oracle_lib as ol
    ag = ol.OAgent(0, B, sims=8, games_per_iteration=games, seed=seed, slot_base=slot_base)
    step = 0
    while ag.games_played < games:
        for _ in range(ag.begin_round()):
            ag.generate_batch()
            pol = np.zeros((B, 7), np.float32); val = np.zeros((B, 3), np.float32)
            for i in range(B):
                pol[i], val[i] = ol.fake_eval(seed, slot_base + i, step, 7, 3)
            ag.process_batch(pol, val); step += 1
        ag.play_moves()
    return ag


def _tallies(ag):
    ws, turns = ag.results()[:2]
    ws = np.asarray(ws)
    return [int(ws[:, 0].sum()), int(ws[:, 1].sum()), int(ws[:, 2].sum()), int(np.sum(turns)), len(turns)]


def _worker(rank, world, port, folder, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from alphazero_general_amd import distributed as D
    from alphazero_general_amd import iteration as I
    D.init_from_env(backend='gloo')
    B, total = 5, 9
    ag = _play(B, D.slot_base(rank, B), 5, D.shard_games(total, rank, world))
    obs, pi, z = [torch.from_numpy(x) for x in ag.samples()]
    (gobs, gpi, gz), t = I.exchange_selfplay(obs, pi, z, _tallies(ag))
    if rank == 0:
        I.write_iteration_files(folder, 7, gobs, gpi, gz)
    # the weights of the net the iteration plays with: rank 0's live module -> every replica, one flat broadcast
    from alphazero_general_amd.nnet import NNetWrapper, DEFAULT_NET_ARGS
    from alphazero_general_amd.envs.connect4 import Game
    torch.manual_seed(100 + rank)                                    # (different weights per rank before the broadcast)
    net = NNetWrapper(Game, dict(DEFAULT_NET_ARGS, num_channels=16, depth=2), device='cpu')
    net.nnet.bn1.num_batches_tracked.fill_(41 + rank)           # an integer entry: travels in the meta record
    sd = net.nnet.state_dict()
    meta = D.broadcast_object(D.state_dict_meta(sd) if rank == 0 else None)
    got = D.broadcast_state_dict(sd if rank == 0 else None, meta)
    D.barrier()
    q.put((rank, obs.numpy(), pi.numpy(), z.numpy(), t.numpy(), {k: v.numpy() for k, v in got.items()}, {k: v.numpy() for k, v in sd.items()}))
    dist.destroy_process_group()


def test_exchange_and_sample_files_world2(tmp_path):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    [p.start() for p in ps]
    outs = sorted([q.get(timeout=240) for _ in range(2)], key=lambda o: o[0])
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    d, p, v = [torch.load(str(tmp_path / ('iteration-0007-%s.pkl' % k)), weights_only=False) for k in ('data', 'policy', 'value')]
    # rank 0's files = the ranks' shards in rank order, each in its own output order (Coach.saveIterationSamples layout, Coach.py:377-383)
    assert (d.numpy() == np.concatenate([outs[0][1], outs[1][1]])).all() and d.dtype == torch.float32 and d.shape[1:] == (4, 6, 7)
    assert (p.numpy() == np.concatenate([outs[0][2], outs[1][2]])).all() and (v.numpy() == np.concatenate([outs[0][3], outs[1][3]])).all()
    assert (outs[0][4] == outs[1][4]).all() and outs[0][4][4] >= 9 and outs[0][4][:3].sum() == outs[0][4][4]   # tallies summed, one winstate per game
    # the broadcast: rank 1 now holds rank 0's weights bit for bit, integer entries included
    for k, w0 in outs[0][6].items():
        assert (outs[1][5][k] == w0).all() and outs[1][5][k].dtype == w0.dtype and outs[1][5][k].shape == w0.shape, k
    assert int(outs[1][5]['bn1.num_batches_tracked']) == 41 and any((outs[1][6][k] != outs[0][6][k]).any() for k in outs[0][6])


def _lead_serve_worker(rank, world, port, q):
    """the command protocol of a Coach-driven job without a GPU: run_iteration / run_arena are replaced by recorders (the engines need a
    device; the real thing runs on the GPU rig, tests/test_gpu_iteration.py) -- what is checked here is what TRAVELS: the picklable slice
    of the args, the temperature table of a schedule that does not pickle, the weights (one broadcast per distinct net, shared nets sent
    once), warm-up iterations without a net, and 'stop'."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from alphazero_general_amd import distributed as D
    from alphazero_general_amd import iteration as I
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import NNetWrapper, DEFAULT_NET_ARGS
    from alphazero_general_amd.utils import dotdict
    D.init_from_env(backend='gloo')
    seen = []

    def digest(n):
        return None if n is None else float(sum(v.double().sum() for v in n.nnet.state_dict().values()))

    def fake_iter(game_cls, nnet, args, iteration, folder=None, **kw):
        seen.append(('selfplay', iteration, digest(nnet), None if nnet is None else int(nnet.args.num_channels), {k: args.get(k) for k in ('numMCTSSims', 'gamesPerIteration', 'cpuct')},
                     list(args.get('_azg_temp_table', []))[:3] if '_azg_temp_table' in args else 'fn', sorted(kw)))
        return {'games': 0}

    def fake_arena(game_cls, nnets, args, num_games, **kw):
        seen.append(('arena', num_games, [digest(n) for n in nnets], [nnets[1] is n for n in nnets], sorted(kw)))
        return ([0, 0], 0, [0, 0])
    I.run_iteration, I.run_arena = fake_iter, fake_arena
    served = None
    if rank == 0:
        torch.manual_seed(5)
        new = NNetWrapper(Game, dict(DEFAULT_NET_ARGS, num_channels=16, depth=2), device='cpu')
        past = NNetWrapper(Game, dict(DEFAULT_NET_ARGS, num_channels=16, depth=2), device='cpu')
        args = dotdict(numMCTSSims=33, gamesPerIteration=12, cpuct=2.5, temp_scaling_fn=lambda t, *_: t * 0.5, startTemp=1.0, baselineTester=object, workers=2)
        I.lead('selfplay', Game, [new], args, iteration=4, folder=None, num_slots=8)
        I.lead('arena', Game, [new, past, past], args, num_games=10, seats='slot')
        I.lead('selfplay', Game, [None], args, iteration=1, warmup=True)
        I.lead('stop', Game, [], args)
        mine = [digest(new), digest(past)]
    else:
        torch.manual_seed(99)
        served = I.serve(Game, device='cpu')
        mine = None
    D.barrier()
    q.put((rank, seen, served, mine))
    dist.destroy_process_group()


def test_lead_and_serve_protocol_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_lead_serve_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    outs = sorted([q.get(timeout=240) for _ in range(2)], key=lambda o: o[0])
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (_, lead_seen, _, (d_new, d_past)), (_, serve_seen, served, _) = outs
    assert served == 3 and len(serve_seen) == len(lead_seen) == 3
    sp, ar, wu = serve_seen
    # self-play: rank 0's weights bit for bit, the architecture, the args that matter; the schedule arrived as its table
    assert sp[:4] == ('selfplay', 4, d_new, 16) and sp[4] == {'numMCTSSims': 33, 'gamesPerIteration': 12, 'cpuct': 2.5} and sp[5] == [0.5, 0.25, 0.125]
    assert 'num_slots' in sp[6] and 'keep_samples' in sp[6]
    # arena [new, past, past]: two distinct nets travelled, the shared one is ONE object on the serving rank too
    assert ar[0] == 'arena' and ar[1] == 10 and ar[2] == [d_new, d_past, d_past] and d_new != d_past and ar[3] == [False, True, True] and 'seats' in ar[4]
    # warm-up: no net, no weights
    assert wu[:3] == ('selfplay', 1, None) and 'warmup' in wu[6]
    assert lead_seen[0][2] == d_new and lead_seen[0][5] == 'fn'       # rank 0 plays its own shard with its own objects and its own callable


def test_adopt_takes_live_weights_and_rebuilds_the_architecture():
    """NNetWrapper.adopt: a live module / wrapper / state_dict, no checkpoint file; a different architecture is rebuilt from the
    source's args like load_checkpoint(use_saved_args=True) does (NNetWrapper.py:252-276)."""
    from alphazero_general_amd.nnet import NNetWrapper, DEFAULT_NET_ARGS
    from alphazero_general_amd.envs.connect4 import Game
    torch.manual_seed(3)
    src = NNetWrapper(Game, dict(DEFAULT_NET_ARGS, num_channels=16, depth=3), device='cpu', fast=False)
    with torch.no_grad():
        for prm in src.nnet.parameters():
            prm.add_(0.01 * torch.randn_like(prm))
    dst = NNetWrapper(Game, None, device='cpu', fast=False)                     # default architecture (32 x 4): must be rebuilt
    x = torch.randn(5, 4, 6, 7)
    for source in (src, src.nnet, src.nnet.state_dict()):
        dst2 = NNetWrapper(Game, None, device='cpu', fast=False)
        dst2.adopt(source, None if source is src else src.args)
        assert dst2.args.num_channels == 16 and dst2.args.depth == 3
        for a, b in zip(dst2.process(x), src.process(x)):
            assert torch.equal(a, b)
    dst.adopt(src)
    with torch.no_grad():
        next(src.nnet.parameters()).mul_(2.0)                                  # the source trains on: the adopted copy is a copy
    assert not torch.equal(dst.process(x)[0], src.process(x)[0])
    with pytest.raises(RuntimeError):
        NNetWrapper(Game, None, device='cpu').adopt({'bogus': torch.zeros(1)}, DEFAULT_NET_ARGS)


def test_winrate_rule_and_seeds():
    from alphazero_general_amd import iteration as I
    # Arena.__update_winrates (Arena.pyx:124-131): draws count half and enter the denominator only with use_draws_for_winrate
    assert I.winrates([6, 2], 2, True) == [0.7, 0.3] and I.winrates([6, 2], 2, False) == [0.75, 0.25] and I.winrates([0, 0], 0, True) == [0, 0]
    assert len({I.iteration_seed(0, i) for i in range(100)}) == 100 and I.iteration_seed(3, 5) == I.iteration_seed(3, 5) < 2 ** 63
    from alphazero_general_amd.utils import dotdict
    assert I.default_slots(dotdict(workers=4, process_batch_size=256), 1) == 1024 and I.default_slots(dotdict(workers=4, process_batch_size=256), 8) == 128
    assert I.default_slots(dotdict(_azg_slots=2048), 8) == 256
    # only picklable keys travel to the serving ranks; the temperature schedule travels as its table
    import pickle
    from alphazero_general_amd.envs.connect4 import Game
    pa = I.portable_args(dotdict(cpuct=4.0, numMCTSSims=100, temp_scaling_fn=lambda t, *_: t * 0.5, baselineTester=object, startTemp=1.0), Game)
    assert 'baselineTester' not in pa and 'temp_scaling_fn' not in pa and pa['_azg_temp_table'][:3] == [0.5, 0.25, 0.125]
    pickle.loads(pickle.dumps(pa))


# ------------------------------------------------------------------------------------------------ against the real reference (build container)
_PROBE = r'''
import ast, inspect, json, os, sys, types
sys.dont_write_bytecode = True
sys.path.insert(0, %(root)r); sys.path.insert(1, %(root)r + '/tests'); sys.path.insert(2, %(ref)r)
import numpy as np, torch
tbx = types.ModuleType('tensorboardX')
class _W:
    def __init__(self, *a, **k): self.scalars = []
    def add_scalar(self, *a, **k): self.scalars.append(a)
    def __getattr__(self, n): return lambda *a, **k: None
tbx.SummaryWriter = _W
sys.modules.setdefault('tensorboardX', tbx)
import pyximport
os.makedirs('/tmp/pyxbld', exist_ok=True)
pyximport.install(setup_args={'include_dirs': np.get_include()}, build_dir='/tmp/pyxbld', language_level=3)
import alphazero_general_amd as azg
azg.install()
import alphazero.Coach as CM
from alphazero.utils import dotdict
from alphazero_general_amd.coach import native_coach, native_arena
from alphazero_general_amd import iteration as I
RefArena = CM.Arena
Native = native_coach(CM.Coach)
out = {'is_subclass': issubclass(Native, CM.Coach), 'arena_rebound': CM.Arena is not RefArena and issubclass(CM.Arena, RefArena),
       'arena_idempotent': native_arena(CM.Arena) is CM.Arena}
# the five methods: defined by the adapter itself, same parameter lists as the reference's source (Coach.py:291,326,364,389,401)
tree = ast.parse(open(CM.__file__).read())
cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'Coach')
ref_sig = {f.name: ([a.arg for a in f.args.args], f.lineno) for f in cls.body if isinstance(f, ast.FunctionDef)}
five = ['generateSelfPlayAgents', 'processSelfPlayBatches', 'saveIterationSamples', 'processGameResults', 'killSelfPlayAgents']
out['five'] = {m: dict(own=m in Native.__dict__, ours=list(inspect.signature(Native.__dict__[m]).parameters), ref=ref_sig[m][0], line=ref_sig[m][1]) for m in five}
out['overrides'] = sorted(k for k in Native.__dict__ if not k.startswith('_'))
# learn()'s self-play phase calls exactly these five, in this order (Coach.py:252-267)
learn = next(f for f in cls.body if isinstance(f, ast.FunctionDef) and f.name == 'learn')
calls = [n.func.attr for n in sorted((n for n in ast.walk(learn) if isinstance(n, ast.Call)), key=lambda n: (n.lineno, n.col_offset))
         if isinstance(n.func, ast.Attribute) and isinstance(n.func.value, ast.Name) and n.func.value.id == 'self']
out['learn_calls'] = [c for c in sorted(set(calls), key=calls.index) if c in five]
# files written by the library are consumed by the REAL Coach.train loader (Coach.py:438-524), CPU
import oracle_lib as ol
ag = ol.OAgent(0, 8, sims=6, games_per_iteration=6, seed=2)
step = 0
while ag.games_played < 6:
    for _ in range(ag.begin_round()):
        ag.generate_batch()
        pol = np.zeros((8, 7), np.float32); val = np.zeros((8, 3), np.float32)
        for i in range(8):
            pol[i], val[i] = ol.fake_eval(2, i, step, 7, 3)
        ag.process_batch(pol, val); step += 1
    ag.play_moves()
obs, pi, z = [torch.from_numpy(x) for x in ag.samples()]
folder = %(tmp)r
I.write_iteration_files(os.path.join(folder, 'data', 'run'), 1, obs, pi, z)
class StubNet:
    seen = []
    def train(self, dataloader, steps):
        for d, p, v in dataloader:
            StubNet.seen.append((tuple(d.shape[1:]), tuple(p.shape[1:]), tuple(v.shape[1:]), int(d.shape[0]), str(d.dtype)))
        StubNet.steps = steps
        return 0.5, 0.25
    def save_checkpoint(self, folder, filename): StubNet.saved = (folder, filename)
coach = object.__new__(Native)
coach.args = dotdict(data=os.path.join(folder, 'data'), run_name='run', checkpoint=os.path.join(folder, 'ckpt'), workers=0, train_batch_size=16,
                     averageTrainSteps=False, autoTrainSteps=True, train_steps_per_iteration=1, train_on_past_data=False, startIter=1,
                     minTrainHistoryWindow=4, trainHistoryIncrementIters=2, maxTrainHistoryWindow=20)
coach.train_net, coach.writer = StubNet(), _W()
coach.train(1)
out['train'] = dict(samples=int(obs.shape[0]), seen=sum(s[3] for s in StubNet.seen), shapes=sorted(set(s[:3] + (s[4],) for s in StubNet.seen)),
                    steps=StubNet.steps, losses=[coach.loss_pi, coach.loss_v], saved=StubNet.saved[1])
# processGameResults / saveIterationSamples on a record as run_iteration returns it
coach._azg_result = dict(wins=[3, 2], draws=1, num_results=6, avg_game_length=20.5, num_samples=int(obs.shape[0]), games=6)
coach.args.use_draws_for_winrate = True
coach.writer = _W()
coach.processGameResults(1); coach.saveIterationSamples(1)
out['scalars'] = [[a[0], round(float(a[1]), 4), a[2]] for a in coach.writer.scalars]
print(json.dumps(out))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'alphazero')), reason='needs the reference checkout (build container only)')
def test_adapter_against_the_real_coach(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, '-c', _PROBE % dict(root=ROOT, ref=REF, tmp=str(tmp_path))], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d['is_subclass'] and d['arena_rebound'] and d['arena_idempotent'], d
    lines = {'generateSelfPlayAgents': 291, 'processSelfPlayBatches': 326, 'saveIterationSamples': 364, 'processGameResults': 389, 'killSelfPlayAgents': 401}
    for m, rec in d['five'].items():
        assert rec['own'] and rec['ours'] == rec['ref'], (m, rec)
        assert abs(rec['line'] - lines[m]) <= 1, (m, rec)                      # (decorator line vs def line)
    assert d['overrides'] == sorted(list(lines) + ['learn']), d['overrides']    # exactly the five (+ learn's `finally: stop the serving ranks`)
    assert d['learn_calls'] == list(lines), d['learn_calls']
    t = d['train']
    assert t['seen'] == t['samples'] > 0 and t['shapes'] == [[[4, 6, 7], [7], [3], 'torch.float32']] and t['steps'] == t['samples'] // 16
    assert t['losses'] == [0.5, 0.25] and t['saved'] == 'iteration-0001.pkl'
    assert d['scalars'] == [['win_rate/player0', round(3.5 / 6, 4), 1], ['win_rate/player1', round(2.5 / 6, 4), 1], ['win_rate/draws', round(1 / 6, 4), 1],
                            ['win_rate/avg_game_length', 20.5, 1]]
