"""The native drivers over the C ABI (selfplay.SelfPlayRunner = Coach.processSelfPlayBatches' loop, Coach.py:291-361;
selfplay.ArenaRunner = the batched branch of Arena.play_games, Arena.pyx:208-328) with a real network in the loop:
launch strategy (hipGraph replay, device-side batch split, stream pipelines) must not change a single move."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _args(**kw):
    from alphazero_general_amd.utils import dotdict, default_temp_scaling
    a = dotdict(numMCTSSims=12, numFastSims=4, probFastSim=0.0, gamesPerIteration=1 << 30, cpuct=4.0, fpu_reduction=0.4,
                root_noise_frac=0.3, root_policy_temp=1.3, min_discount=1.0, add_root_noise=True, add_root_temp=True,
                symmetricSamples=True, mctsResetThreshold=0, startTemp=1.0, arenaTemp=0.25, temp_scaling_fn=default_temp_scaling)
    a.update(kw)
    return a


def _net(seed):
    import torch
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper
    torch.manual_seed(seed)
    return NNetWrapper(Game, CONNECT4_NET_ARGS, device='cuda:0', dtype=torch.float16)


def test_arena_runner_device_split_graph_equals_host_split():
    """ArenaRunner: every model evaluating its slice through a device-side row range inside one captured graph plays
    exactly the games the host-split loop (one .cpu() read of the split per simulation) plays."""
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.selfplay import ArenaRunner
    nets = [_net(0), _net(1)]
    runs = []
    for mode in ('graph', 'eager', 'phase_graph', 'phase_eager', 'host'):    # (graph / eager: the persistent launch, one per move)
        r = ArenaRunner(Game, nets, _args(), num_slots=96, seed=5, use_graph=mode.endswith('graph'), fused_search=not mode.startswith('phase') and mode != 'host')
        assert r.device_split and (r._graph is not None) == mode.endswith('graph') and r.fused_search == (mode in ('graph', 'eager'))
        if mode == 'host':
            r.device_split = False
        acts = []
        for _ in range(14):
            r.play_round()
            acts.append(r.engine.last_actions().cpu().numpy().copy())
        runs.append((np.array(acts), r.engine.counters(), r.results()))
    a0, c0, res0 = runs[0]
    assert c0['games_played'] > 0 and (a0 >= 0).any()
    for a, c, res in runs[1:]:
        assert (a == a0).all()
        assert c['games_played'] == c0['games_played'] and c['sims'] == c0['sims'] and c['expansions'] == c0['expansions']
        assert res == res0
    wins, draws, rates = res0
    assert sum(wins) + draws == c0['games_played']                   # Arena.play_games contract (Arena.pyx:376)


@pytest.mark.parametrize('variant', ['no_graph', 'pipelines2', 'no_graph_fast_rounds_resets', 'no_graph_vs_fused_search'])
def test_selfplay_runner_launch_strategy_is_invisible(variant):
    """SelfPlayRunner: hipGraph replay of the network / two stream pipelines give bit-identical samples and results to
    the plain launch sequence (slots are sharded by global id, so the pipelines play the same games)."""
    import torch
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.selfplay import SelfPlayRunner
    net = _net(3)
    outs = []
    # fast rounds (a second captured round graph with numFastSims, no history) and periodic tree resets ride along in one variant
    extra = dict(probFastSim=0.4, numFastSims=5, mctsResetThreshold=3) if variant.endswith('fast_rounds_resets') else {}
    first = dict(fused_search=True) if variant.endswith('fused_search') else dict(fused_search=False) if variant == 'no_graph' else dict()
    for kw in (first, dict(use_graph=False) if variant.startswith('no_graph') else dict(pipelines=2)):
        r = SelfPlayRunner(Game, net, _args(**extra), num_slots=64, seed=9, example_capacity=64 * 43 * 2 * 4, **kw)
        assert r.fused_search == bool(kw.get('fused_search', kw.get('use_graph', True)))     # on by default when graphs are
        for _ in range(30):
            r.play_round()
        obs, pi, z = r.samples()
        ws, turns, slot = r.results()
        outs.append((obs.cpu().numpy(), pi.cpu().numpy(), z.cpu().numpy(), np.asarray(ws), np.asarray(turns), np.asarray(slot), r.counters()))
    a, b = outs
    assert a[6]['games_played'] == b[6]['games_played'] > 0
    if extra:
        assert len(set(r.sims_per_round)) == 2                    # both kinds of round happened
    if variant.startswith('no_graph'):
        for x, y in zip(a[:6], b[:6]):
            assert x.shape == y.shape and (x == y).all()
    else:                                                            # lanes emit in their own order: compare as multisets
        key = lambda o, p, zz: sorted(map(bytes, np.concatenate([o.reshape(len(o), -1), p, zz], axis=1)))
        assert key(a[0], a[1], a[2]) == key(b[0], b[1], b[2])
        assert sorted(zip(a[5].tolist(), a[4].tolist())) == sorted(zip(b[5].tolist(), b[4].tolist()))


@pytest.mark.parametrize('B,sims,moves', [(203, 17, 30), (1, 2, 44), (2, 3, 44), (5, 2, 12), (7, 4, 6),
                                              (641, 6, 9), (1281, 5, 9)])       # (two and four games per workgroup; a last tile with one game)
def test_fused_search_kernel_equals_three_kernel_path(B, sims, moves):
    """azg_search_f16 (one persistent launch: tree walk, MFMA tower, backup for `sims` simulations) against the same number
    of [azg_select, azg_resnet_policy_value_f16, azg_backup] rounds on a twin engine: identical trees, moves, samples.
    203: the last workgroup owns 3 games, not 4; the tiny cases: fewer games than a workgroup holds, two simulations per move
    (the fewest that leave a visit count to sample from), games that end and restart inside the run."""
    import torch
    from alphazero_general_amd.engine import DeviceEngine
    net = _net(4); net.refresh()
    hip = net._hip
    kw = dict(cpuct=4.0, fpu_reduction=0.4, add_root_noise=True, add_root_temp=True, seed=12, games_per_iteration=1 << 30,
              example_capacity=B * 43 * 2 * 3, sims_hint=sims)
    ea, eb = DeviceEngine(0, B, **kw), DeviceEngine(0, B, **kw)
    obs = torch.zeros((B, 42, 8), dtype=torch.float16, device=ea.device)
    for move in range(moves):
        hip.search(ea, sims)
        for _ in range(sims):
            eb.select(obs)
            p, v = hip.forward_nhwc8(obs)
            eb.backup(p, v)
        ca, cb = ea.root_counts(), eb.root_counts()
        assert torch.equal(ca, cb), move
        assert torch.equal(ea.root_probs(1.0), eb.root_probs(1.0))
        ea.advance(True); eb.advance(True)
        assert torch.equal(ea.last_actions(), eb.last_actions())
    a, b = ea.counters(), eb.counters()
    assert a == b and (a['games_played'] > 0 or moves < 20)
    for x, y in zip(ea.examples(), eb.examples()):
        assert torch.equal(x, y)


def test_warmup_runner_round_graph_equals_eager():
    """warm-up mode (SelfPlayAgent.pyx:48-52,111-114: uniform policy / value, numWarmupSims) through the captured round graph
    and through plain launches: identical games."""
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.selfplay import SelfPlayRunner
    outs = []
    for use_graph in (True, False):
        r = SelfPlayRunner(Game, None, _args(numWarmupSims=7), num_slots=48, seed=4, warmup=True, use_graph=use_graph,
                           example_capacity=48 * 43 * 2 * 4)
        assert r.round_graph == use_graph
        for _ in range(25):
            r.play_round()
        o, p, z = r.samples()
        outs.append((o.cpu().numpy(), p.cpu().numpy(), z.cpu().numpy(), r.counters()))
    assert outs[0][3] == outs[1][3] and outs[0][3]['games_played'] > 0 and outs[0][3]['sims'] == 25 * 7 * 48
    for x, y in zip(outs[0][:3], outs[1][:3]):
        assert x.shape == y.shape and (x == y).all()


@pytest.mark.parametrize('launcher', ['self', 'torchrun'])
def test_bench_two_ranks_on_one_gpu(launcher):
    """bench.py's multi-rank path end to end -- rendezvous, slot sharding by rank, timed loop, all-gather of the example shards,
    tally all-reduce, max-over-ranks -- with two ranks sharing GPU 0 over gloo (RCCL refuses two ranks on one device; the
    8-GPU run itself is the driver's).  launcher = 'self': plain `python bench.py --gpus 2` must start its own two ranks and
    report n_gpus = 2 from two live ranks; 'torchrun': the driver's launch line."""
    import json
    import os
    import signal
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AZG_DIST_BACKEND='gloo', AZG_SINGLE_DEVICE='1')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    tail = [os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '1', '--slots', '256']
    cmd = [sys.executable] + tail if launcher == 'self' else \
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', '29613'] + tail
    p = subprocess.Popen(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        raise
    lines = [l for l in out.decode(errors='replace').splitlines() if l.startswith('{"metric"')]
    assert p.returncode == 0 and len(lines) == 1, out.decode(errors='replace')[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['ranks'] == 2 and d['config']['backend'] == 'gloo'
    assert d['scaling'] == 'weak' and d['value'] > 0 and d['games_finished'] > 0
    assert d['samples_gathered'] >= d['games_finished'] * 7 * 2          # both ranks' shards arrived
    assert 'cpu_baseline' not in d                                       # rank 0, N = 1 only


@pytest.mark.parametrize('workload,slots', [('brandubh', 64), ('arena', 64), ('trimok', 64)])
def test_bench_two_ranks_other_workloads(workload, slots):
    """the N-rank path of the OTHER bench workloads (BASELINE configs 3-5) before their first contact with a multi-GPU node: two
    ranks on GPU 0 over gloo, self-launched; brandubh / trimok (timed on the exact persistent launch) also run their opt-in sparse-heads
    leg on both ranks (its own exchange step), the arena has no example exchange at all."""
    import json
    import os
    import signal
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AZG_DIST_BACKEND='gloo', AZG_SINGLE_DEVICE='1')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--workload', workload, '--steps', '5', '--warmup', '1', '--slots', str(slots),
           '--profile-rounds', '1']
    p = subprocess.Popen(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        raise
    lines = [l for l in out.decode(errors='replace').splitlines() if l.startswith('{"metric"')]
    assert p.returncode == 0 and len(lines) == 1, out.decode(errors='replace')[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['ranks'] == 2 and d['config']['backend'] == 'gloo' and d['value'] > 0
    assert d['config']['games_per_gpu'] == slots and d['roofline'] is not None
    sims = {'brandubh': 200, 'arena': 100, 'trimok': 50}[workload]
    assert d['simulations_per_sec'] * d['ms_per_step'] * 1e-3 * d['steps'] == pytest.approx(2 * slots * sims * 5, rel=1e-3)   # both ranks' work is in the line
    if workload != 'arena':
        assert d['sparse_heads']['value'] > 0 and d['config']['fused_search_launch'] and d['config']['search_heads'] == 'exact'


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """one rank launched, --gpus 2 claimed: bench.py must fail instead of printing n_gpus = 2 for a one-GPU run."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29614', AZG_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--slots', '64'],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert r.returncode != 0 and b'"metric"' not in r.stdout and b'rank(s) were launched' in r.stdout


@pytest.mark.parametrize('game', ['brandubh', 'trimok'])
def test_wide_head_runner_graph_equals_eager_launches(game):
    """networks with factorised heads (brandubh A = 588, 3-player env A = 25): the captured round hands the head FEATURES to the
    tree launch (azg_backup_select_features: sparse heads + softmax + backup + select in one launch).  The graph replay and the
    same launch sequence issued eagerly play the same games and emit the same samples; the plain three-call step (heads ->
    softmax kernel -> backup -> select, all A logits) agrees on the priors to rounding -- see
    test_sparse_heads_equal_full_heads_on_the_valid_actions."""
    import importlib
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.selfplay import SelfPlayRunner
    import torch
    Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
    torch.manual_seed(17)
    net = N.NNetWrapper(Game, N.BRANDUBH_NET_ARGS if game == 'brandubh' else N.DEFAULT_NET_ARGS, device='cuda:0', dtype=torch.float16)
    outs = []
    for eager in (False, True):
        r = SelfPlayRunner(Game, net, _args(numMCTSSims=9, cpuct=1.25, fpu_reduction=0.2), num_slots=40, seed=6, use_graph=True,
                           fused_search=False, search_heads='sparse', example_capacity=40 * 101 * 8 * 2)
        assert r.lanes[0].net.run_features is not None and not r.fused_search
        for _ in range(14):
            r.play_round(eager=eager)
        o, p, z = r.samples()
        outs.append((o.cpu().numpy(), p.cpu().numpy(), z.cpu().numpy(), r.engine.last_actions().cpu().numpy(), r.counters()))
    a, b = outs
    assert a[4] == b[4] and a[4]['sims'] == 14 * 9 * 40
    for x, y in zip(a[:4], b[:4]):
        assert x.shape == y.shape and (x == y).all()


def test_pinned_heads_features_needs_no_second_switch():
    """SelfPlayRunner(heads='features') ALONE (the default search_heads = 'exact' used to downgrade the features hand-over before
    the pinned form was looked at, and the call raised NotImplementedError): a pinned hand-over form is taken as given, and plays the games
    the same form plays when search_heads='sparse' is passed beside it."""
    import torch
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.envs.trimok import Game
    from alphazero_general_amd.selfplay import SelfPlayRunner
    torch.manual_seed(19)
    net = N.NNetWrapper(Game, N.DEFAULT_NET_ARGS, device='cuda:0', dtype=torch.float16)
    outs = []
    for kw in (dict(heads='features'), dict(heads='features', search_heads='sparse'), dict(heads='logits')):
        r = SelfPlayRunner(Game, net, _args(numMCTSSims=8, cpuct=1.25, fpu_reduction=0.2), num_slots=24, seed=3, example_capacity=24 * 26 * 4, **kw)
        assert not r.fused_search
        for _ in range(6):
            r.play_round()
        outs.append((r.engine.last_actions().cpu().numpy(), r.counters()))
    assert (outs[0][0] == outs[1][0]).all() and outs[0][1] == outs[1][1] and outs[0][1]['sims'] == 6 * 8 * 24 == outs[2][1]['sims']


@pytest.mark.parametrize('game,heads', [('brandubh', 'exact'), ('trimok', 'exact'), ('brandubh', 'sparse'), ('trimok', 'sparse')])
def test_wide_head_runner_persistent_launch_with_fast_rounds_and_resets(game, heads):
    """The persistent wide-head launches inside the native runner (azg_search_wide_exact_f16, the default, and azg_search_wide_f16:
    brandubh with four wavefronts per game -- k-split tower, shuffle masks and the rules of the walk on wavefronts of their own)
    against the launch-per-phase runner of the same hand-over (logits / features), with what changes
    the shape of a round riding along: fast rounds (a second simulation count, no history: SelfPlayAgent.pyx:83-92) and periodic tree
    resets (mctsResetThreshold, :172-174).  Samples, results, actions, counters identical."""
    import importlib
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.selfplay import SelfPlayRunner
    import torch
    Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
    torch.manual_seed(23)
    net = N.NNetWrapper(Game, N.BRANDUBH_NET_ARGS if game == 'brandubh' else N.DEFAULT_NET_ARGS, device='cuda:0', dtype=torch.float16)
    B, rounds = 37, 40
    outs = []
    for fused in (True, False):
        r = SelfPlayRunner(Game, net, _args(numMCTSSims=12, numFastSims=5, probFastSim=0.4, mctsResetThreshold=3, cpuct=1.25, fpu_reduction=0.2),
                           num_slots=B, seed=4, use_graph=True, fused_search=fused, search_heads=heads, example_capacity=B * 101 * 8 * 3)
        assert r.fused_search == fused and r.search_exact == (heads == 'exact')
        for _ in range(rounds):
            r.play_round()
        o, p, z = r.samples()
        ws, turns, slot = r.results()
        outs.append((o.cpu().numpy(), p.cpu().numpy(), z.cpu().numpy(), r.engine.last_actions().cpu().numpy(), np.asarray(ws), np.asarray(turns),
                     np.asarray(slot), r.counters(), list(r.sims_per_round)))
    a, b = outs
    assert a[7] == b[7] and a[8] == b[8] and len(set(a[8])) == 2 and a[7]['games_played'] > 0
    for x, y in zip(a[:7], b[:7]):
        assert x.shape == y.shape and (x == y).all()


def test_runner_writes_coach_iteration_files(tmp_path):
    """SelfPlayRunner.save_iteration_samples writes what Coach.saveIterationSamples writes (Coach.py:363-386): three float32 CPU
    tensors that load with torch.load and line up row by row; game_results is get_game_results (utils.py:34-54)."""
    import torch
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.selfplay import SelfPlayRunner
    r = SelfPlayRunner(Game, _net(5), _args(gamesPerIteration=40), num_slots=32, seed=2)
    c = r.run()
    n = r.save_iteration_samples(str(tmp_path / 'run'), 3)
    # (weights_only=False: what torch.load meant in the reference's torch < 2.5, Coach.py:448-450)
    d, p, v = [torch.load(str(tmp_path / 'run' / ('iteration-0003-%s.pkl' % k)), weights_only=False) for k in ('data', 'policy', 'value')]
    assert d.shape == (n, 4, 6, 7) and p.shape == (n, 7) and v.shape == (n, 3) and d.dtype == p.dtype == v.dtype == torch.float32
    assert not d.is_cuda and n == c['num_examples'] > 0
    assert torch.allclose(p.sum(1), torch.ones(n), atol=1e-5) and ((v == 0) | (v == 1)).all() and (v.sum(1) == 1).all()
    wins, draws, avg_len = r.game_results()
    assert sum(wins) + draws == c['num_results'] >= 40 and 7 <= avg_len <= 42


def test_arena_runner_per_slot_seats():
    """seats='slot': every concurrent game has its own seating.  The row map the device builds must equal a numpy restatement
    (model of a game's mover = its slot's permutation), roughly half the games seat model 0 first, graph == eager, and the
    tallies add up per model."""
    import torch
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.selfplay import ArenaRunner
    nets = [_net(0), _net(1)]
    runs = []
    for use_graph, fused in ((True, True), (False, True), (True, False)):   # the persistent launch (graph / eager) and the launch-per-phase form
        r = ArenaRunner(Game, nets, _args(), num_slots=96, seed=8, use_graph=use_graph, seats='slot', fused_search=fused)
        assert r.fused_search == fused
        first = sum(1 for m in r.slot_player_to_index if m[0] == 0)
        assert 24 <= first <= 72                                      # a fair coin per slot
        for rnd in range(12):
            r.play_round()
            if rnd in (0, 5):
                row_of_slot, rpm = r._rows()
                movers = [st[1] for st in r.engine.get_states()]
                model = np.array([r.slot_player_to_index[s][movers[s]] for s in range(96)])
                exp = np.zeros(96, np.int64)
                exp[model == 0] = np.arange((model == 0).sum()); exp[model == 1] = (model == 0).sum() + np.arange((model == 1).sum())
                assert (row_of_slot.cpu().numpy() == exp).all() and rpm.cpu().tolist() == [(model == 0).sum(), (model == 1).sum()]
        runs.append((r.engine.last_actions().cpu().numpy().copy(), r.engine.counters(), r.results()))
    for other in runs[1:]:
        assert (runs[0][0] == other[0]).all() and runs[0][1] == other[1] and runs[0][2] == other[2]
    wins, draws, _ = runs[0][2]
    assert sum(wins) + draws == runs[0][1]['games_played'] > 0


def test_fused_search_kernel_tree_arena_overflow_is_reported():
    """a tree arena that is too small: the persistent search launch must stop expanding, finish, and leave the sticky device
    error for the next counter read -- no out-of-bounds write, no hang."""
    from alphazero_general_amd import _abi
    from alphazero_general_amd.engine import DeviceEngine
    net = _net(4); net.refresh()
    e = DeviceEngine(0, 37, cpuct=4.0, fpu_reduction=0.4, seed=1, sims_hint=4, nodes_per_tree=48)
    net._hip.search(e, 60)
    with pytest.raises(_abi.AzgError) as ei:
        e.counters()
    assert ei.value.code == _abi.E_TREE_FULL
    e.close()
    e2 = DeviceEngine(0, 37, cpuct=4.0, fpu_reduction=0.4, seed=1, sims_hint=60)      # the device is fine afterwards
    net._hip.search(e2, 60)
    assert e2.counters()['sims'] == 37 * 60


def test_example_buffer_overflow_is_reported():
    """a training-example buffer that is too small for the finished games' samples (SelfPlayAgent.pyx:184-196 puts them on a queue
    that cannot fill up): k_finalize must raise the sticky AZG_E_EXAMPLES_FULL BEFORE any sample is written past the buffer, and the
    next counter read must report it."""
    import torch
    from alphazero_general_amd import _abi
    from alphazero_general_amd.engine import DeviceEngine
    B = 16
    e = DeviceEngine(0, B, cpuct=1.25, fpu_reduction=0.2, seed=5, sims_hint=4, example_capacity=8, games_per_iteration=64)
    pol = torch.full((B, 7), 1 / 7, dtype=torch.float32, device=e.device)
    val = torch.full((B, 3), 1 / 3, dtype=torch.float32, device=e.device)
    guard = torch.full((4096,), 7.0, device=e.device)                   # (allocated right after the engine's buffers)
    with pytest.raises(_abi.AzgError) as ei:
        for mv in range(64):                                            # a connect4 game ends within 42 plies: >= 7 positions x 2 symmetries > 8
            e.select(None)
            for s in range(3):
                e.backup_select(pol, val, None)
            e.backup(pol, val)
            e.advance(record_history=True)
            e.counters()
    assert ei.value.code == _abi.E_EXAMPLES_FULL
    assert (guard == 7.0).all()
    e.close()
    e2 = DeviceEngine(0, B, cpuct=1.25, fpu_reduction=0.2, seed=5, sims_hint=4, example_capacity=4096, games_per_iteration=64)
    for mv in range(64):                                                # the same games with room for their samples
        e2.select(None)
        for s in range(3):
            e2.backup_select(pol, val, None)
        e2.backup(pol, val)
        e2.advance(record_history=True)
    c = e2.counters()
    assert c['games_played'] > 0 and c['num_examples'] >= 14 * c['games_played']
    e2.close()


@pytest.mark.parametrize('game', ['brandubh', 'trimok'])
def test_wide_search_kernel_tree_arena_overflow_is_reported(game):
    """the same for the two-wavefront persistent launch (azg_search_wide_f16): walker and helper of a game must both stop on the
    sticky error -- neither may be left waiting for the other's flag -- and the launch must end."""
    import importlib
    import torch
    from alphazero_general_amd import _abi, nnet as N
    from alphazero_general_amd.engine import DeviceEngine
    Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
    torch.manual_seed(3)
    net = N.NNetWrapper(Game, N.BRANDUBH_NET_ARGS if game == 'brandubh' else N.DEFAULT_NET_ARGS, device='cuda:0', dtype=torch.float16)
    net.refresh()
    e = DeviceEngine(Game.AZG_GAME_ID, 21, cpuct=1.25, fpu_reduction=0.2, seed=1, sims_hint=2, nodes_per_tree=200)
    net._hip.search(e, 80)
    with pytest.raises(_abi.AzgError) as ei:
        e.counters()
    assert ei.value.code == _abi.E_TREE_FULL
    e.close()
    e2 = DeviceEngine(Game.AZG_GAME_ID, 21, cpuct=1.25, fpu_reduction=0.2, seed=1, sims_hint=80)     # the device is fine afterwards
    net._hip.search(e2, 80)
    assert e2.counters()['sims'] == 21 * 80


def test_exchange_step_on_rccl_world_of_one():
    """distributed.all_gather_examples through the RCCL branch (all_gather_into_tensor) -- a one-rank nccl group is all a
    single-GPU box can host; the multi-rank layout is covered by the gloo tests on CPU."""
    import os
    import subprocess
    import sys
    code = '''
import os, sys, torch
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
import torch.distributed as dist
from alphazero_general_amd import distributed as D
dist.init_process_group('nccl', rank=0, world_size=1)
torch.cuda.set_device(0)
g = torch.Generator(device='cuda').manual_seed(1)
obs = torch.rand((37, 4, 6, 7), device='cuda', generator=g); pi = torch.rand((37, 7), device='cuda', generator=g); z = torch.rand((37, 3), device='cuda', generator=g)
o, p, v = D.all_gather_examples(obs, pi, z)
assert torch.equal(o, obs) and torch.equal(p, pi) and torch.equal(v, z)
o, p, v = D.all_gather_examples(obs[:0], pi[:0], z[:0])
assert o.shape[0] == 0 and p.shape == (0, 7)
assert D.max_over_ranks(2.5) == 2.5 and D.all_reduce_tallies([3, 4]).tolist() == [3, 4]
recs = D.describe_ranks(0)                       # all_gather_object over RCCL
assert len(recs) == 1 and recs[0]['rank'] == 0 and recs[0]['compute_units'] > 0 and recs[0]['pci_bus_id'] and recs[0]['rccl_version']
D.barrier(); D.shutdown()
print('RCCL_OK')
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert r.returncode == 0 and b'RCCL_OK' in r.stdout, r.stdout.decode(errors='replace')[-2000:]


def test_two_ranks_on_one_gpu_are_refused(tmp_path):
    """distributed.describe_ranks (bench.py calls it before anything is timed): two ranks whose device is the same physical GPU must
    fail loudly -- they would otherwise share it and report a curve that looks like poor scaling; AZG_SINGLE_DEVICE (the one-GPU
    test rig of test_bench_two_ranks_on_one_gpu) is the explicit way around it."""
    import os
    import subprocess
    import sys
    code = '''
import os, sys, torch
sys.path.insert(0, %r)
import torch.multiprocessing as mp

def work(rank, port, single, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK='0')
    if single:
        os.environ['AZG_SINGLE_DEVICE'] = '1'
    else:
        os.environ.pop('AZG_SINGLE_DEVICE', None)
    import torch.distributed as dist
    from alphazero_general_amd import distributed as D
    dist.init_process_group('gloo', rank=rank, world_size=2)
    try:
        recs = D.describe_ranks(0)
        q.put((rank, 'ok', len(recs)))
    except RuntimeError as ex:
        q.put((rank, 'refused', str(ex)))
    dist.destroy_process_group()

if __name__ == '__main__':
    import socket
    ctx = mp.get_context('spawn')
    for single in (False, True):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        q = ctx.Queue()
        ps = [ctx.Process(target=work, args=(r, port, single, q)) for r in range(2)]
        [p.start() for p in ps]
        res = sorted(q.get(timeout=120) for _ in ps)
        [p.join(60) for p in ps]
        if single:
            assert all(r[1] == 'ok' and r[2] == 2 for r in res), res
        else:
            assert all(r[1] == 'refused' and 'share one GPU' in r[2] for r in res), res
    print('REFUSAL_OK')
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'two_ranks.py'                              # (a file: spawned children re-import the main module)
    script.write_text(code)
    r = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b'REFUSAL_OK' in r.stdout, r.stdout.decode(errors='replace')[-2000:]


def test_persistent_launches_refuse_what_they_are_not_built_for():
    """azg_search_arena_f16 on a self-play engine or another game, azg_search_f16 / azg_search_wide_* on an arena engine, a seat map that
    names a model that was not handed over: AZG_E_UNSUPPORTED / AZG_E_INVALID_ARG with a message, nothing launched."""
    import torch
    from alphazero_general_amd import _abi
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.engine import DeviceEngine
    from alphazero_general_amd.nnet import HipResNet
    from alphazero_general_amd.envs.brandubh import Game as BR
    nets = [_net(0), _net(1)]
    for n in nets:
        n.refresh()
    hips = [n._hip for n in nets]
    plain = DeviceEngine(0, 8, sims_hint=10)
    arena = DeviceEngine(0, 8, arena=True, sims_hint=10)
    with pytest.raises(_abi.AzgError) as ei:
        HipResNet.search_arena(hips, plain, 4, player_to_index=[0, 1])
    assert ei.value.code == _abi.E_UNSUPPORTED
    with pytest.raises(_abi.AzgError):
        HipResNet.search_arena(hips, arena, 4, player_to_index=[0, 2])      # model 2 does not exist
    with pytest.raises(_abi.AzgError) as ei:
        hips[0].search(arena, 4)
    assert ei.value.code == _abi.E_UNSUPPORTED
    torch.manual_seed(1)
    bnet = N.NNetWrapper(BR, N.BRANDUBH_NET_ARGS, device='cuda:0', dtype=torch.float16); bnet.refresh()
    barena = DeviceEngine(1, 4, arena=True, sims_hint=10)
    for exact in (True, False):
        with pytest.raises(_abi.AzgError) as ei:
            bnet._hip.search(barena, 4, exact=exact)
        assert ei.value.code == _abi.E_UNSUPPORTED
    HipResNet.search_arena(hips, arena, 3, player_to_index=[1, 0])          # (and the engines are still usable)
    assert arena.counters()['sims'] == 8 * 3
    for e in (plain, arena, barena):
        e.close()
