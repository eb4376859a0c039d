"""The launches bench.py TIMES, at the sizes it times them, directly against the CPU oracle (no launch-form-to-launch-form step in
between):

  config 2   azg_search_f16        connect4, 2048 games x 100 simulations per move
  config 3   azg_search_wide_exact_f16 (and the sparse-heads azg_search_wide_f16)   brandubh, 512 games x 200 simulations per move (the
             8-GPU shard), 1024 and 2048 games (the 4- and 2-GPU shards: two and four games per workgroup)
  config 5   the same launches     3-player env, 256 games x 50 simulations per move (4-GPU shard), 1024 games (1-GPU shard)
  config 4   ArenaRunner's graph   connect4 arena, 256 games x 100 simulations, two nets (per-GPU shard): one multi-model tower launch +
                                   one tree launch per simulation, a whole move replayed as one hipGraph

The oracle (oracle/azg_mcts_ref.c + azg_pool_ref.c: SelfPlayAgent.generateBatch / processBatch / playMoves, SelfPlayAgent.pyx:103-202,
over MCTS.find_leaf / process_results, MCTS.pyx:208-289; pinned to the reference's goldens by tests/test_oracle_golden.py) runs as a
pool of agents on the host threads, every agent owning a contiguous range of the SAME global slots (same random tape), and is fed
the network's evaluation of ITS OWN leaf observations every simulation.

connect4 hands over exact probabilities (fused heads): visit counts, pi, sampled actions every move, then samples / results /
counters must be identical.  The sparse-heads launches (configs 3 and 5) are compared twice: against an oracle fed the sparse
evaluation itself (azg_leaf_heads_sparse_f16 + azg_heads_softmax on an every-stage-its-own-launch twin): identical; and against an
oracle fed NNetWrapper.process (full-width heads, what the reference computes: MCTS.pyx:239-245): equal to rounding, the fraction
of slots that never diverged is recorded (gpurun_out/nn_error.jsonl) and must be >= 0.95."""
import importlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _threads():
    """host threads for the oracle pool: the container's CPU quota (cgroup cpu.max), not the host's CPU count"""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            return max(2, min(32, int(round(int(q) / int(per)))))
    except (OSError, ValueError):
        pass
    return max(2, min(32, os.cpu_count() or 2))


def _net(game, seed):
    import torch
    from alphazero_general_amd import nnet as N
    name, _, width = game.partition(':')                              # 'connect4:32': connect4 with the reference's default net (Coach.py:108-116)
    Game = importlib.import_module('alphazero_general_amd.envs.' + name).Game
    args = {'connect4': N.CONNECT4_NET_ARGS, 'brandubh': N.BRANDUBH_NET_ARGS, 'trimok': N.DEFAULT_NET_ARGS}[name]
    if width:
        args = N.dotdict(dict(N.DEFAULT_NET_ARGS, num_channels=int(width)))
    torch.manual_seed(seed)
    net = N.NNetWrapper(Game, args, device='cuda:0', dtype=torch.float16)
    net.refresh()
    assert net._hip is not None and net._hip.can_search
    return Game, net


def _pool(gid, B, sims, seed, cpuct, fpu):
    n = _threads()
    while B % n:
        n -= 1
    return ol.OPool(gid, n, B // n, sims=sims, games_per_iteration=1 << 30, seed=seed, cpuct=cpuct, fpu_reduction=fpu,
                    add_root_noise=True, add_root_temp=True)


def _sorted_rows(*arrs):
    """the rows of several [n, ...] arrays glued together and sorted: a multiset of samples (the pool's agents each keep their own
    output_queue, the engine has one)"""
    flat = np.concatenate([np.ascontiguousarray(a, np.float32).reshape(a.shape[0], -1) for a in arrs], axis=1)
    v = flat.view(np.uint32)
    return v[np.lexsort(v.T[::-1])]


def test_connect4_search_launch_vs_oracle_at_2048x100():
    """BASELINE config 2 as bench.py times it: 2048 games x 100 simulations, noise + root temperature on, one azg_search_f16 launch
    per move; 26 moves, so that games finish, restart and emit samples."""
    import torch
    from alphazero_general_amd.engine import DeviceEngine
    Game, net = _net('connect4', 0)
    B, sims, moves, seed = 2048, 100, 26, 0
    eng = DeviceEngine(0, B, cpuct=4.0, fpu_reduction=0.4, add_root_noise=True, add_root_temp=True, seed=seed, games_per_iteration=1 << 30,
                       example_capacity=B * 43 * 2 * 2, sims_hint=sims)
    pool = _pool(0, B, sims, seed, 4.0, 0.4)
    probe = list(range(0, B, 37))
    for mv in range(moves):
        net._hip.search(eng, sims)                                # the timed launch
        pool.begin_round()
        for _ in range(sims):
            p, v = net.process(torch.from_numpy(pool.generate()))
            pool.process(p.cpu().numpy(), v.cpu().numpy())
        assert (eng.root_counts().cpu().numpy() == pool.root_counts()).all(), mv
        assert (eng.root_probs(1.0).cpu().numpy()[probe] == pool.root_probs(probe, 1.0)).all(), mv
        pool.play(); eng.advance(True)
        assert (eng.last_actions().cpu().numpy() == pool.last_actions()).all(), mv
    c = eng.counters()
    assert c['sims'] == B * sims * moves == pool.sims_done and c['expansions'] == pool.expansions
    assert c['games_played'] == pool.games_played > 0
    eo, ep, ez = [t.cpu().numpy() for t in eng.examples()]
    oo, op, oz = pool.samples()
    assert eo.shape[0] == oo.shape[0] > 0
    assert (_sorted_rows(eo, ep, ez) == _sorted_rows(oo, op, oz)).all()
    ws, turns, slot = eng.results()
    ows, oturns, oslot = pool.results()
    key = lambda w, t, s: sorted(zip(s.tolist(), t.tolist(), [tuple(x) for x in w.tolist()]))
    assert key(ws, turns, slot) == key(ows, oturns, oslot)
    eng.close()


@pytest.mark.parametrize('game,B,sims,moves', [('brandubh', 512, 200, 5), ('trimok', 256, 50, 8)])
def test_wide_search_launch_vs_oracle_at_bench_size(game, B, sims, moves):
    """BASELINE configs 3 and 5 (per-GPU shard) as bench.py times them.  ea = azg_search_wide_f16; ec = every stage its own launch
    (azg_select, tower, azg_leaf_heads_sparse_f16, azg_heads_softmax, azg_backup), whose per-leaf probabilities feed oracle pool
    `exact`; oracle pool `full` is fed NNetWrapper.process of its own leaves."""
    import torch
    from alphazero_general_amd.engine import DeviceEngine
    Game, net = _net(game, 7)
    hip = net._hip
    gid, seed = Game.AZG_GAME_ID, 29
    gi = ol.game_info(gid)
    kw = dict(cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=seed, games_per_iteration=1 << 30,
              example_capacity=B * (moves + 1) * gi.num_symmetries, sims_hint=sims)
    ea, ec = DeviceEngine(gid, B, **kw), DeviceEngine(gid, B, **kw)
    exact, full = _pool(gid, B, sims, seed, 1.25, 0.2), _pool(gid, B, sims, seed, 1.25, 0.2)
    oc = ec.new_obs(torch.float32)
    same = np.ones(B, bool)
    first_div = None
    for mv in range(moves):
        hip.search(ea, sims)                                      # the timed launch
        exact.begin_round(); full.begin_round()
        for s in range(sims):
            ec.select(oc)
            oobs = exact.generate()
            if s % 20 == 0:
                assert (oc.cpu().numpy() == oobs).all(), (mv, s)
            lg = ec.leaf_heads_sparse(hip.forward_features_nhwc8(hip.to_nhwc8(oc), key=2), hip.head_rows, hip.head2_b)
            pol, val = ec.heads_softmax(lg)
            ec.backup(pol, val)
            exact.process(pol.cpu().numpy(), val.cpu().numpy())
            p, v = net.process(torch.from_numpy(full.generate()))
            full.process(p.cpu().numpy(), v.cpu().numpy())
        cnt = ea.root_counts()
        assert torch.equal(cnt, ec.root_counts()), mv
        cnt = cnt.cpu().numpy()
        assert (cnt == exact.root_counts()).all(), mv             # the tree side of the timed launch: bit for bit
        assert torch.equal(ea.root_probs(1.0), ec.root_probs(1.0)) and torch.equal(ea.root_value(True), ec.root_value(True))
        same &= (cnt == full.root_counts()).all(1)                # the network side: to rounding
        exact.play(); full.play(); ea.advance(True); ec.advance(True)
        act = ea.last_actions().cpu().numpy()
        assert (act == exact.last_actions()).all() and torch.equal(ea.last_actions(), ec.last_actions()), mv
        same &= act == full.last_actions()
        if first_div is None and not same.all():
            first_div = mv
    assert (ea.tape_counters() == ec.tape_counters()).all()
    c = ea.counters()
    assert c == ec.counters() and c['sims'] == B * sims * moves == exact.sims_done and c['expansions'] == exact.expansions
    assert c['games_played'] == exact.games_played
    frac = float(same.mean())
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/nn_error.jsonl', 'a') as fh:
        fh.write(json.dumps({'test': 'bench_size_search_vs_oracle_' + game, 'slots': B, 'sims': sims, 'rounds': moves,
                             'vs_oracle_fed_sparse_evaluation': 'identical', 'slots_never_diverged_vs_full_heads': frac,
                             'first_divergence_round': first_div, 'simulations_compared': B * sims * moves}) + '\n')
    assert frac >= 0.95, (frac, first_div)
    ea.close(); ec.close()


@pytest.mark.parametrize('game,B,sims,moves', [('brandubh', 512, 200, 5), ('brandubh', 1024, 200, 2), ('brandubh', 2048, 200, 3),
                                                 ('trimok', 256, 50, 8), ('trimok', 1024, 50, 5),
                                                 # BASELINE config 1's shape and network (connect4, 32 games x 25 sims, the reference's default
                                                 # net) through whole games, its 64-channel sibling, and the default net at config 2's size
                                                 ('connect4:32', 32, 25, 44), ('connect4:64', 96, 25, 12), ('connect4:32', 2048, 100, 4)])
def test_wide_exact_search_launch_vs_oracle_at_bench_size(game, B, sims, moves):
    """BASELINE configs 3 and 5 as bench.py times them by default, at their 8- / 4-GPU shard size and at the 2- and 1-GPU shard sizes
    (one, two and four games per workgroup) -- and config 1's network on connect4 (round 6: the default net's persistent launch): azg_search_wide_exact_f16 -- all A + P+1 logits inside the launch, softmax over all A, mask,
    renormalise -- against the oracle pool fed NNetWrapper.process of ITS OWN leaves (NNetWrapper.py:225-232 -> MCTS.pyx:239-245, what
    the reference computes).  No slot may diverge: visit counts of every root and pi every move, sampled actions, tape counters,
    samples (as multisets), results, counters -- identical."""
    import torch
    from alphazero_general_amd.engine import DeviceEngine
    Game, net = _net(game, 7)
    hip = net._hip
    gid, seed = Game.AZG_GAME_ID, 31
    gi = ol.game_info(gid)
    eng = DeviceEngine(gid, B, cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=seed, games_per_iteration=1 << 30,
                       example_capacity=B * (moves + 1) * gi.num_symmetries, sims_hint=sims)
    pool = _pool(gid, B, sims, seed, 1.25, 0.2)
    probe = list(range(0, B, 29))
    for mv in range(moves):
        hip.search(eng, sims, exact=True)                         # the timed launch
        pool.begin_round()
        for _ in range(sims):
            p, v = net.process(torch.from_numpy(pool.generate()))
            pool.process(p.cpu().numpy(), v.cpu().numpy())
        assert (eng.root_counts().cpu().numpy() == pool.root_counts()).all(), mv
        assert (eng.root_probs(1.0).cpu().numpy()[probe] == pool.root_probs(probe, 1.0)).all(), mv
        pool.play(); eng.advance(True)
        assert (eng.last_actions().cpu().numpy() == pool.last_actions()).all(), mv
    c = eng.counters()
    assert c['sims'] == B * sims * moves == pool.sims_done and c['expansions'] == pool.expansions
    assert c['games_played'] == pool.games_played
    eo, ep, ez = [t.cpu().numpy() for t in eng.examples()]
    oo, op, oz = pool.samples()
    assert eo.shape[0] == oo.shape[0]
    if eo.shape[0]:
        assert (_sorted_rows(eo, ep, ez) == _sorted_rows(oo, op, oz)).all()
    ws, turns, slot = eng.results()
    ows, oturns, oslot = pool.results()
    key = lambda w, t, s: sorted(zip(s.tolist(), t.tolist(), [tuple(x) for x in w.tolist()]))
    assert key(ws, turns, slot) == key(ows, oturns, oslot)
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/nn_error.jsonl', 'a') as fh:
        fh.write(json.dumps({'test': 'bench_size_exact_search_vs_oracle_' + game, 'slots': B, 'sims': sims, 'rounds': moves,
                             'vs_oracle_fed_NNetWrapper_process': 'identical', 'simulations_compared': B * sims * moves}) + '\n')
    eng.close()


@pytest.mark.parametrize('B,moves,fused', [(256, 24, True), (512, 9, True), (256, 9, False)])
def test_arena_graph_vs_oracle_at_256x100(B, moves, fused):
    """BASELINE config 4 as bench.py times it -- 256 arena games x 100 simulations (the 2-GPU shard) and all 512 on one GPU, two
    differently seeded 128ch x 8 nets, arenaTemp 0.25: the native runner's whole-move hipGraph -- the persistent launch
    azg_search_arena_f16 (one game per workgroup, the mover's tree and model), and the launch-per-phase form (fused=False: device-side row
    split, both models in one tower launch + one tree launch per simulation) -- against the
    oracle's arena agent (SelfPlayAgent.pyx:44-47,117-132,142-151 with the reference's row mis-routing off) fed, every simulation, by
    the same two networks evaluating ITS leaf rows per model: actions every move, then tallies, results, counters -- until games have
    finished and restarted."""
    import torch
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper
    from alphazero_general_amd.selfplay import ArenaRunner
    from alphazero_general_amd.utils import dotdict, default_temp_scaling
    nets = []
    for sd in (0, 1):
        torch.manual_seed(sd)
        n = NNetWrapper(Game, CONNECT4_NET_ARGS, device='cuda:0', dtype=torch.float16); n.refresh(); nets.append(n)
    sims, seed = 100, 0
    args = dotdict(numMCTSSims=sims, numFastSims=20, probFastSim=0.0, gamesPerIteration=1 << 30, cpuct=4.0, fpu_reduction=0.4,
                   root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1.0, add_root_noise=True, add_root_temp=True,
                   symmetricSamples=True, mctsResetThreshold=0, startTemp=1.0, arenaTemp=0.25, temp_scaling_fn=default_temp_scaling)
    r = ArenaRunner(Game, nets, args, num_slots=B, seed=seed, result_capacity=8 * B, fused_search=fused)
    assert r.device_split and r._graph is not None and r.fused_search == fused
    ag = ol.OAgent(0, B, sims=sims, games_per_iteration=1 << 30, seed=seed, cpuct=4.0, fpu_reduction=0.4, is_arena=True, ref_misroute=False)
    assert ag.player_to_index() == r.player_to_index
    for mv in range(moves):
        ag.begin_round()
        for _ in range(sims):
            oobs, rg, rm = ag.generate_batch()
            pol = np.zeros((B, 7), np.float32); val = np.zeros((B, 3), np.float32)
            for m, n in enumerate(nets):
                idx = np.flatnonzero(rm == m)
                if len(idx):
                    p, v = n.process(torch.from_numpy(oobs[idx]))
                    pol[idx], val[idx] = p.cpu().numpy(), v.cpu().numpy()
            ag.process_batch(pol, val)
        ag.play_moves()
        r.play_round()                                            # the timed form: one graph replay per move
        assert (r.engine.last_actions().cpu().numpy() == ag.last_actions()).all(), mv
    c = r.engine.counters()
    assert c['games_played'] == ag.games_played and c['sims'] == B * sims * moves == ag.sims_done and c['expansions'] == ag.expansions
    assert moves < 20 or c['games_played'] > 0
    ws, turns, slot = r.engine.results()
    ows, oturns, oslot = ag.results()
    assert len(ws) == len(ows) and (ws == ows).all() and (turns == oturns).all() and (slot == oslot).all()
