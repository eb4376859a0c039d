"""Parity of the TIMED paths at the sizes bench.py times them (BASELINE.json configs 2 and 4):

  * azg_search_f16 (the persistent launch behind the headline number) against the launch-per-phase path
    azg_select / azg_resnet_policy_value_f16 / azg_backup on a twin engine at 2048 games x 100 simulations;
  * the native arena (ArenaRunner: two 128-channel nets in one multi-model launch, the whole move one hipGraph) at 256 games x
    100 simulations against its host-split form, and at 64 games against the CPU oracle (reference routing bug off) fed by
    the same networks, past the first finished games.

Integer / index work is compared bit-exactly (torch.equal); the networks are the same kernels on both sides, and a board's
probabilities do not depend on its batch position (tests/test_gpu_nnet.py), so the float tree values are bit-equal too."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _net(seed):
    import torch
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper
    torch.manual_seed(seed)
    n = NNetWrapper(Game, CONNECT4_NET_ARGS, device='cuda:0', dtype=torch.float16)
    n.refresh()
    return n


def _wide_game_net(game):
    """'brandubh' / 'trimok' / 'connect4:32' / 'connect4:64' -> (Game class, net args of a factorised-heads network); connect4 x 32 is the
    reference's DEFAULT net (Coach.py:108-116), BASELINE config 1's network"""
    import importlib
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.utils import dotdict
    name, _, width = game.partition(':')
    Game = importlib.import_module('alphazero_general_amd.envs.' + name).Game
    na = dotdict(dict(N.BRANDUBH_NET_ARGS if name == 'brandubh' else N.DEFAULT_NET_ARGS))
    if width:
        na['num_channels'] = int(width)
    return Game, na


def _args(**kw):
    from alphazero_general_amd.utils import dotdict, default_temp_scaling
    a = dotdict(numMCTSSims=100, numFastSims=20, probFastSim=0.0, gamesPerIteration=1 << 30, cpuct=4.0, fpu_reduction=0.4,
                root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1.0, add_root_noise=True, add_root_temp=True,
                symmetricSamples=True, mctsResetThreshold=0, startTemp=1.0, arenaTemp=0.25, temp_scaling_fn=default_temp_scaling)
    a.update(kw)
    return a


def test_search_launch_equals_phase_launches_at_2048x100():
    """the bench's launch (2048 games, 100 simulations per launch, noise + root temperature on) for 4 moves from the start and 3
    more after 12 cheaper moves (mid-game trees, compaction has run): visit counts, pi at T = 1, sampled actions, tape counters,
    counters and the emitted samples must be identical to the three-launch path."""
    import torch
    from alphazero_general_amd.engine import DeviceEngine
    hip = _net(0)._hip
    B, sims = 2048, 100
    kw = dict(cpuct=4.0, fpu_reduction=0.4, add_root_noise=True, add_root_temp=True, seed=0, games_per_iteration=1 << 30,
              example_capacity=B * 43 * 2 * 2, sims_hint=sims)
    ea, eb = DeviceEngine(0, B, **kw), DeviceEngine(0, B, **kw)
    obs = torch.zeros((B, 42, 8), dtype=torch.float16, device=ea.device)

    def move(n):
        hip.search(ea, n)
        for _ in range(n):
            eb.select(obs)
            p, v = hip.forward_nhwc8(obs)
            eb.backup(p, v)
        assert torch.equal(ea.root_counts(), eb.root_counts())
        assert torch.equal(ea.root_probs(1.0), eb.root_probs(1.0))
        assert torch.equal(ea.root_value(False), eb.root_value(False))
        ea.advance(True); eb.advance(True)
        assert torch.equal(ea.last_actions(), eb.last_actions())
        assert (ea.tape_counters() == eb.tape_counters()).all()

    for _ in range(4):
        move(sims)
    for _ in range(12):
        move(9)
    for _ in range(3):
        move(sims)
    a, b = ea.counters(), eb.counters()
    assert a == b and a['sims'] == B * (7 * sims + 12 * 9) and a['games_played'] > 0
    for x, y in zip(ea.examples(), eb.examples()):
        assert x.shape[0] > 0 and torch.equal(x, y)
    for x, y in zip(ea.results(), eb.results()):
        assert (x == y).all()


def test_arena_256x100_graph_equals_host_split():
    """config 4 at its per-GPU size: 256 concurrent games x 100 simulations, two differently seeded 128ch x 8 nets.  The product
    path (device-side row split, both models in one launch, a whole move replayed as one hipGraph) against the host-split loop
    (one .cpu() read of the split per simulation, one launch per model): same moves, tallies and results after 16 moves."""
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.selfplay import ArenaRunner
    nets = [_net(0), _net(1)]
    runs = []
    for mode in ('graph', 'phase', 'host'):       # graph: the persistent launch (azg_search_arena_f16); phase: tower + tree launch per simulation
        r = ArenaRunner(Game, nets, _args(), num_slots=256, seed=3, use_graph=(mode != 'host'), fused_search=(mode == 'graph'))
        assert r.device_split and (r._graph is not None) == (mode != 'host') and r.fused_search == (mode == 'graph')
        if mode == 'host':
            r.device_split = False
        acts = []
        for _ in range(16):
            r.play_round()
            acts.append(r.engine.last_actions().cpu().numpy().copy())
        runs.append((np.array(acts), r.engine.counters(), r.results(), [x.copy() for x in r.engine.results()]))
    (a0, c0, res0, raw0) = runs[0]
    assert c0['sims'] == 256 * 100 * 16 and c0['games_played'] > 0                   # games have ended and restarted
    for (a1, c1, res1, raw1) in runs[1:]:
        assert (a0 == a1).all() and c0 == c1 and res0 == res1
        for x, y in zip(raw0, raw1):
            assert (x == y).all()


def test_arena_64_slots_vs_oracle_with_real_nets():
    """64 arena games x 40 simulations, both sides evaluated by the same two networks, until 70 games have finished (every slot
    has restarted at least once, the batch split between the models changes every simulation): the native runner's graph
    against the CPU oracle with the reference's row mis-routing switched off.  Actions, tallies, results."""
    import torch
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.selfplay import ArenaRunner
    nets = [_net(0), _net(1)]
    B, sims, games, seed = 64, 40, 70, 11
    r = ArenaRunner(Game, nets, _args(numMCTSSims=sims, gamesPerIteration=games), num_slots=B, seed=seed)
    ag = ol.OAgent(0, B, sims=sims, games_per_iteration=games, seed=seed, cpuct=4.0, fpu_reduction=0.4, is_arena=True, ref_misroute=False)
    assert ag.player_to_index() == r.player_to_index
    rounds = 0
    while ag.games_played < games:
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
        r.play_round()
        assert (r.engine.last_actions().cpu().numpy() == ag.last_actions()).all(), rounds
        rounds += 1
    c = r.engine.counters()
    assert c['games_played'] == ag.games_played == games and c['sims'] == ag.sims_done and c['expansions'] == ag.expansions
    ws, turns, slot = r.engine.results()
    ows, oturns, oslot = ag.results()
    assert (ws == ows).all() and (turns == oturns).all() and (slot == oslot).all() and len(ws) >= games


@pytest.mark.parametrize('game,B,sims,games', [('trimok', 48, 20, 52), ('brandubh', 32, 16, 34)])
def test_arena_three_players_and_wide_heads_vs_oracle(game, B, sims, games):
    """The batched Arena for the games whose heads are not fused into the tower (every simulation: device-side grouping of the leaf
    rows by model, one evaluation per model, results routed back) and for THREE players (three trees per game, three models, a seat
    permutation of three; Arena.pyx:208-328, SelfPlayAgent.pyx:44-47,117-132,142-151): the native runner against the CPU oracle fed by
    the same networks until every slot has restarted.  Actions every round, tallies, results."""
    import importlib
    import torch
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.selfplay import ArenaRunner
    Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
    P = Game.num_players()
    nets = []
    for m in range(P):
        torch.manual_seed(20 + m)
        n = N.NNetWrapper(Game, N.BRANDUBH_NET_ARGS if game == 'brandubh' else N.DEFAULT_NET_ARGS, device='cuda:0', dtype=torch.float16)
        n.refresh()
        nets.append(n)
    gid, seed = Game.AZG_GAME_ID, 13
    gi = ol.game_info(gid)
    r = ArenaRunner(Game, nets, _args(numMCTSSims=sims, gamesPerIteration=games), num_slots=B, seed=seed)
    ag = ol.OAgent(gid, B, sims=sims, games_per_iteration=games, seed=seed, cpuct=4.0, fpu_reduction=0.4, is_arena=True, ref_misroute=False)
    assert ag.player_to_index() == r.player_to_index and sorted(r.player_to_index) == list(range(P))
    rounds = 0
    while ag.games_played < games:
        ag.begin_round()
        for _ in range(sims):
            oobs, rg, rm = ag.generate_batch()
            pol = np.zeros((B, gi.action_size), np.float32); val = np.zeros((B, gi.num_players + gi.has_draw), np.float32)
            for m, n in enumerate(nets):
                idx = np.flatnonzero(rm == m)
                if len(idx):
                    p, v = n.process(torch.from_numpy(oobs[idx]))
                    pol[idx], val[idx] = p.cpu().numpy(), v.cpu().numpy()
            ag.process_batch(pol, val)
        ag.play_moves()
        r.play_round()
        assert (r.engine.last_actions().cpu().numpy() == ag.last_actions()).all(), rounds
        rounds += 1
    c = r.engine.counters()
    assert c['games_played'] == ag.games_played == games and c['sims'] == ag.sims_done and c['expansions'] == ag.expansions
    ws, turns, slot = r.engine.results()
    ows, oturns, oslot = ag.results()
    assert (ws == ows).all() and (turns == oturns).all() and (slot == oslot).all() and len(ws) >= games
    wins, draws, _ = r.results()
    assert sum(wins) + draws == len(ws) >= c['games_played'] and len(wins) == P    # (every finished game is put on the result queue, SelfPlayAgent.pyx:178; the cap only stops the count)


@pytest.mark.parametrize('game,B,sims,moves', [(0, 192, 24, 44), (1, 48, 40, 36)])
def test_node_reclamation_is_invisible(game, B, sims, moves):
    """Semi-space compaction of the node store after a move (the GPU counterpart of the reference dropping the played move's
    siblings, MCTS.pyx:185-195) must not change a single result: an engine whose node store is so small that it compacts after
    nearly every move against one that never has to -- same counts, moves, samples, results, tape counters -- while its live
    store stays bounded.  Both the separate and the fused (backup + select, two wavefronts) launches are used."""
    import torch
    from alphazero_general_amd import _abi
    from alphazero_general_amd.engine import DeviceEngine
    gi = _abi.game_info(game)
    A, NV = gi.action_size, gi.num_players + 1
    per_move = sims * gi.max_children
    kw = dict(cpuct=1.5, fpu_reduction=0.3, add_root_noise=True, add_root_temp=True, seed=5, games_per_iteration=1 << 30,
              example_capacity=B * (gi.max_turns + 1) * gi.num_symmetries * 2, sims_hint=sims)
    big = DeviceEngine(game, B, nodes_per_tree=(moves + 2) * per_move, **kw)             # never compacts
    small = DeviceEngine(game, B, nodes_per_tree=4 * per_move, **kw)                    # compacts whenever < 1 move's worth is free
    g = torch.Generator(device='cpu'); g.manual_seed(2)
    obs_b, obs_s = big.new_obs(torch.float16), small.new_obs(torch.float16)
    peak_small = 0
    for mv in range(moves):
        big.select(obs_b); small.select(obs_s)
        for s in range(sims):
            pol = torch.rand((B, A), generator=g) ** 3 + 1e-4                           # peaked priors: deep, reused subtrees
            pol = (pol / pol.sum(1, keepdim=True)).to(big.device)
            val = torch.rand((B, NV), generator=g) + 1e-3
            val = (val / val.sum(1, keepdim=True)).to(big.device)
            if s + 1 < sims:
                big.backup_select(pol, val, obs_b); small.backup_select(pol, val, obs_s)
                if s % 8 == 0:
                    assert torch.equal(obs_b, obs_s), (mv, s, small.counters())
            else:
                big.backup(pol, val); small.backup(pol, val)
        assert torch.equal(big.root_counts(), small.root_counts()), mv
        assert torch.equal(big.root_value(True), small.root_value(True))
        peak_small = max(peak_small, small.counters()['max_nodes_used'])
        big.advance(True); small.advance(True)
        assert torch.equal(big.last_actions(), small.last_actions())
    cb, cs = big.counters(), small.counters()
    assert cb['max_nodes_used'] > 4 * per_move >= peak_small                            # the big store really outgrew the small one
    for k in ('sims', 'expansions', 'games_played', 'num_results', 'num_examples'):
        assert cb[k] == cs[k], k
    assert cb['games_played'] > 0
    assert (big.tape_counters() == small.tape_counters()).all()
    for x, y in zip(big.examples(), small.examples()):
        assert torch.equal(x, y)
    for x, y in zip(big.results(), small.results()):
        assert (x == y).all()


def test_brandubh_4096_games_fit_one_gpu():
    """BASELINE config 3's WHOLE job (4096 games x 200 simulations) on one GPU: with reclaimed node stores the trees take
    4096 x 2 x 307 264 nodes x 32 B = 81 GB (16 moves' worth per semi-space; round 1: 82 MB per tree = 336 GB, did not fit).  A short
    run, properties only."""
    import torch
    from alphazero_general_amd.engine import DeviceEngine
    B, sims = 4096, 200
    eng = DeviceEngine(1, B, cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=1, sims_hint=sims,
                       example_capacity=B * 8, games_per_iteration=1 << 30)
    pol = torch.full((B, eng.A), 1.0 / eng.A, device=eng.device); val = torch.full((B, eng.NV), 1.0 / eng.NV, device=eng.device)
    obs = torch.zeros((B, 49, 8), dtype=torch.float16, device=eng.device)
    for mv in range(2):
        eng.select(obs)
        for s in range(24):
            eng.backup_select(pol, val, obs)
        eng.backup(pol, val)
        eng.advance(True)
    c = eng.counters()
    assert c['sims'] == 2 * 25 * B and 0 < c['max_nodes_used'] <= 16 * sims * 96 + 64
    assert all(t == 2 for (_, _, t) in eng.get_states(B - 4, 4))
    eng.close()


@pytest.mark.parametrize('game,sims,B', [(0, 100, 256), (1, 60, 48)])
def test_sharp_policy_whole_games_fit_the_default_node_store(game, sims, B):
    """A trained, sharp network carries most of a search's visits into the played move, so the subtree kept across moves grows to
    f / (1 - f) moves' worth of nodes (the reference keeps it on the Python heap, MCTS.pyx:185-195).  Synthetic evaluator: 0.95 of
    the prior mass on ONE action per slot at every leaf (renormalised over the legal moves), flat value -- whole games at the
    default nodes_per_tree must finish without AZG_E_TREE_FULL; how many moves' worth of nodes a compaction kept is recorded."""
    import json
    import os
    import torch
    from alphazero_general_amd import _abi
    from alphazero_general_amd.engine import DeviceEngine
    gi = _abi.game_info(game)
    A, NV = gi.action_size, gi.num_players + 1
    eng = DeviceEngine(game, B, cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=8, sims_hint=sims,
                       games_per_iteration=1 << 30, example_capacity=B * 4 * (gi.max_turns + 1) * gi.num_symmetries)
    g = torch.Generator(device='cpu'); g.manual_seed(4)
    val = torch.full((B, NV), 1.0 / NV, device=eng.device)
    obs = eng.new_obs(torch.float16)
    rounds = gi.max_turns + 8
    for mv in range(rounds):
        if mv % 6 == 0:                                                  # the favoured action moves now and then (it may be illegal)
            fav = torch.randint(0, A, (B,), generator=g)
            pol = torch.full((B, A), 0.05 / (A - 1))
            pol[torch.arange(B), fav] = 0.95
            pol = pol.to(eng.device)
        eng.select(obs)
        for s in range(sims):
            if s + 1 < sims:
                eng.backup_select(pol, val, obs)
            else:
                eng.backup(pol, val)
        eng.advance(True)
    c = eng.counters()                                                   # raises on a sticky AZG_E_TREE_FULL
    per_move = sims * gi.max_children
    assert c['sims'] == B * sims * rounds and c['games_played'] >= B // 2
    assert c['max_nodes_used'] <= eng.nodes_per_tree
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/nn_error.jsonl', 'a') as fh:
        fh.write(json.dumps({'test': 'sharp_policy_node_store', 'game': game, 'slots': B, 'sims': sims, 'rounds': rounds,
                             'nodes_per_tree': eng.nodes_per_tree, 'max_nodes_used': c['max_nodes_used'], 'max_nodes_kept': c['max_nodes_kept'],
                             'kept_in_moves_worth': round(c['max_nodes_kept'] / per_move, 3),
                             'used_in_moves_worth': round(c['max_nodes_used'] / per_move, 3)}) + '\n')
    eng.close()


def test_mcts_class_keeps_searching_at_one_root():
    """search() with far more simulations than args.numMCTSSims promised (the engine's reclamation reserve is sized from that hint):
    update_root then leaves less room than the next search needs, and find_leaf's forced compaction must reclaim the played
    moves' dead siblings in time -- same trees as an engine whose store never fills."""
    from alphazero_general_amd.MCTS import MCTS
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.utils import dotdict
    args = dotdict(root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1, fpu_reduction=0.2, cpuct=1.25, _num_players=3,
                   numMCTSSims=10, _azg_seed=77)
    small, big = MCTS(args), MCTS(args)
    g = Game()
    step = [0]

    def nn(obs):
        p, v = ol.fake_eval(77, 0, step[0], 7, 3)
        step[0] += 1
        return p, v
    small._ensure(g); big._ensure(g)
    small._engine.close()
    from alphazero_general_amd.engine import DeviceEngine
    small._engine = DeviceEngine(0, 1, cpuct=1.25, fpu_reduction=0.2, seed=77, sims_hint=10, nodes_per_tree=2000)  # < 3 moves' worth
    for mv in range(8):
        step[0] = 1000 * mv
        small.search(g, nn, 100, False, False)
        step[0] = 1000 * mv
        big.search(g, nn, 100, False, False)
        assert (small.counts(g) == big.counts(g)).all(), mv
        a = big.best_action(g)
        small.update_root(g, a); big.update_root(g, a)
        g.play_action(a)


@pytest.mark.parametrize('game,B,sims', [('brandubh', 203, 23), ('trimok', 131, 17), ('brandubh', 512, 200), ('brandubh', 1, 2),
                                           ('trimok', 3, 2), ('trimok', 1, 5),
                                           # above 512 games: tiles of several boards per workgroup (3 at 700 / 515 / 1300 -- a last tile with one and with
                                           # two games --, 2 at 1024, 4 at 1537 and at BASELINE config 3's 2-GPU shard)
                                           ('brandubh', 700, 23), ('brandubh', 515, 9), ('brandubh', 1024, 9), ('brandubh', 1300, 11),
                                           ('brandubh', 1537, 7), ('brandubh', 2048, 200), ('trimok', 513, 9), ('trimok', 1024, 50),
                                           # connect4 with the reference's default net (32 channels) and its 64-channel sibling: one and two games per workgroup
                                           ('connect4:32', 32, 25), ('connect4:32', 515, 9), ('connect4:32', 1, 3), ('connect4:64', 131, 17), ('connect4:64', 600, 9)])
def test_wide_search_launch_equals_phase_launches(game, B, sims):
    """azg_search_wide_f16 (networks with factorised heads: tree walk by two wavefronts per game, tower, head convolutions and the
    sparse heads all inside one persistent launch) against the launch-per-phase path -- azg_select / azg_backup_select_features,
    azg_resnet_tower_features_f16 -- on a twin engine: the logits are bit-identical by construction (the same dot products in the
    same order), so trees, moves, tape counters and samples must be identical.  The last case is BASELINE config 3's per-GPU
    size."""
    import importlib
    import torch
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.engine import DeviceEngine
    Game, na = _wide_game_net(game)
    torch.manual_seed(21)
    net = N.NNetWrapper(Game, na, device='cuda:0', dtype=torch.float16)
    net.refresh()
    hip = net._hip
    assert hip.fact_head and hip.can_search
    gid = Game.AZG_GAME_ID
    kw = dict(cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=4, games_per_iteration=1 << 30,
              example_capacity=B * 16 * 8, sims_hint=sims)
    ea, eb = DeviceEngine(gid, B, **kw), DeviceEngine(gid, B, **kw)
    hw = Game.observation_size()[1] * Game.observation_size()[2]
    obs = torch.zeros((B, hw, 8), dtype=torch.float16, device=ea.device)
    moves = 3 if B >= 512 else 14
    for move in range(moves):
        hip.search(ea, sims)
        eb.select(obs)
        for s in range(sims):
            eb.backup_select_features(hip.forward_features_nhwc8(obs), hip.head_rows, hip.head2_b, obs, select=s + 1 < sims)
        assert torch.equal(ea.root_counts(), eb.root_counts()), move
        assert torch.equal(ea.root_probs(1.0), eb.root_probs(1.0))
        assert torch.equal(ea.root_value(True), eb.root_value(True))
        ea.advance(True); eb.advance(True)
        assert torch.equal(ea.last_actions(), eb.last_actions())
        assert (ea.tape_counters() == eb.tape_counters()).all()
    a, b = ea.counters(), eb.counters()
    assert a == b and a['sims'] == B * sims * moves
    for x, y in zip(ea.examples(), eb.examples()):
        assert torch.equal(x, y)


@pytest.mark.parametrize('game,B,sims', [('brandubh', 203, 23), ('trimok', 131, 17), ('brandubh', 512, 200), ('brandubh', 1, 2), ('trimok', 3, 2),
                                           ('brandubh', 700, 23), ('brandubh', 1023, 9), ('brandubh', 1300, 11), ('brandubh', 1537, 7),
                                           ('brandubh', 2048, 200), ('trimok', 513, 9), ('trimok', 1024, 50),
                                           ('connect4:32', 32, 25), ('connect4:32', 515, 9), ('connect4:32', 1, 3), ('connect4:32', 2048, 100),
                                           ('connect4:64', 131, 17), ('connect4:64', 600, 9)])
def test_wide_exact_search_launch_equals_logits_phase_launches(game, B, sims):
    """azg_search_wide_exact_f16 -- the persistent launch that computes ALL A + P+1 logits of its boards itself (heads_full_lds: the
    fragments and summation order of k_heads_fact) and takes the softmax over all A, masks, renormalises -- against the launch-per-phase
    form of the same evaluation on a twin engine: azg_select / azg_backup_select_logits fed NNetWrapper.process's logits
    (azg_resnet_tower_features_f16 + azg_policy_value_heads_fact_f16).  That form is what tests/test_gpu_runner_oracle.py holds to the
    oracle fed NNetWrapper.process with frac == 1.0; the persistent launch must be bit-identical to it in every tile shape (one, two,
    three and four games per workgroup by engine size)."""
    import importlib
    import torch
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.engine import DeviceEngine
    Game, na = _wide_game_net(game)
    torch.manual_seed(22)
    net = N.NNetWrapper(Game, na, device='cuda:0', dtype=torch.float16)
    net.refresh()
    hip = net._hip
    assert hip.fact_head and hip.can_search
    gid = Game.AZG_GAME_ID
    kw = dict(cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=5, games_per_iteration=1 << 30,
              example_capacity=B * 16 * 8, sims_hint=sims)
    ea, eb = DeviceEngine(gid, B, **kw), DeviceEngine(gid, B, **kw)
    hw = Game.observation_size()[1] * Game.observation_size()[2]
    obs = torch.zeros((B, hw, 8), dtype=torch.float16, device=ea.device)
    moves = 3 if B >= 512 else 14
    for move in range(moves):
        hip.search(ea, sims, exact=True)
        eb.select(obs)
        for s in range(sims):
            eb.backup_select_logits(hip.forward_logits_nhwc8(obs), obs, select=s + 1 < sims)
        assert torch.equal(ea.root_counts(), eb.root_counts()), move
        assert torch.equal(ea.root_probs(1.0), eb.root_probs(1.0))
        assert torch.equal(ea.root_value(True), eb.root_value(True))
        ea.advance(True); eb.advance(True)
        assert torch.equal(ea.last_actions(), eb.last_actions())
        assert (ea.tape_counters() == eb.tape_counters()).all()
    a, b = ea.counters(), eb.counters()
    assert a == b and a['sims'] == B * sims * moves
    for x, y in zip(ea.examples(), eb.examples()):
        assert torch.equal(x, y)


def test_wide_search_tiles_with_a_deeper_tower():
    """the multi-game tiles keep the layers' parameters in LDS beside the games' scratch; two 4-game workgroups per CU leave room for a
    4-block tower's only.  A 6-block brandubh tower at 1600 games must still search -- on the 3-game tile -- and equal the
    launch-per-phase form, in both hand-overs."""
    import torch
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.engine import DeviceEngine
    from alphazero_general_amd.envs.brandubh import Game
    from alphazero_general_amd.utils import dotdict
    na = dotdict(dict(N.BRANDUBH_NET_ARGS)); na['depth'] = 6
    torch.manual_seed(31)
    net = N.NNetWrapper(Game, na, device='cuda:0', dtype=torch.float16)
    net.refresh()
    hip = net._hip
    assert hip.fact_head and hip.can_search and len(hip.blocks) == 6
    B, sims = 1600, 9
    kw = dict(cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=6, games_per_iteration=1 << 30, example_capacity=B * 8 * 4, sims_hint=sims)
    for exact in (True, False):
        ea, eb = DeviceEngine(1, B, **kw), DeviceEngine(1, B, **kw)
        obs = torch.zeros((B, 49, 8), dtype=torch.float16, device=ea.device)
        for move in range(2):
            hip.search(ea, sims, exact=exact)
            eb.select(obs)
            for s in range(sims):
                if exact:
                    eb.backup_select_logits(hip.forward_logits_nhwc8(obs), obs, select=s + 1 < sims)
                else:
                    eb.backup_select_features(hip.forward_features_nhwc8(obs), hip.head_rows, hip.head2_b, obs, select=s + 1 < sims)
            assert torch.equal(ea.root_counts(), eb.root_counts()), (exact, move)
            ea.advance(True); eb.advance(True)
            assert torch.equal(ea.last_actions(), eb.last_actions())
        assert ea.counters() == eb.counters()
        ea.close(); eb.close()


@pytest.mark.parametrize('game,B', [('brandubh', 96), ('trimok', 64)])
def test_sparse_heads_equal_full_heads_on_the_valid_actions(game, B):
    """azg_backup_select_features computes, inside the tree launch, only the logits process_results uses: the value logits and
    the policy logits of the leaf's valid actions.  Against the full-width path (azg_policy_value_heads_fact_f16 -> all A logits ->
    azg_backup_select_logits: softmax over all A, mask, renormalise) on a twin engine fed the SAME leaves, the priors written
    to the new children must agree to rounding (the full softmax's normaliser cancels in the renormalisation) and every value
    backed up likewise; recorded beside the network errors in gpurun_out/nn_error.jsonl."""
    import importlib
    import json
    import os
    import torch
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.engine import DeviceEngine
    Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
    torch.manual_seed(33)
    net = N.NNetWrapper(Game, N.BRANDUBH_NET_ARGS if game == 'brandubh' else N.DEFAULT_NET_ARGS, device='cuda:0', dtype=torch.float16)
    net.refresh()
    hip = net._hip
    assert hip.fact_head
    kw = dict(cpuct=1.25, fpu_reduction=0.2, add_root_noise=False, add_root_temp=False, seed=9, games_per_iteration=1 << 30, sims_hint=8)
    ea, eb = DeviceEngine(Game.AZG_GAME_ID, B, **kw), DeviceEngine(Game.AZG_GAME_ID, B, **kw)
    hw = Game.observation_size()[1] * Game.observation_size()[2]
    oa = torch.zeros((B, hw, 8), dtype=torch.float16, device=ea.device); ob = torch.zeros_like(oa)
    worst_p = worst_q = 0.0
    same = np.ones(B, bool)              # slots whose two trees still see the same leaves (a prior that differs in the last bit can
    for move in range(6):                #  flip a PUCT near-tie; such a slot is dropped from the comparison from then on)
        ea.select(oa); eb.select(ob)
        for s in range(8):
            same &= (oa == ob).reshape(B, -1).all(1).cpu().numpy()
            ea.backup_select_features(hip.forward_features_nhwc8(oa), hip.head_rows, hip.head2_b, oa, select=s < 7)
            eb.backup_select_logits(hip.forward_logits_nhwc8(ob), ob, select=s < 7)
            for slot in np.nonzero(same)[0][::5]:
                ka, kb = ea.root_children(int(slot)), eb.root_children(int(slot))
                if not ((ka['a'] == kb['a']).all() and (ka['n'] == kb['n']).all()):
                    same[slot] = False
                    continue
                worst_p = max(worst_p, float(np.abs(ka['p'] - kb['p']).max())); worst_q = max(worst_q, float(np.abs(ka['q'] - kb['q']).max()))
        same &= (ea.root_counts() == eb.root_counts()).all(1).cpu().numpy()
        ea.advance(True); eb.advance(True)
        same &= (ea.last_actions() == eb.last_actions()).cpu().numpy()
    assert same.mean() > 0.9, same.mean()
    assert worst_p < 2e-6 and worst_q < 2e-5, (worst_p, worst_q)
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/nn_error.jsonl', 'a') as fh:
        fh.write(json.dumps({'test': 'sparse_vs_full_heads_' + game, 'boards': B, 'max_abs_prior': worst_p, 'max_abs_q': worst_q,
                             'slots_never_diverged': float(same.mean()), 'tolerance': 2e-6}) + '\n')
