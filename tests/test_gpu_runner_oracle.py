"""The native self-play runner (selfplay.SelfPlayRunner: whole rounds replayed as one hipGraph) against the CPU oracle, both fed by
the SAME real network, for every hand-over form between the network and the tree launch that a bench line or a runner uses:

  probs     azg_backup_select            (k_backup_select2<G, IN_PROBS>: probabilities from the network launch)
  logits    azg_backup_select_logits     (k_backup_select2<G, IN_LOGITS>: both softmaxes inside the tree launch -- the same
                                          heads_softmax_row arithmetic as the network's own softmax launch, so bit-identical)
  features  azg_backup_select_features   (sparse heads: logits of the valid actions only, equal to rounding -- see below)
  search    azg_search_wide_f16 / azg_search_f16 (one persistent launch per move)

The oracle (oracle/azg_mcts_ref.c, pinned to the reference's goldens) plays SelfPlayAgent.generateBatch / processBatch / playMoves
(SelfPlayAgent.pyx:103-202, MCTS.pyx:208-289) with the probabilities NNetWrapper.process returns for ITS leaf observations.  For
probs / logits every action, sample, result and counter must be identical until every slot has finished a game and restarted.
The sparse-heads forms (features, wide search) evaluate a leaf as softmax over its VALID actions' logits (fp32 dot products
instead of the MFMA chains of the full-width heads), which equals NNetWrapper.process followed by the mask + renormalisation
of MCTS.pyx:239-245 only to rounding (~1e-8 on a prior).  Two tests therefore:
  * test_sparse_heads_launches_vs_oracle_bit_exact: the oracle is fed the sparse evaluation itself (azg_leaf_heads_sparse_f16 +
    azg_heads_softmax, the same arithmetic as its own launch) -- then azg_search_wide_f16, azg_backup_select_features and the
    oracle must agree bit for bit, which pins the TREE side of the timed launches of configs 3 and 5;
  * test_runner_timed_launches_vs_oracle_with_real_net: the oracle is fed NNetWrapper.process; every slot is followed until its
    first differing move and the fraction that never diverged is recorded -- a statement about the NETWORK rounding (a 1e-8
    prior difference flips a PUCT near-tie about once per 2e5 simulations), bounded, not required to be zero."""
import importlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _setup(game, seed_net):
    import torch
    from alphazero_general_amd import nnet as N
    name, _, width = game.partition(':')                              # 'connect4:32' = connect4 with the reference's default net (config 1)
    Game = importlib.import_module('alphazero_general_amd.envs.' + name).Game
    net_args = {'connect4': N.CONNECT4_NET_ARGS, 'brandubh': N.BRANDUBH_NET_ARGS, 'trimok': N.DEFAULT_NET_ARGS}[name]
    if width:
        net_args = N.dotdict(dict(N.DEFAULT_NET_ARGS, num_channels=int(width)))
    torch.manual_seed(seed_net)
    net = N.NNetWrapper(Game, net_args, device='cuda:0', dtype=torch.float16)
    net.refresh()
    assert net._hip is not None
    return Game, net


def _args(sims, games, **kw):
    from alphazero_general_amd.utils import dotdict, default_temp_scaling
    a = dotdict(numMCTSSims=sims, numFastSims=20, probFastSim=0.0, gamesPerIteration=games, cpuct=1.25, fpu_reduction=0.2,
                root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1.0, add_root_noise=True, add_root_temp=True,
                symmetricSamples=True, mctsResetThreshold=0, startTemp=1.0, arenaTemp=0.25, temp_scaling_fn=default_temp_scaling)
    a.update(kw)
    return a


def _oracle_round(ag, net, sims):
    import torch
    ag.begin_round()
    for _ in range(sims):
        oobs, _, _ = ag.generate_batch()
        p, v = net.process(torch.from_numpy(oobs))
        ag.process_batch(p.cpu().numpy(), v.cpu().numpy())
    ag.play_moves()


@pytest.mark.parametrize('game,heads,B,sims', [
    ('brandubh', 'probs', 48, 24), ('brandubh', 'logits', 48, 40),
    ('trimok', 'probs', 64, 16), ('trimok', 'logits', 64, 24),
    ('connect4', 'probs', 64, 24),
])
def test_runner_phase_launches_vs_oracle_with_real_net(game, heads, B, sims):
    """bit-exact forms: actions every round, then samples (x symmetries), results and counters, until every slot has restarted."""
    from alphazero_general_amd.selfplay import SelfPlayRunner
    Game, net = _setup(game, 5)
    seed, games = 23, 3 * B
    kw = dict(cpuct=4.0, fpu_reduction=0.4) if game == 'connect4' else {}
    r = SelfPlayRunner(Game, net, _args(sims, games, **kw), num_slots=B, seed=seed, fused_search=False, heads=heads)
    assert r.round_graph and not r.fused_search
    ag = ol.OAgent(Game.AZG_GAME_ID, B, sims=sims, games_per_iteration=games, seed=seed, add_root_noise=True, add_root_temp=True,
                   cpuct=kw.get('cpuct', 1.25), fpu_reduction=kw.get('fpu_reduction', 0.2))
    rounds = 0
    while ag.games_played < games and len(set(ag.results()[2].tolist())) < B:
        _oracle_round(ag, net, sims)
        r.play_round()
        assert (r.engine.last_actions().cpu().numpy() == ag.last_actions()).all(), rounds
        rounds += 1
    c = r.engine.counters()
    assert c['games_played'] == ag.games_played and c['sims'] == ag.sims_done and c['expansions'] == ag.expansions
    assert len(set(ag.results()[2].tolist())) == B or ag.games_played >= games
    oo, op, oz = ag.samples()
    eo, ep, ez = [t.cpu().numpy() for t in r.engine.examples()]
    assert eo.shape == oo.shape and eo.shape[0] > 0
    assert (eo == oo).all() and (ep == op).all() and (ez == oz).all()
    for x, y in zip(r.engine.results(), ag.results()):
        assert (x == y).all()


@pytest.mark.parametrize('game,form,B,sims,rounds', [
    ('brandubh', 'search', 48, 200, 60), ('brandubh', 'features', 48, 40, 110),
    ('trimok', 'search', 64, 50, 60), ('trimok', 'features', 64, 24, 60),
    ('connect4', 'search', 64, 100, 50),
    ('brandubh', 'exact', 48, 200, 60), ('trimok', 'exact', 64, 50, 60),
    ('connect4:32', 'exact', 32, 25, 50),                        # BASELINE config 1: the default net's persistent launch (round 6)
])
def test_runner_timed_launches_vs_oracle_with_real_net(game, form, B, sims, rounds):
    """The launches bench.py times (--workload brandubh | trimok: azg_search_wide_f16 with sparse heads at 200 / 50 simulations per
    move; connect4: azg_search_f16) and the launch-per-phase sparse-heads form, against the oracle fed by NNetWrapper.process
    (full-width heads: softmax over all A, mask, renormalise -- what the reference computes, MCTS.pyx:239-245).  connect4's fused
    heads hand over exact probabilities: bit-identical, asserted.  The sparse heads agree to rounding only: a slot is followed
    until its first differing move; the fraction that never diverged is recorded (gpurun_out/nn_error.jsonl) and must stay
    at or above 0.95 (observed: 47 of 48 brandubh slots after 60 moves x 200 simulations = 0.979, all slots in the other cases).
    form 'exact' = azg_search_wide_exact_f16, the runners' and bench.py's default for configs 3 and 5 since round 5 (all A + P+1 logits
    inside the persistent launch): like connect4, no slot may ever diverge."""
    import torch
    from alphazero_general_amd.selfplay import SelfPlayRunner
    Game, net = _setup(game, 7)
    seed, games = 29, 1 << 30
    kw = dict(cpuct=4.0, fpu_reduction=0.4) if game == 'connect4' else {}
    gi = ol.game_info(Game.AZG_GAME_ID)
    cap = B * (rounds + 1) * gi.num_symmetries
    r = SelfPlayRunner(Game, net, _args(sims, games, **kw), num_slots=B, seed=seed, example_capacity=cap,
                       fused_search=(form in ('search', 'exact')), heads=(None if form in ('search', 'exact') else form),
                       search_heads='exact' if form == 'exact' else 'sparse')
    assert r.fused_search == (form in ('search', 'exact'))
    ag = ol.OAgent(Game.AZG_GAME_ID, B, sims=sims, games_per_iteration=games, seed=seed, add_root_noise=True, add_root_temp=True,
                   cpuct=kw.get('cpuct', 1.25), fpu_reduction=kw.get('fpu_reduction', 0.2))
    same = np.ones(B, bool)
    first_div = None
    A = gi.action_size
    for rnd in range(rounds):
        ag.begin_round()
        for _ in range(sims):
            oobs, _, _ = ag.generate_batch()
            p, v = net.process(torch.from_numpy(oobs))
            ag.process_batch(p.cpu().numpy(), v.cpu().numpy())
        # one round of the runner WITHOUT its advance: the root statistics are compared before the move is played
        e = r.engine
        if form in ('search', 'exact'):
            net._hip.search(e, sims, exact=(form == 'exact'))
        else:
            e.select(r.lanes[0].obs)
            for i in range(sims):
                feat, rows, hb = r.lanes[0].net.run_features()
                e.backup_select_features(feat, rows, hb, r.lanes[0].obs, select=i + 1 < sims)
        cnt = e.root_counts().cpu().numpy()
        ocnt = np.zeros((B, A), np.int32)
        for i in range(B):
            ch = ag.root_children(i)
            ocnt[i, ch['a']] = ch['n']
        same &= (cnt == ocnt).all(1)
        ag.play_moves()
        e.advance(True)
        same &= e.last_actions().cpu().numpy() == ag.last_actions()
        if first_div is None and not same.all():
            first_div = rnd
    frac = float(same.mean())
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/nn_error.jsonl', 'a') as fh:
        fh.write(json.dumps({'test': 'runner_%s_vs_oracle_%s' % (form, game), 'slots': B, 'sims': sims, 'rounds': rounds,
                             'slots_never_diverged': frac, 'first_divergence_round': first_div,
                             'games_finished': int(ag.games_played)}) + '\n')
    assert ag.games_played > 0
    if game == 'connect4' or form == 'exact':
        assert frac == 1.0, (frac, first_div)
    else:
        assert frac >= 0.95, (frac, first_div)
        if frac < 1.0:
            return                                              # (diverged slots play different games: the totals below differ)
    c = r.engine.counters()
    assert c['games_played'] == ag.games_played and c['sims'] == ag.sims_done and c['expansions'] == ag.expansions
    oo, op, oz = ag.samples()
    eo, ep, ez = [t.cpu().numpy() for t in r.engine.examples()]
    assert eo.shape == oo.shape and (eo == oo).all() and (ez == oz).all()
    assert (ep == op).all()                                     # pi(T=1) is counts / sum: identical counts, identical pi


@pytest.mark.parametrize('game,B,sims', [('brandubh', 48, 40), ('trimok', 64, 24)])
def test_sparse_heads_launches_vs_oracle_bit_exact(game, B, sims):
    """Three engines and the oracle on the same seed and network, every round until every slot has restarted:
      ea  azg_search_wide_f16                (the launch --workload brandubh | trimok times: tree, tower, sparse heads in one launch)
      eb  azg_backup_select_features         (the same per phase: tower launch + two-wave tree launch with the sparse heads inside)
      ec  azg_select, tower, azg_leaf_heads_sparse_f16, azg_heads_softmax, azg_backup  (every stage its own launch)
      oracle  generateBatch / processBatch / playMoves fed the probabilities ec's softmax launch produced for the same leaves.
    Leaf observations (ec vs oracle), visit counts, actions, tape counters, samples, results, counters: identical."""
    import torch
    from alphazero_general_amd.engine import DeviceEngine
    Game, net = _setup(game, 9)
    hip = net._hip
    assert hip.fact_head and hip.can_search
    gid, seed, games = Game.AZG_GAME_ID, 31, 3 * B
    gi = ol.game_info(gid)
    kw = dict(cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=seed, games_per_iteration=games,
              example_capacity=4 * B * (gi.max_turns + 1) * gi.num_symmetries, sims_hint=sims)
    ea, eb, ec = DeviceEngine(gid, B, **kw), DeviceEngine(gid, B, **kw), DeviceEngine(gid, B, **kw)
    ag = ol.OAgent(gid, B, sims=sims, games_per_iteration=games, seed=seed, add_root_noise=True, add_root_temp=True)
    hw = gi.obs_h * gi.obs_w
    ob = torch.zeros((B, hw, 8), dtype=torch.float16, device=ea.device)
    oc = ec.new_obs(torch.float32)                                # f32 planes: compared with the oracle's, then fed to the network
    rounds = 0
    while ag.games_played < games and len(set(ag.results()[2].tolist())) < B:
        hip.search(ea, sims)
        eb.select(ob)
        ag.begin_round()
        for s in range(sims):
            eb.backup_select_features(hip.forward_features_nhwc8(ob, key=1), hip.head_rows, hip.head2_b, ob, select=s + 1 < sims)
            oobs, _, _ = ag.generate_batch()
            ec.select(oc)
            if s % 5 == 0:
                assert (oc.cpu().numpy() == oobs).all(), (rounds, s)
            lg = ec.leaf_heads_sparse(hip.forward_features_nhwc8(hip.to_nhwc8(oc), key=2), hip.head_rows, hip.head2_b)
            pol, val = ec.heads_softmax(lg)
            ec.backup(pol, val)
            ag.process_batch(pol.cpu().numpy(), val.cpu().numpy())
        cnt = ea.root_counts()
        assert torch.equal(cnt, eb.root_counts()) and torch.equal(cnt, ec.root_counts()), rounds
        assert torch.equal(ea.root_probs(1.0), ec.root_probs(1.0)) and torch.equal(ea.root_value(True), ec.root_value(True))
        ag.play_moves()
        for e in (ea, eb, ec):
            e.advance(True)
            assert (e.last_actions().cpu().numpy() == ag.last_actions()).all(), rounds
        assert (ea.tape_counters() == ec.tape_counters()).all() and (eb.tape_counters() == ec.tape_counters()).all()
        rounds += 1
    c = ea.counters()
    assert c == eb.counters() and c == ec.counters()
    assert c['games_played'] == ag.games_played and c['sims'] == ag.sims_done and c['expansions'] == ag.expansions and c['games_played'] > 0
    oo, op, oz = ag.samples()
    for e in (ea, eb, ec):
        eo, ep, ez = [t.cpu().numpy() for t in e.examples()]
        assert eo.shape == oo.shape and (eo == oo).all() and (ep == op).all() and (ez == oz).all()
        for x, y in zip(e.results(), ag.results()):
            assert (x == y).all()


def test_sparse_heads_with_more_than_64_children_vs_oracle():
    """The same three launch forms from roots with 65-70 legal moves (ol.br_wide_positions: playouts never get there): the sparse heads'
    MFMA passes over five subtiles of 16 children, the shuffle without the pre-computed 64-key masks, two-chunk priors -- one move's
    simulations against the oracle's MCTS fed the probabilities of the split launches, every engine against every other."""
    import torch
    from alphazero_general_amd.engine import DeviceEngine
    Game, net = _setup('brandubh', 9)
    hip = net._hip
    gid, seed, M, sims = Game.AZG_GAME_ID, 41, 24, 64
    pos = ol.br_wide_positions(M, 11)
    ks = [int(g.valid_moves().sum()) for g in pos]
    assert min(ks) > 64
    kw = dict(cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=seed, sims_hint=sims)
    ea, eb, ec = DeviceEngine(gid, M, **kw), DeviceEngine(gid, M, **kw), DeviceEngine(gid, M, **kw)
    oms = [ol.OMCTS(gid, seed=seed, stream=r) for r in range(M)]
    for e in (ea, eb, ec):
        e.set_states([(g.cells(), g.player, g.turns, g.s.aux[0]) for g in pos])
    ob = torch.zeros((M, 49, 8), dtype=torch.float16, device=ea.device)
    oc = ec.new_obs(torch.float32)
    hip.search(ea, sims)
    eb.select(ob)
    for s in range(sims):
        eb.backup_select_features(hip.forward_features_nhwc8(ob, key=1), hip.head_rows, hip.head2_b, ob, select=s + 1 < sims)
        ec.select(oc)
        o = oc.cpu().numpy()
        lg = ec.leaf_heads_sparse(hip.forward_features_nhwc8(hip.to_nhwc8(oc), key=2), hip.head_rows, hip.head2_b)
        pol, val = ec.heads_softmax(lg)
        ec.backup(pol, val)
        pn, vn = pol.cpu().numpy(), val.cpu().numpy()
        for r in range(M):
            leaf, _ = oms[r].find_leaf(pos[r])
            assert (ec.last_path(r) == oms[r].last_path()).all() and (o[r] == leaf.observation().reshape(o[r].shape)).all(), (r, s)
            oms[r].process_results(vn[r], pn[r], noise=True, temp=True)
    for r in range(M):
        och = oms[r].root_children()
        assert len(och['a']) == ks[r]
        for e in (ea, eb, ec):
            ch = e.root_children(r)
            for f in ('a', 'n', 'q', 'p', 'v'):
                assert (ch[f] == och[f]).all(), (f, r)
    assert (ea.tape_counters() == ec.tape_counters()).all() and (eb.tape_counters() == ec.tape_counters()).all()
    c = ea.counters()
    assert c == eb.counters() and c == ec.counters() and c['sims'] == M * sims
    for e in (ea, eb, ec):
        e.close()


@pytest.mark.parametrize('game', ['connect4', 'brandubh', 'trimok'])
def test_persistent_launches_at_finished_and_almost_finished_roots(game):
    """find_leaf at a root whose game is over returns the root itself (MCTS.pyx:213) and process_results backs the win state up
    (:234-235) -- a simulation without an evaluation; one or two plies earlier most leaves are terminal.  The persistent launches
    (azg_search_f16 / azg_search_wide_f16: the tower runs for the workgroup whether or not its game's leaf takes the result) against
    the launch-per-phase form from such roots, and the visit count of a finished root against the oracle's."""
    import torch
    from alphazero_general_amd.engine import DeviceEngine
    Game, net = _setup(game, 9)
    hip = net._hip
    gid, sims = Game.AZG_GAME_ID, 16
    gi = ol.game_info(gid)
    rng = np.random.RandomState(3)
    pos = []
    while len(pos) < 24:
        g = ol.OGame(gid); hist = [g.clone()]
        while not g.win_state().any():
            g.play(int(rng.choice(np.flatnonzero(g.valid_moves())))); hist.append(g.clone())
        pos += [hist[-1], hist[-2], hist[max(0, len(hist) - 3)]]
    B = len(pos)
    st = [(g.cells(), g.player, g.turns, g.s.aux[0]) if game == 'brandubh' else (g.cells(), g.player, g.turns) for g in pos]
    ea, eb = DeviceEngine(gid, B, seed=1, sims_hint=sims), DeviceEngine(gid, B, seed=1, sims_hint=sims)
    ea.set_states(st); eb.set_states(st)
    hip.search(ea, sims)
    if hip.fact_head:
        ob = torch.zeros((B, gi.obs_h * gi.obs_w, 8), dtype=torch.float16, device=ea.device)
        eb.select(ob)
        for s in range(sims):
            eb.backup_select_features(hip.forward_features_nhwc8(ob, key=1), hip.head_rows, hip.head2_b, ob, select=s + 1 < sims)
    else:
        obs = eb.new_obs()
        for s in range(sims):
            eb.select(obs)
            eb.backup(*net.process(obs))
    assert torch.equal(ea.root_counts(), eb.root_counts()) and torch.equal(ea.root_value(True), eb.root_value(True))
    pa, pb = ea.root_probs(1.0), eb.root_probs(1.0)                # (a finished root has no children: counts / 0, never asked for by playMoves)
    assert torch.equal(torch.isnan(pa), torch.isnan(pb)) and torch.equal(torch.nan_to_num(pa), torch.nan_to_num(pb))
    assert (ea.tape_counters() == eb.tape_counters()).all()
    c = ea.counters()
    assert c == eb.counters() and c['sims'] == B * sims
    uni_p, uni_v = np.ones(gi.action_size, np.float32), np.full(gi.num_players + gi.has_draw, 1.0, np.float32)
    for r, g in enumerate(pos):
        if g.win_state().any():
            m = ol.OMCTS(gid, seed=1, stream=r)
            for s in range(sims):
                leaf, exp = m.find_leaf(g)
                m.process_results(uni_v, uni_p)
            ti = ea.tree_info(r)
            assert ti['n'] == m.root_n == sims and ti['nodes_used'] == eb.tree_info(r)['nodes_used'] and ti['max_depth'] == 0, r
    ea.close(); eb.close()



@pytest.mark.parametrize('B', [300, 600, 1024, 1600, 2100])    # 1, 3, 2, 4, 3 games per workgroup
def test_wide_tiles_from_wide_finished_and_almost_finished_roots(B):
    """Every tile shape of the persistent wide-head launches (1, 2, 3 and 4 brandubh games per workgroup by engine size; the last two walk
    a game's tree with ONE wavefront on the compact LDS scratch) from the roots that take the rare paths: 65-70 legal moves (two lane
    chunks in the move list, the shuffle, best_child, the priors' numpy-order sum; five 16-child subtiles in the sparse heads), games that
    are over (find_leaf returns the root, the win state is backed up, MCTS.pyx:213,234-235) and one or two plies before the end.  Both
    hand-overs (exact: all logits inside the launch; sparse) against their launch-per-phase twins: counts, values, pi, tape counters,
    counters identical."""
    import torch
    from alphazero_general_amd.engine import DeviceEngine
    Game, net = _setup('brandubh', 9)
    hip = net._hip
    gid, sims = Game.AZG_GAME_ID, 10
    rng = np.random.RandomState(5)
    pos = list(ol.br_wide_positions(16, 11))
    assert min(int(g.valid_moves().sum()) for g in pos) > 64
    while len(pos) < 64:
        g = ol.OGame(gid); hist = [g.clone()]
        while not g.win_state().any():
            g.play(int(rng.choice(np.flatnonzero(g.valid_moves())))); hist.append(g.clone())
        pos += [hist[-1], hist[-2], hist[max(0, len(hist) - 3)], hist[len(hist) // 2]]
    st = [(g.cells(), g.player, g.turns, g.s.aux[0]) for g in pos]
    st = [st[(i * 7) % len(st)] for i in range(B)]                # (mixed within every tile)
    kw = dict(cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=3, sims_hint=sims)
    for exact in (True, False):
        ea, eb = DeviceEngine(gid, B, **kw), DeviceEngine(gid, B, **kw)
        ea.set_states(st); eb.set_states(st)
        hip.search(ea, sims, exact=exact)
        ob = torch.zeros((B, 49, 8), dtype=torch.float16, device=ea.device)
        eb.select(ob)
        for s in range(sims):
            if exact:
                eb.backup_select_logits(hip.forward_logits_nhwc8(ob, key=1), ob, select=s + 1 < sims)
            else:
                eb.backup_select_features(hip.forward_features_nhwc8(ob, key=1), hip.head_rows, hip.head2_b, ob, select=s + 1 < sims)
        assert torch.equal(ea.root_counts(), eb.root_counts()), exact
        assert torch.equal(ea.root_value(True), eb.root_value(True)) and torch.equal(ea.root_value(False), eb.root_value(False))
        pa, pb = ea.root_probs(1.0), eb.root_probs(1.0)
        assert torch.equal(torch.isnan(pa), torch.isnan(pb)) and torch.equal(torch.nan_to_num(pa), torch.nan_to_num(pb))
        assert (ea.tape_counters() == eb.tape_counters()).all()
        for r in range(0, B, 97):
            ca, cb = ea.root_children(r), eb.root_children(r)
            for f in ('a', 'n', 'q', 'p', 'v'):
                assert (ca[f] == cb[f]).all(), (exact, f, r)
        c = ea.counters()
        assert c == eb.counters() and c['sims'] == B * sims
        ea.close(); eb.close()
