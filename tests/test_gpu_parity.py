"""GPU parity tests (pytest -m gpu): the HIP engine, called through the C ABI, against
  (1) the committed golden vectors produced by the actual reference (tests/golden/*.npz), and
  (2) the CPU oracle run live on the same seeded inputs at sizes the oracle finishes in seconds,
plus size-independent properties at BASELINE.json's full size (2048 games x 100 sims).
Bars: visit counts, actions, paths, pi(T=1), samples bit-exact; q/v bit-exact in the exact tier (no root
temperature), atol 1e-5 otherwise; pi(T not in {1, .5, 0}) within rtol 3e-7 (powf tier)."""
import os
import zlib

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
C4 = 0


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@pytest.fixture(scope='module')
def torch_mod():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch


def engine(**kw):
    from alphazero_general_amd.engine import DeviceEngine
    game = kw.pop('game', C4)
    B = kw.pop('B')
    return DeviceEngine(game, B, **kw)


def ostate(g):
    return (g.cells(), g.player, g.turns)


def fake_batch(torch, seed, slots, step, A, NV, dev):
    pol = np.zeros((len(slots), A), np.float32); val = np.zeros((len(slots), NV), np.float32)
    for r, s in enumerate(slots):
        pol[r], val[r] = ol.fake_eval(seed, int(s), step, A, NV)
    return torch.from_numpy(pol).to(dev), torch.from_numpy(val).to(dev)


# ------------------------------------------------------------------------------------------------- tree goldens
@pytest.mark.parametrize('cname', ['default', 'c4train', 'noise', 'noise_temp'])
def test_c4_tree_vs_reference_goldens(torch_mod, cname):
    torch = torch_mod
    d = dict(np.load(os.path.join(G, 'c4_tree.npz')))              # (NpzFile decompresses an array on EVERY d[key])
    cpuct, fpu, noise, temp, sims = d[cname + '_cfg']
    noise, temp, sims = bool(noise), bool(temp), int(sims)
    seed = int(d[cname + '_seed'])
    R = d['prefix'].shape[0]
    A, NV = 7, 3
    exact = not temp
    eng = engine(B=R, cpuct=cpuct, fpu_reduction=fpu, add_root_noise=noise, add_root_temp=temp, seed=seed, sims_hint=sims)
    states = []
    for r in range(R):
        g = ol.OGame(C4)
        for a in d['prefix'][r]:
            if a >= 0:
                g.play(a)
        states.append(ostate(g))
    eng.set_states(states)
    obs = eng.new_obs()
    for s in range(sims):
        eng.select(obs)
        for r in range(0, R, 7):                      # spot-check leaf paths every sim on a subset of roots
            path = eng.last_path(r)
            assert len(path) == d[cname + '_depth'][r, s]
            assert (path[:24] == d[cname + '_paths'][r, s][:len(path)]).all(), (r, s)
        pol, val = fake_batch(torch, seed, range(R), s, A, NV, eng.device)
        eng.backup(pol, val)
        cnt = eng.root_counts().cpu().numpy()
        assert (cnt == d[cname + '_rootn'][:, s]).all(), s
    for r in range(R):
        ch = eng.root_children(r)
        k = len(ch['a'])
        assert (ch['a'] == d[cname + '_a'][r][:k]).all() and (d[cname + '_a'][r][k:] == -1).all()
        assert (ch['n'] == d[cname + '_n'][r][:k]).all()
        for f in ('q', 'p', 'v'):
            if exact:
                assert (ch[f] == d[cname + '_' + f][r][:k]).all(), (f, r)
            else:
                assert np.allclose(ch[f], d[cname + '_' + f][r][:k], atol=1e-5), (f, r)
        info = eng.tree_info(r)
        assert info['n'] == d[cname + '_root_n'][r] and info['max_depth'] == d[cname + '_maxdepth'][r]
    assert (eng.root_counts().cpu().numpy() == d[cname + '_counts']).all()
    for ti, t in enumerate(d['prob_temps']):
        pr = eng.root_probs(float(t)).cpu().numpy()
        ref = d[cname + '_probs'][:, ti]
        if t in (1.0, 0.5, 0.0):
            assert (pr == ref).all(), t
        else:
            assert np.allclose(pr, ref, rtol=3e-7, atol=1e-12), t
    assert (eng.root_value(False).cpu().numpy() == d[cname + '_vmax']).all()
    assert (eng.root_value(True).cpu().numpy() == d[cname + '_vavg']).all()
    assert (eng.tape_counters() == d[cname + '_ctr']).all()
    eng.counters()
    eng.close()


def test_c4_search_under_numpys_mt19937_seed(torch_mod):
    """"Identical seeds", literally: tests/golden/c4_mt19937.npz was produced by the reference's MCTS.search (MCTS.pyx:165-173) under
    np.random.seed(s) on numpy's own MT19937 stream, with every np.random.shuffle of Node.add_children (:76-79) observed (test_oracle_golden.py
    re-derives the recorded permutations from the seed alone).  The engine replays the recorded shuffles (azg_set_shuffle_tape) and must
    reproduce the reference's trees: root children in list order with a / n / q / p / v, counts, pi at T = 1 and T = 0, values, depth."""
    torch = torch_mod
    d = dict(np.load(os.path.join(G, 'c4_mt19937.npz')))
    R, sims, eseed = d['prefix'].shape[0], int(d['sims']), int(d['eval_seed'])
    A, NV = 7, 3
    eng = engine(B=R, cpuct=float(d['cfg'][0]), fpu_reduction=float(d['cfg'][1]), add_root_noise=False, add_root_temp=False, seed=12345, sims_hint=sims)
    states = []
    for r in range(R):
        g = ol.OGame(C4)
        for a in d['prefix'][r]:
            if a >= 0:
                g.play(a)
        states.append(ostate(g))
    eng.set_states(states)
    eng.set_shuffle_tape(d['ranks'])
    obs = eng.new_obs()
    for launch in ('phase', 'fused'):
        if launch == 'fused':                              # the two-wavefront launch (backup k + select k + 1) replays the same tape
            eng.set_states(states); eng.set_tape_counters([0] * R)
        if launch == 'phase':
            for s in range(sims):
                eng.select(obs)
                pol, val = fake_batch(torch, eseed, range(R), s, A, NV, eng.device)
                eng.backup(pol, val)
        else:
            eng.select(obs)
            for s in range(sims):
                pol, val = fake_batch(torch, eseed, range(R), s, A, NV, eng.device)
                if s + 1 < sims:
                    eng.backup_select(pol, val, obs)
                else:
                    eng.backup(pol, val)
        for r in range(R):
            ch = eng.root_children(r)
            k = len(ch['a'])
            assert (ch['a'] == d['a'][r][:k]).all() and (d['a'][r][k:] == -1).all(), (launch, r)
            assert (ch['n'] == d['n'][r][:k]).all(), (launch, r)
            for f in ('q', 'p', 'v'):
                assert (ch[f] == d[f][r][:k]).all(), (launch, f, r)
            info = eng.tree_info(r)
            assert info['n'] == d['root_n'][r] and info['max_depth'] == d['maxdepth'][r]
        assert (eng.root_counts().cpu().numpy() == d['counts']).all()
        assert (eng.root_probs(1.0).cpu().numpy() == d['probs1']).all() and (eng.root_probs(0.0).cpu().numpy() == d['probs0']).all()
        assert (eng.root_value(False).cpu().numpy() == d['vmax']).all() and (eng.root_value(True).cpu().numpy() == d['vavg']).all()
        used = (d['expansion_children'] > 0).sum(1)
        assert (eng.tape_counters() == d['expansion_children'].sum(1)).all() and used.min() > 0
    eng.set_shuffle_tape(None)                             # back to the counter-based tape
    eng.set_states(states); eng.set_tape_counters([0] * R)
    eng.select(obs)
    eng.counters()
    eng.close()


@pytest.mark.parametrize('name', ['c4', 'br'])
@pytest.mark.parametrize('launch', ['phase', 'fused'])
def test_c4_selfplay_agent_under_numpys_mt19937_seed(torch_mod, launch, name):
    """A whole SelfPlayAgent under np.random.seed(s), move for move: tests/golden/c4_mt19937_agent.npz is the REFERENCE's
    agent -- 4 concurrent connect4 games, root noise and root temperature on, 6 games -- on numpy's own MT19937 stream with shuffle /
    dirichlet / choice observed per game slot (tests/test_oracle_golden.py re-derives every draw from the seed alone).  The engine replays
    all three (azg_set_random_tape) and must reproduce: visit counts and the sampled action of every slot in every round, the games_played
    trajectory, the (state, pi, z) samples incl. symmetries and the results in queue order, and end with its tape counters at the end of
    each slot's recorded draws.  (Temperatures other than 1 go through powf: counts and actions are compared exactly, as in the agent
    goldens.)  name = 'br': the same for the reference's second game -- br_mt19937_agent.npz, 4 brandubh games over 53 rounds, 3 329 recorded
    shuffles of 40-100 children (lists longer than a wavefront), 8-fold symmetries in the samples."""
    torch = torch_mod
    d = dict(np.load(os.path.join(G, name + '_mt19937_agent.npz')))
    B, sims, games, eseed = int(d['B']), int(d['sims']), int(d['games']), int(d['eval_seed'])
    cpuct, fpu, nfrac, rtemp = [float(x) for x in d['cfg']]
    eng = engine(game={'c4': C4, 'br': ol.GAME_BRANDUBH}[name], B=B, cpuct=cpuct, fpu_reduction=fpu, root_noise_frac=nfrac, root_policy_temp=rtemp,
                 add_root_noise=True, add_root_temp=True, seed=987654321, games_per_iteration=games, example_capacity=4096, sims_hint=sims)
    eng.set_random_tape(d['tape_ranks'], d['tape_u'], d['tape_noise_off'], d['tape_noise_pool'])
    rec = run_engine_agent(torch, eng, eseed, 0, sims, games, launch=launch)
    n = len(d['actions'])
    assert len(rec['actions']) == n and (np.array(rec['games_played']) == d['games_played']).all()
    for r in range(n):
        assert (np.asarray(rec['counts'][r]) == d['counts'][r]).all(), (launch, r)
        assert (np.asarray(rec['actions'][r]) == d['actions'][r]).all(), (launch, r)
    eo, ep, ez = [t.cpu().numpy() for t in eng.examples()]
    assert eo.shape == d['s_obs'].shape and (eo == d['s_obs']).all() and (ez == d['s_z']).all()
    assert np.allclose(ep, d['s_pi'], rtol=0, atol=0) and (ep == d['s_pi']).all()         # pi at T = 1 is counts / sum: exact
    ws, turns, _ = eng.results()
    assert (ws == d['r_ws']).all() and (turns == d['r_turns']).all()
    used = [int((d['call_order'][(d['call_order'][:, 1] == sl) & (d['call_order'][:, 0] == 0), 2]).sum()
                + ((d['call_order'][:, 1] == sl) & (d['call_order'][:, 0] > 0)).sum()) for sl in range(B)]
    assert (eng.tape_counters() == np.array(used, np.uint64)).all()
    eng.set_shuffle_tape(None)
    eng.close()


# ------------------------------------------------------------------------------------------------ agent goldens
AGENT_CFGS = {
    'plain': dict(),
    'noisy': dict(add_root_noise=True, add_root_temp=True, cpuct=4.0, fpu_reduction=0.4),
    'fastmix': dict(symmetric_samples=False),
    'reset': dict(mcts_reset_threshold=3),
    'warmup': dict(),
    'config1': dict(),
}
AGENT_ROUND = {'fastmix': dict(prob_fast=0.5, fast_sims=6), 'warmup': dict(warmup=True, warmup_sims=5)}


def run_engine_agent(torch, eng, seed, slot_base, sims, games, prob_fast=0.0, fast_sims=20, warmup=False, warmup_sims=5,
                     max_rounds=500, launch='phase'):
    """SelfPlayAgent.run (SelfPlayAgent.pyx:79-101) on the engine.  launch = 'phase': generateBatch / processBatch as azg_select /
    azg_backup; 'fused': backup k and select k + 1 share a launch (azg_backup_select: k_backup_select2, two wavefronts per tree) --
    the form every runner and bench line uses."""
    from alphazero_general_amd.utils import AGENT_STREAM
    from alphazero_general_amd import _abi
    L = _abi.lib()
    B, A, NV = eng.B, eng.A, eng.NV
    rec = dict(actions=[], counts=[], obs_crc=[], games_played=[], sims=[])
    obs = eng.new_obs()
    step, actr = 0, 0
    wp = torch.full((B, A), 1 / A, dtype=torch.float32, device=eng.device)
    wv = torch.full((B, NV), 1 / NV, dtype=torch.float32, device=eng.device)
    gp = 0
    for _ in range(max_rounds):
        if gp >= games:
            break
        fast = L.azg_tape_uniform(seed, AGENT_STREAM + slot_base, actr) < prob_fast
        actr += 1
        ns = fast_sims if fast else (warmup_sims if warmup else sims)
        rec['sims'].append(ns)
        if launch == 'fused':
            eng.select(None if warmup else obs)
        for s in range(ns):
            if launch == 'phase':
                eng.select(None if warmup else obs)
            if warmup:
                pol, val = wp, wv
            else:
                o = obs.cpu().numpy()
                rec['obs_crc'].append([crc(o[i]) for i in range(B)])
                pol, val = fake_batch(torch, seed, [slot_base + i for i in range(B)], step, A, NV, eng.device)
            if launch == 'fused' and s + 1 < ns:
                eng.backup_select(pol, val, None if warmup else obs)
            else:
                eng.backup(pol, val)
            step += 1
        rec['counts'].append(eng.root_counts().cpu().numpy())
        eng.advance(record_history=not fast)
        rec['actions'].append(eng.last_actions().cpu().numpy())
        gp = eng.counters()['games_played']
        rec['games_played'].append(gp)
    return rec


LAUNCHES = ['phase', 'fused']


@pytest.mark.parametrize('launch', LAUNCHES)
@pytest.mark.parametrize('cname', list(AGENT_CFGS))
def test_c4_agent_vs_reference_goldens(torch_mod, cname, launch):
    torch = torch_mod
    d = dict(np.load(os.path.join(G, 'c4_agent.npz')))              # (NpzFile decompresses an array on EVERY d[key])
    B, sims, games = int(d[cname + '_B']), int(d[cname + '_sims']), int(d[cname + '_games'])
    seed, slot_base = int(d[cname + '_seed']), int(d[cname + '_slot_base'])
    eng = engine(B=B, seed=seed, slot_base=slot_base, games_per_iteration=games, example_capacity=16384, sims_hint=sims,
                 **AGENT_CFGS[cname])
    rec = run_engine_agent(torch, eng, seed, slot_base, sims, games, launch=launch, **AGENT_ROUND.get(cname, {}))
    assert (np.array(rec['sims']) == d[cname + '_round_sims']).all()
    assert (np.array(rec['counts']) == d[cname + '_counts']).all()
    assert (np.array(rec['actions']) == d[cname + '_actions']).all()
    assert (np.array(rec['games_played']) == d[cname + '_games_played']).all()
    if cname != 'warmup':
        assert (np.array(rec['obs_crc'], np.uint32) == d[cname + '_obs_crc']).all()
    obs, pi, z = [t.cpu().numpy() for t in eng.examples()]
    assert obs.shape == d[cname + '_s_obs'].shape
    assert (obs == d[cname + '_s_obs']).all()
    assert (pi == d[cname + '_s_pi']).all()
    assert (z == d[cname + '_s_z']).all()
    ws, turns, slot = eng.results()
    assert (ws == d[cname + '_r_ws']).all() and (turns == d[cname + '_r_turns']).all()
    eng.close()


# ----------------------------------------------------------------------------------- live oracle, larger sizes
@pytest.mark.parametrize('launch', LAUNCHES)
@pytest.mark.parametrize('B,sims,games,kw', [
    (256, 40, 300, dict(cpuct=4.0, fpu_reduction=0.4)),
    (96, 30, 120, dict(add_root_noise=True, cpuct=1.25)),
])
def test_c4_selfplay_vs_oracle_live(torch_mod, B, sims, games, kw, launch):
    torch = torch_mod
    seed = 4242
    okw = dict(kw)
    ag = ol.OAgent(C4, B, sims=sims, games_per_iteration=games, seed=seed, **okw)
    eng = engine(B=B, seed=seed, games_per_iteration=games, example_capacity=200000, sims_hint=sims, **kw)
    obs = eng.new_obs()
    A, NV = 7, 3
    step = 0
    while ag.games_played < games:
        ag.begin_round()
        if launch == 'fused':
            eng.select(obs)
        for s in range(sims):
            oobs, rg, rm = ag.generate_batch()
            if launch == 'phase':
                eng.select(obs)
            if step % 7 == 0:
                assert (obs.cpu().numpy() == oobs).all(), step
            pol = np.zeros((B, A), np.float32); val = np.zeros((B, NV), np.float32)
            for i in range(B):
                pol[i], val[i] = ol.fake_eval(seed, i, step, A, NV)
            ag.process_batch(pol, val)
            tp, tv = torch.from_numpy(pol).to(eng.device), torch.from_numpy(val).to(eng.device)
            if launch == 'fused' and s + 1 < sims:
                eng.backup_select(tp, tv, obs)
            else:
                eng.backup(tp, tv)
            step += 1
        ag.play_moves()
        eng.advance(True)
        assert (eng.last_actions().cpu().numpy() == ag.last_actions()).all()
    c = eng.counters()
    assert c['games_played'] == ag.games_played
    assert c['sims'] == ag.sims_done and c['expansions'] == ag.expansions
    oo, op, oz = ag.samples()
    eo, ep, ez = [t.cpu().numpy() for t in eng.examples()]
    assert eo.shape == oo.shape and (eo == oo).all() and (ep == op).all() and (ez == oz).all()
    ws, turns, slot = eng.results()
    ows, oturns, oslot = ag.results()
    assert (ws == ows).all() and (turns == oturns).all() and (slot == oslot).all()
    eng.close()


# ------------------------------------------------------------------- full-size properties (2048 games x 100 sims)
def test_c4_full_size_properties(torch_mod):
    torch = torch_mod
    B, sims = 2048, 100
    seed = 7
    eng = engine(B=B, seed=seed, cpuct=4.0, fpu_reduction=0.4, games_per_iteration=1 << 30, example_capacity=400000, sims_hint=sims)
    eng2 = engine(B=B // 2, seed=seed, slot_base=B // 2, cpuct=4.0, fpu_reduction=0.4, games_per_iteration=1 << 30,
                  example_capacity=200000, sims_hint=sims)
    g = torch.Generator(device='cpu'); g.manual_seed(0)
    obs, obs2 = eng.new_obs(torch.float16), eng2.new_obs(torch.float16)
    for move in range(3):
        for s in range(sims):
            pol = torch.rand((B, 7), generator=g) + 1e-3
            pol = (pol / pol.sum(1, keepdim=True)).to(eng.device)
            val = torch.rand((B, 3), generator=g) + 1e-3
            val = (val / val.sum(1, keepdim=True)).to(eng.device)
            eng.select(obs); eng.backup(pol, val)
            eng2.select(obs2); eng2.backup(pol[B // 2:].contiguous(), val[B // 2:].contiguous())
        cnt = eng.root_counts()
        # every simulation adds exactly one visit below the root: sum(child n) == root.n - 1 for a fresh root (Q8)
        if move == 0:
            assert (cnt.sum(1) == sims - 1).all()
        pr = eng.root_probs(1.0)
        assert torch.allclose(pr.sum(1), torch.ones(B, device=eng.device), atol=1e-5)
        assert ((pr > 0) == (cnt > 0)).all()
        # sharding invariance: slots [B/2, B) of the big engine == an engine created with slot_base = B/2
        assert (cnt[B // 2:] == eng2.root_counts()).all()
        eng.advance(True); eng2.advance(True)
        assert (eng.last_actions()[B // 2:] == eng2.last_actions()).all()
        # the sampled action is always a visited child
        act = eng.last_actions().long()
        assert (cnt.gather(1, act[:, None]) > 0).all()
    c = eng.counters()
    assert c['sims'] == 3 * sims * B and c['expansions'] <= c['sims']
    st = eng.get_states(0, 16)
    assert all(t == 3 for (_, _, t) in st)
    eng.close(); eng2.close()


def test_errors_surface(torch_mod):
    eng = engine(B=4, sims_hint=1, nodes_per_tree=16)
    obs = eng.new_obs()
    torch = torch_mod
    pol = torch.full((4, 7), 1 / 7, device=eng.device); val = torch.full((4, 3), 1 / 3, device=eng.device)
    with pytest.raises(ValueError):
        eng.update_root(0, 9)                      # not a legal action -> ValueError like MCTS.pyx:195
    from alphazero_general_amd._abi import AzgError
    with pytest.raises(AzgError):
        for _ in range(10):
            eng.select(obs); eng.backup(pol, val)
        eng.counters()                             # tree arena overflow is reported, not silently dropped
    eng.close()
    # one simulation per move: no child of the root has a visit when playMoves asks for MCTS.probs, counts / 0.  The reference runs
    # under np.seterr(all='raise') (MCTS.pyx:23) and dies with FloatingPointError at :320; the engine raises the same, not NaNs
    eng = engine(B=4, sims_hint=1)
    obs = eng.new_obs()
    eng.select(obs); eng.backup(pol, val)
    eng.advance(True)
    with pytest.raises(FloatingPointError):
        eng.counters()
    eng.close()
    # a policy whose valid entries sum to 0: policy /= np.sum(policy) (MCTS.pyx:245) is x / 0 -> FloatingPointError there and here
    eng = engine(B=4, sims_hint=4)
    obs = eng.new_obs()
    eng.select(obs); eng.backup(torch.zeros((4, 7), device=eng.device), val)
    with pytest.raises(FloatingPointError):
        eng.counters()
    eng.close()


# ------------------------------------------------------------------ single-tree MCTS class (reference API surface)
def test_mcts_class_api_vs_oracle(torch_mod):
    from alphazero_general_amd.MCTS import MCTS
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.utils import dotdict
    seed = 31337
    args = dotdict(root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1, fpu_reduction=0.2, cpuct=1.25,
                   _num_players=3, numMCTSSims=50, _azg_seed=seed)
    m = MCTS(args)
    g = Game()
    for a in (3, 3, 2):
        g.play_action(a)
    og = ol.OGame(C4)
    for a in (3, 3, 2):
        og.play(a)
    om = ol.OMCTS(C4, seed=seed, stream=0)
    step = [0]

    def nn(obs):
        assert obs.shape == (4, 6, 7)
        p, v = ol.fake_eval(seed, 0, step[0], 7, 3)
        step[0] += 1
        return p, v
    m.search(g, nn, 50, False, False)
    for s in range(50):
        leaf, _ = om.find_leaf(og)
        p, v = ol.fake_eval(seed, 0, s, 7, 3)
        om.process_results(v, p)
    assert (m.counts(g) == om.counts()).all()
    assert (m.probs(g, 1.0) == om.probs(1.0)).all()
    assert m.best_action(g) == int(np.argmax(om.counts()))
    assert m.value() == om.value(False) and m.value(True) == om.value(True)
    assert m.max_depth == om.max_depth
    ch = {c.a: (c.n, c.q) for c in m._root._children}
    och = om.root_children()
    assert ch == {int(a): (int(n), float(q)) for a, n, q in zip(och['a'], och['n'], och['q'])}
    a = m.best_action(g)
    m.update_root(g, a); om.update_root(og, a)
    g.play_action(a); og.play(a)
    m.raw_search(g, 20, False, False); om.raw_search(og, 20)
    assert (m.counts(g) == om.counts()).all()
    with pytest.raises(ValueError):
        full = Game()
        m2 = MCTS(args)
        m2.update_root(full, 7)


# ----------------------------------------------------------------------------------- arena mode (BASELINE config 4)
def test_c4_arena_vs_oracle_live(torch_mod):
    """Arena engine (one tree per player per game, rows grouped by model, correct row<->game routing) against the
    oracle with ref_misroute off.  Until the first game ends all games are synchronised, where the reference's own
    routing is correct too: that prefix is additionally pinned to the reference golden."""
    torch = torch_mod
    from alphazero_general_amd.utils import AGENT_STREAM
    d = dict(np.load(os.path.join(G, 'c4_arena.npz')))              # (NpzFile decompresses an array on EVERY d[key])
    B, sims, games, seed = int(d['arena_B']), int(d['arena_sims']), int(d['arena_games']), int(d['arena_seed'])
    A, NV = 7, 3
    ag = ol.OAgent(C4, B, sims=sims, games_per_iteration=games, seed=seed, is_arena=True, ref_misroute=False)
    p2i = ag.player_to_index()
    assert p2i == list(d['arena_player_to_index'])
    eng = engine(B=B, arena=True, seed=seed, games_per_iteration=games, sims_hint=sims, arena_temp=0.25)
    obs = eng.new_obs()
    step, rnd = 0, 0
    synced = True
    while ag.games_played < games:
        ag.begin_round()
        for s in range(sims):
            oobs, rg, rm = ag.generate_batch()
            row_of_slot, rpm = eng.arena_rows(p2i)
            eng.select(obs, row_of_slot)
            assert (np.argsort(row_of_slot.cpu().numpy(), kind='stable') == rg).all()
            assert (rpm.cpu().numpy() == np.bincount(rm, minlength=2)).all()
            assert (obs.cpu().numpy() == oobs).all()
            if synced:
                assert [crc(oobs[i]) for i in range(B)] == list(d['arena_obs_crc'][step])
            pol = np.zeros((B, A), np.float32); val = np.zeros((B, NV), np.float32)
            for row in range(B):
                pol[row], val[row] = ol.fake_eval(seed, rg[row], step, A, NV)
            ag.process_batch(pol, val)
            eng.backup(torch.from_numpy(pol).to(eng.device), torch.from_numpy(val).to(eng.device), row_of_slot)
            step += 1
        cnt = eng.root_counts().cpu().numpy()
        for i in range(B):
            ch = ag.root_children(i, ag.state(i).player)
            c = np.zeros(A, np.int32); c[ch['a']] = ch['n']
            assert (c == cnt[i]).all()
        if synced:
            assert (cnt == d['arena_counts'][rnd]).all()
        ag.play_moves(); eng.advance(False)
        assert (eng.last_actions().cpu().numpy() == ag.last_actions()).all()
        if synced:
            assert (ag.last_actions() == d['arena_actions'][rnd]).all()
        if ag.games_played > 0:
            synced = False
        rnd += 1
    c = eng.counters()
    assert c['games_played'] == ag.games_played and c['num_examples'] == 0
    ws, turns, slot = eng.results()
    ows, oturns, oslot = ag.results()
    assert (ws == ows).all() and (turns == oturns).all() and (slot == oslot).all()
    eng.close()


# ------------------------------------------------------------------------------------------------ brandubh (config 3)
BR = 1


def _br_positions(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        g = ol.OGame(BR)
        L = rng.randint(0, 70)
        for _ in range(L):
            if g.win_state().any():
                break
            v = np.flatnonzero(g.valid_moves())
            g.play(int(rng.choice(v)))
        out.append(g)
    return out


def test_br_rules_fuzz(torch_mod):
    """Device rule kernels against the oracle on random playout positions: expanding a root exposes valid_moves, win_state
    and observation; the second simulation descends one ply and exposes play_action (captures, surround, king flags)."""
    torch = torch_mod
    N = 1536
    pos = _br_positions(N, 5)
    eng = engine(game=BR, B=N, seed=3, sims_hint=4)
    eng.set_states([(g.cells(), g.player, g.turns, g.s.aux[0]) for g in pos])
    obs = eng.new_obs()
    eng.select(obs)
    o = obs.cpu().numpy()
    uni_p = torch.full((N, 588), 1 / 588, device=eng.device); uni_v = torch.full((N, 3), 1 / 3, device=eng.device)
    for i, g in enumerate(pos):
        ch = eng.root_children(i)
        assert (np.sort(ch['a']) == np.flatnonzero(g.valid_moves())).all(), i
        ws = g.win_state()
        assert eng.tree_info(i)['e'] == int(ws[0]) + 2 * int(ws[1]) + 4 * int(ws[2]), i
        assert (o[i] == g.observation()).all(), i
    eng.backup(uni_p, uni_v)
    eng.select(obs)
    o = obs.cpu().numpy()
    leaves = eng.get_leaf_states(full=True)
    for i, g in enumerate(pos):
        path = eng.last_path(i)
        h = g.clone()
        if g.win_state().any():
            assert len(path) == 0
        else:
            assert len(path) == 1
            h.play(int(path[0]))
        cells, player, turns, kc = leaves[i]
        assert (cells == h.cells()).all() and player == h.player and turns == h.turns and kc == h.s.aux[0], i
        assert (o[i] == h.observation()).all(), i
    eng.counters()
    eng.close()


@pytest.mark.parametrize('cname', ['default', 'noise_temp'])
def test_br_tree_vs_reference_goldens(torch_mod, cname):
    torch = torch_mod
    d = dict(np.load(os.path.join(G, 'br_tree.npz')))              # (NpzFile decompresses an array on EVERY d[key])
    cpuct, fpu, noise, temp, sims = d[cname + '_cfg']
    noise, temp, sims = bool(noise), bool(temp), int(sims)
    seed = int(d[cname + '_seed'])
    R, A, NV = d['prefix'].shape[0], 588, 3
    exact = not temp
    eng = engine(game=BR, B=R, cpuct=cpuct, fpu_reduction=fpu, add_root_noise=noise, add_root_temp=temp, seed=seed, sims_hint=sims)
    states = []
    for r in range(R):
        g = ol.OGame(BR)
        for a in d['prefix'][r]:
            if a >= 0:
                g.play(a)
        states.append((g.cells(), g.player, g.turns, g.s.aux[0]))
    eng.set_states(states)
    obs = eng.new_obs()
    for s in range(sims):
        eng.select(obs)
        for r in range(0, R, 5):
            path = eng.last_path(r)
            assert len(path) == d[cname + '_depth'][r, s]
            assert (path[:24] == d[cname + '_paths'][r, s][:len(path)]).all(), (r, s)
        pol, val = fake_batch(torch, seed, range(R), s, A, NV, eng.device)
        eng.backup(pol, val)
        cnt = eng.root_counts().cpu().numpy()
        assert (cnt == d[cname + '_rootn'][:, s]).all(), s
    for r in range(R):
        ch = eng.root_children(r)
        k = len(ch['a'])
        assert (ch['a'] == d[cname + '_a'][r][:k]).all() and (ch['n'] == d[cname + '_n'][r][:k]).all()
        for f in ('q', 'p', 'v'):
            if exact:
                assert (ch[f] == d[cname + '_' + f][r][:k]).all(), (f, r)
            else:
                assert np.allclose(ch[f], d[cname + '_' + f][r][:k], atol=1e-5), (f, r)
    for ti, t in enumerate(d['prob_temps']):
        pr = eng.root_probs(float(t)).cpu().numpy()
        ref = d[cname + '_probs'][:, ti]
        if t in (1.0, 0.5, 0.0):
            assert (pr == ref).all(), t                 # exercises the numpy pairwise-sum plan for A = 588
        else:
            assert np.allclose(pr, ref, rtol=3e-7, atol=1e-12), t
    assert (eng.root_value(False).cpu().numpy() == d[cname + '_vmax']).all()
    assert (eng.root_value(True).cpu().numpy() == d[cname + '_vavg']).all()
    assert (eng.tape_counters() == d[cname + '_ctr']).all()
    eng.close()


@pytest.mark.parametrize('noise', [False, True])
def test_br_nodes_with_more_than_64_children_vs_oracle(torch_mod, noise):
    """The two-chunk paths of the tree kernels (a node's children spread over two lanes-of-64 passes: valid list, add_children with
    the rank-by-key shuffle of k > 64 keys, best_child, leaf policy with numpy-order renormalisation, root noise over k > 64
    children) are never reached by playouts from the start position.  Crafted positions with 65-74 legal moves, every simulation
    against the oracle's MCTS on the same tape: paths, counts, q, priors bit for bit; both launch forms."""
    torch = torch_mod
    M, sims, seed, A, NV = 24, 48, 77, 588, 3
    pos = ol.br_wide_positions(M, 11)
    ks = [int(g.valid_moves().sum()) for g in pos]
    assert min(ks) > 64 and max(ks) >= 70
    oms = [ol.OMCTS(BR, seed=seed, stream=r) for r in range(M)]
    engs = [engine(game=BR, B=M, seed=seed, add_root_noise=noise, sims_hint=sims) for _ in range(2)]    # phase launches, fused launch
    for e in engs:
        e.set_states([(g.cells(), g.player, g.turns, g.s.aux[0]) for g in pos])
    obs = engs[0].new_obs()
    engs[1].select(None)
    for s in range(sims):
        engs[0].select(obs)
        o = obs.cpu().numpy()
        pol, val = fake_batch(torch, seed, range(M), s, A, NV, engs[0].device)
        for r in range(M):
            leaf, _ = oms[r].find_leaf(pos[r])
            assert (engs[0].last_path(r) == oms[r].last_path()).all() and (engs[1].last_path(r) == oms[r].last_path()).all(), (r, s)
            assert (o[r] == leaf.observation()).all(), (r, s)
            p, v = ol.fake_eval(seed, r, s, A, NV)
            oms[r].process_results(v, p, noise=noise, temp=False)
        engs[0].backup(pol, val)
        if s + 1 < sims:
            engs[1].backup_select(pol, val, None)
        else:
            engs[1].backup(pol, val)
    for e in engs:
        for r in range(M):
            ch, och = e.root_children(r), oms[r].root_children()
            assert len(ch['a']) == ks[r]
            for f in ('a', 'n', 'q', 'p', 'v'):
                assert (ch[f] == och[f]).all(), (f, r)
        pr = e.root_probs(1.0).cpu().numpy()
        assert all((pr[r] == oms[r].probs(1.0)).all() for r in range(M))
        e.counters()
        e.close()


@pytest.mark.parametrize('game,spread', [(BR, 95.0), (BR, 12.0), (0, 95.0), (2, 40.0)])
def test_prior_spreads_from_denormal_to_one_vs_oracle(torch_mod, game, spread):
    """Python's sum() over the visited children's priors (MCTS.pyx:91, double, list order) is taken by a reduction tree when the
    priors' exponents prove every partial sum exact, by the serial loop otherwise (csrc/azg_kernels.h best_child).  Policies whose
    entries span e^-spread .. 1 (spread 95: down to float32 denormals and exact zeros after renormalisation; spread 12: everything
    inside the 22 binades of the exact case) put both forms, and the switch between them from one simulation to the next, against
    the oracle: every path, and a / n / q / p / v of every root child at the end, bit for bit."""
    torch = torch_mod
    gi = ol.game_info(game)
    A, NV, M, sims, seed = gi.action_size, gi.num_players + gi.has_draw, 32, 160, 5
    rng = np.random.RandomState(int(spread) + game)
    pos = []
    for r in range(M):
        g = ol.OGame(game)
        for _ in range(rng.randint(0, 6)):
            g.play(int(rng.choice(np.flatnonzero(g.valid_moves()))))
        pos.append(g)
    oms = [ol.OMCTS(game, seed=seed, stream=r, cpuct=1.0, fpu_reduction=-1.0) for r in range(M)]    # (a first-play BONUS: every child gets visited)
    engs = [engine(game=game, B=M, seed=seed, cpuct=1.0, fpu_reduction=-1.0, sims_hint=sims) for _ in range(2)]
    for e in engs:
        e.set_states([(g.cells(), g.player, g.turns, g.s.aux[0]) if game == BR else ostate(g) for g in pos])
    engs[1].select(None)
    visited_max = 0
    for s in range(sims):
        engs[0].select(None)
        pol = np.exp(-rng.uniform(0.0, spread, size=(M, A))).astype(np.float32)
        val = rng.dirichlet(np.ones(NV), size=M).astype(np.float32)
        for r in range(M):
            oms[r].find_leaf(pos[r])
            assert (engs[0].last_path(r) == oms[r].last_path()).all() and (engs[1].last_path(r) == oms[r].last_path()).all(), (r, s)
            oms[r].process_results(val[r], pol[r])
        tp, tv = torch.from_numpy(pol).to(engs[0].device), torch.from_numpy(val).to(engs[0].device)
        engs[0].backup(tp, tv)
        if s + 1 < sims:
            engs[1].backup_select(tp, tv, None)
        else:
            engs[1].backup(tp, tv)
    for r in range(M):
        och = oms[r].root_children()
        visited_max = max(visited_max, int((och['n'] > 0).sum()))
        for e in engs:
            ch = e.root_children(r)
            for f in ('a', 'n', 'q', 'p', 'v'):
                assert (ch[f] == och[f]).all(), (f, r)
    assert visited_max > 6                                         # (the reduction tree is only taken above six visited children)
    for e in engs:
        e.counters()
        e.close()


@pytest.mark.parametrize('launch', LAUNCHES)
@pytest.mark.parametrize('cname,kw', [('plain', dict()), ('noisy', dict(add_root_noise=True, add_root_temp=True)), ('wide', dict())])
def test_br_agent_vs_reference_goldens(torch_mod, cname, kw, launch):
    torch = torch_mod
    d = dict(np.load(os.path.join(G, 'br_agent.npz')))              # (NpzFile decompresses an array on EVERY d[key])
    B, sims, games = int(d[cname + '_B']), int(d[cname + '_sims']), int(d[cname + '_games'])
    seed, slot_base = int(d[cname + '_seed']), int(d[cname + '_slot_base'])
    eng = engine(game=BR, B=B, seed=seed, slot_base=slot_base, games_per_iteration=games, example_capacity=20000, sims_hint=sims, **kw)
    rec = run_engine_agent(torch, eng, seed, slot_base, sims, games, launch=launch)
    assert (np.array(rec['counts']) == d[cname + '_counts']).all()
    assert (np.array(rec['actions']) == d[cname + '_actions']).all()
    assert (np.array(rec['games_played']) == d[cname + '_games_played']).all()
    assert (np.array(rec['obs_crc'], np.uint32) == d[cname + '_obs_crc']).all()
    obs, pi, z = [t.cpu().numpy() for t in eng.examples()]
    assert obs.shape == d[cname + '_s_obs'].shape
    assert (obs == d[cname + '_s_obs']).all() and (pi == d[cname + '_s_pi']).all() and (z == d[cname + '_s_z']).all()
    ws, turns, slot = eng.results()
    assert (ws == d[cname + '_r_ws']).all() and (turns == d[cname + '_r_turns']).all()
    eng.close()


# ------------------------------------------------------------------------------------------------ trimok, 3 players (config 5)
TM = 2


@pytest.mark.parametrize('cname', ['default', 'noise_temp'])
def test_tm_tree_vs_reference_goldens(torch_mod, cname):
    torch = torch_mod
    d = dict(np.load(os.path.join(G, 'tm_tree.npz')))              # (NpzFile decompresses an array on EVERY d[key])
    cpuct, fpu, noise, temp, sims = d[cname + '_cfg']
    noise, temp, sims = bool(noise), bool(temp), int(sims)
    seed = int(d[cname + '_seed'])
    R, A, NV = d['prefix'].shape[0], 25, 4
    exact = not temp
    eng = engine(game=TM, B=R, cpuct=cpuct, fpu_reduction=fpu, add_root_noise=noise, add_root_temp=temp, seed=seed, sims_hint=sims)
    states = []
    for r in range(R):
        g = ol.OGame(TM)
        for a in d['prefix'][r]:
            if a >= 0:
                g.play(a)
        states.append(ostate(g))
    eng.set_states(states)
    obs = eng.new_obs()
    for s in range(sims):
        eng.select(obs)
        pol, val = fake_batch(torch, seed, range(R), s, A, NV, eng.device)
        eng.backup(pol, val)
        assert (eng.root_counts().cpu().numpy() == d[cname + '_rootn'][:, s]).all(), s
    for r in range(R):
        ch = eng.root_children(r)
        k = len(ch['a'])
        assert (ch['a'] == d[cname + '_a'][r][:k]).all() and (ch['n'] == d[cname + '_n'][r][:k]).all()
        for f in ('q', 'p', 'v'):
            if exact:
                assert (ch[f] == d[cname + '_' + f][r][:k]).all(), (f, r)
            else:
                assert np.allclose(ch[f], d[cname + '_' + f][r][:k], atol=1e-5), (f, r)
    assert (eng.root_probs(1.0).cpu().numpy() == d[cname + '_probs'][:, 0]).all()
    assert (eng.root_value(False).cpu().numpy() == d[cname + '_vmax']).all()
    assert (eng.tape_counters() == d[cname + '_ctr']).all()
    eng.close()


@pytest.mark.parametrize('launch', LAUNCHES)
@pytest.mark.parametrize('cname,kw', [('plain', dict()), ('noisy', dict(add_root_noise=True, add_root_temp=True)), ('wide', dict())])
def test_tm_agent_vs_reference_goldens(torch_mod, cname, kw, launch):
    torch = torch_mod
    d = dict(np.load(os.path.join(G, 'tm_agent.npz')))              # (NpzFile decompresses an array on EVERY d[key])
    B, sims, games = int(d[cname + '_B']), int(d[cname + '_sims']), int(d[cname + '_games'])
    seed, slot_base = int(d[cname + '_seed']), int(d[cname + '_slot_base'])
    eng = engine(game=TM, B=B, seed=seed, slot_base=slot_base, games_per_iteration=games, example_capacity=8000, sims_hint=sims, **kw)
    rec = run_engine_agent(torch, eng, seed, slot_base, sims, games, launch=launch)
    assert (np.array(rec['counts']) == d[cname + '_counts']).all()
    assert (np.array(rec['actions']) == d[cname + '_actions']).all()
    assert (np.array(rec['games_played']) == d[cname + '_games_played']).all()
    assert (np.array(rec['obs_crc'], np.uint32) == d[cname + '_obs_crc']).all()
    obs, pi, z = [t.cpu().numpy() for t in eng.examples()]
    assert (obs == d[cname + '_s_obs']).all() and (pi == d[cname + '_s_pi']).all() and (z == d[cname + '_s_z']).all()
    ws, turns, slot = eng.results()
    assert (ws == d[cname + '_r_ws']).all() and (turns == d[cname + '_r_turns']).all()
    eng.close()


# --------------------------------------------------- full-size properties of the other BASELINE configs (3 and 5)
@pytest.mark.parametrize('game,B,sims,cpuct,fpu', [(1, 512, 200, 1.25, 0.2), (2, 256, 50, 1.25, 0.2)])
def test_other_configs_full_size_properties(torch_mod, game, B, sims, cpuct, fpu):
    """brandubh 512 games x 200 sims (config 3, per-GPU shard) and the 3-player env 256 x 50 (config 5): visit-count
    conservation, pi a distribution over visited children, sharding invariance, sampled action visited, noise on."""
    torch = torch_mod
    from alphazero_general_amd import _abi
    gi = _abi.game_info(game)
    A, NV = gi.action_size, gi.num_players + 1
    seed = 11
    kw = dict(game=game, seed=seed, cpuct=cpuct, fpu_reduction=fpu, games_per_iteration=1 << 30, sims_hint=sims,
              add_root_noise=True, add_root_temp=True)
    eng = engine(B=B, example_capacity=4096, **kw)
    eng2 = engine(B=B // 4, slot_base=3 * B // 4, example_capacity=1024, **kw)
    g = torch.Generator(device='cpu'); g.manual_seed(1)
    obs, obs2 = eng.new_obs(torch.float16), eng2.new_obs(torch.float16)
    for move in range(2):
        for s in range(sims):
            pol = torch.rand((B, A), generator=g) + 1e-3
            pol = (pol / pol.sum(1, keepdim=True)).to(eng.device)
            val = torch.rand((B, NV), generator=g) + 1e-3
            val = (val / val.sum(1, keepdim=True)).to(eng.device)
            eng.select(obs); eng.backup(pol, val)
            eng2.select(obs2); eng2.backup(pol[3 * B // 4:].contiguous(), val[3 * B // 4:].contiguous())
        cnt = eng.root_counts()
        if move == 0:
            assert (cnt.sum(1) == sims - 1).all()                    # Q8
        pr = eng.root_probs(1.0)
        assert torch.allclose(pr.sum(1), torch.ones(B, device=eng.device), atol=1e-5)
        assert ((pr > 0) == (cnt > 0)).all()
        assert (cnt[3 * B // 4:] == eng2.root_counts()).all()        # sharding invariance (global slot ids)
        assert torch.equal(obs[3 * B // 4:], obs2)                   # ... down to the last leaf observation
        eng.advance(True); eng2.advance(True)
        act = eng.last_actions().long()
        assert (act[3 * B // 4:] == eng2.last_actions().long()).all()
        assert (cnt.gather(1, act[:, None]) > 0).all()
    c = eng.counters()
    assert c['sims'] == 2 * sims * B and c['expansions'] <= c['sims'] and c['max_nodes_used'] > 0
    assert all(t == 2 for (_, _, t) in eng.get_states(0, 8))
    eng.close(); eng2.close()


@pytest.mark.parametrize('game', [0, 2])
def test_whole_games_sample_invariants(torch_mod, game):
    """play whole games on the device (uniform evaluator, few sims) and check what Coach would save (Coach.py:377-383):
    z rows are the one-hot winstate of their game, pi rows are distributions over legal moves, every finished game
    contributes (its length x symmetries) samples, results and counters agree."""
    torch = torch_mod
    from alphazero_general_amd import _abi
    gi = _abi.game_info(game)
    A, NV, B, sims = gi.action_size, gi.num_players + 1, 128, 6
    nsym = gi.num_symmetries
    eng = engine(game=game, B=B, seed=3, games_per_iteration=1 << 30, sims_hint=sims,
                 example_capacity=B * 6 * (gi.max_turns + 1) * nsym)
    pol = torch.full((B, A), 1.0 / A, device=eng.device); val = torch.full((B, NV), 1.0 / NV, device=eng.device)
    rounds = 3 * gi.max_turns // 2
    for _ in range(rounds):
        for s in range(sims):
            eng.select(None); eng.backup(pol, val)
        eng.advance(True)
    c = eng.counters()
    assert c['games_played'] >= B                                    # every slot finished at least one game
    o, p, z = [t.cpu().numpy() for t in eng.examples()]
    ws, turns, slot = eng.results()
    assert len(ws) == c['num_results'] == c['games_played'] and o.shape[0] == c['num_examples'] == int(turns.sum()) * nsym
    assert ((z == 0) | (z == 1)).all() and (z.sum(1) == 1).all()
    assert np.allclose(p.sum(1), 1.0, atol=1e-5) and (p >= 0).all()
    # samples come game by game in result order: block g has turns[g] * nsym rows, all with z == winstate of game g
    off = 0
    for gidx in range(len(ws)):
        n = int(turns[gidx]) * nsym
        assert (z[off:off + n] == ws[gidx].astype(np.float32)).all()
        off += n
    assert off == o.shape[0]
    eng.close()


def test_abi_argument_validation(torch_mod):
    """every entry point rejects bad arguments with a status code and a message (azg_last_error) instead of touching the GPU."""
    import ctypes as C
    from alphazero_general_amd import _abi
    L = _abi.lib()
    eng = engine(B=8, sims_hint=4)
    h, null = eng.h, C.c_void_p(0)
    st = C.c_void_p(torch_mod.cuda.current_stream().cuda_stream)
    pol = torch_mod.zeros((8, 7), device=eng.device); val = torch_mod.zeros((8, 3), device=eng.device)
    vp = lambda t: C.c_void_p(t.data_ptr())

    def bad(rc, code=_abi.E_INVALID_ARG):
        assert rc == code, (rc, L.azg_last_error())
        assert len(L.azg_last_error()) > 0

    bad(L.azg_select(null, st, null, 0, null))
    bad(L.azg_select(h, st, null, 7, null))                                   # unknown obs dtype
    bad(L.azg_backup(h, st, null, vp(val), null, -1))
    bad(L.azg_backup_select(h, st, vp(pol), null, null, -1, null, 0))
    bad(L.azg_backup_select_logits(h, st, vp(pol), 7, null, -1, null, 0, 1))  # stride < A + P + 1
    bad(L.azg_set_states(h, st, 6, 5, _abi.states_array(5), 1))               # slot range out of bounds
    bad(L.azg_resnet_tower_f16(st, 0, null, null, null, null, null, null, 4, 1, 128))
    bad(L.azg_resnet_tower_f16(st, 0, vp(pol), vp(pol), vp(val), vp(val), vp(val), vp(pol), 4, 1, 96), _abi.E_UNSUPPORTED)   # no 96-channel tower
    bad(L.azg_resnet_policy_value_f16(st, 1, vp(pol), vp(pol), vp(val), vp(val), vp(val), 4, 1, vp(pol), vp(val), 588, 3, vp(pol), vp(val)),
        _abi.E_UNSUPPORTED)                                                   # heads too wide to fuse
    bad(L.azg_policy_value_heads_f16(st, vp(pol), vp(pol), vp(val), 4, 100, 7, 3, vp(pol), vp(pol), vp(val)))   # k not a multiple of 32
    bad(L.azg_search_f16(null, st, vp(pol), vp(val), vp(val), vp(val), 1, vp(pol), vp(val), 4))
    br = engine(game=1, B=4, sims_hint=4)
    bad(L.azg_search_f16(br.h, st, vp(pol), vp(val), vp(val), vp(val), 1, vp(pol), vp(val), 4), _abi.E_UNSUPPORTED)   # connect4 only
    cfg = _abi.Config(); cfg.abi_version = _abi.ABI_VERSION + 1; cfg.num_slots = 4
    out = C.c_void_p()
    bad(L.azg_engine_create(C.byref(cfg), C.byref(out)))                      # ABI version mismatch
    cfg.abi_version = _abi.ABI_VERSION; cfg.game = 99
    bad(L.azg_engine_create(C.byref(cfg), C.byref(out)), _abi.E_UNSUPPORTED)  # unknown game
    # the engines are still healthy
    eng.select(None); eng.backup(pol + 1.0 / 7, val + 1.0 / 3)
    assert eng.counters()['sims'] == 8
    eng.close(); br.close()
