"""Compat mode on the GPU: the SelfPlayAgent process class with the reference's constructor and queue / event /
shared-tensor protocol (alphazero/SelfPlayAgent.pyx:13-202), driven by a stand-in for the parent loop
Coach.processSelfPlayBatches (alphazero/Coach.py:326-361) -- /root/reference does not exist on the GPU box, so the
parent side is restated here line for line: ready_queue.get -> evaluate input_tensors[id] -> copy into
policy/value tensors -> batch_ready[id].set(), until completed == workers."""
import queue

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _evaluate(batch):
    """deterministic stand-in for nnet.process: probabilities from the observation planes (CPU, float32)."""
    import torch
    b = batch.reshape(batch.shape[0], -1)
    A, NV = 7, 3
    w = torch.linspace(-1, 1, b.shape[1] * A).reshape(b.shape[1], A).sin()
    u = torch.linspace(-2, 2, b.shape[1] * NV).reshape(b.shape[1], NV).cos()
    return torch.softmax(b @ w * 0.3, 1), torch.softmax(b @ u * 0.3, 1)


def _args(games, sims, **kw):
    from alphazero_general_amd.utils import dotdict, default_temp_scaling
    a = dotdict(cpuct=1.25, fpu_reduction=0.2, root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1, _num_players=3,
                numMCTSSims=sims, numFastSims=4, numWarmupSims=3, probFastSim=0.0, gamesPerIteration=games,
                add_root_noise=False, add_root_temp=False, symmetricSamples=True, mctsResetThreshold=None, startTemp=1,
                arenaTemp=0.25, temp_scaling_fn=default_temp_scaling, _azg_seed=2024)
    a.update(kw)
    return a


def _serve(agents, input_tensors, policy_tensors, value_tensors, batch_ready, ready_queue, completed, workers, batch_queues=None,
           timeout=300):
    import time
    t0 = time.time()
    while completed.value != workers:
        assert time.time() - t0 < timeout, 'agents did not finish'
        try:
            i = ready_queue.get(timeout=1)
        except queue.Empty:
            continue
        if batch_queues is None:
            p, v = _evaluate(input_tensors[i])
        else:                                            # Arena.play_games :269-281
            data = batch_queues[i].get()
            ps, vs = [], []
            for m in range(len(data)):
                if not isinstance(data[m], list):
                    p_, v_ = _evaluate(data[m] * (1.0 + 0.5 * m))       # two different "models"
                    ps.append(p_); vs.append(v_)
            import torch
            p, v = torch.cat(ps), torch.cat(vs)
        policy_tensors[i].copy_(p); value_tensors[i].copy_(v)
        batch_ready[i].set()


def test_compat_selfplay_agents_two_workers():
    import torch
    import torch.multiprocessing as mp
    from alphazero_general_amd.SelfPlayAgent import SelfPlayAgent
    from alphazero_general_amd.envs.connect4 import Game
    torch.zeros(1, device='cuda:0')                       # the parent owns a HIP context before forking, like Coach
    workers, B, games = 2, 8, 9
    args = _args(games, 6)
    ready_queue, file_queue, result_queue = mp.Queue(), mp.Queue(), mp.Queue()
    completed, games_played = mp.Value('i', 0), mp.Value('i', 0)
    stop, pause = mp.Event(), mp.Event()
    inputs, pols, vals, ready, agents = [], [], [], [], []
    for i in range(workers):                               # Coach.generateSelfPlayAgents :291-323
        inputs.append(torch.zeros([B, 4, 6, 7]).share_memory_())
        pols.append(torch.zeros([B, 7]).share_memory_())
        vals.append(torch.zeros([B, 3]).share_memory_())
        ready.append(mp.Event())
        agents.append(SelfPlayAgent(i, Game, ready_queue, ready[i], inputs[i], pols[i], vals[i], file_queue, result_queue,
                                    completed, games_played, stop, pause, args))
        agents[i].daemon = True
        agents[i].start()
    samples, results = [], []

    def drain():
        while True:
            try:
                samples.append(file_queue.get_nowait())
            except queue.Empty:
                break
        while True:
            try:
                results.append(result_queue.get_nowait())
            except queue.Empty:
                break
    import threading
    stop_drain = threading.Event()

    def drainer():
        while not stop_drain.is_set():
            drain(); stop_drain.wait(0.05)
    th = threading.Thread(target=drainer); th.start()
    try:
        _serve(agents, inputs, pols, vals, ready, ready_queue, completed, workers)
    finally:
        stop.set(); stop_drain.set(); th.join()
    import time
    time.sleep(0.5); drain()
    for a in agents:
        a.join(30)
    assert games_played.value == games                     # the cap is exact across agents (lock semantics)
    assert len(results) >= games
    assert len(samples) > 0
    for state, ws, aid in results:
        assert isinstance(state, Game) and ws.dtype == np.uint8 and ws.sum() == 1 and aid in (0, 1)
        assert (state.win_state() == ws).all()
    for obs, pi, z in samples:
        assert obs.shape == (4, 6, 7) and pi.shape == (7,) and z.shape == (3,)
        assert abs(pi.sum() - 1) < 1e-5 and z.sum() == 1


def test_compat_agent_matches_direct_engine():
    """one compat agent == the same engine driven in-process with the same seed and the same evaluator."""
    import torch
    import torch.multiprocessing as mp
    from alphazero_general_amd.SelfPlayAgent import SelfPlayAgent
    from alphazero_general_amd.engine import DeviceEngine
    from alphazero_general_amd.envs.connect4 import Game
    B, games, sims = 6, 5, 7
    args = _args(games, sims)
    ready_queue, file_queue, result_queue = mp.Queue(), mp.Queue(), mp.Queue()
    completed, games_played = mp.Value('i', 0), mp.Value('i', 0)
    stop, pause = mp.Event(), mp.Event()
    inp, pol, val, ev = torch.zeros([B, 4, 6, 7]).share_memory_(), torch.zeros([B, 7]).share_memory_(), torch.zeros([B, 3]).share_memory_(), mp.Event()
    ag = SelfPlayAgent(0, Game, ready_queue, ev, inp, pol, val, file_queue, result_queue, completed, games_played, stop, pause, args)
    ag.daemon = True; ag.start()
    got = []
    import threading
    done = threading.Event()

    def drainer():
        while not done.is_set():
            try:
                got.append(file_queue.get(timeout=0.05))
            except queue.Empty:
                pass
    th = threading.Thread(target=drainer); th.start()
    _serve([ag], [inp], [pol], [val], [ev], ready_queue, completed, 1)
    import time
    time.sleep(0.5); done.set(); th.join(); stop.set(); ag.join(30)
    # direct run
    eng = DeviceEngine(0, B, seed=2024, games_per_iteration=games, example_capacity=5000, sims_hint=sims)
    obs = eng.new_obs(torch.float32)
    while eng.counters()['games_played'] < games:
        for _ in range(sims):
            eng.select(obs)
            p, v = _evaluate(obs.cpu())
            eng.backup(p.to(eng.device).contiguous(), v.to(eng.device).contiguous())
        eng.advance(True)
    o, p, z = [t.cpu().numpy() for t in eng.examples()]
    assert len(got) == o.shape[0]
    assert all((g[0] == o[i]).all() and (g[1] == p[i]).all() and (g[2] == z[i]).all() for i, g in enumerate(got))


def test_compat_arena_agent():
    import torch
    import torch.multiprocessing as mp
    from alphazero_general_amd.SelfPlayAgent import SelfPlayAgent
    from alphazero_general_amd.envs.connect4 import Game
    B, games = 8, 10
    args = _args(games, 5)
    ready_queue, result_queue, bq = mp.Queue(), mp.Queue(), mp.Queue()
    completed, games_played = mp.Value('i', 0), mp.Value('i', 0)
    stop, pause = mp.Event(), mp.Event()
    pol, val, ev = torch.zeros([B, 7]).share_memory_(), torch.zeros([B, 3]).share_memory_(), mp.Event()
    ag = SelfPlayAgent(0, Game, ready_queue, ev, [[], []], pol, val, bq, result_queue, completed, games_played, stop, pause,
                       args, _is_arena=True)
    assert sorted(ag.player_to_index) == [0, 1]
    ag.daemon = True; ag.start()
    _serve([ag], None, [pol], [val], [ev], ready_queue, completed, 1, batch_queues=[bq])
    stop.set(); ag.join(30)
    assert games_played.value == games
    n = 0
    while True:
        try:
            state, ws, aid = result_queue.get(timeout=0.5); n += 1
            assert ws.sum() == 1 and (state.win_state() == ws).all()
        except queue.Empty:
            break
    assert n >= games


@pytest.mark.timeout(180)
def test_compat_agent_realistic_batch_with_gpu_net_in_parent():
    """Coach's real arrangement at a realistic size: the parent has a GPU network AND a live CPU thread pool before it forks
    the agent, batches are 1024 leaves.  (A Tensor.copy_ of that size in the forked agent used to wait forever for the
    parent's intra-op threads; small batches never showed it.)"""
    import time
    import torch
    import torch.multiprocessing as mp
    from alphazero_general_amd.SelfPlayAgent import SelfPlayAgent
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper
    torch.manual_seed(0)
    net = NNetWrapper(Game, CONNECT4_NET_ARGS, device='cuda:0'); net.refresh()
    (torch.randn(1024, 1024) @ torch.randn(1024, 1024)).sum().item()           # the parent's intra-op pool is up
    B, sims, games = 1024, 8, 1100
    args = _args(games, sims, add_root_noise=True, add_root_temp=True, cpuct=4.0, fpu_reduction=0.4)
    ready_queue, file_queue, result_queue = mp.Queue(), mp.Queue(), mp.Queue()
    completed, games_played = mp.Value('i', 0), mp.Value('i', 0)
    stop, pause = mp.Event(), mp.Event()
    inp, pol, val, ev = torch.zeros([B, 4, 6, 7]).share_memory_(), torch.zeros([B, 7]).share_memory_(), torch.zeros([B, 3]).share_memory_(), mp.Event()
    ag = SelfPlayAgent(0, Game, ready_queue, ev, inp, pol, val, file_queue, result_queue, completed, games_played, stop, pause, args)
    ag.daemon = True; ag.start()
    nsamples, nresults, steps, t0 = 0, 0, 0, time.time()
    try:
        while completed.value != 1:
            assert time.time() - t0 < 150, 'agent did not finish (%d batches served)' % steps
            for q, which in ((file_queue, 0), (result_queue, 1)):
                try:
                    while True:
                        q.get_nowait()
                        if which: nresults += 1
                        else: nsamples += 1
                except queue.Empty:
                    pass
            try:
                ready_queue.get(timeout=0.2)
            except queue.Empty:
                continue
            p, v = net.process(inp)                                    # Coach.processSelfPlayBatches :337-342
            pol.copy_(p); val.copy_(v); ev.set(); steps += 1
    finally:
        stop.set()
    time.sleep(0.5)
    for q, which in ((file_queue, 0), (result_queue, 1)):
        try:
            while True:
                q.get_nowait()
                if which: nresults += 1
                else: nsamples += 1
        except queue.Empty:
            pass
    ag.join(30)
    assert games_played.value == games and nresults >= games and nsamples >= games * 7 * 2 and steps % sims == 0
