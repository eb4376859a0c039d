"""bench.py -- self-play throughput of the MI355X engine on BASELINE.json's headline config.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): connect4, 2048 concurrent games per GPU x 100 MCTS sims per move, fp16 ResNet 128ch x 8
(envs/connect4/train.py net + search hyper-parameters), random-init weights, games from the empty board, root noise +
root temperature on, probFastSim = 0.  One "step" = one self-play round of the hot path over
the whole batch: 100 x [select -> network -> backup] + advance, i.e. 204 800 simulations per GPU.  value = MCTS node
expansions per second summed over all GPUs (games/s is reported next to it).  Everything is resident in HBM; the
only host traffic in the timed region is one 40-byte counter read per round.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from alphazero_general_amd import distributed as D  # noqa: E402
from alphazero_general_amd.envs.connect4 import Game  # noqa: E402
from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper  # noqa: E402
from alphazero_general_amd.selfplay import SelfPlayRunner  # noqa: E402
from alphazero_general_amd.utils import dotdict, default_temp_scaling  # noqa: E402

B_PER_GPU, SIMS = 2048, 100
NN_REPS = 8                                                           # back-to-back tower launches timed by one event pair
# algorithmic figures (DESIGN.md "Roofline"): bytes one simulation moves through the tree kernels / FLOPs per leaf
C4_SELECT_BYTES_PER_SIM = 5 * (32 + 7 * 32) + (32 + 7 * 32) + 2 * 80 + 336 + 5 * 4   # D=5 levels read, expand write, states, fp16 obs, path
C4_BACKUP_BYTES_PER_SIM = 7 * 4 + 12 + 7 * 4 + 5 * (4 + 16) + 32
C4_NET_FLOPS_PER_LEAF = 205e6
HBM_PEAK_GBS, MFMA_F16_PEAK_TFLOPS = 8000.0, 2500.0                                   # MI355X_MICROARCH.md
TOWER_TRAFFIC_BYTES = 50850000   # PMC per launch @2048 boards, (2 x FETCH_SIZE + WRITE_SIZE) KB (profiles/r01_pmc_summary.csv, rows
                                 # tower2_r1c): 43 MB fetched at the L2 <-> fabric boundary = the 4.7 MB weight stream once per XCD
                                 # (8 x 4.7 = 38 MB; Infinity-Cache hits are counted) + inputs, 6.4 MB written (52 B/lane of spills)


SEARCH_TRAFFIC_BYTES = 7169000000  # one azg_search_f16 launch, 2048 games x 100 sims


def selfplay_args(games):
    return dotdict(cpuct=4.0, fpu_reduction=0.4, root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1.0,
                   numMCTSSims=SIMS, numFastSims=20, numWarmupSims=5, probFastSim=0.0, gamesPerIteration=games,
                   add_root_noise=True, add_root_temp=True, symmetricSamples=True, mctsResetThreshold=None,
                   startTemp=1.0, arenaTemp=0.25, temp_scaling_fn=default_temp_scaling)


def library_gemm_tflops(dev, n=8192, reps=10):
    """fp16 n^3 GEMM through torch (hipBLASLt), TFLOP/s."""
    x = torch.randn(n, n, device=dev, dtype=torch.float16); y = torch.randn(n, n, device=dev, dtype=torch.float16)
    best = 0.0
    for _ in range(2):
        for _ in range(3):
            x @ y
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            x @ y
        e1.record(); torch.cuda.synchronize()
        best = max(best, 2 * n ** 3 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best


def cpu_baseline_threads(net, threads=16, seconds=6.0):
    """The reference's own arrangement (Coach.py:291-342): `workers` agent processes, each searching its batch of games
    on one host core, all of them queueing on ONE GPU network.  Here: `threads` oracle agents on `threads` host cores
    (the C oracle runs outside the GIL), 256 games each, the GPU net behind a lock.  Bounded sample."""
    import threading
    import oracle_lib as ol
    Bc = 256
    threads = max(1, min(threads, (os.cpu_count() or 1) - 2))
    lock = threading.Lock()
    agents = [ol.OAgent(0, Bc, sims=SIMS, games_per_iteration=1 << 30, seed=100 + i, cpuct=4.0, fpu_reduction=0.4,
                        add_root_noise=True, add_root_temp=True) for i in range(threads)]
    stop = time.time() + seconds

    def work(ag):
        while time.time() < stop:
            ag.begin_round()
            for s in range(SIMS):
                obs, _, _ = ag.generate_batch()
                with lock:
                    p, v = net.process(torch.from_numpy(obs))
                    p, v = p.cpu().numpy(), v.cpu().numpy()
                ag.process_batch(p, v)
            ag.play_moves()

    t0 = time.time()
    ts = [threading.Thread(target=work, args=(ag,)) for ag in agents]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.time() - t0
    total = sum(ag.expansions for ag in agents)
    return {'value': round(total / dt, 1), 'unit': 'expansions/s', 'cores': threads,
            'sample': '%d oracle agents x %d games x %d sims on %d host cores for %.1f s, one shared GPU net' % (threads, Bc, SIMS, threads, dt)}


def cpu_tree_only_threads(threads=64, seconds=4.0):
    """Tree-only leg on many host cores (BASELINE.md section 3 variant i): warm-up evaluator semantics -- uniform policy and
    value (SelfPlayAgent.pyx:48-52,111-114) -- so only select / expand / backup / playMoves run; one oracle agent of 256 games
    per thread, no shared resource."""
    import threading
    import oracle_lib as ol
    Bc = 256
    threads = max(1, min(threads, (os.cpu_count() or 1) - 2))
    agents = [ol.OAgent(0, Bc, sims=SIMS, games_per_iteration=1 << 30, seed=200 + i, cpuct=4.0, fpu_reduction=0.4,
                        add_root_noise=True, add_root_temp=True) for i in range(threads)]
    pol = np.full((Bc, 7), 1 / 7, np.float32); val = np.full((Bc, 3), 1 / 3, np.float32)
    stop = time.time() + seconds

    L = ol.lib()

    def work(ag):
        obs = np.zeros((Bc, ag.O), np.float32); rg = np.zeros(Bc, np.int32); rm = np.zeros(Bc, np.int32)   # (preallocated: the
        while time.time() < stop:                                    #  threads only meet at the GIL between C calls)
            ag.begin_round()
            for s in range(SIMS):
                L.azo_agent_generate_batch(ag.h, obs, rg, rm)
                L.azo_agent_process_batch(ag.h, pol, val)
            ag.play_moves()

    t0 = time.time()
    ts = [threading.Thread(target=work, args=(ag,)) for ag in agents]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.time() - t0
    return {'value': round(sum(ag.expansions for ag in agents) / dt, 1), 'unit': 'expansions/s', 'cores': threads,
            'sample': '%d oracle agents x %d games x %d sims, uniform evaluator, %.1f s' % (threads, Bc, SIMS, dt)}


def cpu_baseline(net, seconds=12.0):
    """The CPU path timed beside the GPU number: the C oracle (bit-exact restatement of the reference Cython path,
    oracle/) drives the same workload on ONE host core, leaves evaluated by the same GPU network through host
    buffers (the reference's own arrangement, Coach.py:337-342).  Bounded sample."""
    import oracle_lib as ol
    Bc = 256
    ag = ol.OAgent(0, Bc, sims=SIMS, games_per_iteration=1 << 30, seed=1, cpuct=4.0, fpu_reduction=0.4,
                   add_root_noise=True, add_root_temp=True)
    t_tree = t_all = 0.0
    sims_done = 0
    t_start = time.time()
    while time.time() - t_start < seconds:
        ag.begin_round()
        for s in range(SIMS):
            t0 = time.time()
            obs, _, _ = ag.generate_batch()
            t1 = time.time()
            p, v = net.process(torch.from_numpy(obs))
            p, v = p.cpu().numpy(), v.cpu().numpy()
            t2 = time.time()
            ag.process_batch(p, v)
            t3 = time.time()
            t_tree += (t1 - t0) + (t3 - t2); t_all += t3 - t0
            sims_done += Bc
        t0 = time.time(); ag.play_moves(); dt = time.time() - t0
        t_tree += dt; t_all += dt
    many = cpu_baseline_threads(net)
    tree_many = cpu_tree_only_threads()
    return {'value': round(ag.expansions / t_all, 1), 'unit': 'expansions/s', 'cores': 1, 'kind': 'port', 'many_cores': many, 'tree_only_many_cores': tree_many,
            'sample': 'connect4 %d games x %d sims, %d simulations in %.1f s on one host core, leaves evaluated by the same GPU net '
                      'through host buffers' % (Bc, SIMS, sims_done, t_all),
            'tree_only_value': round(ag.expansions / t_tree, 1), 'host_cpus': os.cpu_count(),
            # BASELINE.md section 3 calibration, measured in the build container (same core for both; the reference cannot travel):
            'vs_reference_cython': 'tree-only, connect4 256 games x 100 sims, one core: C oracle 201.6 k sims/s, reference Cython '
                                   'MCTS/SelfPlayAgent 8.2 k sims/s -> the port is 24.6x the reference'}


WORKLOADS = {
    # name: (game module, net args, games/GPU, sims, cpuct, fpu)   -- BASELINE.json configs 2..5
    'connect4': ('connect4', 'CONNECT4_NET_ARGS', 2048, 100, 4.0, 0.4),
    'brandubh': ('brandubh', 'BRANDUBH_NET_ARGS', 512, 200, 1.25, 0.2),
    'trimok': ('trimok', 'DEFAULT_NET_ARGS', 256, 50, 1.25, 0.2),
    'arena': ('connect4', 'CONNECT4_NET_ARGS', 256, 100, 4.0, 0.4),
}


# algorithmic bytes one simulation moves through the tree kernels (SURVEY.md 8d formula with 32-B node records):
# D*(32 + k*32) select reads + (32 + k*32) expansion writes + 2*80 states + fp16 obs + D*4 path ; backup: 4*A + 4*(P+1) + 4*k + D*20 + 32
TREE_BYTES = {'brandubh': (4 * (32 + 40 * 32) + (32 + 40 * 32) + 160 + 49 * 16 + 16, 4 * 588 + 12 + 160 + 80 + 32),
              'trimok': (4 * (32 + 20 * 32) + (32 + 20 * 32) + 160 + 25 * 16 + 16, 4 * 25 + 16 + 80 + 80 + 32)}


def tree_roofline_other(workload, prof, B):
    if prof is None or workload not in TREE_BYTES or not prof['select_n']:
        return None
    sb, bb = TREE_BYTES[workload]
    sel_us = prof['select_ms'] * 1e3 / prof['select_n']; bak_us = prof['backup_ms'] * 1e3 / max(prof['backup_n'], 1)
    g = sb * B / (sel_us * 1e-6) / 1e9
    return {'kernel': 'k_select<%s>' % {'brandubh': 'BR', 'trimok': 'TM'}[workload], 'bound': 'hbm', 'achieved': round(g, 2), 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': round(g / HBM_PEAK_GBS, 6), 'avg_launch_us': round(sel_us, 2), 'algorithmic_bytes_per_launch': sb * B,
            'backup_us': round(bak_us, 2), 'backup_GBps': round(bb * B / (bak_us * 1e-6) / 1e9, 2), 'traffic': None}


def run_other_workload(a, rank, local_rank, world):
    """configs 3-5: not the headline bench line; same timing protocol, reported with their own config.workload."""
    import importlib
    from alphazero_general_amd import nnet as nn_mod
    from alphazero_general_amd.selfplay import ArenaRunner
    gmod, netargs, B, sims, cpuct, fpu = WORKLOADS[a.workload]
    B = a.slots or B
    Game_ = importlib.import_module('alphazero_general_amd.envs.' + gmod).Game
    dev = torch.device('cuda', local_rank)
    args = selfplay_args(1 << 30)
    args.update(cpuct=cpuct, fpu_reduction=fpu, numMCTSSims=sims)
    if a.workload == 'arena':
        nets = []
        for sd in (0, 1):                                            # two differently seeded random-init nets
            torch.manual_seed(sd)
            nets.append(NNetWrapper(Game_, getattr(nn_mod, netargs), device=dev, dtype=torch.float16))
        runner = ArenaRunner(Game_, nets, args, num_slots=B, seed=0, slot_base=D.slot_base(rank, B), device=local_rank,
                             result_capacity=B * (a.steps + a.warmup + 8) // 5 + 2 * B)
        counters = lambda: runner.engine.counters()
    else:
        torch.manual_seed(0)
        net = NNetWrapper(Game_, getattr(nn_mod, netargs), device=dev, dtype=torch.float16)
        per_game = (Game_.max_turns() + 1) * len(Game_().symmetries(np.zeros(Game_.action_size(), np.float32)))
        runner = SelfPlayRunner(Game_, net, args, num_slots=B, seed=0, slot_base=D.slot_base(rank, B), device=local_rank,
                                example_capacity=int(B * (a.steps + a.warmup + 4) / 5.0 + 2 * B) * per_game)
        counters = lambda: runner.counters()
    if hasattr(runner, 'prepare'):
        runner.prepare()                                             # graph capture stays out of the timed region even at --warmup 0
    for _ in range(a.warmup):
        runner.play_round()
    c0 = counters()
    D.barrier(); torch.cuda.synchronize()
    t0 = time.time()
    prof = None
    eng = getattr(runner, 'engine', None)
    for k in range(a.steps):
        timed = a.workload != 'arena' and eng is not None and k == a.steps // 2      # one eagerly launched round with HIP events
        if timed:
            torch.cuda.synchronize(); eng.profile(True)
        runner.play_round()
        if timed:
            prof = eng.profile_read(); eng.profile(False)
    c1 = counters()
    torch.cuda.synchronize(); D.barrier()
    dt = D.max_over_ranks(time.time() - t0)
    tall = D.all_reduce_tallies([c1['expansions'] - c0['expansions'], c1['sims'] - c0['sims'], c1['games_played'] - c0['games_played']])
    if rank == 0:
        exp, sm, gm = [int(x) for x in tall]
        print(json.dumps({'metric': 'mcts_node_expansions_per_sec', 'value': round(exp / dt, 1), 'unit': 'expansions/s',
                          'n_gpus': a.gpus, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt * 1e3 / a.steps, 3),
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 tree / f16 net',
                          'data': 'synthetic', 'games_per_sec': round(gm / dt, 2), 'simulations_per_sec': round(sm / dt, 1),
                          'config': {'workload': '%s, %d games/GPU x %d sims/move, net %s' % (a.workload, B, sims, netargs)},
                          'tree_roofline': tree_roofline_other(a.workload, prof, B)}))


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (RCCL), same
    arguments; rank 0 of the child job prints the JSON line.  Returns the job's exit code."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=45)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--slots', type=int, default=0)
    ap.add_argument('--workload', default='connect4', choices=sorted(WORKLOADS))
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--pipelines', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fused-search', action='store_true',
                    help='2 launches per simulation (tower, backup+select) instead of one persistent launch per move (azg_search_f16)')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(a.gpus)                                    # one rank per GPU under torch.distributed.run
    rank, local_rank, world = D.init_from_env()
    assert world == a.gpus, '--gpus %d but %d rank(s) were launched (torch.distributed.run --nproc-per-node must equal --gpus)' % (a.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a HIP device (there is no CPU fallback)'
    if not os.environ.get('AZG_SINGLE_DEVICE'):
        assert torch.cuda.device_count() >= world, '%d ranks but only %d visible GPU(s)' % (world, torch.cuda.device_count())
    a.gpus = world                                                    # n_gpus in the output = the ranks actually initialised
    torch.cuda.set_device(local_rank)
    if a.workload != 'connect4':
        return run_other_workload(a, rank, local_rank, world)
    dev = torch.device('cuda', local_rank)
    torch.manual_seed(0)                                            # same random-init weights on every rank
    B = a.slots or B_PER_GPU
    net = NNetWrapper(Game, CONNECT4_NET_ARGS, device=dev, dtype=torch.float16)
    games = 1 << 30
    per_game = 43 * 2
    runner = SelfPlayRunner(Game, net, selfplay_args(games), num_slots=B, seed=0, slot_base=D.slot_base(rank, B),
                            device=local_rank, use_graph=not a.no_graph, pipelines=a.pipelines, fused_search=False if (a.no_fused_search or a.pipelines > 1) else None,
                            example_capacity=int(B * (a.steps + a.warmup + 8) / 7.0 + 2 * B) * per_game)
    eng = runner.engine
    lanes = runner.lanes
    runner.prepare()                                                 # graph capture stays out of the timed region even at --warmup 0
    for _ in range(a.warmup):
        runner.play_round()
    c0 = runner.counters()
    ex0 = [ln.engine.counters()['num_examples'] for ln in lanes]
    ev_nn, ev_search = [], []
    D.barrier(); torch.cuda.synchronize()
    t0 = time.time()
    for k in range(a.steps):
        if k == a.steps // 2:                                       # HIP-event timing of the kernels for ONE round
            torch.cuda.synchronize()
            if runner.use_graph:                                    # (events around every launch perturb the pipeline)
                with torch.cuda.stream(lanes[0].stream):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for _ in range(8):                               # warm launch path and clocks (a short burst after a pause
                        lanes[0].net.replay()                        #  runs at boost clock: 0.24 ms instead of the sustained 0.28)
                    e0.record()
                    for _ in range(NN_REPS):
                        lanes[0].net.replay()
                    e1.record(); ev_nn.append((e0, e1))
                torch.cuda.synchronize()
            eng.profile(True)
        if runner.fused_search and k == a.steps // 2 + 1:           # HIP events around ONE search launch (= all simulations of a move)
            with torch.cuda.stream(lanes[0].stream):
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record(); net._hip.search(eng, SIMS); s1.record()
                eng.advance(True)
            runner.sims_per_round.append(SIMS); runner._actr += 1    # (probFastSim = 0: the agent-level coin of this round is unused)
            ev_search.append((s0, s1))
        else:
            runner.play_round()
        if k == a.steps // 2:
            prof = eng.profile_read()
            eng.profile(False)
    c1 = runner.counters()
    # the exchange step of an iteration: all-gather the example shards (RCCL) + tallies
    obs, pi, z = runner.samples(ex0)
    gobs, gpi, gz = D.all_gather_examples(obs, pi, z)
    torch.cuda.synchronize(); D.barrier()
    dt = D.max_over_ranks(time.time() - t0)
    tall = D.all_reduce_tallies([c1['expansions'] - c0['expansions'], c1['sims'] - c0['sims'],
                                 c1['games_played'] - c0['games_played'], gobs.shape[0] if rank == 0 else 0])
    if rank != 0:
        return
    expansions, sims, games_done, nsamples = [int(x) for x in tall]
    sel_us = prof['select_ms'] * 1e3 / max(prof['select_n'], 1)
    bak_us = prof['backup_ms'] * 1e3 / max(prof['backup_n'], 1)
    adv_us = prof['advance_ms'] * 1e3 / max(prof['advance_n'], 1)
    Bl = B // a.pipelines                                           # slots per launch
    sel_gbs = C4_SELECT_BYTES_PER_SIM * Bl / (sel_us * 1e-6) / 1e9
    nn_ms = ev_nn[0][0].elapsed_time(ev_nn[0][1]) / NN_REPS if ev_nn else None
    tree = {'kernel': 'k_select<C4>', 'bound': 'hbm', 'achieved': round(sel_gbs, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(sel_gbs / HBM_PEAK_GBS, 6),
            # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof_tree.py):
            # 2 x FETCH_SIZE (gfx950 wide-load correction) + WRITE_SIZE, profiles/r01_pmc_summary.csv; 2048-slot launch only
            'traffic': 5995000 if Bl == 2048 else None, 'avg_launch_us': round(sel_us, 2),
            'algorithmic_bytes_per_launch': C4_SELECT_BYTES_PER_SIM * Bl,
            'backup_us': round(bak_us, 2), 'advance_us': round(adv_us, 2),
            'backup_GBps': round(C4_BACKUP_BYTES_PER_SIM * Bl / (bak_us * 1e-6) / 1e9, 2)}
    if nn_ms is not None:
        # the dominant kernel of the step (~94 % of the time): the network tower + heads, ONE launch (k_tower2), MFMA-bound
        tf = C4_NET_FLOPS_PER_LEAF * Bl / (nn_ms * 1e-3) / 1e12
        roof = {'kernel': 'k_tower2<6,7,4> (ResNet 128ch x 8 tower + heads, one launch per evaluation)', 'bound': 'mfma',
                'achieved': round(tf, 1), 'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / MFMA_F16_PEAK_TFLOPS, 4),
                'avg_launch_us': round(nn_ms * 1e3, 1), 'algorithmic_flops_per_launch': C4_NET_FLOPS_PER_LEAF * Bl,
                # HBM bytes per launch (PMC, profiles/r01_pmc_summary.csv): weights + input planes + probabilities
                'traffic': TOWER_TRAFFIC_BYTES if Bl == 2048 else None}
    else:
        roof = tree
    if ev_search:
        # default path: the whole simulation loop of a move is ONE persistent launch (azg_search_f16): tree walk, tower + heads and
        # backup of every simulation.  Its FLOPs are the network's; the tree phases ride inside the launch time.
        sms = ev_search[0][0].elapsed_time(ev_search[0][1])
        stf = C4_NET_FLOPS_PER_LEAF * Bl * SIMS / (sms * 1e-3) / 1e12
        roof = {'kernel': 'k_tower2<6,7,4,128,1,SearchArgs<C4>> (azg_search_f16: %d x [find_leaf, ResNet 128ch x 8 + heads, backup] '
                          'on every game, one persistent launch per move)' % SIMS, 'bound': 'mfma', 'achieved': round(stf, 1),
                'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(stf / MFMA_F16_PEAK_TFLOPS, 4),
                'avg_launch_us': round(sms * 1e3, 1), 'algorithmic_flops_per_launch': C4_NET_FLOPS_PER_LEAF * Bl * SIMS,
                # PMC per launch (2 x FETCH_SIZE + WRITE_SIZE, profiles/r01_pmc_summary.csv rows search_r1d): 72 MB per simulation
                # = the tower's 51 MB (weight stream per XCD) + the trees' ~8 MB + ~13 MB of spill traffic
                'traffic': SEARCH_TRAFFIC_BYTES if (Bl == 2048 and SIMS == 100) else None,
                # the same tower + heads as its own launch (one evaluation), timed as a burst of NN_REPS launches: after the
                # lighter search launches the chip boosts, so this runs faster than the same kernel does in a sustained stream
                # (0.28 ms = 60 % with --no-fused-search, where it is launched 100 times per move)
                'net_eval_only': roof}
    out = {
        'metric': 'mcts_node_expansions_per_sec', 'value': round(expansions / dt, 1), 'unit': 'expansions/s',
        'n_gpus': a.gpus, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt * 1e3 / a.steps, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 tree / f16 net', 'data': 'synthetic',
        'config': {'workload': 'connect4 self-play, %d games/GPU x %d sims/move, fp16 ResNet 128ch x 8, random-init, noise+temp on'
                               % (B, SIMS), 'games_per_gpu': B, 'sims_per_move': SIMS, 'hipgraph_net': bool(runner.use_graph),
                   'stream_pipelines': a.pipelines, 'fused_search_launch': bool(runner.fused_search)},
        'games_per_sec': round(games_done / dt, 2), 'simulations_per_sec': round(sims / dt, 1),
        'games_finished': games_done, 'samples_gathered': nsamples,
        'roofline': roof, 'tree_roofline': tree,
    }
    if nn_ms is not None:
        # context for the MFMA fraction: the best plain fp16 GEMM the vendor library reaches on this very GPU (outside the
        # timed region; SURVEY.md 8d asks for it next to the datasheet peak)
        lib_tf = library_gemm_tflops(dev)
        tgt = out['roofline'].get('net_eval_only', out['roofline'])
        tgt['library_gemm_tflops'] = round(lib_tf, 1)
        tgt['vs_library_gemm'] = round(tf / lib_tf, 3)
    if a.gpus == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(net)
    print(json.dumps(out))


if __name__ == '__main__':
    try:
        rc = main()
    finally:
        D.shutdown()
    sys.exit(rc or 0)
