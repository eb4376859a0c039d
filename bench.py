"""bench.py -- self-play throughput of the MI355X engine on BASELINE.json's configurations.

    python bench.py [--workload connect4|brandubh|arena|trimok] --gpus N --steps K --warmup W

Default workload = the headline config (BASELINE.json configs[1]): connect4, 2048 concurrent games per GPU x 100 MCTS sims per
move, fp16 ResNet 128ch x 8 (envs/connect4/train.py net + search hyper-parameters), random-init weights, games from the empty
board, root noise + root temperature on, probFastSim = 0.  The other workloads are configs 3-5 at their
per-GPU size.  One "step" = one self-play round of the hot path over the whole batch: sims x [find_leaf -> network ->
process_results] + playMoves.  value = MCTS node expansions per second summed over all GPUs (games/s next to it).  Everything is
resident in HBM; the only host traffic in the timed region is one counter read per run.

N > 1: `python bench.py --gpus N` starts its own N ranks (torch.distributed.run, one per GPU, RCCL); under an external
torch.distributed.run it uses the ranks it was given.  n_gpus in the output is the number of ranks that actually ran.

Measured inside the run (HIP events on the launch stream, rounds launched eagerly AFTER the timed region, which holds product rounds only):
`roofline` = the dominant kernel of a simulation step, `tree_roofline` = the tree launch.  `traffic` (HBM bytes per launch) comes
from the committed rocprofv3 --pmc summary profiles/r03_pmc.json (tools/collect_profiles.py, run on the GPU box), never from a
constant in this file.  `cpu_baseline` (rank 0, N = 1 only) = the C oracle on the host cores, bounded sample.
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
from alphazero_general_amd import nnet as nn_mod  # noqa: E402
from alphazero_general_amd.nnet import HipResNet, NNetWrapper  # noqa: E402
from alphazero_general_amd.iteration import ArenaIteration, SelfPlayIteration  # noqa: E402  (the library's iteration objects: per-rank runner + exchange step)
from alphazero_general_amd.utils import dotdict, default_temp_scaling  # noqa: E402

HBM_PEAK_GBS, MFMA_F16_PEAK_TFLOPS = 8000.0, 2500.0                                   # MI355X_MICROARCH.md
PMC_FILE = next((p for p in (os.path.join(ROOT, 'profiles', 'r%02d_pmc.json' % r) for r in (6, 5, 4, 3, 2)) if os.path.exists(p)),
                os.path.join(ROOT, 'profiles', 'r06_pmc.json'))
PHASE_FILE = next((p for p in (os.path.join(ROOT, 'profiles', 'r%02d_phase_budget.json' % r) for r in (6, 5, 4)) if os.path.exists(p)),
                  os.path.join(ROOT, 'profiles', 'r06_phase_budget.json'))
CALIBRATION_FILE = os.path.join(ROOT, 'profiles', 'cpu_oracle_vs_reference.json')
CLOCK_GHZ_NOMINAL = 2.4                                                               # MI355X_MICROARCH.md (peak engine clock)


def csrc_sha():
    """content hash of everything the library is built from -- csrc/*.h, csrc/*.hip, include/azg.h, the compile flags
    (alphazero_general_amd.build.source_sha) -- over the WORKING TREE: the committed PMC summary and phase budget are stamped with
    the hash of the sources they were measured on (tools/collect_profiles.py); a counter taken on other kernels is not quoted next
    to this run's launch time (there is no .git on the GPU box, so the stamp is a content hash, not a commit)."""
    from alphazero_general_amd.build import source_sha
    return source_sha()

# name: game module, net args, games / GPU, sims / move, cpuct, fpu reduction, typical (children, depth) of the tree bytes model
WORKLOADS = {
    'connect4': dict(game='connect4', net='CONNECT4_NET_ARGS', B=2048, sims=100, cpuct=4.0, fpu=0.4, kbar=7, depth=5, oracle_game=0),
    'brandubh': dict(game='brandubh', net='BRANDUBH_NET_ARGS', B=512, sims=200, cpuct=1.25, fpu=0.2, kbar=40, depth=4, oracle_game=1),
    'trimok': dict(game='trimok', net='DEFAULT_NET_ARGS', B=256, sims=50, cpuct=1.25, fpu=0.2, kbar=20, depth=4, oracle_game=2),
    'arena': dict(game='connect4', net='CONNECT4_NET_ARGS', B=256, sims=100, cpuct=4.0, fpu=0.4, kbar=7, depth=5, oracle_game=0),
}


WORKLOAD_TOTAL_GAMES = {'brandubh': 4096, 'trimok': 1024, 'arena': 512}              # BASELINE.json configs 3, 5, 4 (whole job)


def selfplay_args(W, games=1 << 30):
    return dotdict(cpuct=W['cpuct'], fpu_reduction=W['fpu'], root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1.0,
                   numMCTSSims=W['sims'], numFastSims=20, numWarmupSims=5, probFastSim=0.0, gamesPerIteration=games,
                   add_root_noise=True, add_root_temp=True, symmetricSamples=True, mctsResetThreshold=None,
                   startTemp=1.0, arenaTemp=0.25, temp_scaling_fn=default_temp_scaling)


# ---------------------------------------------------------------------------------------------- algorithmic figures
def net_flops_per_leaf(game_cls, a):
    """FLOPs of one evaluation of the reference's ResNet (NNetArchitecture.py:69-120), 2 per multiply-add: stem + 2 convs per
    block + the two 1x1 head convs + the dense chains (205 MFLOP for connect4 128ch x 8)."""
    C, H, W = game_cls.observation_size()
    hw, ch = H * W, a.num_channels
    f = 2 * 9 * C * ch * hw + a.depth * 2 * (2 * 9 * ch * ch * hw)
    for heads, dense, out in ((a.value_head_channels, a.value_dense_layers, game_cls.num_players() + game_cls.has_draw()),
                              (a.policy_head_channels, a.policy_dense_layers, game_cls.action_size())):
        f += 2 * ch * heads * hw
        sizes = [hw * heads] + list(dense) + [out]
        f += sum(2 * sizes[i] * sizes[i + 1] for i in range(len(sizes) - 1))
    return float(f)


def tree_bytes_per_sim(game_cls, W, feat_k=None):
    """Algorithmic bytes one simulation moves through the tree launch (SURVEY.md 8d formula with this build's 32-byte node records,
    64-byte tree header, 16-byte path entries): find_leaf = header + D child blocks read + one child block written + root and leaf
    state (2 x 80) + the fp16 NHWC8 observation + D path entries; process_results = header + policy and value row + k priors
    written + D x (path entry read + (n, q) written).  feat_k: the launch is fed head features and computes its logits itself
    (sparse heads): the row it reads is the board's fp16 [2][feat_k] features instead of A + P+1 logits; the head matrix
    (A + P+1 rows of feat_k halves, shared by every board of the launch) is returned separately, counted once per launch."""
    k, Dp = W['kbar'], W['depth']
    C, H, Wd = game_cls.observation_size()
    A, NV = game_cls.action_size(), game_cls.num_players() + 1
    select = 64 + Dp * k * 32 + (32 + k * 32) + 2 * 80 + H * Wd * 16 + Dp * 16
    row = 4 * (A + NV) if feat_k is None else 2 * feat_k * 2
    backup = 64 + row + 4 * k + k * 2 + Dp * (16 + 8) + 12
    shared = 0 if feat_k is None else (A + NV) * (feat_k * 2 + 4)
    return select, backup, shared


def load_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def profile_key(c):
    """the key of a workload in the committed PMC / phase files: its name at the default shard size, name_<games> at another"""
    return c.name if c.B == WORKLOADS[c.name]['B'] else '%s_%d' % (c.name, c.B)


def lib_source_sha():
    """the kernel sources the LOADED library was built from (azg_source_sha, stamped by build.py)"""
    from alphazero_general_amd import _abi
    try:
        return _abi.lib().azg_source_sha().decode()
    except Exception:                                                # noqa: BLE001
        return None


def measured_traffic(workload, kernel_substr):
    """HBM bytes per launch of a kernel from the committed PMC summary (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes;
    MI355X_MICROARCH.md 'HBM': FETCH_SIZE counts half the bytes of wide loads on gfx950).  None when no profile is committed."""
    pmc = load_json(PMC_FILE)
    if not pmc:
        return None, None, None
    same = pmc.get('csrc_sha') == csrc_sha() == lib_source_sha()      # were the counters taken on the kernels this run's binary holds?
    # (the set-up of a persistent wide launch times every tile shape once: those trial launches -- two dispatches each -- are in the profile
    #  too; the kernel the rounds ran is the one with the most dispatches)
    cands = [(rec.get('dispatches', 0), name, rec) for name, rec in pmc.get('workloads', {}).get(workload, {}).items() if kernel_substr in name]
    if cands:
        _, name, rec = max(cands, key=lambda c: c[0])
        return (int(rec['traffic_bytes']), 'profiles/%s @%s (%s, %d dispatches%s)' % (os.path.basename(PMC_FILE), pmc.get('git', '?'), name, rec.get('dispatches', 0),
                                                                                      '' if same else '; kernel sources have changed since'),
                rec.get('mfma_busy_cycles') if same else None)
    return None, None, None


def library_gemm_tflops(dev, n=8192, reps=10):
    """best plain fp16 n^3 GEMM through torch (hipBLASLt) on this GPU, TFLOP/s: context for the MFMA fraction."""
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


# ---------------------------------------------------------------------------------------------- CPU baseline (rank 0, N = 1)
def cpu_baseline(W, nets, arena=False, budget_s=24.0):
    """The CPU path timed beside the GPU number: the C oracle -- the bit-exact restatement of the reference's
    Cython MCTS / SelfPlayAgent, oracle/ -- with one agent per host core (oracle/azg_pool_ref.c: the reference's `workers`
    processes, Coach.py:291-342), same game, network and search parameters, same number of concurrent games.
      end to end: every simulation step the agents' leaves form ONE batch evaluated by the same GPU network through host
                  buffers (the reference's arrangement, Coach.py:337-342), so only where the tree lives differs;
      tree only : the warm-up evaluator's uniform policy / value (SelfPlayAgent.pyx:48-52), agents free-running.
    Thread counts from all logical host CPUs down to half the container's CPU quota (cgroup cpu.max; the GPU boxes give this container
    16 CPUs' worth of time on a 256-CPU host) are sampled for ~2 s each and the best is reported (`cores` = threads used)."""
    import oracle_lib as ol
    ncpu = os.cpu_count() or 1
    quota = None                                                     # the container's CPU-time quota (cgroup v2 cpu.max), in CPUs
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        quota = None if q == 'max' else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    G, games, sims = W['oracle_game'], W['B'], W['sims']
    kw = dict(sims=sims, games_per_iteration=1 << 30, seed=1, cpuct=W['cpuct'], fpu_reduction=W['fpu'], add_root_noise=not arena,
              add_root_temp=not arena, is_arena=arena)
    net = nets[0]

    def evaluate(pool, obs):
        if not arena:
            p, v = net.process(torch.from_numpy(obs))
            return p.cpu().numpy(), v.cpu().numpy()
        rm = pool.row_models()                                       # every model evaluates its own rows (Arena.pyx:262-281)
        p = np.zeros((obs.shape[0], pool.A), np.float32); v = np.zeros((obs.shape[0], pool.NV), np.float32)
        for m, nn_ in enumerate(nets):
            idx = np.flatnonzero(rm == m)
            if len(idx):
                pm, vm = nn_.process(torch.from_numpy(obs[idx]))
                p[idx], v[idx] = pm.cpu().numpy(), vm.cpu().numpy()
        return p, v

    qn = int(quota) if quota else 16
    counts = sorted({max(1, min(n, games)) for n in (ncpu, 4 * qn, 2 * qn, qn, max(qn // 2, 1))}, reverse=True)
    per = max(1.5, budget_s / (2 * len(counts) + 1))
    best_e2e, best_tree, tried = None, None, []
    for n in counts:
        Bc = (games + n - 1) // n                                    # games per agent: the config's concurrent games spread over n cores
        pool = ol.OPool(G, n, Bc, **kw)
        # ---- end to end ----
        t0 = time.time(); e0 = pool.expansions
        while time.time() - t0 < per:
            pool.begin_round()
            for _ in range(sims):
                p, v = evaluate(pool, pool.generate())
                pool.process(p, v)
            pool.play()
        dt = time.time() - t0
        e2e = (pool.expansions - e0) / dt
        # ---- tree only ----
        e1 = pool.expansions
        dt2 = pool.run_tree_only(per)
        tree = (pool.expansions - e1) / dt2
        tried.append({'threads': n, 'games_per_thread': Bc, 'end_to_end': round(e2e, 1), 'tree_only': round(tree, 1)})
        if best_e2e is None or e2e > best_e2e[0]:
            best_e2e = (e2e, n, Bc, dt)
        if best_tree is None or tree > best_tree[0]:
            best_tree = (tree, n, Bc, dt2)
        del pool
    # one core, as the per-core figure
    one = ol.OPool(G, 1, min(games, 256), **kw)
    dt1 = one.run_tree_only(per)
    tree1 = one.expansions / dt1
    out = {'value': round(best_e2e[0], 1), 'unit': 'expansions/s', 'cores': best_e2e[1], 'kind': 'port', 'host_cpus': ncpu, 'cgroup_cpu_quota': quota,
           'sample': '%s%s: %d oracle agents x %d games x %d sims/move on %d host threads for %.1f s, leaves of all agents evaluated as one '
                     'batch by the same GPU net(s) through host buffers' % (W['game'], ' arena' if arena else '', best_e2e[1], best_e2e[2], sims, best_e2e[1], best_e2e[3]),
           'tree_only': {'value': round(best_tree[0], 1), 'cores': best_tree[1], 'per_core_1_thread': round(tree1, 1),
                         'sample': '%d agents x %d games, uniform evaluator, free-running for %.1f s' % (best_tree[1], best_tree[2], best_tree[3])},
           'thread_counts_tried': tried}
    cal = load_json(CALIBRATION_FILE)
    if cal:
        out['vs_reference_cython'] = cal                             # BASELINE.md section 3 calibration (tools/calibrate_oracle.py)
    return out


# ---------------------------------------------------------------------------------------------- launcher
def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (RCCL), same
    arguments; rank 0 of the child job prints the JSON line.  Returns the job's exit code."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # (NCCL_DEBUG=VERSION: RCCL prints its version line once on stderr -- the first thing to read when an N-GPU run misbehaves)
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                                        NCCL_DEBUG=os.environ.get('NCCL_DEBUG', 'VERSION'))).returncode


class Ctx:
    """one workload set up on this rank's GPU: runner, network(s), engines"""


def build(workload, a, rank, local_rank, dev, rounds, slots=0, search_heads=None):
    import importlib
    c = Ctx()
    c.name = workload
    c.W = W = dict(WORKLOADS[workload])
    c.B = W['B'] = slots or (a.slots if workload == a.workload else 0) or W['B']
    c.sims = W['sims']
    c.Game = Game = importlib.import_module('alphazero_general_amd.envs.' + W['game']).Game
    netargs = getattr(nn_mod, W['net'])
    c.arena = workload == 'arena'
    args = selfplay_args(W)
    nsym = len(Game().symmetries(np.zeros(Game.action_size(), np.float32)))
    c.pipelines = a.pipelines if workload == a.workload else 1
    if c.arena:
        c.nets = []
        for sd in (0, 1):                                            # two differently seeded random-init nets
            torch.manual_seed(sd)
            c.nets.append(NNetWrapper(Game, netargs, device=dev, dtype=torch.float16))
        c.net = c.nets[0]
        c.iter = ArenaIteration(Game, c.nets, args, 1 << 30, num_slots=c.B, seed=0, seats='agent', device=local_rank,
                                use_graph=not a.no_graph, fused_search=False if a.no_fused_search else None, result_capacity=c.B * rounds // 5 + 2 * c.B)
        c.runner = c.iter.runner
        c.engines = [c.runner.engine]
        c.counters = c.runner.engine.counters
        c.fused_search = bool(c.runner.fused_search)
        c.search_heads = None
    else:
        torch.manual_seed(0)                                         # same random-init weights on every rank
        c.net = NNetWrapper(Game, netargs, device=dev, dtype=torch.float16)
        c.nets = [c.net]
        per_game = (Game.max_turns() + 1) * nsym
        c.iter = SelfPlayIteration(Game, c.net, args, num_slots=c.B, seed=0, device=local_rank,
                                   use_graph=not a.no_graph, pipelines=c.pipelines,
                                   fused_search=False if (a.no_fused_search or c.pipelines > 1) else None, search_heads=search_heads or a.search_heads,
                                   example_capacity=int(c.B * rounds / 5.0 + 2 * c.B) * per_game)
        c.runner = c.iter.runner
        c.engines = [ln.engine for ln in c.runner.lanes]
        c.counters = c.runner.counters
        c.fused_search = bool(c.runner.fused_search)
        c.search_heads = ('exact' if c.runner.search_exact else 'sparse') if (c.fused_search and c.net._hip is not None and c.net._hip.fact_head) else None
        c.runner.prepare()                                           # graph capture stays out of the timed region even at --warmup 0
    return c


def timed_region(c, steps, warmup, world, rank):
    """W warm-up rounds, then EXACTLY `steps` rounds -- every one the product's launch form (a replayed hipGraph unless --no-graph)
    -- and the iteration's exchange step, bracketed by barrier + synchronize; nothing else runs inside.  The rounds and the exchange
    are the library's own (alphazero_general_amd.iteration: what run_iteration / run_arena and the Coach adapter execute)."""
    c.rounds_before = getattr(c, 'rounds_played', 0)                 # (rounds this runner has played before this region)
    for _ in range(warmup):
        c.iter.play_round()
    c.rounds_played = c.rounds_before + warmup + steps
    c.iter.begin()                                                   # the iteration's marks: what exchange() hands over is what follows
    if world > 1 and not c.arena:                                    # (the collectives' one-time set-up stays out of the timed region)
        o_, p_, z_ = c.iter.local_samples()
        D.all_gather_examples(o_[:1], p_[:1], z_[:1])
    D.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        c.iter.play_round()
    torch.cuda.synchronize()
    t_search = time.perf_counter() - t0                                      # this rank's own rounds (before it waits for anybody)
    ex = c.iter.exchange()                                           # all-gather of the example shards + tallies (arena: tallies only)
    torch.cuda.synchronize()
    t_exchange = time.perf_counter() - t0 - t_search
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    r = dict(dt=dt, steps=steps, rank_ms_per_step_max=D.max_over_ranks(t_search) * 1e3 / steps,
             rank_ms_per_step_min=-D.max_over_ranks(-t_search) * 1e3 / steps, exchange_ms=D.max_over_ranks(t_exchange) * 1e3)
    r['expansions'], r['sims'], r['games'], r['samples'] = ex['expansions'], ex['sims'], ex['games'], ex.get('num_samples', 0)
    # steady state: all games start from the empty board at round 0, so the first finishes come in a burst; every slot plays one
    # move per round, so a game's finishing round is the running sum of its slot's game lengths (result records, read after the
    # region) -- games that finished in the SECOND half of the timed rounds, over that half's share of the time
    half = steps // 2
    late = 0
    if half > 0:
        _, turns, slot = c.runner.results() if not c.arena else c.runner.engine.results()
        rounds_done = warmup + steps + c.rounds_before
        end = {}
        for t, sl in zip(turns.tolist(), slot.tolist()):
            end[sl] = end.get(sl, 0) + int(t)
            if end[sl] > rounds_done - half:
                late += 1
    late = int(D.all_reduce_tallies([late])[0])
    r['games_second_half'], r['second_half_steps'] = late, half
    return r


def profile_rounds(c, persistent_rounds=10):
    """OUTSIDE the timed region: rounds launched eagerly with the library's profile hooks on (single kernels go through
    hipExtLaunchKernelGGL, whose events carry the dispatch's own begin / end timestamps).  With a persistent search launch:
    `persistent_rounds` rounds of it, then ONE round of the launch-per-phase form of the same move loop, which shows the tree
    launch and the tower launch on their own; otherwise one round."""
    netprof, prof = {}, None
    if c.pipelines != 1:
        return netprof, prof
    kinds = ['search'] * persistent_rounds + ['phase'] if c.fused_search else ['phase']
    for kind in kinds:
        torch.cuda.synchronize()
        if c.fused_search and kind == 'phase':
            c.runner.fused_search = False
        c.engines[0].profile(True); HipResNet.profile(True)
        c.runner.play_round(eager=True)
        rd = HipResNet.profile_read()
        for kk, vv in rd.items():                                    # (GPU ms and launch counts per family: tower, wide heads, search)
            netprof[kk] = netprof.get(kk, 0) + vv
        if kind == 'search' and rd.get('search_n'):                  # one persistent launch per round: its own duration, launch by launch
            netprof.setdefault('search_each_us', []).append(rd['search_ms'] * 1e3 / rd['search_n'])
        HipResNet.profile(False)
        if kind == 'phase':
            prof = c.engines[0].profile_read()
        c.engines[0].profile(False)
        if c.fused_search and kind == 'phase':
            c.runner.fused_search = True
    torch.cuda.synchronize()
    return netprof, prof


def rooflines(c, netprof, prof):
    """roofline of the network launch (MFMA) and of the tree launch (HBM) from the profile rounds' events"""
    W, Game, net, arena, sims = c.W, c.Game, c.net, c.arena, c.sims
    Bl = c.B // c.pipelines                                          # slots per launch
    flops_leaf = net_flops_per_leaf(Game, net.args)

    def issued_frac():
        """MFMA work the 4-board connect4 tile actually issues / the algorithmic count: border-class subtiles drop the taps
        that only read zero padding, on 176 lanes for 168 pixels"""
        if c.name != 'connect4':
            return None
        _, H, Wd = Game.observation_size()
        sub = lambda n: (n + 15) // 16
        n5 = [sub(4 * (H - 2) * (Wd - 2)), sub(4 * Wd), sub(4 * (H - 2))]
        return round((n5[0] * 9 + 2 * n5[1] * 6 + 2 * n5[2] * 6) * 16 / (9.0 * 4 * H * Wd), 4)

    def mfma_roof(fam, kname, kmatch, units):
        n = int(netprof.get(fam + '_n', 0))
        if not n:
            return None
        us = netprof[fam + '_ms'] * 1e3 / n
        each = sorted(netprof.get(fam + '_each_us', []))
        tf = flops_leaf * units / (us * 1e-6) / 1e12
        traffic, src, busy = measured_traffic(profile_key(c), kmatch)
        r = {'kernel': kname, 'bound': 'mfma', 'achieved': round(tf, 1), 'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
             'frac': round(tf / MFMA_F16_PEAK_TFLOPS, 4), 'avg_launch_us': round(us, 2), 'launches_timed': n,
             # (the timed launches one by one: avg_launch_us <= ms_per_step of the graph-replayed rounds can be checked against the spread)
             'launch_us_min_median_max': [round(each[0], 2), round(each[len(each) // 2], 2), round(each[-1], 2)] if each else None,
             'algorithmic_flops_per_launch': flops_leaf * units, 'mfma_issued_over_algorithmic': issued_frac(),
             'traffic': traffic, 'traffic_source': src}
        if fam == 'search':
            r['operand_stream'] = operand_stream(us, units // max(Bl, 1))
        if busy:
            # FLOPs the MFMA pipes really executed in the profiled launch (SQ_VALU_MFMA_BUSY_CYCLES / 16 cycles per v_mfma_f32_16x16x32_f16
            # x 16 384 FLOP) over THIS run's launch time: what `frac` would be without credit for the skipped zero products
            ex = busy / 16.0 * 16384.0
            r['executed_flops_per_launch'] = ex
            r['executed_frac'] = round(ex / (us * 1e-6) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)
        return r

    def operand_stream(us, nsims):
        """The bound that applies to a SMALL-shard persistent launch (one or two boards per workgroup): every workgroup streams the whole
        parameter set -- tower fragments + head operands, HipResNet.stream_bytes -- from L2 through its CU's L1 miss path once per
        simulation (nothing of it fits LDS beside the image), whatever the MFMA pipes could do with it.  achieved = bytes x workgroups x
        simulations / launch time; peak = 64 B/clk/CU (the path's width) x the nominal clock x the device's CUs; the ceiling MEASURED on this
        access pattern is 46-57 B/clk/CU (profiles/notes_r02_r04_experiments.md section 3d), i.e. frac 0.72-0.89 is the practical limit."""
        hip = net._hip
        if hip is None:
            return None
        exact = c.search_heads != 'sparse'
        tile = hip.search_tile(c.engines[0], exact=exact) if (hip.fact_head and not arena) else None
        cus = torch.cuda.get_device_properties(c.engines[0].device).multi_processor_count
        gpw = tile['games_per_workgroup'] if tile else (1 if arena or 2 * Bl <= 5 * cus else 2 if Bl <= 5 * cus else 4)   # (connect4 x 128: azg_search_f16's rule)
        sb = hip.stream_bytes(exact=exact, kbar=W['kbar'])
        nwg = (Bl + gpw - 1) // gpw
        total = float(sb['total']) * nwg * nsims
        gbs = total / (us * 1e-6) / 1e9
        peak = 64.0 * CLOCK_GHZ_NOMINAL * cus
        return {'bound': 'l2->l1 operand stream: every workgroup re-reads the network parameters once per simulation', 'games_per_workgroup': gpw,
                'tile': tile, 'workgroups': nwg, 'bytes_per_workgroup_per_sim': sb, 'bytes_per_launch': total, 'achieved': round(gbs, 1),
                'peak': round(peak, 1), 'unit': 'GB/s', 'frac': round(gbs / peak, 4), 'B_per_clk_per_cu_at_nominal_clock': round(gbs / (CLOCK_GHZ_NOMINAL * cus), 2),
                'peak_B_per_clk_per_cu': 64, 'measured_ceiling_B_per_clk_per_cu': [46, 57], 'cus': cus, 'clock_ghz': CLOCK_GHZ_NOMINAL}

    skind = ('Arena', 'C4', 'azg_search_arena_f16') if arena else ('Args', 'C4', 'azg_search_f16') if net._hip is not None and net._hip.fused_head else \
        ('Wide', W['game'], 'azg_search_wide_exact_f16' if c.search_heads == 'exact' else 'azg_search_wide_f16')
    roof_search = mfma_roof('search', 'k_tower2<...,Search%s<%s>> (%s: %d x [find_leaf, ResNet + heads, backup] on every game, one persistent launch '
                            'per move)' % (skind + (sims,)), 'Search' + skind[0] + '<', Bl * sims)
    roof_net = mfma_roof('tower', 'k_tower2 (%s, one launch per simulation)' % ('both models on their row ranges' if arena else 'ResNet tower'
                                                                                     + ('' if net._hip is None or net._hip.wide_head else ' + heads')),
                         'NoSearch', Bl)
    if roof_net is not None and netprof.get('heads_n'):             # wide heads: their own launch behind the tower
        roof_net['heads_launch_us'] = round(netprof['heads_ms'] * 1e3 / netprof['heads_n'], 2)
    roof_tree = None
    if prof is not None and prof['backup_n'] > 0:
        feat_k = net._hip.feat_k if (not arena and net._hip is not None and net._hip.fact_head) else None
        sel_b, bak_b, shared_b = tree_bytes_per_sim(Game, W, feat_k)
        us = prof['backup_ms'] * 1e3 / prof['backup_n']              # backup k + select k + 1 share a launch
        gbs = ((sel_b + bak_b) * Bl + shared_b) / (us * 1e-6) / 1e9
        traffic, src, _ = measured_traffic(profile_key(c), 'k_backup_select2')
        roof_tree = {'kernel': 'k_backup_select2 (process_results of simulation k + find_leaf of k + 1, two wavefronts per tree%s)'
                               % (', sparse heads on the head features' if feat_k else ''),
                     'bound': 'hbm', 'achieved': round(gbs, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 6),
                     'avg_launch_us': round(us, 2), 'launches_timed': prof['backup_n'],
                     'algorithmic_bytes_per_launch': (sel_b + bak_b) * Bl + shared_b,
                     'traffic': traffic, 'traffic_source': src,
                     'advance_us': round(prof['advance_ms'] * 1e3 / max(prof['advance_n'], 1), 1)}
    # the dominant kernel of a simulation step: the persistent search launch if that is what ran, else the longer of the two launches
    if roof_search:
        roofline = roof_search
    elif roof_net and roof_tree:
        roofline = roof_net if roof_net['avg_launch_us'] >= roof_tree['avg_launch_us'] else roof_tree
    else:
        roofline = roof_net or roof_tree
    return roofline, roof_tree, roof_net


def sparse_heads_run(c, a, rank, local_rank, dev, world, steps=8, warmup=2):
    """configs 3 and 5 once more with the OPT-IN sparse heads (SelfPlayRunner(search_heads='sparse'), azg_search_wide_f16): the tree
    phase computes only the logits of each leaf's valid actions and takes the softmax over those -- equal to the default exact
    launch (all A + P+1 logits inside the launch: NNetWrapper.process's bits, MCTS.pyx:239-245) to rounding only (~1e-8 on a prior;
    a PUCT near-tie flips about once per 5e5 simulations) -- and saves the stream of the full head matrix."""
    hip = c.net._hip
    if c.arena or hip is None or not hip.fact_head or c.search_heads != 'exact':
        return None
    args = selfplay_args(c.W)
    nsym = len(c.Game().symmetries(np.zeros(c.Game.action_size(), np.float32)))
    x = Ctx()
    x.name, x.W, x.B, x.sims, x.Game, x.net, x.nets, x.arena, x.pipelines = c.name + '_sparse', c.W, c.B, c.sims, c.Game, c.net, c.nets, False, 1
    x.iter = SelfPlayIteration(c.Game, c.net, args, num_slots=c.B, seed=0, device=local_rank,
                               use_graph=not a.no_graph, search_heads='sparse',
                               example_capacity=int(c.B * (steps + warmup) / 5.0 + 2 * c.B) * (c.Game.max_turns() + 1) * nsym)
    x.runner = x.iter.runner
    x.engines = [ln.engine for ln in x.runner.lanes]
    x.counters = x.runner.counters
    x.fused_search = bool(x.runner.fused_search)
    x.runner.prepare()
    t = timed_region(x, steps, warmup, world, rank)
    out = {'value': round(t['expansions'] / t['dt'], 1), 'unit': 'expansions/s', 'ms_per_step': round(t['dt'] * 1e3 / steps, 3), 'steps': steps,
           'warmup': warmup, 'form': 'azg_search_wide_f16: the same persistent launch, policy logits of the leaf\'s valid actions only (softmax over '
                                      'those); equal to the exact launch to rounding, not bit for bit'}
    for e in x.engines:
        e.close()
    return out


def phase_budget(c, roofline):
    """The bound that applies to a persistent wide-head launch: ONE workgroup per game runs tree phase -> tower -> head convolutions
    back to back, `sims` times -- a chain of phase latencies, not an MFMA- or HBM-limited stream.  Cycles per simulation and phase
    come from the s_memtime stamps of the measurement build (tools/phase_budget.py -> profiles/r05_phase_budget.json, stamped
    with the kernel sources' hash); the floors beside them: the tower's MFMA issue time for one board on the workgroup's SIMDs, and
    the walk's dependent-load chain (one child block per level)."""
    pb = load_json(PHASE_FILE)
    if not pb or profile_key(c) not in pb.get('workloads', {}) or roofline is None:
        return None
    w = dict(pb['workloads'][profile_key(c)])
    if w.get('search_heads', 'sparse') != (c.search_heads or 'sparse'):
        return None
    w['source'] = 'profiles/%s @%s%s' % (os.path.basename(PHASE_FILE), pb.get('git', '?'), '' if pb.get('csrc_sha') == csrc_sha() else ' (kernel sources have changed since)')
    cyc = w['tree'] + w['tower'] + w['headconv'] + w.get('heads', 0)
    gpw = int(w.get('games_per_workgroup', 1))
    w['bound'] = ('phase-latency chain: a workgroup (%d game%s) runs tree phase -> tower -> head convolutions%s back to back, so a move takes '
                  'sims x cycles_per_sim / shader clock; the MFMA fraction above only says how much of that chain is the tower'
                  % (gpw, '' if gpw == 1 else 's', ' -> full-width heads' if w.get('heads') else ''))
    w['cycles_per_sim'], w['sims'], w['chain_cycles_per_move'] = cyc, c.sims, cyc * c.sims
    w['measured_launch_us'] = roofline['avg_launch_us']
    w['implied_clock_ghz'] = round(cyc * c.sims / (roofline['avg_launch_us'] * 1e3), 3)      # chain cycles / measured launch time
    w['chain_us_at_nominal_clock'] = round(cyc * c.sims / (CLOCK_GHZ_NOMINAL * 1e3), 1)
    # floor of the tower phase: one board's MFMAs issued back to back on the CU's four SIMDs (one v_mfma_f32_16x16x32_f16 = 16 384
    # FLOP per 16 cycles per SIMD = the 2.5 PFLOP/s peak over 1 024 SIMDs at 2.4 GHz)
    per_simd_cycle = MFMA_F16_PEAK_TFLOPS * 1e12 / 1024 / (CLOCK_GHZ_NOMINAL * 1e9)
    w['tower_mfma_floor_cycles'] = int(gpw * net_flops_per_leaf(c.Game, c.net.args) / per_simd_cycle / 4)   # (the workgroup's boards)
    w['tower_over_mfma_floor'] = round(w['tower'] / max(w['tower_mfma_floor_cycles'], 1), 2)
    w['tower_share_of_chain'] = round(w['tower'] / cyc, 3)
    return w


def compat_run(W, net, seconds=12.0, workers=2, games_per_worker=None):
    """What an UNMODIFIED Coach gets (compat mode, Coach.py:291-361): `workers` SelfPlayAgent processes with the reference's
    constructor and queue / event / shared-tensor protocol, each driving a device engine through its worker interpreter, the
    parent serving their batches with the GPU network exactly like Coach.processSelfPlayBatches (:337-342: ready_queue.get ->
    nnet.process(input_tensors[id]) -> copy into the shared policy / value tensors -> batch_ready[id].set()).  Every simulation
    pays two process hops and an H2D + D2H of the batch.  Config 2's games split over the workers; runs for `seconds`."""
    import importlib
    import queue
    import torch.multiprocessing as mp
    from alphazero_general_amd.SelfPlayAgent import SelfPlayAgent
    Game = importlib.import_module('alphazero_general_amd.envs.' + W['game']).Game
    B = int(games_per_worker or W['B'] // workers)
    args = selfplay_args(W)
    args.update(_num_players=Game.num_players() + 1, _azg_seed=0)
    C, H, Wd = Game.observation_size()
    A, NV = Game.action_size(), Game.num_players() + 1
    ready_queue, file_queue, result_queue = mp.Queue(), mp.Queue(), mp.Queue()
    completed, games_played = mp.Value('i', 0), mp.Value('i', 0)
    stop, pause = mp.Event(), mp.Event()
    inputs, pols, vals, ready, agents = [], [], [], [], []
    for i in range(workers):                                         # Coach.generateSelfPlayAgents :291-323
        inputs.append(torch.zeros([B, C, H, Wd]).share_memory_()); pols.append(torch.zeros([B, A]).share_memory_())
        vals.append(torch.zeros([B, NV]).share_memory_()); ready.append(mp.Event())
        agents.append(SelfPlayAgent(i, Game, ready_queue, ready[i], inputs[i], pols[i], vals[i], file_queue, result_queue, completed,
                                    games_played, stop, pause, args))
        agents[i].daemon = True
        agents[i].start()
    served, nsamples, t_first, t_end = 0, 0, None, None
    from alphazero_general_amd.nnet import pin_stats
    pin_stats(reset=True)                                            # (how the shared batches travel: page-locked in place, or staged -- never silently)
    t_wait = t_net = t_copy = 0.0                                    # the parent's own time per batch: idle, evaluation (H2D + net + D2H), hand-back
    deadline = time.time() + float(os.environ.get("AZG_COMPAT_DEADLINE", "240"))

    def drain():
        n = 0
        for q in (file_queue, result_queue):
            try:
                while True:
                    q.get_nowait(); n += q is file_queue
            except queue.Empty:
                pass
        return n
    try:
        while completed.value != workers and time.time() < deadline:
            if served % 32 == 0:                                     # (Coach.processSelfPlayBatches does not touch the sample queue in this loop at all)
                nsamples += drain()
            t0 = time.perf_counter()
            try:
                i = ready_queue.get(timeout=0.5)
            except queue.Empty:
                continue
            t1 = time.perf_counter()
            p, v = net.process(inputs[i])
            p, v = p.cpu(), v.cpu()                                  # (the D2H the copy_ below would do, timed with the evaluation)
            t2 = time.perf_counter()
            pols[i].copy_(p); vals[i].copy_(v); ready[i].set()
            t3 = time.perf_counter()
            if t_first is None:
                t_first = time.time()                                # (the workers' start-up is not self-play)
            else:
                served += 1
                t_wait += t1 - t0; t_net += t2 - t1; t_copy += t3 - t2
            t_end = time.time()
            if t_end - t_first >= seconds:
                break
    finally:
        stop.set()
        for ev in ready:
            ev.set()
    t1 = time.time()
    while time.time() - t1 < 20 and completed.value != workers:
        drain()
        try:
            i = ready_queue.get(timeout=0.2)
            ready[i].set()
        except queue.Empty:
            pass
    drain()
    for ag in agents:
        ag.join(10)
        if ag.is_alive():
            ag.terminate()
    if not served or t_first is None:
        return {'error': 'no batch was served'}
    dt = t_end - t_first
    return {'value': round(served * B / dt, 1), 'unit': 'simulations/s (leaf evaluations served; ~ expansions/s: only revisits of terminal nodes differ)',
            'workers': workers, 'games_per_worker': B, 'sims_per_move': W['sims'], 'batches_served': served, 'seconds': round(dt, 2),
            'ms_per_batch': round(dt * 1e3 / served, 3), 'games_finished': int(games_played.value), 'samples_received': nsamples,
            # where the parent's time per batch goes (it serves the agents one batch at a time, Coach.py:337-342): waiting for a ready agent,
            # nnet.process incl. H2D of the batch and D2H of policy / value, copying into the shared tensors + batch_ready.set()
            'parent_us_per_batch': {'wait_for_agent': round(t_wait * 1e6 / served, 1), 'evaluate_h2d_net_d2h': round(t_net * 1e6 / served, 1),
                                    'hand_back': round(t_copy * 1e6 / served, 1)},
            'host_batches': pin_stats(),                             # 'dma': page-locked in place (hipHostRegister); 'staged': pageable copies
            'path': 'alphazero_general_amd.SelfPlayAgent (reference constructor / queue protocol) x %d processes, parent serves batches like '
                    'Coach.processSelfPlayBatches with the GPU net: per simulation two process hops + H2D + D2H of the batch' % workers}


def projected_scaling(out, others, shards):
    """BASELINE's metric is quoted at 1/2/4/8 GPUs.  Games shard with ZERO communication during search, so an N-GPU job
    runs N copies of the per-GPU shard and its throughput is N x the shard's 1-GPU value -- minus the exchange step once per iteration.
    Configs 3-5 name TOTAL games: more GPUs = SMALLER shards, and small shards run below the large ones' rate (one or two boards per
    workgroup: bound by the parameter stream, `operand_stream`).  Every figure here is computed from shards TIMED ON ONE GPU in this run
    (`strong_scaling_shards`, `other_workloads`): a projection, stated as one, until an 8-GPU node measures it."""
    def val(name, B):
        rec = shards.get('%s_%d' % (name, B)) if shards else None
        if rec is None and others and WORKLOADS[name]['B'] == B:
            rec = others.get(name)
        return rec.get('value') if rec and 'value' in rec else None
    proj = {}
    for label, name, total, gpus in (('config3_brandubh_4096_games_200_sims', 'brandubh', 4096, (1, 2, 4, 8)),
                                     ('config5_trimok_1024_games_50_sims', 'trimok', 1024, (1, 2, 4)),
                                     ('config4_arena_512_games_100_sims', 'arena', 512, (1, 2))):
        v1 = val(name, total)
        rows = []
        for n in gpus:
            vs = val(name, total // n)
            if v1 and vs:
                rows.append({'gpus': n, 'games_per_gpu': total // n, 'shard_value_1gpu': vs, 'projected_value': round(n * vs, 1),
                             'speedup': round(n * vs / v1, 3), 'efficiency': round(vs / v1, 3)})
        if rows:
            proj[label] = {'scaling': 'strong (total games fixed)', 'points': rows, 'at_config_gpus': rows[-1]}
    k = max(out['steps'], 1)
    ex = out.get('exchange_ms') or 0.0
    proj['config2_connect4_2048_games_per_gpu'] = {
        'scaling': 'weak (games per GPU fixed)', 'per_gpu_value': out['value'],
        'points': [{'gpus': n, 'projected_value': round(n * out['value'] * (out['ms_per_step'] * k) / (out['ms_per_step'] * k + ex), 1)} for n in (1, 2, 4, 8)],
        'efficiency': round((out['ms_per_step'] * k) / (out['ms_per_step'] * k + ex), 4),
        'note': 'no communication during search; the only N-rank cost is the exchange step once per iteration (here: %.3f ms per %d rounds at world 1; '
                'the all-gather moves each rank\'s shard over its own xGMI link)' % (ex, k)}
    proj['basis'] = 'shards timed on ONE GPU in this run; N-GPU value = N x shard value (independent ranks); not a multi-GPU measurement'
    return proj


def workload_label(c):
    return '%s %s, %d games/GPU x %d sims/move, fp16 ResNet %dch x %d, random-init, %s' % (
        c.W['game'], 'arena (two nets)' if c.arena else 'self-play', c.B, c.sims, c.net.args.num_channels, c.net.args.depth,
        'arenaTemp 0.25' if c.arena else 'noise+temp on')


def release(c):
    for e in c.engines:
        e.close()
    c.runner = c.iter = c.engines = c.net = c.nets = None
    import gc
    gc.collect(); torch.cuda.empty_cache()


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
    ap.add_argument('--no-library-gemm', action='store_true', help='skip the hipBLASLt context GEMM (keeps profiler traces to our own kernels)')
    ap.add_argument('--no-fused-search', action='store_true',
                    help='2 launches per simulation (tower, backup+select) instead of one persistent launch per move (azg_search_f16)')
    ap.add_argument('--no-other-workloads', action='store_true',
                    help='skip the short runs of BASELINE configs 3-5 that the default (connect4, 1 GPU) line carries as other_workloads')
    ap.add_argument('--profile-rounds', type=int, default=10, help='eager rounds of the persistent launch timed after the timed region')
    ap.add_argument('--compat', action='store_true', help='also time compat mode (unmodified-Coach protocol: SelfPlayAgent processes served by the parent)')
    ap.add_argument('--search-heads', default=None, choices=['exact', 'sparse'],
                    help='wide-head workloads, persistent launch: all A + P+1 logits inside the launch (bit-exact) or the valid actions only')
    ap.add_argument('--no-exact-heads', '--no-sparse-heads', dest='no_sparse_heads', action='store_true',
                    help='skip the second run of the wide-head workloads with the opt-in sparse heads')
    ap.add_argument('--no-shards', action='store_true', help='skip the 1 / 2 / 4-GPU shard sizes of configs 3-5 on the default line')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(a.gpus)                                    # one rank per GPU under torch.distributed.run
    rank, local_rank, world = D.init_from_env()
    assert world == a.gpus, '--gpus %d but %d rank(s) were launched (torch.distributed.run --nproc-per-node must equal --gpus)' % (a.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a HIP device (there is no CPU fallback)'
    if not os.environ.get('AZG_SINGLE_DEVICE'):
        assert torch.cuda.device_count() >= world, '%d ranks but only %d visible GPU(s)' % (world, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    rank_devices = D.describe_ranks(local_rank)                       # (raises if two ranks share a GPU; the records go on the line)

    c = build(a.workload, a, rank, local_rank, dev, a.steps + a.warmup + a.profile_rounds + 8)
    t = timed_region(c, a.steps, a.warmup, world, rank)
    netprof, prof = profile_rounds(c, a.profile_rounds)              # after the timed region
    ex = None if a.no_sparse_heads else sparse_heads_run(c, a, rank, local_rank, dev, world)   # (every rank: it has its own exchange step)
    if rank != 0:
        return 0
    roofline, roof_tree, roof_net = rooflines(c, netprof, prof)
    dt = t['dt']
    out = {
        'metric': 'mcts_node_expansions_per_sec', 'value': round(t['expansions'] / dt, 1), 'unit': 'expansions/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt * 1e3 / a.steps, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 tree / f16 net', 'data': 'synthetic',
        'config': {'workload': workload_label(c),
                   'games_per_gpu': c.B, 'sims_per_move': c.sims, 'hipgraph_rounds': not a.no_graph, 'stream_pipelines': a.pipelines,
                   'fused_search_launch': c.fused_search, 'search_heads': c.search_heads, 'mfma_tower': c.net._hip is not None, 'ranks': world,
                   'backend': torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                   'lib_source_sha': lib_source_sha(), 'tree_source_sha': csrc_sha(),
                   'rank_devices': rank_devices},
        'games_per_sec': round(t['games'] / dt, 2), 'simulations_per_sec': round(t['sims'] / dt, 1),
        'games_finished': t['games'], 'samples_gathered': t['samples'],
        # (all games start from the empty board: the whole-region figure includes the first burst of finishes)
        'games_per_sec_steady': None if not t['second_half_steps'] else round(t['games_second_half'] / (dt * t['second_half_steps'] / a.steps), 2),
        'games_finished_second_half': t['games_second_half'],
        # where the step time of an N-rank run goes: the slowest / fastest rank's own rounds, and the iteration's exchange step
        # (example all-gather + tallies, once per timed region)
        'rank_ms_per_step': {'max': round(t['rank_ms_per_step_max'], 3), 'min': round(t['rank_ms_per_step_min'], 3)},
        'exchange_ms': round(t['exchange_ms'], 3),
        'roofline': roofline, 'tree_roofline': roof_tree,
    }
    if roof_net is not None and roofline is not roof_net:
        out['net_roofline'] = roof_net
    pbud = phase_budget(c, roofline)
    if pbud is not None:
        roofline['phase_budget'] = pbud
    if ex is not None:
        out['sparse_heads'] = ex
    if world == 1 and roofline is not None and roofline['bound'] == 'mfma' and not a.no_library_gemm:
        lib_tf = library_gemm_tflops(dev)                            # outside the timed region
        roofline['library_gemm_tflops'] = round(lib_tf, 1)
        roofline['vs_library_gemm'] = round(roofline['achieved'] / lib_tf, 3)
    if world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(c.W, c.nets, c.arena)
    if world == 1 and a.workload == 'connect4' and not a.no_other_workloads and not a.slots:
        # BASELINE configs 3-5 at their per-GPU size: 8 graph-replayed rounds each after 2 warm-up rounds (their own full lines:
        # --workload NAME; profiles/r03_bench_*.json)
        release(c)
        others = {}
        for name in ('brandubh', 'arena', 'trimok'):
            try:                                                     # (a failure here must not take the headline line with it)
                oc = build(name, a, rank, local_rank, dev, 8 + 2 + 2 + 8)
                ot = timed_region(oc, 8, 2, world, rank)
                onp, opf = profile_rounds(oc, 1)
                orf, otr, _ = rooflines(oc, onp, opf)
                opb = phase_budget(oc, orf)
                oex = None if a.no_sparse_heads else sparse_heads_run(oc, a, rank, local_rank, dev, world)
                others[name] = {'workload': workload_label(oc), 'value': round(ot['expansions'] / ot['dt'], 1), 'unit': 'expansions/s',
                                'games_per_sec': round(ot['games'] / ot['dt'], 2), 'steps': 8, 'warmup': 2,
                                'ms_per_step': round(ot['dt'] * 1e3 / 8, 3), 'fused_search_launch': oc.fused_search, 'search_heads': oc.search_heads,
                                'roofline': None if orf is None else {k: orf.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac',
                                                                                             'executed_frac', 'avg_launch_us', 'launches_timed', 'traffic', 'operand_stream')},
                                'tree_launch_us': None if otr is None else otr['avg_launch_us']}
                if opb is not None:
                    others[name]['phase_budget'] = opb
                if oex is not None:
                    others[name]['sparse_heads'] = oex
                release(oc)
            except Exception as ex:                                  # noqa: BLE001
                others[name] = {'error': '%s: %s' % (type(ex).__name__, ex)}
        try:                                                         # compat mode: what an unmodified Coach gets (config 2's games, 2 workers)
            torch.manual_seed(0)
            cnet = NNetWrapper(__import__('alphazero_general_amd.envs.connect4', fromlist=['Game']).Game, nn_mod.CONNECT4_NET_ARGS, device=dev, dtype=torch.float16)
            cnet.refresh()
            # config 2's own shape (its 2048 games on two agents) and four agents x 2048 games: the parent serves ONE batch at a time
            # (Coach.py:337-342), so what it needs is enough agents to always find one ready and batches large enough to amortise its
            # fixed cost per batch (tools/compat_sweep.py: profiles/r05_compat_sweep.txt)
            runs = [compat_run(WORKLOADS['connect4'], cnet, seconds=6.0, workers=w_, games_per_worker=g_) for w_, g_ in ((2, 1024), (4, 2048))]
            # 'compat' is config 2's OWN shape (2048 games on two agents), like for like with the headline; the larger sweep point sits
            # under its own key
            others['compat'] = dict(runs[0], runs=[{k_: r_.get(k_) for k_ in ('workers', 'games_per_worker', 'value', 'ms_per_batch', 'parent_us_per_batch', 'error')} for r_ in runs])
            others['compat_best_shape'] = {k_: max(runs, key=lambda r_: r_.get('value', 0)).get(k_) for k_ in ('workers', 'games_per_worker', 'value', 'unit', 'ms_per_batch')}
            del cnet
        except Exception as ex:                                      # noqa: BLE001
            others['compat'] = {'error': '%s: %s' % (type(ex).__name__, ex)}
        out['other_workloads'] = others
        if not a.no_shards:
            # BASELINE's metric is "at 1/2/4/8 MI355X" and configs 3-5 name TOTAL games: their 1- / 2- / 4-GPU points are larger shards
            # per GPU (config 3: 4096 / 2048 / 1024 brandubh games, config 5: 1024 / 512, config 4: 512).  Each timed here on ONE GPU, 4
            # graph-replayed rounds after 1 warm-up round (the shard sizes in other_workloads above are the 8- / 4- / 2-GPU points)
            shards = {}
            for name, sizes in (('brandubh', (4096, 2048, 1024)), ('trimok', (1024, 512)), ('arena', (512,))):
                for Bs in sizes:
                    key = '%s_%d' % (name, Bs)
                    try:
                        oc = build(name, a, rank, local_rank, dev, 4 + 1 + 1 + 8, slots=Bs)
                        ot = timed_region(oc, 4, 1, world, rank)
                        onp, opf = profile_rounds(oc, 1)
                        orf, _, _ = rooflines(oc, onp, opf)
                        shards[key] = {'workload': workload_label(oc), 'games_per_gpu': Bs, 'gpus_at_this_shard_size': WORKLOAD_TOTAL_GAMES[name] // Bs,
                                       'value': round(ot['expansions'] / ot['dt'], 1), 'unit': 'expansions/s', 'games_per_sec': round(ot['games'] / ot['dt'], 2),
                                       'steps': 4, 'warmup': 1, 'ms_per_step': round(ot['dt'] * 1e3 / 4, 3), 'search_heads': oc.search_heads,
                                       'roofline': None if orf is None else {k: orf.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us', 'launches_timed', 'operand_stream')}}
                        release(oc)
                    except Exception as ex:                          # noqa: BLE001
                        shards[key] = {'error': '%s: %s' % (type(ex).__name__, ex)}
            out['strong_scaling_shards'] = shards
            out['projected_scaling'] = projected_scaling(out, others, shards)
    elif world == 1 and a.compat and not c.arena:
        out['compat'] = compat_run(c.W, c.net)
    print(json.dumps(out))
    return 0


if __name__ == '__main__':
    try:
        rc = main()
    finally:
        D.shutdown()
    sys.exit(rc or 0)
