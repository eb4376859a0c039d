"""Native self-play driver: the device engine and the network share one GPU, so a simulation step is
select -> network -> backup on one HIP stream with no host hop.  This is the build-owned replacement of the
parent-side loop Coach.processSelfPlayBatches (alphazero/Coach.py:326-361) + the agent loop SelfPlayAgent.run
(alphazero/SelfPlayAgent.pyx:79-101); outputs land in the three-tensor layout Coach.saveIterationSamples writes
(Coach.py:364-386).  Multi-GPU: one process per GPU, game slots sharded by rank (slot_base = rank * B), no
communication during search; finished examples are all-gathered once per iteration (distributed.py).
"""
import time

import torch

from . import _abi
from .engine import DeviceEngine
from .Game import azg_game_id
from .utils import AGENT_STREAM, default_temp_scaling


class _Lane:
    """One software-pipeline lane: a device engine over a contiguous range of game slots, its own HIP stream and its
    own captured network graph."""

    def __init__(self, engine, stream):
        self.engine, self.stream = engine, stream
        self.obs = self.policy = self.value = self.net = None


class SelfPlayRunner:
    """num_slots games on one GPU.  `pipelines` > 1 splits them into equal slot ranges that run the same lock-step loop
    on separate HIP streams, so that the tree kernels of one range execute underneath the network evaluation of
    another (the network tower occupies one workgroup per CU and leaves most wave slots free).  Games are
    independent and every random draw is keyed by the global slot id, so the split changes no result."""

    def __init__(self, game_cls, nnet, args, *, num_slots, seed=0, slot_base=0, device=None, example_capacity=None,
                 use_graph=True, obs_dtype=torch.float16, warmup=False, pipelines=1):
        self.game_cls, self.nnet, self.args = game_cls, nnet, args
        self.game = azg_game_id(game_cls)
        self.B = int(num_slots)
        self.warmup = bool(warmup)
        self.pipelines = int(pipelines)
        assert self.B % self.pipelines == 0
        Bl = self.B // self.pipelines
        sims = max(int(args.get('numMCTSSims', 100)), int(args.get('numFastSims', 0) or 0), int(args.get('numWarmupSims', 0) or 0))
        gi = _abi.game_info(self.game)
        if example_capacity is None:
            per_game = (gi.max_turns + 1) * (gi.num_symmetries if args.get('symmetricSamples', True) else 1)
            example_capacity = (int(args.get('gamesPerIteration', self.B)) + self.B) * per_game
        self.seed, self.slot_base = int(seed), int(slot_base)
        self._actr = 0
        self.use_graph = bool(use_graph) and not self.warmup and nnet is not None
        self.lanes = []
        for li in range(self.pipelines):
            eng = DeviceEngine(
                self.game, Bl, cpuct=args.get('cpuct', 1.25), fpu_reduction=args.get('fpu_reduction', 0.2),
                root_noise_frac=args.get('root_noise_frac', 0.1), root_policy_temp=args.get('root_policy_temp', 1.1),
                min_discount=args.get('min_discount', 1.0), add_root_noise=args.get('add_root_noise', True),
                add_root_temp=args.get('add_root_temp', True), symmetric_samples=args.get('symmetricSamples', True),
                mcts_reset_threshold=args.get('mctsResetThreshold', 0) or 0,
                games_per_iteration=int(args.get('gamesPerIteration', 1 << 30)), start_temp=args.get('startTemp', 1.0),
                arena_temp=args.get('arenaTemp', 0.25), temp_fn=args.get('temp_scaling_fn', default_temp_scaling),
                seed=seed, slot_base=self.slot_base + li * Bl, device=device,
                example_capacity=example_capacity // self.pipelines + 1, sims_hint=sims)
            dev = eng.device
            with torch.cuda.device(dev):
                lane = _Lane(eng, torch.cuda.Stream(device=dev) if self.pipelines > 1 else torch.cuda.current_stream(dev))
            with torch.cuda.stream(lane.stream):
                if self.warmup:                                      # SelfPlayAgent.pyx:48-52: uniform policy / value
                    lane.policy = torch.full((Bl, eng.A), 1 / eng.A, dtype=torch.float32, device=dev)
                    lane.value = torch.full((Bl, eng.NV), 1 / eng.NV, dtype=torch.float32, device=dev)
                elif self.use_graph:
                    lane.net = nnet.capture_net(Bl, in_dtype=obs_dtype)
                    lane.obs, lane.policy, lane.value = lane.net.x, lane.net.policy, lane.net.value
                else:
                    lane.obs = eng.new_obs(obs_dtype)
            self.lanes.append(lane)
        self.engine = self.lanes[0].engine                           # single-lane convenience (tests, smoke)
        self.device = self.engine.device
        self.sims_per_round = []

    @property
    def obs(self):
        return self.lanes[0].obs

    def _sims_for_round(self):
        """SelfPlayAgent.run :84-86: one fast coin per round for the whole batch (agent-level tape stream)."""
        a = self.args
        u = _abi.lib().azg_tape_uniform(self.seed, AGENT_STREAM + self.slot_base, self._actr)
        self._actr += 1
        fast = u < float(a.get('probFastSim', 0.0) or 0.0)
        if fast:
            return int(a.get('numFastSims', 20)), True
        return int(a.get('numWarmupSims', 5) if self.warmup else a.get('numMCTSSims', 100)), False

    def _step_lane(self, ln):
        e = ln.engine
        e.select(ln.obs)
        if self.warmup:
            e.backup(ln.policy, ln.value)
        elif self.use_graph:
            ln.net.replay()
            e.backup(ln.policy, ln.value)
        else:
            p, v = self.nnet.process(ln.obs)
            e.backup(p.contiguous(), v.contiguous())

    def step(self):
        """one simulation on every slot: generateBatch -> network -> processBatch (SelfPlayAgent.pyx:87-92)."""
        if self.pipelines == 1:
            self._step_lane(self.lanes[0])
            return
        for ln in self.lanes:
            with torch.cuda.stream(ln.stream):
                self._step_lane(ln)

    def play_round(self):
        sims, fast = self._sims_for_round()
        for _ in range(sims):
            self.step()
        for ln in self.lanes:                                        # playMoves :153-202
            with torch.cuda.stream(ln.stream):
                ln.engine.advance(record_history=not fast)
        self.sims_per_round.append(sims)
        return sims

    def counters(self):
        tot = None
        for ln in self.lanes:
            with torch.cuda.stream(ln.stream):
                c = ln.engine.counters()
            if tot is None:
                tot = dict(c)
            else:
                for k in ('sims', 'expansions', 'games_played', 'num_results', 'num_examples'):
                    tot[k] += c[k]
                tot['max_nodes_used'] = max(tot['max_nodes_used'], c['max_nodes_used'])
        return tot

    def run(self, games=None, max_rounds=None, poll_every=1):
        """Play rounds until `games` (default args.gamesPerIteration) games have finished.  Returns counters."""
        games = int(self.args.get('gamesPerIteration') if games is None else games)
        t0 = time.time()
        rounds = 0
        while True:
            self.play_round()
            rounds += 1
            if rounds % poll_every == 0:
                c = self.counters()                                  # one small D2H per move (not per simulation)
                if c['games_played'] >= games:
                    break
            if max_rounds is not None and rounds >= max_rounds:
                c = self.counters()
                break
        c['rounds'], c['seconds'] = rounds, time.time() - t0
        return c

    def samples(self, first_per_lane=None):
        """(data, policy, value) tensors as saved by Coach.saveIterationSamples (Coach.py:377-383); lanes in slot order."""
        outs = []
        for i, ln in enumerate(self.lanes):
            with torch.cuda.stream(ln.stream):
                f = 0 if first_per_lane is None else first_per_lane[i]
                outs.append(ln.engine.examples(f))
                ln.stream.synchronize()
        return tuple(torch.cat([o[j] for o in outs]) for j in range(3))

    def results(self):
        import numpy as np
        rs = [ln.engine.results() for ln in self.lanes]
        Bl = self.B // self.pipelines
        return (np.concatenate([r[0] for r in rs]), np.concatenate([r[1] for r in rs]),
                np.concatenate([r[2] + i * Bl for i, r in enumerate(rs)]))
