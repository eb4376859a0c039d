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


class SelfPlayRunner:
    def __init__(self, game_cls, nnet, args, *, num_slots, seed=0, slot_base=0, device=None, example_capacity=None,
                 use_graph=True, obs_dtype=torch.float16, warmup=False):
        self.game_cls, self.nnet, self.args = game_cls, nnet, args
        self.game = azg_game_id(game_cls)
        self.B = int(num_slots)
        self.warmup = bool(warmup)
        sims = max(int(args.get('numMCTSSims', 100)), int(args.get('numFastSims', 0) or 0), int(args.get('numWarmupSims', 0) or 0))
        gi = _abi.game_info(self.game)
        if example_capacity is None:
            per_game = (gi.max_turns + 1) * (gi.num_symmetries if args.get('symmetricSamples', True) else 1)
            example_capacity = (int(args.get('gamesPerIteration', self.B)) + self.B) * per_game
        self.engine = DeviceEngine(
            self.game, self.B, cpuct=args.get('cpuct', 1.25), fpu_reduction=args.get('fpu_reduction', 0.2),
            root_noise_frac=args.get('root_noise_frac', 0.1), root_policy_temp=args.get('root_policy_temp', 1.1),
            min_discount=args.get('min_discount', 1.0), add_root_noise=args.get('add_root_noise', True),
            add_root_temp=args.get('add_root_temp', True), symmetric_samples=args.get('symmetricSamples', True),
            mcts_reset_threshold=args.get('mctsResetThreshold', 0) or 0,
            games_per_iteration=int(args.get('gamesPerIteration', 1 << 30)), start_temp=args.get('startTemp', 1.0),
            arena_temp=args.get('arenaTemp', 0.25), temp_fn=args.get('temp_scaling_fn', default_temp_scaling),
            seed=seed, slot_base=slot_base, device=device, example_capacity=example_capacity, sims_hint=sims)
        self.seed, self.slot_base = int(seed), int(slot_base)
        self._actr = 0
        dev = self.engine.device
        self.use_graph = bool(use_graph) and not self.warmup and nnet is not None
        if self.warmup:                                              # SelfPlayAgent.pyx:48-52: uniform policy / value
            self.policy = torch.full((self.B, self.engine.A), 1 / self.engine.A, dtype=torch.float32, device=dev)
            self.value = torch.full((self.B, self.engine.NV), 1 / self.engine.NV, dtype=torch.float32, device=dev)
            self.obs = None
        elif self.use_graph:
            self.obs, self.policy, self.value = nnet.capture(self.B, in_dtype=obs_dtype)
        else:
            self.obs = self.engine.new_obs(obs_dtype)
        self.sims_per_round = []

    def _sims_for_round(self):
        """SelfPlayAgent.run :84-86: one fast coin per round for the whole batch (agent-level tape stream)."""
        a = self.args
        u = _abi.lib().azg_tape_uniform(self.seed, AGENT_STREAM + self.slot_base, self._actr)
        self._actr += 1
        fast = u < float(a.get('probFastSim', 0.0) or 0.0)
        if fast:
            return int(a.get('numFastSims', 20)), True
        return int(a.get('numWarmupSims', 5) if self.warmup else a.get('numMCTSSims', 100)), False

    def step(self):
        """one simulation on every slot: generateBatch -> network -> processBatch (SelfPlayAgent.pyx:87-92)."""
        e = self.engine
        e.select(self.obs)
        if self.warmup:
            e.backup(self.policy, self.value)
        elif self.use_graph:
            self.nnet.replay()
            e.backup(self.policy, self.value)
        else:
            p, v = self.nnet.process(self.obs)
            e.backup(p.contiguous(), v.contiguous())

    def play_round(self):
        sims, fast = self._sims_for_round()
        for _ in range(sims):
            self.step()
        self.engine.advance(record_history=not fast)                # playMoves :153-202
        self.sims_per_round.append(sims)
        return sims

    def run(self, games=None, max_rounds=None, poll_every=1):
        """Play rounds until `games` (default args.gamesPerIteration) games have finished.  Returns counters."""
        games = int(self.args.get('gamesPerIteration') if games is None else games)
        t0 = time.time()
        rounds = 0
        while True:
            self.play_round()
            rounds += 1
            if rounds % poll_every == 0:
                c = self.engine.counters()                           # one small D2H per move (not per simulation)
                if c['games_played'] >= games:
                    break
            if max_rounds is not None and rounds >= max_rounds:
                c = self.engine.counters()
                break
        c['rounds'], c['seconds'] = rounds, time.time() - t0
        return c

    def samples(self):
        """(data, policy, value) tensors as saved by Coach.saveIterationSamples (Coach.py:377-383)."""
        return self.engine.examples()

    def results(self):
        return self.engine.results()
