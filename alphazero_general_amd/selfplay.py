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
from .nnet import HipResNet
from .Game import azg_game_id
from .utils import AGENT_STREAM, default_temp_scaling


# The persistent launch of a wide-head network hands the tree NNetWrapper.process's own bits by default ('exact': visit counts and pi
# bit-exact against the reference's evaluation, BASELINE north_star); 'sparse' (the logits of the valid actions only: equal to rounding,
# 10-35 % faster by shard size) is the opt-in: SelfPlayRunner(search_heads='sparse')
SEARCH_HEADS_DEFAULT = 'exact'


class _Lane:
    """One software-pipeline lane: a device engine over a contiguous range of game slots, its own HIP stream and its
    own captured network graph."""

    def __init__(self, engine, stream):
        self.engine, self.stream = engine, stream
        self.obs = self.policy = self.value = self.net = None
        self.round_graphs = {}


class SelfPlayRunner:
    """num_slots games on one GPU.  `pipelines` > 1 splits them into equal slot ranges that run the same lock-step loop
    on separate HIP streams, so that the tree kernels of one range execute underneath the network evaluation of
    another (the network tower occupies one workgroup per CU and leaves most wave slots free).  Games are
    independent and every random draw is keyed by the global slot id, so the split changes no result."""

    def __init__(self, game_cls, nnet, args, *, num_slots, seed=0, slot_base=0, device=None, example_capacity=None,
                 use_graph=True, obs_dtype=torch.float16, warmup=False, pipelines=1, round_graph=None, result_capacity=None,
                 fused_search=None, heads=None, nodes_per_tree=0, search_heads=None):
        """heads: what the tree launch is fed when the search is launched per phase -- None follows search_heads ('exact', the default:
        'logits' for wide heads -- all A + P+1 logits, softmax inside the tree launch --, else 'probs'; 'sparse': 'features' for
        factorised heads -- the launch computes the logits of the valid actions itself); tests pin each form against the oracle.
        search_heads: the same choice for the persistent launch of a factorised-heads network.  nodes_per_tree: node store of a tree
        (0 = the library's default, include/azg.h)."""
        assert heads in (None, 'probs', 'logits', 'features')
        # the persistent launch of a factorised-heads network: 'exact' = all A + P+1 logits inside the launch (NNetWrapper.process's bits),
        # 'sparse' = only the logits of each leaf's valid actions (equal to rounding)
        assert search_heads in (None, 'exact', 'sparse')
        self.search_exact = (search_heads or SEARCH_HEADS_DEFAULT) == 'exact'
        if heads is not None:                                        # a pinned hand-over form IS the launch-per-phase search
            if fused_search:
                raise ValueError('heads=%r pins a launch-per-phase form; it cannot be combined with fused_search=True' % heads)
            fused_search = False
        self.heads = heads
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
            games = int(args.get('gamesPerIteration', self.B))
            # (no cap -- the benchmark's and the tests' "play K rounds": room for four generations of games; pass example_capacity to size it)
            example_capacity = ((games if games < (1 << 30) else 3 * self.B) + self.B) * per_game
        if result_capacity is None:                                  # one record per finished game; a game is at least a few moves
            result_capacity = example_capacity // max(gi.num_symmetries if args.get('symmetricSamples', True) else 1, 1) // 4 + 4 * self.B + 1024
        self.seed, self.slot_base = int(seed), int(slot_base)
        self._actr = 0
        self.use_graph = bool(use_graph) and not self.warmup and nnet is not None
        # whole-round graphs need a capturable evaluation: the captured net's launch sequence, or warm-up's constants
        self.round_graph = (self.use_graph or (self.warmup and bool(use_graph))) if round_graph is None else bool(round_graph)
        self.lanes = []
        from .distributed import shard_games
        total_games = int(args.get('gamesPerIteration', 1 << 30))
        for li in range(self.pipelines):
            # the games cap is split into per-lane quotas that sum to exactly gamesPerIteration (a lane that has filled its
            # quota idles its finished slots, like the reference's agents once games_played reaches the cap)
            quota = total_games if total_games >= (1 << 30) else shard_games(total_games, li, self.pipelines)
            eng = DeviceEngine(
                self.game, Bl, cpuct=args.get('cpuct', 1.25), fpu_reduction=args.get('fpu_reduction', 0.2),
                root_noise_frac=args.get('root_noise_frac', 0.1), root_policy_temp=args.get('root_policy_temp', 1.1),
                min_discount=args.get('min_discount', 1.0), add_root_noise=args.get('add_root_noise', True),
                add_root_temp=args.get('add_root_temp', True), symmetric_samples=args.get('symmetricSamples', True),
                mcts_reset_threshold=args.get('mctsResetThreshold', 0) or 0,
                games_per_iteration=quota, start_temp=args.get('startTemp', 1.0),
                arena_temp=args.get('arenaTemp', 0.25), temp_fn=args.get('temp_scaling_fn', default_temp_scaling),
                temp_table_override=args.get('_azg_temp_table'),     # (iteration.serve: the schedule of a callable that does not pickle)
                seed=seed, slot_base=self.slot_base + li * Bl, device=device,
                example_capacity=example_capacity // self.pipelines + 1, result_capacity=int(result_capacity) // self.pipelines + 1,
                sims_hint=sims, nodes_per_tree=nodes_per_tree)
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
        # the whole simulation loop of a move as ONE persistent launch where the network has one (azg_search_f16 / azg_search_wide_f16)
        hip = getattr(nnet, '_hip', None) if nnet is not None else None
        self.fused_search = (fused_search is None or bool(fused_search)) and self.round_graph and not self.warmup \
            and hip is not None and hip.can_search and self.game == hip.game

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

    def _issue_round(self, ln, sims, fast):
        """The launch sequence of one whole round of a lane: sims x (find_leaf, network, process_results) + playMoves, with
        backup k and select k + 1 sharing a launch (or the whole loop in one persistent launch)."""
        e = ln.engine
        if self.fused_search:
            self.nnet._hip.search(e, sims, exact=self.search_exact)
            e.advance(record_history=not fast)
            return
        e.select(ln.obs)
        logits_path = not self.warmup and ln.net.run_logits is not None    # wide heads: softmax inside the tree launch
        feat_path = not self.warmup and getattr(ln.net, 'run_features', None) is not None   # factorised: the heads too
        if self.heads is None and feat_path and logits_path and self.search_exact:   # (the default hand-over is the exact one in every
            feat_path = False                                                        #  launch form; a pinned heads= form is taken as given)
        if self.heads is not None and not self.warmup:
            if (self.heads == 'features' and not feat_path) or (self.heads == 'logits' and not logits_path):
                raise NotImplementedError('this network has no %s hand-over to the tree launch' % self.heads)
            feat_path, logits_path = self.heads == 'features', self.heads == 'logits'
        for i in range(sims):                                    # backup k and select k + 1 share a launch
            if feat_path:
                feat, rows, hb = ln.net.run_features()
                e.backup_select_features(feat, rows, hb, ln.obs, select=i + 1 < sims)
                continue
            if logits_path:
                e.backup_select_logits(ln.net.run_logits(), ln.obs, select=i + 1 < sims)
                continue
            p, v = (ln.policy, ln.value) if self.warmup else ln.net.run()
            if i + 1 < sims:
                e.backup_select(p, v, ln.obs)
            else:
                e.backup(p, v)
        e.advance(record_history=not fast)

    def _round_graph(self, ln, sims, fast):
        """One whole round of a lane as ONE hipGraph: every launch is stream-ordered and nothing is read on the host, so the
        host issues one replay per move instead of 3 calls per simulation (the small configs are otherwise bound by host
        launch rate, not by the GPU)."""
        key = (sims, fast)
        if key not in ln.round_graphs:
            e = ln.engine
            if self.fused_search:
                self.nnet._hip.search(e, 0, exact=self.search_exact)  # one-time setup outside the capture
            torch.cuda.synchronize(e.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                self._issue_round(ln, sims, fast)
            ln.round_graphs[key] = g
        return ln.round_graphs[key]

    def prepare(self):
        """Capture the round graph(s) now (nothing is executed), so that the first play_round does not pay for it."""
        if self.round_graph:
            a = self.args
            kinds = [(int(a.get('numWarmupSims', 5) if self.warmup else a.get('numMCTSSims', 100)), False)]
            if float(a.get('probFastSim', 0.0) or 0.0) > 0:
                kinds.append((int(a.get('numFastSims', 20)), True))
            for ln in self.lanes:
                with torch.cuda.stream(ln.stream):
                    for sims, fast in kinds:
                        self._round_graph(ln, sims, fast)
        return self

    def play_round(self, eager=False):
        """eager=True issues the captured launch sequence as plain launches (measurement: events between the launches)."""
        sims, fast = self._sims_for_round()
        if self.round_graph and eager:
            for ln in self.lanes:
                with torch.cuda.stream(ln.stream), torch.no_grad():
                    self._issue_round(ln, sims, fast)
        elif self.round_graph and not any(getattr(ln.engine, 'profiling', False) for ln in self.lanes):
            for ln in self.lanes:
                with torch.cuda.stream(ln.stream):
                    self._round_graph(ln, sims, fast).replay()
        else:
            for _ in range(sims):
                self.step()
            for ln in self.lanes:                                    # playMoves :153-202
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
                tot['max_nodes_kept'] = max(tot['max_nodes_kept'], c['max_nodes_kept'])
        return tot

    def run(self, games=None, max_rounds=None, poll_every=1):
        """Play rounds until `games` (default args.gamesPerIteration) games have finished.  Returns counters."""
        games = int(self.args.get('gamesPerIteration') if games is None else games)
        cap = int(self.args.get('gamesPerIteration', 1 << 30))
        if games > cap:                                              # the lanes' quotas sum to gamesPerIteration: more can never be counted
            raise ValueError('run(games=%d) exceeds args.gamesPerIteration=%d, the cap the engines were created with' % (games, cap))
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

    def save_iteration_samples(self, folder, iteration, first_per_lane=None):
        """Write the iteration's samples the way Coach.saveIterationSamples does (Coach.py:363-386): three CPU float32 tensors
        `iteration-NNNN-{data,policy,value}.pkl` (torch.save; iteration.write_iteration_files) that the unchanged Coach.train loads
        (:444-456).  Returns the number of samples."""
        from .iteration import write_iteration_files
        data, policy, value = self.samples(first_per_lane)
        write_iteration_files(folder, iteration, data, policy, value)
        return int(data.shape[0])

    def game_results(self, first_per_lane=None):
        """(wins per player, draws, average game length) -- utils.get_game_results (:34-54), what Coach.processGameResults
        logs (Coach.py:388-398) -- over every finished game."""
        ws, turns, _ = self.results(first_per_lane)
        P = self.game_cls.num_players()
        wins = [int(ws[:, p].sum()) for p in range(P)] if len(ws) else [0] * P
        draws = int(ws[:, P].sum()) if len(ws) else 0
        return wins, draws, (float(turns.sum()) / len(turns) if len(turns) else 0)

    def results(self, first_per_lane=None):
        """(winstate u8[n, P+1], turns i32[n], slot i32[n]) of every finished game in result_queue order, lanes in slot order;
        first_per_lane: skip the records a lane held before (the marks of iteration.SelfPlayIteration.begin)."""
        import numpy as np
        rs = [ln.engine.results(0 if first_per_lane is None else first_per_lane[i]) for i, ln in enumerate(self.lanes)]
        Bl = self.B // self.pipelines
        return (np.concatenate([r[0] for r in rs]), np.concatenate([r[1] for r in rs]),
                np.concatenate([r[2] + i * Bl for i, r in enumerate(rs)]))


class ArenaRunner:
    """Native batched Arena (BASELINE config 4): the replacement of the batched branch of Arena.play_games
    (alphazero/Arena.pyx:208-328) + the arena mode of SelfPlayAgent (SelfPlayAgent.pyx:44-47,60-73,117-132,142-151,158,
    167-168).  One engine with one tree per player per game; every simulation the leaf rows are grouped by model
    (player_to_index[mover]), each model evaluates its own contiguous slice, and the results are routed back with the
    correct row <-> game map (the reference's mis-routing, SURVEY.md Q15, is not reproduced).  Returns the
    (wins, draws, winrates) contract of Arena.play_games (:376) via get_game_results semantics (utils.py:34-54)."""

    def __init__(self, game_cls, nnets, args, *, num_slots, seed=0, slot_base=0, device=None, use_graph=True, result_capacity=None,
                 seats='agent', nodes_per_tree=0, fused_search=None):
        self.game_cls, self.nnets, self.args = game_cls, list(nnets), args
        self.game = azg_game_id(game_cls)
        self.B = int(num_slots)
        P = game_cls.num_players()
        assert len(self.nnets) == P
        self.seed, self.slot_base = int(seed), int(slot_base)
        # seat permutation, one per agent, drawn from the agent-level tape stream (SelfPlayAgent.pyx:44-47)
        import ctypes as C
        pos = (C.c_int32 * P)()
        _abi.lib().azg_tape_shuffle_pos(self.seed, AGENT_STREAM + self.slot_base, 0, P, pos)
        self.player_to_index = [0] * P
        for i in range(P):
            self.player_to_index[pos[i]] = i
        self.engine = DeviceEngine(self.game, self.B, arena=True, cpuct=args.get('cpuct', 1.25),
                                   fpu_reduction=args.get('fpu_reduction', 0.2), arena_temp=args.get('arenaTemp', 0.25),
                                   games_per_iteration=int(args.get('gamesPerIteration', 1 << 30)), seed=seed,
                                   slot_base=slot_base, device=device, sims_hint=int(args.get('numMCTSSims', 100)),
                                   nodes_per_tree=nodes_per_tree, result_capacity=int(result_capacity if result_capacity is not None else
                                                       min(int(args.get('gamesPerIteration', 1 << 30)), 1 << 20) + 4 * self.B + 1024))
        e = self.engine
        # seats = 'agent': the one permutation above for every game (the reference, SelfPlayAgent.pyx:44-47);
        # seats = 'slot': every concurrent game draws its own seating from its slot's stream of the tape, so that
        # one engine of B games is as balanced as the reference's many agents
        assert seats in ('agent', 'slot')
        self.seats = seats
        self.slot_seats = None
        if seats == 'slot':
            perm = []
            for sl in range(self.B):
                _abi.lib().azg_tape_shuffle_pos(self.seed ^ 0x5EA7, AGENT_STREAM + self.slot_base + 1 + sl, 0, P, pos)
                m = [0] * P
                for i in range(P):
                    m[pos[i]] = i
                perm.append(m)
            self.slot_player_to_index = perm                        # [slot][player] -> model
            packed = [sum(m[p] << (4 * p) for p in range(P)) for m in perm]
            self.slot_seats = torch.tensor(packed, dtype=torch.int32, device=e.device)
        hip = all(getattr(n, '_hip', None) is not None or (n.refresh() and n._hip is not None) for n in self.nnets)
        self.nhwc8 = bool(hip)
        self.obs = (torch.zeros((self.B, e.gi.obs_h * e.gi.obs_w, 8), dtype=torch.float16, device=e.device) if self.nhwc8
                    else e.new_obs(torch.float16))
        self.policy = torch.zeros((self.B, e.A), dtype=torch.float32, device=e.device)
        self.value = torch.zeros((self.B, e.NV), dtype=torch.float32, device=e.device)
        # fused tower + heads on every model: the per-model batch split never leaves the device
        self.device_split = bool(hip) and all(n._hip.fused_head for n in self.nnets)
        # a whole move as ONE persistent launch (azg_search_arena_f16: one game per workgroup, the mover's tree, the mover's model) where the
        # models have it -- connect4 x 128 channels --, else one multi-model tower launch + one tree launch per simulation
        can = self.device_split and self.game == 0 and all(n._hip.CH == 128 for n in self.nnets)
        if fused_search and not can:
            raise NotImplementedError('no persistent arena launch for these models / this game')
        self.fused_search = can if fused_search is None else bool(fused_search)
        self._graph = None
        if self.device_split and use_graph:
            self.capture()

    def _rows(self):
        e = self.engine
        return e.arena_rows_seats(self.slot_seats) if self.slot_seats is not None else e.arena_rows(self.player_to_index)

    def step(self):
        if self.device_split:
            self._step_device_split()
            return
        e = self.engine
        row_of_slot, rpm = self._rows()
        e.select(self.obs, row_of_slot)
        counts = rpm.cpu().tolist()                                  # host read of the batch split, once per simulation
        off = 0
        for mi, n in enumerate(counts):
            if n:
                x = self.obs[off:off + n]
                p, v = (self.nnets[mi]._hip.forward_nhwc8(x, key=100 + mi) if self.nhwc8 else self.nnets[mi].process(x))
                self.policy[off:off + n] = p; self.value[off:off + n] = v
            off += n
        e.backup(self.policy, self.value, row_of_slot)

    def _step_device_split(self):
        """rows -> select -> every model on its own slice (the split stays on the device) -> backup: no host read, so the
        whole simulation step is one hipGraph."""
        e = self.engine
        row_of_slot, rpm = self._rows()
        e.select(self.obs, row_of_slot)
        HipResNet.forward_models([n._hip for n in self.nnets], self.obs, self.policy, self.value, rpm)    # one launch
        e.backup(self.policy, self.value, row_of_slot)

    def _round_device_split(self, sims):
        """A whole move: the mover of every game -- hence the row <-> game map and the per-model split -- is fixed until
        advance, so the rows are laid out once; then select, and per simulation ONE tower launch for all models plus one
        tree launch (backup k + select k + 1); advance."""
        e = self.engine
        if self.fused_search:
            HipResNet.search_arena([n._hip for n in self.nnets], e, sims, None if self.slot_seats is not None else self.player_to_index, self.slot_seats)
            e.advance(record_history=False)
            return
        row_of_slot, rpm = self._rows()
        e.select(self.obs, row_of_slot)
        nets = [n._hip for n in self.nnets]
        for i in range(sims):
            HipResNet.forward_models(nets, self.obs, self.policy, self.value, rpm)
            if i + 1 < sims:
                e.backup_select(self.policy, self.value, self.obs, row_of_slot)
            else:
                e.backup(self.policy, self.value, row_of_slot)
        e.advance(record_history=False)

    def capture(self):
        """Capture one whole round (all simulations of a move + advance) as a hipGraph (device-side split only)."""
        assert self.device_split
        self._step_device_split()                                    # warm: lazy allocations happen outside the capture
        if self.fused_search:
            HipResNet.search_arena([n._hip for n in self.nnets], self.engine, 0, None if self.slot_seats is not None else self.player_to_index, self.slot_seats)
        self.engine.reset()                                          # (the warm step is not part of any game)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._round_device_split(int(self.args.get('numMCTSSims', 100)))
        self._graph = g

    def play_round(self, eager=False):
        if self.device_split and (self._graph is not None or self.fused_search):
            if eager or self._graph is None:
                self._round_device_split(int(self.args.get('numMCTSSims', 100)))
            else:
                self._graph.replay()
            return
        for _ in range(int(self.args.get('numMCTSSims', 100))):
            self.step()
        self.engine.advance(record_history=False)

    def run(self, games=None):
        games = int(self.args.get('gamesPerIteration') if games is None else games)
        while True:
            self.play_round()
            c = self.engine.counters()
            if c['games_played'] >= games:
                return c

    def results(self, first=0):
        """(wins per MODEL, draws, winrates) like Arena.play_games: winstate index -> model via player_to_index.  first: skip the
        result records the engine held before (iteration.ArenaIteration.begin)."""
        ws, turns, slot = self.engine.results(first)
        P = self.game_cls.num_players()
        wins, draws = [0] * P, 0
        for w, sl in zip(ws, slot):
            p2i = self.slot_player_to_index[int(sl)] if self.slot_seats is not None else self.player_to_index
            for p in range(P + 1):
                if w[p]:
                    if p == P:
                        draws += 1
                    else:
                        wins[p2i[p]] += 1
        n = max(len(ws), 1)
        return wins, draws, [x / n for x in wins]
