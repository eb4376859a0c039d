"""One self-play iteration / one arena comparison as a library component, on 1..N GPUs (SURVEY.md 8e + 8f-1 + 8f-3).

This is what stands behind the reference's `Coach.learn()` in NATIVE mode (alphazero/Coach.py:225-288): the five calls
generateSelfPlayAgents -> processSelfPlayBatches -> saveIterationSamples -> processGameResults -> killSelfPlayAgents
(:291,326,364,389,401) and the batched branch of gating (compareToPast :528-572 -> Arena.play_games, Arena.pyx:208-328, return
contract :376) map onto

    SelfPlayIteration / run_iteration   per-rank SelfPlayRunner (quota = shard_games, slot_base = rank * B), zero communication
                                        while games are played, then ONE exchange step: all-gather of the (state, pi, z) shards +
                                        all-reduce of the win / draw / length tallies; rank 0 writes iteration-NNNN-{data,policy,
                                        value}.pkl (the files the unchanged Coach.train loads, :442-456)
    ArenaIteration / run_arena          per-rank ArenaRunner over its share of the games, all-reduce of the tallies ->
                                        (wins per player object, draws, winrates) with the reference's winrate rule (Arena.pyx:124-131)
    lead / serve                        a Coach that lives on rank 0 drives the other ranks: the command and the net's weights are
                                        broadcast (distributed.broadcast_state_dict), every rank plays its shard

`bench.py` times exactly these objects (its timed region is `SelfPlayIteration.play_round` x K + `exchange()`), and
`coach.native_coach` binds them to the reference's method names.  Everything here is collective: every rank of the group calls the
same function with the same arguments; without an initialised process group it is the 1-GPU case.
"""
import os
import time

import torch
import torch.distributed as dist

from . import distributed as D
from .utils import dotdict, default_temp_scaling, temp_table


def _rank_world(group=None):
    if dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def iteration_seed(base, iteration):
    """the tape seed of iteration i: every iteration plays different games, the same (base, i) the same ones on any number of ranks"""
    return (int(base) * 0x9E3779B1 + int(iteration) * 0x85EBCA77 + 1) & 0x7FFFFFFFFFFFFFFF


def default_slots(args, world=1):
    """concurrent games per rank: the reference keeps workers x process_batch_size games in flight (Coach.py:294-297), split over ranks"""
    total = int(args.get('_azg_slots') or int(args.get('workers', 1)) * int(args.get('process_batch_size', 256)))
    return max((total + world - 1) // world, 1)


# ---------------------------------------------------------------------------------------------------- the exchange step, file format
def exchange_selfplay(obs, pi, z, tallies, group=None):
    """THE communication of a self-play iteration: variable-length all-gather of the example shards (rank order, each
    rank's samples in its own output_queue order) + the sum of the small integer tallies.  Returns ((obs, pi, z), tallies)."""
    g = D.all_gather_examples(obs, pi, z, group)
    return g, D.all_reduce_tallies(tallies, group)


def write_iteration_files(folder, iteration, data, policy, value):
    """Coach.saveIterationSamples' three files (Coach.py:373-383; name: utils.get_iter_file :15-16): float32 CPU tensors written with
    torch.save.  Pickle protocol: the reference asks for pickle.HIGHEST_PROTOCOL, which its own loader -- a bare torch.load
    (Coach.py:448-450) -- only reads back on the torch it pins (< 2.5); from torch 2.6 on torch.load defaults to the weights-only
    unpickler, which refuses protocol 5 files.  torch.save's default protocol (2) holds the same three tensors and is read by both, so
    that is what is written: the unchanged Coach.train consumes these files on either torch (tests/test_iteration_cpu.py runs the real
    loader on them).  Returns the file stem."""
    os.makedirs(folder, exist_ok=True)
    stem = os.path.join(folder, 'iteration-%04d' % int(iteration))
    for name, t in (('data', data), ('policy', policy), ('value', value)):
        torch.save(t.detach().to('cpu', torch.float32).contiguous(), '%s-%s.pkl' % (stem, name))
    return stem


def winrates(wins, draws, use_draws):
    """Arena.__update_winrates (Arena.pyx:124-131) + _PlayerStats.update (:32-36)"""
    n = sum(wins) + (draws if use_draws else 0)
    return [((w + 0.5 * (draws if use_draws else 0)) / n) if n else 0 for w in wins]


# ---------------------------------------------------------------------------------------------------- self-play
class SelfPlayIteration:
    """This rank's share of one self-play iteration.  args.gamesPerIteration is the WHOLE job's cap; this rank counts the first
    shard_games(cap, rank, world) games it finishes (quotas instead of a global atomic) on `num_slots` concurrent games
    whose global slot ids start at rank * num_slots -- so N ranks of B slots play the games one engine of N x B slots would."""

    def __init__(self, game_cls, nnet, args, *, num_slots, seed=0, group=None, warmup=False, device=None, **runner_kw):
        from .selfplay import SelfPlayRunner
        self.group = group
        self.rank, self.world = _rank_world(group)
        self.game_cls, self.args = game_cls, args
        self.total_games = int(args.get('gamesPerIteration', 1 << 30))
        self.quota = self.total_games if self.total_games >= (1 << 30) else D.shard_games(self.total_games, self.rank, self.world)
        self.B = int(num_slots)
        rargs = dotdict(args)
        rargs['gamesPerIteration'] = self.quota
        self.runner = None
        if self.quota > 0:                                           # (more ranks than games: a rank without a quota only joins the exchange)
            self.runner = SelfPlayRunner(game_cls, nnet, rargs, num_slots=self.B, seed=seed, slot_base=D.slot_base(self.rank, self.B),
                                         device=device, warmup=warmup, **runner_kw)
        self.device = self.runner.device if self.runner is not None else torch.device('cuda', torch.cuda.current_device() if device is None else device)
        self.rounds = 0
        self.begin()

    # -- marks: an iteration is what happens between begin() and exchange()
    def begin(self):
        r = self.runner
        self._c0 = r.counters() if r is not None else None
        self._ex0 = [ln.engine.counters()['num_examples'] for ln in r.lanes] if r is not None else None
        self._res0 = [ln.engine.counters()['num_results'] for ln in r.lanes] if r is not None else None
        self._t0 = time.time()
        self._rounds0 = self.rounds
        return self

    def prepare(self):
        if self.runner is not None:
            self.runner.prepare()
        return self

    def play_round(self):
        if self.runner is not None:
            self.runner.play_round()
        self.rounds += 1

    def games_counted(self):
        return 0 if self.runner is None else self.runner.counters()['games_played'] - self._c0['games_played']

    def done(self):
        """has this rank filled its quota? (one small D2H)"""
        return self.runner is None or self.runner.counters()['games_played'] >= self.quota

    def play(self, max_rounds=None, stop=None, poll_every=1):
        """SelfPlayAgent.run's outer loop (SelfPlayAgent.pyx:79-94) for this rank: rounds until the quota is counted (or `stop()`)."""
        n = 0
        while self.runner is not None:
            self.play_round()
            n += 1
            if n % poll_every == 0 and self.done():
                break
            if (max_rounds is not None and n >= max_rounds) or (stop is not None and stop()):
                break
        return n

    def local_samples(self):
        if self.runner is None:
            C, H, W = self.game_cls.observation_size()
            A, NV = self.game_cls.action_size(), self.game_cls.num_players() + 1
            return (torch.zeros((0, C, H, W), dtype=torch.float32, device=self.device), torch.zeros((0, A), dtype=torch.float32, device=self.device),
                    torch.zeros((0, NV), dtype=torch.float32, device=self.device))
        return self.runner.samples(self._ex0)

    def local_tallies(self):
        """[wins per player ..., draws, sum of game lengths, finished games, counted games, expansions, simulations, samples]"""
        P = self.game_cls.num_players()
        if self.runner is None:
            return [0] * (P + 7)
        ws, turns, _ = self.runner.results(self._res0)
        c1 = self.runner.counters()
        wins = [int(ws[:, p].sum()) for p in range(P)] if len(ws) else [0] * P
        return wins + [int(ws[:, P].sum()) if len(ws) else 0, int(turns.sum()), int(len(turns)), c1['games_played'] - self._c0['games_played'],
                       c1['expansions'] - self._c0['expansions'], c1['sims'] - self._c0['sims'], c1['num_examples'] - self._c0['num_examples']]

    def exchange(self):
        """The iteration's exchange step.  Every rank gets the gathered samples and the summed tallies:
        dict(samples=(data, policy, value), wins, draws, avg_game_length, num_results, games, expansions, sims, num_samples)."""
        P = self.game_cls.num_players()
        (gobs, gpi, gz), t = exchange_selfplay(*self.local_samples(), self.local_tallies(), self.group)
        t = [int(x) for x in t]
        nres = t[P + 2]
        return dict(samples=(gobs, gpi, gz), wins=t[:P], draws=t[P], avg_game_length=(t[P + 1] / nres if nres else 0), num_results=nres,
                    games=t[P + 3], expansions=t[P + 4], sims=t[P + 5], num_samples=int(gobs.shape[0]), rounds=self.rounds - self._rounds0,
                    seconds=time.time() - self._t0, ranks=self.world)

    def close(self):
        if self.runner is not None:
            for ln in self.runner.lanes:
                ln.engine.close()
            self.runner = None


def run_iteration(game_cls, nnet, args, iteration, folder=None, *, num_slots=None, seed=None, warmup=False, group=None, stop=None,
                  max_rounds=None, keep_samples=True, **runner_kw):
    """generateSelfPlayAgents + processSelfPlayBatches + saveIterationSamples + processGameResults + killSelfPlayAgents
    (Coach.py:291-435) in native mode: play args.gamesPerIteration games over all ranks, exchange, rank 0 writes the three sample
    files under `folder` (skipped when None), and every rank returns the same record (see SelfPlayIteration.exchange; 'samples' is
    dropped unless keep_samples)."""
    rank, world = _rank_world(group)
    B = int(num_slots) if num_slots else default_slots(args, world)
    sd = iteration_seed(args.get('_azg_seed', 0) if seed is None else seed, iteration)
    it = SelfPlayIteration(game_cls, nnet, args, num_slots=B, seed=sd, group=group, warmup=warmup, **runner_kw)
    try:
        it.play(max_rounds=max_rounds, stop=stop)
        out = it.exchange()
    finally:
        it.close()
    out['iteration'], out['slots_per_rank'] = int(iteration), B
    if folder is not None and rank == 0:
        out['files'] = write_iteration_files(folder, iteration, *out['samples'])
    if not keep_samples:
        out.pop('samples')
    return out


# ---------------------------------------------------------------------------------------------------- arena
class ArenaIteration:
    """This rank's share of one batched arena comparison (Arena.play_games' batched branch, Arena.pyx:208-328): num_games over all
    ranks, every rank an ArenaRunner with all models resident.  seats: 'slot' (default: every concurrent game draws its own
    seating -- one engine is then as balanced as the reference's many agents) or 'agent' (the reference: one permutation per
    agent, SelfPlayAgent.pyx:44-47)."""

    def __init__(self, game_cls, nnets, args, num_games, *, num_slots, seed=0, group=None, seats='slot', device=None, **runner_kw):
        from .selfplay import ArenaRunner
        self.group = group
        self.rank, self.world = _rank_world(group)
        self.game_cls, self.args = game_cls, args
        self.total_games = int(num_games)
        self.quota = D.shard_games(self.total_games, self.rank, self.world)
        self.B = max(min(int(num_slots), max(self.quota, 1)), 1)
        rargs = dotdict(args)
        rargs['gamesPerIteration'] = self.quota                      # (Arena.play_games :233 sets it to num)
        self.runner = None
        if self.quota > 0:
            self.runner = ArenaRunner(game_cls, nnets, rargs, num_slots=self.B, seed=seed, slot_base=D.slot_base(self.rank, int(num_slots)),
                                      device=device, seats=seats, **runner_kw)
        self.rounds = 0
        self.begin()

    def begin(self):
        e = self.runner.engine if self.runner is not None else None
        self._c0 = e.counters() if e is not None else None
        self._t0, self._rounds0 = time.time(), self.rounds
        return self

    def play_round(self):
        if self.runner is not None:
            self.runner.play_round()
        self.rounds += 1

    def done(self):
        return self.runner is None or self.runner.engine.counters()['games_played'] >= self.quota

    def play(self, max_rounds=None, stop=None):
        n = 0
        while self.runner is not None:
            self.play_round()
            n += 1
            if self.done() or (max_rounds is not None and n >= max_rounds) or (stop is not None and stop()):
                break
        return n

    def local_tallies(self):
        """[wins per MODEL ..., draws, finished games, counted games, expansions, simulations]"""
        P = self.game_cls.num_players()
        if self.runner is None:
            return [0] * (P + 5)
        wins, draws, _ = self.runner.results(self._c0['num_results'])
        c1 = self.runner.engine.counters()
        return list(wins) + [draws, c1['num_results'] - self._c0['num_results'], c1['games_played'] - self._c0['games_played'],
                             c1['expansions'] - self._c0['expansions'], c1['sims'] - self._c0['sims']]

    def exchange(self):
        """all-reduce of the tallies (the arena has no example exchange).  dict(wins, draws, winrates, num_results, games, ...)"""
        P = self.game_cls.num_players()
        t = [int(x) for x in D.all_reduce_tallies(self.local_tallies(), self.group)]
        wins, draws = t[:P], t[P]
        return dict(wins=wins, draws=draws, winrates=winrates(wins, draws, bool(self.args.get('use_draws_for_winrate', True))),
                    num_results=t[P + 1], games=t[P + 2], expansions=t[P + 3], sims=t[P + 4], rounds=self.rounds - self._rounds0,
                    seconds=time.time() - self._t0, ranks=self.world)

    def close(self):
        if self.runner is not None:
            self.runner.engine.close()
            self.runner = None


def run_arena(game_cls, nnets, args, num_games, *, num_slots=None, seed=None, seats='slot', group=None, stop=None, max_rounds=None,
              details=False, **runner_kw):
    """Arena.play_games(num) for model players, batched branch (Arena.pyx:208-328), over all ranks -> (wins, draws, winrates), the
    return contract of :376 (wins / winrates indexed like `nnets`, i.e. like Arena.players).  details=True returns the whole record."""
    rank, world = _rank_world(group)
    if num_slots is None:
        total = int(args.get('_azg_arena_slots') or int(args.get('workers', 1)) * int(args.get('arena_batch_size', 64)))
        num_slots = max((min(total, int(num_games)) + world - 1) // world, 1)
    sd = iteration_seed(args.get('_azg_seed', 0) if seed is None else seed, 0x41524E41)
    it = ArenaIteration(game_cls, nnets, args, num_games, num_slots=num_slots, seed=sd, group=group, seats=seats, **runner_kw)
    try:
        it.play(max_rounds=max_rounds, stop=stop)
        out = it.exchange()
    finally:
        it.close()
    return out if details else (out['wins'], out['draws'], out['winrates'])


# ---------------------------------------------------------------------------------------------------- a Coach on rank 0 drives N ranks
# args keys the runners read (selfplay.SelfPlayRunner / ArenaRunner): only these travel to the other ranks -- a Coach's args hold
# things that do not pickle (baselineTester classes, lambdas)
ARG_KEYS = ('cpuct', 'fpu_reduction', 'root_noise_frac', 'root_policy_temp', 'min_discount', 'add_root_noise', 'add_root_temp',
            'symmetricSamples', 'mctsResetThreshold', 'gamesPerIteration', 'startTemp', 'arenaTemp', 'numMCTSSims', 'numFastSims',
            'numWarmupSims', 'probFastSim', 'workers', 'process_batch_size', 'arena_batch_size', 'use_draws_for_winrate',
            '_azg_seed', '_azg_slots', '_azg_arena_slots')


def portable_args(args, game_cls):
    """the picklable slice of a Coach's args; the temperature schedule (a callable, utils.py:19-31) travels as its table"""
    out = {k: args[k] for k in ARG_KEYS if k in args}
    out['_azg_temp_table'] = temp_table(args.get('temp_scaling_fn', default_temp_scaling), args.get('startTemp', 1.0),
                                        game_cls.max_turns()).tolist()
    return out


def _net_from(game_cls, sd, net_args, device):
    from .nnet import NNetWrapper
    return NNetWrapper(game_cls, dotdict(net_args), device=device).adopt(sd, net_args)


def lead(op, game_cls, nnets, args, *, iteration=0, folder=None, num_games=None, group=None, stop=None, **kw):
    """rank 0 (where the Coach lives): tell the serving ranks what to play, hand them the weights, play this rank's own shard.
    op = 'selfplay' (nnets = [net]) -> run_iteration's record; 'arena' -> run_arena's triple; 'stop' ends serve() on the others.
    `nnets`: this package's NNetWrappers, already holding the weights to play with (NNetWrapper.adopt).  `stop` (a callable polled
    once per round) is local to this rank: the others finish their quota, the exchange step waits for them."""
    from .nnet import DEFAULT_NET_ARGS
    rank, world = _rank_world(group)
    assert rank == 0, 'lead() is called by rank 0 only; the other ranks sit in serve()'
    if world > 1:
        if op == 'stop':
            D.broadcast_object({'op': 'stop'}, group=group)
            return None
        uniq, index = [], []                                         # (gating: [new] + [past] * (P - 1) -- every distinct net travels once)
        for n in nnets:
            j = -1 if n is None else next((j for j, m in enumerate(uniq) if m is n), None)     # (None: a warm-up iteration plays without a net)
            if j is None:
                j = len(uniq); uniq.append(n)
            index.append(j)
        cmd = {'op': op, 'args': portable_args(args, game_cls), 'iteration': int(iteration), 'num_games': num_games, 'kw': kw,
               'index': index, 'net_args': [{k: n.args[k] for k in DEFAULT_NET_ARGS if k in n.args} for n in uniq], 'meta': [D.state_dict_meta(n.nnet.state_dict()) for n in uniq]}
        D.broadcast_object(cmd, group=group)
        for n, m in zip(uniq, cmd['meta']):
            D.broadcast_state_dict(n.nnet.state_dict(), m, group=group)
    elif op == 'stop':
        return None
    if op == 'selfplay':
        return run_iteration(game_cls, nnets[0], args, iteration, folder, group=group, stop=stop, **kw)
    return run_arena(game_cls, nnets, args, num_games, group=group, stop=stop, **kw)


def serve(game_cls, device=None, group=None):
    """ranks != 0 of a Coach-driven job: wait for rank 0's command, receive the weights, play this rank's shard, join the exchange;
    returns when rank 0 sends 'stop' (coach.native_coach does at the end of learn()).  Returns the number of commands served."""
    rank, world = _rank_world(group)
    assert world > 1 and rank != 0, 'serve() is for the non-zero ranks of an initialised process group'
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    served = 0
    while True:
        cmd = D.broadcast_object(None, group=group)
        if cmd['op'] == 'stop':
            return served
        args = dotdict(cmd['args'])
        uniq = [_net_from(game_cls, D.broadcast_state_dict(None, m, group=group), na, dev) for na, m in zip(cmd['net_args'], cmd['meta'])]
        nnets = [uniq[j] if j >= 0 else None for j in cmd['index']]
        if cmd['op'] == 'selfplay':
            run_iteration(game_cls, nnets[0], args, cmd['iteration'], None, group=group, **dict(cmd['kw'], keep_samples=False))
        else:
            run_arena(game_cls, nnets, args, cmd['num_games'], group=group, **cmd['kw'])
        served += 1
