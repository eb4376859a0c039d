"""NATIVE mode behind the reference's unchanged `Coach.learn()` (alphazero/Coach.py:225-288).

    from alphazero.Coach import Coach                      # the reference's own class, untouched
    from alphazero_general_amd.coach import native_coach
    coach = native_coach(Coach)(Game, nnet, args)          # `nnet`: the reference's NNetWrapper (it keeps training on it)
    coach.learn()

`native_coach` returns a subclass that overrides EXACTLY the five methods learn() calls for self-play -- generateSelfPlayAgents,
processSelfPlayBatches, saveIterationSamples, processGameResults, killSelfPlayAgents (Coach.py:291,326,364,389,401) -- and rebinds the
name `Arena` in the Coach's module to `native_arena(Arena)`, whose play_games takes the batched branch (Arena.pyx:223-328) to the
device when every player is a model player of a game with device rules.  Everything else -- train(), gating, checkpoints, the
TensorBoard writer, compareToBaseline against non-model players -- runs the reference's code.  No agent processes, no shared
tensors, no queues: the games live on the GPU(s) (iteration.run_iteration / run_arena), the live net's weights are adopted in
memory (NNetWrapper.adopt), and the three sample files the unchanged Coach.train loads (:442-456) are written by rank 0.

Multi-GPU: start the training script under torch.distributed.run; rank 0 builds the Coach, the other ranks call
`alphazero_general_amd.iteration.serve(Game)`.
"""
import os
import sys
import time

from . import iteration as it_mod
from .Game import has_device_rules


def _ours(game_cls):
    """this package's GameState class for a game with device rules (the reference's own env classes are recognised by module name,
    Game.azg_game_id), or None: the runners read the static protocol (observation_size, max_turns, symmetries ...) from it -- the
    reference's brandubh class lacks has_draw / max_turns at this snapshot"""
    import importlib
    from .Game import azg_game_id
    if not has_device_rules(game_cls):
        return None
    return importlib.import_module(__package__ + '.envs.' + ('connect4', 'brandubh', 'trimok')[azg_game_id(game_cls)]).Game


class _NetCache:
    """this package's NNetWrapper per reference wrapper, re-adopting the live weights before every use"""

    def __init__(self, game_cls, device=None):
        self.game_cls, self.device, self.map = game_cls, device, {}

    def get(self, ref_net):
        from .nnet import NNetWrapper
        if isinstance(ref_net, NNetWrapper):
            return ref_net
        w = self.map.get(id(ref_net))
        if w is None:
            w = self.map[id(ref_net)] = NNetWrapper(self.game_cls, ref_net.args, device=self.device)
        return w.adopt(ref_net)


def native_coach(Coach, *, device=None, install_arena=True):
    """class factory: the reference's Coach with its self-play phase on the device engine (see the module docstring)"""
    cmod = sys.modules[Coach.__module__]
    TrainState = getattr(cmod, 'TrainState', None)

    def _state(self, name):
        if TrainState is not None and hasattr(TrainState, name):
            self.state = getattr(TrainState, name)

    class NativeCoach(Coach):
        __doc__ = 'native_coach(%s): self-play and batched gating on the MI355X engine' % Coach.__name__

        def _azg_setup(self):
            if getattr(self, '_azg_game', None) is None:
                g = _ours(self.game_cls)
                if g is None:
                    raise NotImplementedError('%r has no device rules (alphazero_general_amd.envs): native mode cannot play it' % (self.game_cls,))
                self._azg_game, self._azg_nets, self._azg_result = g, _NetCache(g, device), None
            return self._azg_game

        def generateSelfPlayAgents(self):                            # Coach.py:291-323
            _state(self, 'INIT_AGENTS')
            self._azg_setup()
            src = self.self_play_net if self.args.model_gating else self.train_net      # (:334)
            self._azg_net = None if self.warmup else self._azg_nets.get(src)
            self.agents = []                                         # (learn()'s final `if self.agents: killSelfPlayAgents()` :286-287)
            _state(self, 'STANDBY')

        def processSelfPlayBatches(self, iteration):                 # Coach.py:326-361
            _state(self, 'SELF_PLAY')
            t0 = time.time()
            folder = os.path.join(self.args.data, self.args.run_name)
            # (the sample files are written here, by rank 0 right behind the exchange step; saveIterationSamples reports them)
            self._azg_result = r = it_mod.lead('selfplay', self._azg_game, [self._azg_net], self.args, iteration=iteration, folder=folder,
                                               warmup=bool(self.warmup), keep_samples=False, stop=self.stop_train.is_set)
            self.sample_time = (time.time() - t0) / max(r['games'], 1)
            self.iter_time = time.time() - t0
            self.writer.add_scalar('loss/sample_time', self.sample_time, iteration)
            _state(self, 'STANDBY')

        def saveIterationSamples(self, iteration):                   # Coach.py:364-386
            _state(self, 'SAVE_SAMPLES')
            print('Saving %d samples' % self._azg_result['num_samples'])
            _state(self, 'STANDBY')

        def processGameResults(self, iteration):                     # Coach.py:389-398
            _state(self, 'PROCESS_RESULTS')
            r = self._azg_result
            n = max(r['num_results'], 1)
            for i, w in enumerate(r['wins']):
                self.writer.add_scalar('win_rate/player%d' % i, (w + (0.5 * r['draws'] if self.args.use_draws_for_winrate else 0)) / n, iteration)
            self.writer.add_scalar('win_rate/draws', r['draws'] / n, iteration)
            self.writer.add_scalar('win_rate/avg_game_length', r['avg_game_length'], iteration)
            _state(self, 'STANDBY')

        def killSelfPlayAgents(self):                                # Coach.py:401-435
            _state(self, 'KILL_AGENTS')
            self._azg_net = None
            self.agents = []
            _state(self, 'STANDBY')

        def learn(self):
            try:
                return Coach.learn(self)
            finally:
                it_mod.lead('stop', getattr(self, '_azg_game', None), [], self.args)   # the serving ranks return from serve()

    NativeCoach.__name__ = NativeCoach.__qualname__ = 'Native' + Coach.__name__
    if install_arena and hasattr(cmod, 'Arena'):
        cmod.Arena = native_arena(cmod.Arena, device=device)
    return NativeCoach


def native_arena(Arena, *, device=None):
    """class factory: the reference's Arena whose batched play_games runs on the device when it can (all players carry a model `.nn`,
    Arena was built with use_batched_mcts, the game has device rules); anything else falls through to the reference's own code."""
    if getattr(Arena, '_azg_native', False):
        return Arena

    class NativeArena(Arena):
        _azg_native = True

        def play_games(self, num, verbose=False, shuffle_players=True):          # Arena.pyx:188-376
            g = _ours(self.game_cls)
            nets = [getattr(p, 'nn', None) for p in self.players]
            if not self.use_batched_mcts or g is None or any(n is None for n in nets):
                return Arena.play_games(self, num, verbose, shuffle_players)
            cache = _NetCache(g, device)
            ours = [cache.get(n) for n in nets]                      # (one wrapper per distinct reference net: [new] + [past] * (P - 1))
            self.total_games = num
            r = it_mod.lead('arena', g, ours, self.args, num_games=num, details=True,
                            seats='slot' if shuffle_players else 'agent', stop=self.stop_event.is_set)
            self.draws, self.games_played = r['draws'], r['games']
            stats = getattr(self, '_Arena__player_stats', None)      # (so that wins() / winrates() answer afterwards, :133-137)
            if stats is not None:
                for s, w, wr in zip(stats, r['wins'], r['winrates']):
                    s.wins, s.winrate = w, wr
            return r['wins'], r['draws'], r['winrates']

    NativeArena.__name__ = NativeArena.__qualname__ = 'Native' + Arena.__name__
    return NativeArena
