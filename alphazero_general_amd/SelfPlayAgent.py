"""`SelfPlayAgent` with the reference's constructor and process protocol (alphazero/SelfPlayAgent.pyx:13-202), so that
the reference's Coach.py (Coach.py:291-361) and Arena.pyx (Arena.pyx:236-328) can spawn it unchanged ("compat mode"):

    SelfPlayAgent(id, game_cls, ready_queue, batch_ready, batch_tensor, policy_tensor, value_tensor, output_queue,
                  result_queue, complete_count, games_played, stop_event, pause_event, args, _is_arena, _is_warmup)

It is an mp.Process; the caller owns every queue / event / shared tensor.  Per simulation the agent writes the leaf
observations into `batch_tensor` (self-play) or puts the per-model list on `output_queue` (arena, :125-132), puts its
id on `ready_queue`, waits for `batch_ready`, and reads `policy_tensor` / `value_tensor`; finished games go to
`result_queue` as (final_state, winstate, id) and, while `games_played < gamesPerIteration` (tested under the
caller's lock, :179-183), their samples to `output_queue` as (observation, pi, winstate float32).

Where the search runs: Coach forks its agents AFTER initialising the GPU in the parent, and a forked child cannot
use HIP.  The agent process therefore keeps the reference's role (queues, events, shared tensors) and drives a
separate worker process (a fresh interpreter, alphazero_general_amd/_engine_worker.py) that owns the device engine; the two exchange the batch through POSIX shared memory
and a pipe.  Compat mode pays that host hop per simulation by construction (the network lives in the parent); the
throughput path is alphazero_general_amd.selfplay.SelfPlayRunner.  Errors in the worker are re-raised in the agent,
which prints the traceback and exits like the reference (:100-101) -- but never leaves the parent waiting: it still
counts itself complete.
"""
import itertools
import os
import time
import traceback
from multiprocessing import shared_memory

import numpy as np
import torch
import torch.multiprocessing as mp

from .Game import azg_game_id, has_device_rules
from .utils import default_temp_scaling, temp_table


from ._engine_worker import engine_worker  # noqa: F401  (the device-side half; runs in its own process)


class SelfPlayAgent(mp.Process):
    def __new__(cls, id=None, game_cls=None, *a, **k):
        """Dispatch per game (SURVEY.md 8b "Game plugin"): an env without device rule kernels is handed to the REFERENCE'S OWN
        SelfPlayAgent (alphazero_general_amd.reference_class; reference side, not a CPU engine of this package) -- the object
        returned is then not an instance of this class, so __init__ below does not run on it."""
        if cls is SelfPlayAgent and game_cls is not None and not has_device_rules(game_cls):
            from . import reference_class
            ref = reference_class('SelfPlayAgent')
            if ref is not None and ref is not SelfPlayAgent:
                return ref(id, game_cls, *a, **k)
        return super().__new__(cls)

    def __init__(self, id, game_cls, ready_queue, batch_ready, batch_tensor, policy_tensor, value_tensor, output_queue,
                 result_queue, complete_count, games_played, stop_event, pause_event, args, _is_arena=False, _is_warmup=False):
        super().__init__()
        self.id = id
        self.game_cls = game_cls
        self.ready_queue = ready_queue
        self.batch_ready = batch_ready
        self.batch_tensor = batch_tensor
        self.batch_size = policy_tensor.shape[0] if _is_arena else self.batch_tensor.shape[0]     # :23-26
        self.policy_tensor = policy_tensor
        self.value_tensor = value_tensor
        self.output_queue = output_queue
        self.result_queue = result_queue
        self.games_played = games_played
        self.complete_count = complete_count
        self.stop_event = stop_event
        self.pause_event = pause_event
        self.args = args
        self._is_arena = _is_arena
        self._is_warmup = _is_warmup
        if _is_arena:
            self.player_to_index = list(range(game_cls.num_players()))        # :44-47 (read by Arena.play_games)
            np.random.shuffle(self.player_to_index)
            self.batch_indices = None
        self.fast = False
        self._conn = None
        self._more_sims = False          # run(): another simulation of this move follows the processBatch being made
        self._ahead = None               # the worker already ran that simulation's find_leaf (fused with the backup): its reply

    # ---- worker plumbing ----
    def _call(self, cmd, arg=None):
        self._conn.send((cmd, arg))
        status, out = self._conn.recv()
        if status == 'error':
            raise RuntimeError('device engine worker failed:\n' + out)
        return out

    def _check_pause(self):
        while self.pause_event.is_set():
            time.sleep(.1)

    def _start_worker(self):
        a = self.args
        gid = azg_game_id(self.game_cls)
        B, A, NV = self.batch_size, self.game_cls.action_size(), self.game_cls.num_players() + 1
        O = int(np.prod(self.game_cls.observation_size()))
        self._shm = {k: shared_memory.SharedMemory(create=True, size=4 * B * n) for k, n in (('obs', O), ('pol', A), ('val', NV))}
        self._obs_h = np.ndarray((B, O), np.float32, buffer=self._shm['obs'].buf)
        self._pol_h = np.ndarray((B, A), np.float32, buffer=self._shm['pol'].buf)
        self._val_h = np.ndarray((B, NV), np.float32, buffer=self._shm['val'].buf)
        sims = max(int(a.get('numMCTSSims', 100) or 0), int(a.get('numFastSims', 0) or 0), int(a.get('numWarmupSims', 0) or 0))
        tt = temp_table(a.get('temp_scaling_fn', default_temp_scaling), a.get('startTemp', 1.0), self.game_cls.max_turns())
        seed = int(a.get('_azg_seed', int.from_bytes(os.urandom(7), 'little'))) + 7919 * int(self.id)    # :81 np.random.seed()
        cfg = dict(game=gid, B=B, A=A, NV=NV, O=O, arena=bool(self._is_arena), temp_table=tt,
                   player_to_index=list(getattr(self, 'player_to_index', [])),
                   shm={k: v.name for k, v in self._shm.items()},
                   engine=dict(cpuct=a.cpuct, fpu_reduction=a.fpu_reduction, root_noise_frac=a.root_noise_frac,
                               root_policy_temp=a.root_policy_temp, min_discount=a.get('min_discount', 1.0),
                               add_root_noise=bool(a.get('add_root_noise', True)), add_root_temp=bool(a.get('add_root_temp', True)),
                               symmetric_samples=bool(a.get('symmetricSamples', True)),
                               mcts_reset_threshold=a.get('mctsResetThreshold', 0) or 0, games_per_iteration=1 << 30,
                               start_temp=a.get('startTemp', 1.0), arena_temp=a.get('arenaTemp', 0.25), seed=seed,
                               example_capacity=0 if self._is_arena else 8 * B * (self.game_cls.max_turns() or 64) + 1024,
                               sims_hint=sims))
        # a fresh interpreter (not a fork: HIP cannot be used in a forked child of a process that initialised it; not an
        # mp child: Coach makes the agents daemonic and daemonic processes may not have mp children)
        import subprocess
        import sys
        # the rendezvous is an AF_UNIX socket of our own with an accept timeout (a worker that dies before connecting -- bad
        # PYTHONPATH, exec failure -- must not leave the agent, and the Coach polling complete_count, waiting forever), wrapped
        # into a multiprocessing Connection with the same authkey handshake multiprocessing.connection.Listener.accept performs
        import socket
        import tempfile
        try:                                                       # (public names of multiprocessing.connection since Python 3.3; their
            from multiprocessing.connection import Connection, answer_challenge, deliver_challenge   # signatures are (connection, authkey) in 3.8-3.13)
        except ImportError as ex:
            raise RuntimeError('this Python\'s multiprocessing.connection lacks Connection / deliver_challenge / answer_challenge (%s): the '
                               'compat agent needs CPython 3.8-3.13' % ex) from ex
        key = os.urandom(16)
        # an AF_UNIX path is limited to ~107 bytes: a long TMPDIR falls back to /tmp
        sockdir = tempfile.mkdtemp(prefix='azg-agent-')
        address = os.path.join(sockdir, 'worker.sock')
        if len(address.encode()) > 100:
            try:
                os.rmdir(sockdir)
            except OSError:
                pass
            sockdir = tempfile.mkdtemp(prefix='azg-agent-', dir='/tmp')
            address = os.path.join(sockdir, 'worker.sock')
        srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        deadline = time.time() + float(os.environ.get('AZG_WORKER_START_TIMEOUT', '300'))
        try:
            try:
                srv.bind(address); srv.listen(1); srv.settimeout(1.0)
            except OSError as ex:
                raise RuntimeError('cannot bind the agent <-> worker rendezvous socket at %r (%s; AF_UNIX paths are limited to ~107 bytes: '
                                   'check TMPDIR)' % (address, ex)) from ex
            env = dict(os.environ, AZG_WORKER_KEY=key.hex(),
                       PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] + sys.path))
            self._worker = subprocess.Popen([sys.executable, '-m', 'alphazero_general_amd._engine_worker', address], env=env)
            while True:
                try:
                    peer, _ = srv.accept()
                    break
                except socket.timeout:
                    rc = self._worker.poll()
                    if rc is not None:
                        raise RuntimeError('device engine worker exited with code %s before connecting' % rc)
                    if time.time() > deadline:
                        self._worker.kill()
                        raise RuntimeError('device engine worker did not connect in time')
        finally:
            srv.close()
            try:
                if sockdir:
                    if os.path.exists(address):
                        os.unlink(address)
                    os.rmdir(sockdir)
            except OSError:
                pass
        peer.setblocking(True)
        self._conn = Connection(peer.detach())
        deliver_challenge(self._conn, key)
        answer_challenge(self._conn, key)
        self._conn.send(cfg)
        status, out = self._conn.recv()
        if status == 'error':
            raise RuntimeError('device engine worker failed to start:\n' + out)

    def _stop_worker(self):
        try:
            if self._conn is not None:
                self._call('close')
            self._worker.wait(10)
        except Exception:
            pass
        for s in getattr(self, '_shm', {}).values():
            try:
                s.close(); s.unlink()
            except Exception:
                pass

    # ---- the reference's loop (:79-101) ----
    def run(self):
        try:
            np.random.seed()
            torch.set_num_threads(1)                                  # forked child: never touch the parent's thread pools
            self._start_worker()
            a = self.args
            while not self.stop_event.is_set() and self.games_played.value < a.gamesPerIteration:
                self._check_pause()
                self.fast = np.random.random_sample() < a.probFastSim
                sims = a.numFastSims if self.fast else a.numMCTSSims if not self._is_warmup else a.numWarmupSims
                for i in range(sims):
                    if self.stop_event.is_set(): break
                    self.generateBatch()
                    if self.stop_event.is_set(): break
                    # (processBatch of simulation i and generateBatch of i + 1 are adjacent and touch the same trees: when another
                    #  simulation follows, the worker runs them as ONE launch -- azg_backup_select -- and one pipe round trip)
                    self._more_sims = i + 1 < sims
                    self.processBatch()
                    self._more_sims = False
                if self.stop_event.is_set(): break
                self.playMoves()
        except Exception:
            print(traceback.format_exc())
        finally:
            # unlike the reference (an agent exception leaves the parent polling forever) the agent
            # always reports completion
            with self.complete_count.get_lock():
                self.complete_count.value += 1
            if not self._is_arena:
                try:
                    self.output_queue.close()
                    self.output_queue.join_thread()
                except Exception:
                    pass
            self._stop_worker()

    def generateBatch(self):                                           # :103-135
        self._check_pause()
        if self._is_warmup:
            self._call('select_noobs')
            return
        if self._ahead is not None:                                    # (done together with the previous processBatch)
            rows, self._ahead = self._ahead[0], None
        else:
            rows = self._call('select')
        shape = tuple(self.game_cls.observation_size())
        if self._is_arena:
            row_of_slot, rpm = rows
            obs = torch.from_numpy(self._obs_h.reshape((self.batch_size,) + shape).copy())
            batch, off = [], 0
            for mi in range(self.game_cls.num_players()):               # list indexed by model, [] when a model has no rows
                n = int(rpm[mi])
                batch.append(obs[off:off + n] if n else [])
                off += n
            self.output_queue.put(batch)
            order = np.argsort(row_of_slot, kind='stable')               # row -> game, what the reference keeps (:132)
            self.batch_indices = list(order)
        else:
            # numpy, not Tensor.copy_: a large copy_ goes to torch's intra-op thread pool, whose threads do not exist in this
            # forked child (the parent created them) -- it would wait for them forever
            self.batch_tensor.numpy()[...] = self._obs_h.reshape((self.batch_size,) + shape)
        self.ready_queue.put(self.id)

    def processBatch(self):                                            # :137-151
        if self._is_warmup:                                            # :48-52,111-114 uniform policy / value
            self._pol_h[:] = 1.0 / self._pol_h.shape[1]
            self._val_h[:] = 1.0 / self._val_h.shape[1]
        else:
            self.batch_ready.wait()
            self.batch_ready.clear()
            self._pol_h[:] = self.policy_tensor.data.numpy()
            self._val_h[:] = self.value_tensor.data.numpy()
        if self._more_sims and not self._is_warmup:
            self._ahead = (self._call('backup_select'),)
        else:
            self._call('backup')

    def playMoves(self):                                               # :153-202
        self._check_pause()
        fin, states = self._call('advance_begin', (not self.fast) and (not self._is_arena))
        idx = np.flatnonzero(fin)
        counted = np.zeros(self.batch_size, np.int32)
        from .MCTS import decode_state
        template = self.game_cls()
        NV = self.game_cls.num_players() + 1
        winstates = {}
        for k, i in enumerate(idx):
            ws = np.array([(int(fin[i]) >> j) & 1 for j in range(NV)], dtype=np.uint8)
            winstates[int(i)] = ws
            self.result_queue.put((decode_state(template, *states[k]), ws, self.id))                 # :178
            lock = self.games_played.get_lock()
            lock.acquire()
            if self.games_played.value < self.args.gamesPerIteration:                                # :181-183
                self.games_played.value += 1
                counted[i] = 1
            lock.release()
        out = self._call('advance_commit', counted.tolist())
        if out is not None and not self._is_arena:
            o, p, z = out
            for j in range(o.shape[0]):                                                             # :184-196
                self._check_pause()
                self.output_queue.put((o[j], p[j], z[j]))
