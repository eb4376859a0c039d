"""Device-side half of the compat SelfPlayAgent: a fresh interpreter that owns the DeviceEngine and serves the agent
process over a multiprocessing.connection (see SelfPlayAgent.py).  Started as
    python -m alphazero_general_amd._engine_worker <listener address>        (authkey in AZG_WORKER_KEY)
"""
import os
import sys
import traceback
from multiprocessing import shared_memory

import numpy as np


def engine_worker(conn, cfg):
    """Runs in a fresh (spawned) process: owns the DeviceEngine, serves commands from the agent process."""
    try:
        import torch as _t
        from alphazero_general_amd.engine import DeviceEngine
        B = cfg['B']
        shm = {k: shared_memory.SharedMemory(name=n) for k, n in cfg['shm'].items()}
        obs_h = np.ndarray((B, cfg['O']), np.float32, buffer=shm['obs'].buf)
        pol_h = np.ndarray((B, cfg['A']), np.float32, buffer=shm['pol'].buf)
        val_h = np.ndarray((B, cfg['NV']), np.float32, buffer=shm['val'].buf)
        eng = DeviceEngine(cfg['game'], B, arena=cfg['arena'], temp_table_override=cfg['temp_table'], **cfg['engine'])
        dev = eng.device
        obs = eng.new_obs(_t.float32)
        pol = _t.zeros((B, cfg['A']), dtype=_t.float32, device=dev)
        val = _t.zeros((B, cfg['NV']), dtype=_t.float32, device=dev)
        # the shared-memory buffers page-locked in place (hipHostRegister): the observation batch goes device -> shm and the policy /
        # value rows shm -> device by DMA, without a pageable staging copy each (the hops of every simulation in compat mode)
        obs_t, pol_t, val_t = _t.from_numpy(obs_h), _t.from_numpy(pol_h), _t.from_numpy(val_h)
        pinned = True
        for t_ in (obs_t, pol_t, val_t):
            try:
                pinned = pinned and int(_t.cuda.cudart().cudaHostRegister(t_.data_ptr(), t_.numel() * 4, 0)) == 0
            except Exception:                                    # noqa: BLE001
                pinned = False

        def obs_out():
            if pinned:
                obs_t.copy_(obs.reshape(B, -1), non_blocking=True)
                _t.cuda.current_stream().synchronize()
            else:
                obs_h[:] = obs.reshape(B, -1).cpu().numpy()

        def pv_in():
            pol.copy_(pol_t, non_blocking=pinned); val.copy_(val_t, non_blocking=pinned)
        n_ex = 0
        row_of_slot = None
        conn.send(('ready', None))
        while True:
            cmd, arg = conn.recv()
            if cmd == 'select':
                rows = None
                if cfg['arena']:
                    row_of_slot, rpm = eng.arena_rows(cfg['player_to_index'])
                eng.select(obs, row_of_slot)
                obs_out()
                if cfg['arena']:
                    rows = (row_of_slot.cpu().numpy().copy(), rpm.cpu().numpy().copy())
                conn.send(('ok', rows))
            elif cmd == 'select_noobs':
                eng.select(None)
                conn.send(('ok', None))
            elif cmd == 'backup':
                pv_in()
                if pinned:
                    _t.cuda.current_stream().synchronize()      # (the rows have left the shared buffers before the agent is answered)
                eng.backup(pol, val, row_of_slot if cfg['arena'] else None)
                conn.send(('ok', None))
            elif cmd == 'backup_select':                         # processBatch of this simulation + generateBatch of the next: one launch
                pv_in()
                # (arena: the movers -- hence the row <-> game map -- do not change until the move is played)
                eng.backup_select(pol, val, obs, row_of_slot if cfg['arena'] else None)
                obs_out()
                conn.send(('ok', (row_of_slot.cpu().numpy().copy(), rpm.cpu().numpy().copy()) if cfg['arena'] else None))
            elif cmd == 'advance_begin':
                fin = eng.advance_begin(record_history=arg)
                idx = np.flatnonzero(fin)
                states = [eng.get_states_full(int(i), 1)[0] for i in idx]
                conn.send(('ok', (fin, states)))
            elif cmd == 'advance_commit':
                eng.advance_commit(arg)
                c = eng.counters()
                new = c['num_examples'] - n_ex
                out = None
                if new > 0:
                    o, p, z = eng.examples(n_ex, new)
                    out = (o.cpu().numpy(), p.cpu().numpy(), z.cpu().numpy())
                # everything emitted so far has been handed over: rewind the engine's result / example cursors, so that an agent
                # can finish any number of games (the reference's queues are unbounded; games_cap is 2^30 here, so zeroing the
                # games counter with them changes nothing)
                eng.clear_outputs()
                n_ex = 0
                conn.send(('ok', out))
            elif cmd == 'close':
                conn.send(('ok', None))
                break
    except Exception:
        try:
            conn.send(('error', traceback.format_exc()))
        except Exception:
            pass




def main():
    from multiprocessing.connection import Client
    conn = Client(sys.argv[1], family='AF_UNIX', authkey=bytes.fromhex(os.environ['AZG_WORKER_KEY']))
    cfg = conn.recv()
    engine_worker(conn, cfg)


if __name__ == '__main__':
    main()
