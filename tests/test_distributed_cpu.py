"""The N > 1 path on CPU: two gloo ranks shard game slots, exchange variable-length example shards with the
all-gather used by bench.py / the native driver, and sum tallies.  (The engine itself needs a GPU; here the shards
are produced by the CPU oracle so that the exchange is tested on real self-play data, including the sharding
invariance property: rank r with slot_base = r*B reproduces the second half of a 2B-slot run.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _play(B, slot_base, seed, games):
    import oracle_lib as ol
    ag = ol.OAgent(0, B, sims=8, games_per_iteration=games, seed=seed, slot_base=slot_base)
    step = 0
    while ag.games_played < games:
        ns = ag.begin_round()
        for _ in range(ns):
            obs, rg, rm = ag.generate_batch()
            pol = np.zeros((B, 7), np.float32); val = np.zeros((B, 3), np.float32)
            for i in range(B):
                pol[i], val[i] = ol.fake_eval(seed, slot_base + i, step, 7, 3)
            ag.process_batch(pol, val); step += 1
        ag.play_moves()
    return ag


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from alphazero_general_amd import distributed as D
    r, lr, w = D.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    B = 6
    games = D.shard_games(9, rank, world) + 2 * rank        # unequal quotas (5, 6) -> unequal shard lengths
    ag = _play(B, D.slot_base(rank, B), 5, games)
    obs, pi, z = [torch.from_numpy(x) for x in ag.samples()]
    gobs, gpi, gz = D.all_gather_examples(obs, pi, z)
    tall = D.all_reduce_tallies([ag.games_played, obs.shape[0]])
    tmax = D.max_over_ranks(float(rank + 1))
    D.barrier()
    q.put((rank, obs.shape[0], gobs.numpy(), gpi.numpy(), gz.numpy(), tall.numpy(), tmax, obs.numpy(), pi.numpy()))
    dist.destroy_process_group()


def test_gloo_allgather_examples_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n0, n1 = res[0][1], res[1][1]
    assert n0 != n1 and n0 > 0 and n1 > 0
    for r in res:
        assert r[2].shape[0] == n0 + n1
        assert (r[2][:n0] == res[0][7]).all() and (r[2][n0:] == res[1][7]).all()       # rank order, own order kept
        assert (r[3][:n0] == res[0][8]).all() and (r[3][n0:] == res[1][8]).all()
        assert r[4].shape == (n0 + n1, 3)
        assert r[5][1] == n0 + n1 and r[6] == 2.0
    assert (res[0][2] == res[1][2]).all()


def _worker_unequal(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from alphazero_general_amd import distributed as D
    D.init_from_env(backend='gloo')
    n = [0, 1, 0, 977][rank] if world == 4 else 0
    g = torch.Generator().manual_seed(100 + rank)
    obs, pi, z = torch.rand((n, 5, 7, 7), generator=g), torch.rand((n, 588), generator=g), torch.rand((n, 3), generator=g)
    outs = []
    for rnd in range(2):                                     # twice: the second iteration's exchange reuses the group
        gobs, gpi, gz = D.all_gather_examples(obs, pi, z)
        outs.append((gobs.numpy(), gpi.numpy(), gz.numpy()))
    tall = D.all_reduce_tallies([n, rank])
    recs = D.describe_ranks(rank)
    D.barrier()
    q.put((rank, obs.numpy(), outs, tall.numpy(), [r['rank'] for r in recs]))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [4, 3])
def test_gloo_allgather_with_empty_and_lopsided_shards(world):
    """the exchange step after a SHORT iteration: ranks that finished no game at all (n_i = 0, first and in the middle), one with a single
    sample, one holding nearly everything -- and every rank empty (world 3: the padded gather of zero rows).  Rank order and each
    rank's own order are kept, empty ranks contribute nothing, the tallies add up, describe_ranks returns one record per rank."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_unequal, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.concatenate([r[1] for r in res])
    total = want.shape[0]
    assert total == (978 if world == 4 else 0)
    for r in res:
        for gobs, gpi, gz in r[2]:
            assert gobs.shape == (total, 5, 7, 7) and gpi.shape == (total, 588) and gz.shape == (total, 3)
            assert (gobs == want).all()
        assert r[3][0] == total and r[3][1] == sum(range(world)) and r[4] == list(range(world))


def test_sharding_invariance_oracle():
    """slot_base makes shards reproduce the big run: slots [6,12) of a 12-slot agent == a 6-slot agent at slot_base 6."""
    import oracle_lib as ol
    big = ol.OAgent(0, 12, sims=6, games_per_iteration=1 << 30, seed=3)
    small = ol.OAgent(0, 6, sims=6, games_per_iteration=1 << 30, seed=3, slot_base=6)
    for rnd in range(5):
        big.begin_round(); small.begin_round()
        for s in range(6):
            big.generate_batch(); small.generate_batch()
            pol = np.zeros((12, 7), np.float32); val = np.zeros((12, 3), np.float32)
            for i in range(12):
                pol[i], val[i] = ol.fake_eval(3, i, rnd * 6 + s, 7, 3)
            big.process_batch(pol, val); small.process_batch(pol[6:].copy(), val[6:].copy())
        big.play_moves(); small.play_moves()
        assert (big.last_actions()[6:] == small.last_actions()).all()


def test_game_quotas_sum_exactly():
    """per-rank / per-lane quotas sum to exactly gamesPerIteration (the reference counts exactly that many games,
    SelfPlayAgent.pyx:179-183), differ by at most one, and the remainder goes to the lowest ranks."""
    from alphazero_general_amd import distributed as D
    for total in (0, 1, 7, 8, 9, 100, 4096, 4099):
        for world in (1, 2, 3, 4, 8):
            q = [D.shard_games(total, r, world) for r in range(world)]
            assert sum(q) == total and max(q) - min(q) <= 1 and q == sorted(q, reverse=True)


def test_oracle_pool_equals_one_agent():
    """oracle/azg_pool_ref.c (bench.py's all-cores CPU baseline): 4 agents x 6 games on 4 threads, stepped in lock step with one
    leaf batch, play exactly what one agent of 24 games plays (global slot ids key the tape); the free-running tree-only loop
    makes progress on every thread."""
    import oracle_lib as ol
    kw = dict(sims=6, games_per_iteration=1 << 30, seed=3)
    big, pool = ol.OAgent(0, 24, **kw), ol.OPool(0, 4, 6, **kw)
    for rnd in range(5):
        big.begin_round(); pool.begin_round()
        for s in range(6):
            ob, _, _ = big.generate_batch()
            assert (ob == pool.generate()).all()
            pol = np.zeros((24, 7), np.float32); val = np.zeros((24, 3), np.float32)
            for i in range(24):
                pol[i], val[i] = ol.fake_eval(3, i, rnd * 6 + s, 7, 3)
            big.process_batch(pol.copy(), val.copy()); pool.process(pol.copy(), val.copy())
        big.play_moves(); pool.play()
        assert (np.concatenate([pool.agent_last_actions(i) for i in range(4)]) == big.last_actions()).all()
    assert pool.expansions == big.expansions and pool.sims_done == big.sims_done == 24 * 6 * 5
    free = ol.OPool(0, 3, 4, sims=5, games_per_iteration=1 << 30, seed=9)
    dt = free.run_tree_only(0.3)
    assert dt >= 0.3 and free.sims_done >= 3 * 4 * 5 and free.expansions > 0
