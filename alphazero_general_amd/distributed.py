"""Multi-GPU layer: one process per GPU, game slots sharded by rank, ZERO communication during search.  The only
exchange step is once per iteration: all-gather the finished (state, pi, z) example shards (variable length per
rank) and sum the win/draw tallies -- what Coach.saveIterationSamples / processGameResults consume
(alphazero/Coach.py:364-398).  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.

Message sizes are tiny (connect4: 712 B per sample), so the exchange is latency-bound: one all_gather of the
counts (one int per rank) and one padded all_gather per tensor, instead of per-sample traffic.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
    # under torch.distributed.run (WORLD_SIZE set) the group is initialised even for one rank, so that a 1-GPU launch
    # exercises the same RCCL calls as an 8-GPU one
    if (world > 1 or 'WORLD_SIZE' in os.environ and 'MASTER_ADDR' in os.environ) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('AZG_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if os.environ.get('AZG_SINGLE_DEVICE'):              # test rig: every rank on GPU 0 (needs AZG_DIST_BACKEND=gloo;
            local_rank = 0                                   #  RCCL refuses two ranks on one device)
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif os.environ.get('AZG_SINGLE_DEVICE'):
        local_rank = 0
    return rank, local_rank, world


def shard_games(total_games, rank, world):
    """Per-rank game quota (quotas instead of a global atomic): floor(total / world), the remainder going to the
    lowest ranks, so that the quotas sum to exactly `total_games` -- the reference counts exactly gamesPerIteration games
    (SelfPlayAgent.pyx:179-183)."""
    total_games, world = int(total_games), int(world)
    return total_games // world + (1 if rank < total_games % world else 0)


def slot_base(rank, slots_per_rank):
    """Global id of a rank's slot 0: shards reproduce the trajectories a single engine of world*B slots would play."""
    return rank * int(slots_per_rank)


def all_gather_examples(obs, pi, z, group=None):
    """Variable-length all-gather of example shards.  Returns (obs, pi, z) holding every rank's samples, rank order,
    each rank's samples in its own output order."""
    if not dist.is_initialized():
        return obs, pi, z
    world = dist.get_world_size(group)
    if dist.get_backend(group) != 'nccl':                    # gloo (CPU tests, single-device rig): gather on the host
        obs, pi, z = obs.cpu(), pi.cpu(), z.cpu()
    n = torch.tensor([obs.shape[0]], dtype=torch.int64, device=obs.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(max(counts), 1)
    out = []
    for t in (obs, pi, z):
        pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        if dist.get_backend(group) == 'nccl':                # RCCL: ONE output buffer, no per-rank staging copies
            allb = torch.empty((world,) + tuple(pad.shape), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(allb, pad, group=group)
            bufs = list(allb.unbind(0))
        else:
            bufs = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(bufs, pad, group=group)
        out.append(torch.cat([b[:c] for b, c in zip(bufs, counts)]))
    return tuple(out)


def all_reduce_tallies(values, group=None):
    """Sum small integer tallies (wins per player, draws, game-length sum, games, expansions ...) over ranks."""
    t = torch.as_tensor(values, dtype=torch.int64)
    if not dist.is_initialized():
        return t
    dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
    t = t.to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu()


def broadcast_object(obj, src=0, group=None):
    """a small picklable object from rank `src` to every rank (the command of iteration.serve)"""
    if not dist.is_initialized():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def broadcast_state_dict(sd, meta=None, src=0, group=None, device=None):
    """The weights of the net an iteration plays with, from rank `src` to every replica: ONE broadcast of a flat float32 buffer
    (RCCL over xGMI: connect4 128ch x 8 is 9.6 MB) -- the only traffic of an iteration besides the example all-gather.  `sd`: the
    state_dict on `src` (None elsewhere); `meta` = state_dict_meta(sd), which the other ranks need to cut the buffer up again
    (sent with the command, iteration.lead).  Integer entries (BatchNorm's num_batches_tracked) travel in `meta`."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if meta is None:
        meta = state_dict_meta(sd)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {k: v for k, v in sd.items()}
    dev = device if device is not None else ('cuda' if dist.get_backend(group) == 'nccl' else 'cpu')
    n = sum(m[2] for m in meta['float'])
    if rank == src:
        flat = torch.cat([sd[k].detach().reshape(-1).to(dev, torch.float32) for k, _, _, _ in meta['float']]) if n else torch.zeros(0, device=dev)
    else:
        flat = torch.empty(n, dtype=torch.float32, device=dev)
    if n:
        dist.broadcast(flat, src=src, group=group)
    out, off = {}, 0
    for k, shape, numel, dt in meta['float']:
        out[k] = flat[off:off + numel].reshape(shape).to(getattr(torch, dt))
        off += numel
    for k, shape, vals, dt in meta['int']:
        out[k] = torch.tensor(vals, dtype=getattr(torch, dt)).reshape(shape)
    return {k: out[k] for k in meta['order']}


def state_dict_meta(sd):
    """{'order': keys, 'float': [(key, shape, numel, dtype)], 'int': [(key, shape, values, dtype)]} -- plain Python, picklable"""
    fl, it = [], []
    for k, v in sd.items():
        dt = str(v.dtype).replace('torch.', '')
        if v.is_floating_point():
            fl.append((k, tuple(v.shape), int(v.numel()), dt))
        else:
            it.append((k, tuple(v.shape), v.detach().reshape(-1).cpu().tolist(), dt))
    return {'order': list(sd.keys()), 'float': fl, 'int': it}


def describe_ranks(local_rank, group=None):
    """One record per rank, rank order -- which physical GPU every rank computes on (PCI bus id, name, CU count, memory), the
    host and process, and the collective library's version -- so that an N-GPU line explains its own placement; and the check
    that matters before any scaling number: two ranks on ONE device would still run (every rank then gets a share of it) and
    report a curve that looks like poor scaling.  Raises RuntimeError unless AZG_SINGLE_DEVICE (the one-GPU test rig) is set."""
    import socket
    rec = {'rank': dist.get_rank(group) if dist.is_initialized() else 0, 'local_rank': int(local_rank), 'host': socket.gethostname(), 'pid': os.getpid()}
    if torch.cuda.is_available():
        p = torch.cuda.get_device_properties(local_rank)
        # (a torch build whose device properties carry no pci_* / uuid fields gives no identifier: the shared-GPU check below is then
        #  skipped for that rank instead of refusing a correct one-process-per-GPU job)
        bus = ('%04x:%02x:%02x' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id & 0xFF, getattr(p, 'pci_device_id', 0) & 0xFF)
               if hasattr(p, 'pci_bus_id') else None)
        rec.update(device=p.name, pci_bus_id=bus, uuid=str(getattr(p, 'uuid', '')), compute_units=p.multi_processor_count,
                   memory_gb=round(p.total_memory / 2 ** 30, 1), visible_devices=os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('CUDA_VISIBLE_DEVICES')))
    try:
        rec['rccl_version'] = '.'.join(str(x) for x in torch.cuda.nccl.version())
    except Exception:                                       # noqa: BLE001 (a build without the nccl bindings)
        rec['rccl_version'] = None
    recs = [rec]
    if dist.is_initialized():
        recs = [None] * dist.get_world_size(group)
        dist.all_gather_object(recs, rec, group=group)
    seen = {}
    for r in recs:
        key = (r['host'], r.get('pci_bus_id'), r.get('uuid'))
        if not (r.get('pci_bus_id') or r.get('uuid')):              # no identifier for this rank's device: nothing to compare
            continue
        if key in seen and not os.environ.get('AZG_SINGLE_DEVICE'):
            raise RuntimeError('ranks %d and %d share one GPU (%s on %s): one process per GPU -- check LOCAL_RANK / HIP_VISIBLE_DEVICES'
                               % (seen[key], r['rank'], r['pci_bus_id'], r['host']))
        seen[key] = r['rank']
    return recs


def max_over_ranks(x, group=None):
    if not dist.is_initialized():
        return float(x)
    dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def barrier(group=None):
    if dist.is_initialized():
        dist.barrier(group=group)


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
