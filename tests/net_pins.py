"""Bit pins of the HIP network (test infrastructure).

The tree is compared bit for bit with an oracle that is fed by THIS network, and the network itself is only held to a tolerance against
fp32 PyTorch (fp16 by north_star) -- so a change of the network's arithmetic would move both sides of every tree test together.  These pins
close that gap: for deterministic weights (integer hashing only: no libm, no torch RNG) and deterministic 0/1 input planes, the CRC-32 of
`NNetWrapper.process`'s float32 policy and value rows is recorded per (game, network, batch size) in tests/golden/hip_net_pins.json -- the
batch sizes walk through the tower's tile shapes (1 / 2 / 4 boards per workgroup, pixel- and k-split), whose outputs are identical per board.
Generated ON an MI355X by tests/golden/make_hip_net_pins.py (MFMA arithmetic is a property of the hardware: the pins are gfx950's);
checked by tests/test_gpu_nnet.py::test_hip_network_bits_are_pinned (seen to hold on four different GPUs of the pool: unique ids 0x3efe7164df93d021,
0x41930ee287244ba2, 0xbe04e15bd387da5f, 0x7e7f52cea602aca9).  A deliberate change of the summation order regenerates them."""
import zlib

import numpy as np

CONFIGS = [                                                      # (key, env module, NNetWrapper args name, batch sizes)
    ('connect4_128x8', 'connect4', 'CONNECT4_NET_ARGS', (1, 5, 700, 1400, 2048)),
    ('connect4_32x4', 'connect4', 'DEFAULT_NET_ARGS', (3, 515)),
    ('brandubh_64x4', 'brandubh', 'BRANDUBH_NET_ARGS', (1, 7, 600, 1100, 2048)),
    ('trimok_32x4', 'trimok', 'DEFAULT_NET_ARGS', (2, 300, 1024)),
]


def det_fill(sd):
    """weights from the (sorted) key order and the element index by integer hashing: exact on every host"""
    import torch
    out = {}
    for i, k in enumerate(sorted(sd)):
        t = sd[k]
        if not t.dtype.is_floating_point:
            out[k] = t.clone()
            continue
        n = t.numel()
        j = np.arange(n, dtype=np.uint64)
        h = (j * np.uint64(2654435761) + np.uint64(i + 1) * np.uint64(40503) * np.uint64(65599)) & np.uint64(0xFFFFFFFF)
        h = ((h ^ (h >> np.uint64(15))) * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
        u = ((h >> np.uint64(8)) & np.uint64(0xFFFF)).astype(np.float64) / 65536.0 - 0.5          # (-0.5, 0.5), 16 bits: exact in fp32
        if k.endswith('running_var'):
            x = np.abs(u) * 1.5 + 0.5
        elif k.endswith('running_mean') or k.endswith('.bias'):
            x = u * 0.25
        elif 'bn' in k and k.endswith('.weight'):
            x = u * 0.5 + 1.0
        else:
            fan = max(int(t[0].numel()), 1) if t.dim() > 0 else 1
            x = u * 2.0 ** (2 - fan.bit_length() // 2)                                             # ~ 4 / sqrt(fan) as a power of two, from integer arithmetic only
        out[k] = torch.from_numpy(x.reshape(tuple(t.shape))).to(t.dtype)
    return out


def planes(shape, seed):
    return np.random.RandomState(seed).randint(0, 2, size=shape).astype(np.float32)


def compute():
    """{key: {str(B): [crc(policy), crc(value)]}} on cuda:0"""
    import importlib
    import torch
    from alphazero_general_amd import nnet as N
    out = {}
    for key, env, argname, sizes in CONFIGS:
        Game = importlib.import_module('alphazero_general_amd.envs.' + env).Game
        net = N.NNetWrapper(Game, getattr(N, argname), device='cuda:0', dtype=torch.float16)
        net.adopt(det_fill(net.nnet.state_dict()))
        net.refresh()
        assert net._hip is not None, 'the pins are the MFMA path\'s'
        C, H, W = Game.observation_size()
        rec = {}
        for B in sizes:
            x = torch.from_numpy(planes((B, C, H, W), 1000 + B)).to('cuda:0')
            p, v = net.process(x)
            p, v = p.float().cpu().numpy(), v.float().cpu().numpy()
            assert np.isfinite(p).all() and np.isfinite(v).all() and abs(float(p.sum(1).mean()) - 1.0) < 1e-3
            rec[str(B)] = [zlib.crc32(np.ascontiguousarray(p).tobytes()), zlib.crc32(np.ascontiguousarray(v).tobytes()),
                           # (a board's rows do not depend on the batch it is evaluated in: the first row again, alone)
                           zlib.crc32(np.ascontiguousarray(net.process(x[:1])[0].float().cpu().numpy()).tobytes()) ==
                           zlib.crc32(np.ascontiguousarray(p[:1]).tobytes())]
        out[key] = rec
    return out
