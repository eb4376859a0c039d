"""Numerics of the network path on the GPU: the folded PyTorch inference net and the hand-written MFMA tower
(csrc/azg_conv.h) against the plain fp32 PyTorch reference of the same architecture (NNetArchitecture.py:69-120).
Tolerance: probabilities within 3e-3 absolute (fp16 activations/weights, fp32 accumulation) -- the parity bar for the
floating-point network; the tree itself is checked bit-exactly with the SAME (p, v) fed to oracle and engine."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 3e-3        # absolute, on probabilities: the parity bar of the floating-point network (fp16 activations and weights, fp32 accumulation)


def check_probs(name, p, rp, v, rv, tol=TOL):
    """Assert the tolerance AND record the error actually achieved: max-abs and mean KL(reference || ours) for both heads, printed
    (pytest -s / -rA shows it) and appended to gpurun_out/nn_error.jsonl so that DESIGN.md can quote measured numbers."""
    import json
    import os
    import torch
    p, rp, v, rv = [t.detach().float().cpu() for t in (p, rp, v, rv)]
    kl = lambda r, q: float((r * (torch.log(r.clamp_min(1e-30)) - torch.log(q.clamp_min(1e-30)))).sum(1).mean())
    rec = {'test': name, 'boards': int(p.shape[0]), 'max_abs_policy': float((p - rp).abs().max()), 'max_abs_value': float((v - rv).abs().max()),
           'kl_policy': kl(rp, p), 'kl_value': kl(rv, v), 'tolerance': tol}
    print('NNERR ' + json.dumps(rec))
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'nn_error.jsonl'), 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass
    assert rec['max_abs_policy'] < tol and rec['max_abs_value'] < tol, rec
    return rec


def _randomize(net, torch, seed=0):
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.8 + 0.4)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) * 0.6 + 0.7)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)


def _boards(torch, B, seed=0):
    rng = np.random.RandomState(seed)
    from alphazero_general_amd.envs.connect4 import Game
    obs = []
    for b in range(B):
        g = Game()
        for _ in range(rng.randint(0, 30)):
            v = np.flatnonzero(g.valid_moves())
            if len(v) == 0 or g.win_state().any():
                break
            g.play_action(int(rng.choice(v)))
        obs.append(g.observation())
    return torch.from_numpy(np.array(obs, np.float32))


@pytest.mark.parametrize('backend', ['torch', 'hip', 'hip_tower_only'])
def test_inference_paths_vs_fp32_reference(backend):
    tower_only = backend == 'hip_tower_only'                          # tower launch + wide-head kernel instead of the fused heads
    backend = 'hip' if tower_only else backend
    import torch
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper
    torch.manual_seed(3)
    net = NNetWrapper(Game, CONNECT4_NET_ARGS, device='cuda:0', backend=backend)
    _randomize(net.nnet.cpu(), torch); net.nnet.to('cuda:0')
    x = _boards(torch, 1061 if backend == 'hip' else 16)            # 37: not a multiple of the 4-board workgroup tile
    with torch.no_grad():
        lp, lv = net.nnet(x.to('cuda:0'))
        rp, rv = torch.exp(lp).cpu(), torch.exp(lv).cpu()
    if tower_only:
        net.refresh(); net._hip.fused_head = False
    p, v = net.process(x)
    assert (net._hip is not None) == (backend == 'hip')
    assert p.shape == rp.shape and v.shape == rv.shape and p.dtype == torch.float32
    check_probs('c4_128x8_' + ('tower_only' if tower_only else backend), p, rp, v, rv)
    assert torch.allclose(p.sum(1).cpu(), torch.ones(p.shape[0]), atol=1e-4)


def test_tower_multi_tile_loop_and_determinism():
    """more tiles than resident workgroups (each workgroup loops over several tiles) + run-to-run bit determinism."""
    import torch
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper
    torch.manual_seed(5)
    net = NNetWrapper(Game, CONNECT4_NET_ARGS, device='cuda:0', backend='hip')
    _randomize(net.nnet.cpu(), torch, seed=2); net.nnet.to('cuda:0'); net.refresh()
    base = _boards(torch, 300, seed=4)
    x = base.repeat(8, 1, 1, 1)[:2101]                      # 526 tiles of 4 boards > 512 resident workgroups
    p, v = net.process(x)
    p, v = p.clone(), v.clone()
    p2, v2 = net.process(x)
    assert torch.equal(p, p2) and torch.equal(v, v2)
    assert torch.equal(p[:300], p[300:600]) and torch.equal(p[:1], p[2100:2101])     # same board -> same output in any tile
    with torch.no_grad():
        lp, lv = net.nnet(x[:300].to('cuda:0'))
    check_probs('c4_128x8_multi_tile', p[:300], torch.exp(lp), v[:300], torch.exp(lv))


def test_mfma_tower_vs_reference_golden():
    """the hand-written tower (fp16, MFMA) against outputs of the REFERENCE ResNet (fp32) on the committed fixture."""
    import os
    import torch
    from test_nnet_cpu import fill_deterministic, G
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper
    d = dict(np.load(os.path.join(G, 'c4_net.npz')))              # (NpzFile decompresses an array on EVERY d[key])
    w = NNetWrapper(Game, CONNECT4_NET_ARGS, device='cuda:0', backend='hip')
    w.nnet.load_state_dict(fill_deterministic(w.nnet.state_dict()))
    w.refresh()
    assert w._hip is not None
    p, v = w.process(torch.from_numpy(d['obs']))
    check_probs('c4_128x8_vs_reference_net_golden', p, torch.from_numpy(d['c4train_policy']), v, torch.from_numpy(d['c4train_value']))


def test_mfma_tower_brandubh_64ch_vs_fp32_reference():
    """the same tower kernel at 64 channels on the 7x7 board (brandubh net of envs/hnefatafl/train_brandubh.py:50-55),
    heads unfused (A + NV = 591)."""
    import torch
    from alphazero_general_amd.envs.brandubh import Game
    from alphazero_general_amd.nnet import BRANDUBH_NET_ARGS, NNetWrapper
    torch.manual_seed(9)
    net = NNetWrapper(Game, BRANDUBH_NET_ARGS, device='cuda:0', backend='hip')
    _randomize(net.nnet.cpu(), torch, seed=5); net.nnet.to('cuda:0'); net.refresh()
    assert net._hip is not None and net._hip.CH == 64 and not net._hip.fused_head
    rng = np.random.RandomState(1)
    obs = []
    for b in range(301):
        g = Game()
        for _ in range(rng.randint(0, 40)):
            if g.win_state().any():
                break
            g.play_action(int(rng.choice(np.flatnonzero(g.valid_moves()))))
        obs.append(g.observation())
    x = torch.from_numpy(np.array(obs, np.float32))
    with torch.no_grad():
        lp, lv = net.nnet(x.to('cuda:0'))
    p, v = net.process(x)
    assert p.shape == (301, 588) and v.shape == (301, 3)
    check_probs('brandubh_64x4', p, torch.exp(lp), v, torch.exp(lv))


@pytest.mark.parametrize('game', ['connect4', 'brandubh'])
def test_tower_tile_shapes_agree(game):
    """the tower picks 1, 2 or 4 boards per workgroup tile by batch size; a board's outputs must not depend on the tile it
    is evaluated in (same MFMA accumulation order per pixel): bit-identical across the shapes."""
    import torch
    if game == 'connect4':
        from alphazero_general_amd.envs.connect4 import Game
        from alphazero_general_amd.nnet import CONNECT4_NET_ARGS as NA, NNetWrapper
        base, sizes = _boards(torch, 200, seed=11), (200, 900, 1400)      # 1, 2, 4 boards per tile
    else:
        from alphazero_general_amd.envs.brandubh import Game
        from alphazero_general_amd.nnet import BRANDUBH_NET_ARGS as NA, NNetWrapper
        rng = np.random.RandomState(2)
        obs = []
        for b in range(200):
            g = Game()
            for _ in range(rng.randint(0, 30)):
                if g.win_state().any():
                    break
                g.play_action(int(rng.choice(np.flatnonzero(g.valid_moves()))))
            obs.append(g.observation())
        base, sizes = torch.from_numpy(np.array(obs, np.float32)), (200, 1100)   # 1, 2 boards per tile
    torch.manual_seed(13)
    net = NNetWrapper(Game, NA, device='cuda:0', backend='hip')
    _randomize(net.nnet.cpu(), torch, seed=6); net.nnet.to('cuda:0'); net.refresh()
    outs = []
    for n in sizes:
        x = base.repeat((n + 199) // 200, 1, 1, 1)[:n]
        p, v = net.process(x)
        outs.append((p[:200].clone(), v[:200].clone()))
    with torch.no_grad():
        lp, lv = net.nnet(base.to('cuda:0'))
    assert float((outs[0][0].cpu() - torch.exp(lp).cpu()).abs().max()) < 3e-3
    for p, v in outs[1:]:
        if game == 'connect4':                     # heads fused in the kernel: the whole evaluation is tile-independent
            assert torch.equal(p, outs[0][0]) and torch.equal(v, outs[0][1])
        else:                                      # heads GEMM is a library call whose k-split depends on the batch size
            assert float((p - outs[0][0]).abs().max()) < 1e-4 and float((v - outs[0][1]).abs().max()) < 1e-4
    if game == 'brandubh':
        # the tower itself (head features): the k-split 1-board tile (<= 512 boards: wave = cout group x k group, partial sums through
        # LDS), the 2-board tile in two pixel groups (<= 1024) and the unsplit 2-board tile all sum in k-half order -- bit-identical
        hip = net._hip
        feats = []
        for n in (200, 500, 900, 2100):
            x = hip.to_nhwc8(base.repeat((n + 199) // 200, 1, 1, 1)[:n].to('cuda:0'))
            feats.append(hip.forward_features_nhwc8(x, key=n)[:200].clone())
        for f in feats[1:]:
            assert torch.equal(f, feats[0])


def test_trimok_tower_tile_shapes_agree():
    """the 5x5 x 32-channel tower's tile shapes -- 2 boards in two pixel groups (<= 2048 boards), 5 boards unsplit (above); the one-board
    tile of the persistent search launch is compared with these in tests/test_gpu_fullsize.py -- must give a board bit-identical head
    features (same MFMA accumulation order per output everywhere)."""
    import torch
    from alphazero_general_amd.envs.trimok import Game
    from alphazero_general_amd.nnet import DEFAULT_NET_ARGS as NA, NNetWrapper
    rng = np.random.RandomState(4)
    obs = []
    for b in range(200):
        g = Game()
        for _ in range(rng.randint(0, 12)):
            if g.win_state().any():
                break
            g.play_action(int(rng.choice(np.flatnonzero(g.valid_moves()))))
        obs.append(g.observation())
    base = torch.from_numpy(np.array(obs, np.float32))
    torch.manual_seed(17)
    net = NNetWrapper(Game, NA, device='cuda:0', backend='hip')
    _randomize(net.nnet.cpu(), torch, seed=8); net.nnet.to('cuda:0'); net.refresh()
    hip = net._hip
    assert hip.fact_head
    feats = []
    for n in (200, 1500, 2100, 4300):
        x = hip.to_nhwc8(base.repeat((n + 199) // 200, 1, 1, 1)[:n].to('cuda:0'))
        feats.append(hip.forward_features_nhwc8(x, key=n)[:200].clone())
    assert float(feats[0].float().abs().max()) > 0
    for f in feats[1:]:
        assert torch.equal(f, feats[0])


def test_mfma_tower_trimok_32ch_vs_fp32_reference():
    """the tower at 32 channels (one wave per workgroup) on the 5x5 three-player board, 5 input planes, default net of
    Coach.py:108-116; heads through the wide-head kernel (A + NV = 29)."""
    import torch
    from alphazero_general_amd.envs.trimok import Game
    from alphazero_general_amd.nnet import DEFAULT_NET_ARGS, NNetWrapper
    torch.manual_seed(21)
    net = NNetWrapper(Game, DEFAULT_NET_ARGS, device='cuda:0', backend='auto')
    _randomize(net.nnet.cpu(), torch, seed=8); net.nnet.to('cuda:0'); net.refresh()
    assert net._hip is not None and net._hip.CH == 32 and net._hip.wide_head
    rng = np.random.RandomState(4)
    obs = []
    for b in range(2500):                                   # 2500 boards: both tile shapes (2 and 5 boards per workgroup)
        g = Game()
        for _ in range(rng.randint(0, 20)):
            if g.win_state().any():
                break
            g.play_action(int(rng.choice(np.flatnonzero(g.valid_moves()))))
        obs.append(g.observation())
    x = torch.from_numpy(np.array(obs, np.float32))
    with torch.no_grad():
        lp, lv = net.nnet(x.to('cuda:0'))
    for n in (2500, 301):
        p, v = net.process(x[:n])
        assert p.shape == (n, 25) and v.shape == (n, 4)
        check_probs('trimok_32x4_%d' % n, p, torch.exp(lp[:n]), v, torch.exp(lv[:n]))


def test_mfma_tower_connect4_default_net_32ch():
    """BASELINE config 1's network (Coach.py:108-116 defaults, 32 channels x 4 blocks) on the connect4 board: backend='auto' must
    pick the MFMA tower (no silent MIOpen fallback on a BASELINE shape), both tile shapes, against the fp32 reference."""
    import torch
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import DEFAULT_NET_ARGS, NNetWrapper
    torch.manual_seed(31)
    net = NNetWrapper(Game, DEFAULT_NET_ARGS, device='cuda:0', backend='auto')
    _randomize(net.nnet.cpu(), torch, seed=12); net.nnet.to('cuda:0'); net.refresh()
    assert net._hip is not None and net._hip.CH == 32 and net._hip.wide_head
    x = _boards(torch, 1500, seed=21)
    with torch.no_grad():
        lp, lv = net.nnet(x.to('cuda:0'))
    outs = []
    for n in (1500, 32):                                    # 4 and 2 boards per workgroup tile
        p, v = net.process(x[:n])
        check_probs('c4_32x4_%d' % n, p, torch.exp(lp[:n]), v, torch.exp(lv[:n]))
        outs.append((p[:32].clone(), v[:32].clone()))
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 1e-4


@pytest.mark.parametrize('game,base,depth,boards', [('connect4', 'CONNECT4_NET_ARGS', 1, 200), ('connect4', 'CONNECT4_NET_ARGS', 3, 64),
                                                    ('brandubh', 'BRANDUBH_NET_ARGS', 1, 300), ('brandubh', 'BRANDUBH_NET_ARGS', 6, 300)])
def test_one_board_tiles_at_other_depths(game, base, depth, boards):
    """one-board tiles keep every layer's biases and affines in an LDS area the LAUNCH sizes from the network's depth
    (csrc/azg_conv.h tower_param_bytes): depths other than the BASELINE networks', small batches (one board per workgroup),
    against the fp32 reference."""
    import importlib
    import torch
    from alphazero_general_amd import nnet as N
    Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
    from alphazero_general_amd.utils import dotdict
    args = dotdict(dict(getattr(N, base))); args.depth = depth
    torch.manual_seed(100 + depth)
    net = N.NNetWrapper(Game, args, device='cuda:0', backend='hip')
    _randomize(net.nnet.cpu(), torch, seed=depth); net.nnet.to('cuda:0'); net.refresh()
    assert net._hip is not None
    rng = np.random.RandomState(depth)
    obs = []
    for b in range(boards):
        g = Game()
        for _ in range(rng.randint(0, 25)):
            if g.win_state().any():
                break
            g.play_action(int(rng.choice(np.flatnonzero(g.valid_moves()))))
        obs.append(g.observation())
    x = torch.from_numpy(np.array(obs, np.float32))
    with torch.no_grad():
        lp, lv = net.nnet(x.to('cuda:0'))
    p, v = net.process(x)
    check_probs('%s_depth%d_%d' % (game, depth, boards), p, torch.exp(lp), v, torch.exp(lv))


def test_one_board_tile_refuses_a_tower_that_does_not_fit_lds():
    """(12 * nblocks + 8) * C bytes of parameters beside the image: a 128-channel tower of 110 blocks cannot be a one-board tile --
    the launch must say so (AZG_E_INVALID_ARG), not overrun LDS."""
    import ctypes as C
    import torch
    from alphazero_general_amd import _abi
    L = _abi.lib()
    nb, ch, B = 110, 128, 8
    dev = 'cuda:0'
    need = int(L.azg_tower_weights_size(ch, nb))
    w = torch.zeros(need, dtype=torch.float16, device=dev)
    bias = torch.zeros((2 * nb + 1) * ch, dtype=torch.float32, device=dev)
    sc = torch.ones(nb * ch, dtype=torch.float32, device=dev); sh = torch.zeros(nb * ch, dtype=torch.float32, device=dev)
    x = torch.zeros((B, 42, 8), dtype=torch.float16, device=dev)
    y = torch.zeros((B, 42, ch), dtype=torch.float16, device=dev)
    rc = L.azg_resnet_tower_f16(C.c_void_p(torch.cuda.current_stream().cuda_stream), 0, C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()),
                                C.c_void_p(sc.data_ptr()), C.c_void_p(sh.data_ptr()), C.c_void_p(y.data_ptr()), B, nb, ch)
    torch.cuda.synchronize()
    assert rc == -1, (rc, L.azg_last_error())                  # AZG_E_INVALID_ARG


def test_shared_batch_tensors_are_page_locked_for_their_lifetime():
    """NNetWrapper.process page-locks a caller-owned SHARED CPU batch tensor in place (nnet.pin_shared: what Coach's input tensors are,
    Coach.py:294-300) so that the H2D copy is a DMA instead of a staged pageable copy; the registration follows the tensor object -- it
    is dropped when the tensor is collected -- and the evaluation equals the one of a private copy bit for bit.  Plain CPU tensors are
    never registered."""
    import gc
    import torch
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.envs.connect4 import Game
    torch.manual_seed(3)
    net = N.NNetWrapper(Game, N.CONNECT4_NET_ARGS, device='cuda:0', dtype=torch.float16)
    g = torch.Generator().manual_seed(5)
    base = (torch.rand((256, 4, 6, 7), generator=g) > 0.5).float()
    before = set(N._PINNED)                                      # (registrations other tests' still-living tensors hold)
    shared = base.clone().share_memory_()
    key = shared.untyped_storage().data_ptr()
    p0, v0 = net.process(base)                                   # private pageable tensor: the staged path
    assert base.untyped_storage().data_ptr() not in N._PINNED
    for _ in range(3):                                           # the same shared tensor, call after call (one registration)
        p1, v1 = net.process(shared)
        assert torch.equal(p0, p1) and torch.equal(v0, v1)
    assert N._PINNED[key][1] == 'registered' and shared.is_pinned() and N._PINNED[key][3] == 3     # one registration served the three calls
    shared[0, 0, 0, 0] = 1 - shared[0, 0, 0, 0]                  # the caller rewrites the batch in place between calls (SelfPlayAgent.pyx:116-123)
    p2, _ = net.process(shared)
    q2, _ = net.process(shared.clone())
    assert torch.equal(p2, q2)
    del shared, p1, v1, p2
    gc.collect()
    assert key not in N._PINNED                                  # unregistered before the memory went away
    for _ in range(3):                                           # new shared tensors keep working (fresh registrations, no leak of old ones)
        t = base.clone().share_memory_()
        p3, _ = net.process(t)
        assert torch.equal(p3, p0)
        del t
    gc.collect()
    assert all(v[1] != 'registered' for k_, v in N._PINNED.items() if k_ not in before)
    # a caller that wraps the same memory in a NEW tensor object for every call (each registration serves one call) is left on the pageable
    # path after a few of them; a long-lived tensor at an address the allocator hands out again is registered afresh
    keep = base.clone().share_memory_()
    k2 = keep.untyped_storage().data_ptr()
    N._PIN_COUNT.pop(k2, None)
    import warnings
    N._PIN_STATS['warned'] = 0
    N.pin_stats(reset=True)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        for i in range(8):
            view = keep[:]                                        # same storage, new tensor object
            p4, _ = net.process(view)
            assert torch.equal(p4, p0)
            del view
            gc.collect()
    assert N._PIN_COUNT.get(k2, 0) == 4 and N._PINNED.get(k2, [0, 'no'])[1] == 'no'
    # ... and the fall-back is not silent: counted, and announced once
    st = N.pin_stats()
    assert st['registrations'] == 4 and st['dma'] == 4 and st['staged'] == 4, st
    assert sum('pageable (staged) path' in str(w.message) for w in caught) == 1, [str(w.message) for w in caught]
    del keep
    gc.collect()


def test_hip_network_bits_are_pinned():
    """The HIP network's OUTPUT BITS on deterministic weights and inputs (tests/net_pins.py; pins generated on an MI355X by
    tests/golden/make_hip_net_pins.py): the tree tests compare the engine with an oracle fed by this same network, and the network is held
    to fp32 PyTorch only within a tolerance -- a change of its arithmetic (summation order, a fused epilogue, a different tile's rounding)
    would move both sides of every tree test together.  This is the test that sees it: every BASELINE network on its game, batch sizes that
    walk through the tile shapes."""
    import json
    import net_pins
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'hip_net_pins.json')))['pins']
    got = net_pins.compute()
    assert set(got) == set(want)
    for key in sorted(want):
        for B in sorted(want[key], key=int):
            assert list(got[key][B]) == list(want[key][B]), (key, B, got[key][B], want[key][B])
