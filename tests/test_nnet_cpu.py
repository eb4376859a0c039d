"""The network architecture against the REFERENCE ResNet (alphazero/NNetArchitecture.py:69-120): golden outputs made by
tests/golden/make_goldens.py with deterministic weights; checkpoint keys and shapes interchange.  CPU, fp32."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def fill_deterministic(sd):
    out = {}
    for i, k in enumerate(sorted(sd)):
        t = sd[k]
        if t.dtype in (torch.int64, torch.int32):
            out[k] = t.clone()
            continue
        n = t.numel()
        x = torch.sin(torch.arange(n, dtype=torch.float64) * 0.37 + i * 1.7)
        if k.endswith('running_var'):
            x = x.abs() * 0.8 + 0.4
        elif k.endswith('running_mean') or k.endswith('.bias'):
            x = x * 0.1
        elif 'bn' in k and k.endswith('.weight'):
            x = x * 0.3 + 1.0
        else:
            x = x * (1.5 / max(t[0].numel(), 1) ** 0.5)
        out[k] = x.reshape(t.shape).to(t.dtype)
    return out


def _check(name, args):
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import FoldedResNet, NNetWrapper
    d = dict(np.load(os.path.join(G, 'c4_net.npz')))              # (NpzFile decompresses an array on EVERY d[key])
    w = NNetWrapper(Game, args, device='cpu', fast=False)
    sd = w.nnet.state_dict()
    assert sorted(sd.keys()) == list(d[name + '_keys'])                       # checkpoints interchange key for key
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == list(d[name + '_shapes'])
    w.nnet.load_state_dict(fill_deterministic(sd))
    x = torch.from_numpy(d['obs'])
    p, v = w.process(x)
    assert np.allclose(p.numpy(), d[name + '_policy'], atol=2e-6) and np.allclose(v.numpy(), d[name + '_value'], atol=2e-6)
    fp, fv = FoldedResNet(w.nnet).eval()(x)                                   # BN folding + collapsed linear chains
    assert np.allclose(fp.detach().numpy(), d[name + '_policy'], atol=2e-5) and np.allclose(fv.detach().numpy(), d[name + '_value'], atol=2e-5)
    pp, pv = w.predict(d['obs'][0])
    assert np.allclose(pp, d[name + '_policy'][0], atol=2e-6) and np.allclose(pv, d[name + '_value'][0], atol=2e-6)


def test_default_net_vs_reference():
    from alphazero_general_amd.nnet import DEFAULT_NET_ARGS
    _check('default', DEFAULT_NET_ARGS)


def test_connect4_train_net_vs_reference():
    from alphazero_general_amd.nnet import CONNECT4_NET_ARGS
    _check('c4train', CONNECT4_NET_ARGS)


def test_load_checkpoint_written_by_the_reference(tmp_path):
    """tests/golden/c4_ref_checkpoint.pth.tar was written by the reference's NNetWrapper.save_checkpoint (make_goldens.py
    c4_ckpt); c4_ckpt.npz holds the reference's own process() outputs.  Loading it here (no reference importable) must rebuild
    the saved architecture and reproduce those outputs; a save / load round trip through this wrapper must too."""
    import sys
    import torch
    assert not any(m == 'alphazero' or m.startswith('alphazero.') for m in sys.modules), 'the reference must not be importable here'
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import NNetWrapper
    d = dict(np.load(os.path.join(G, 'c4_ckpt.npz')))
    w = NNetWrapper(Game, device='cpu', backend='torch')
    saved = w.load_checkpoint(G, 'c4_ref_checkpoint.pth.tar')
    assert saved.num_channels == 8 and saved.depth == 2 and w.args.num_channels == 8 and saved.value_loss_weight == 1.5
    p, v = w.process(torch.from_numpy(d['obs']))
    assert np.abs(p.numpy() - d['policy']).max() < 1e-6 and np.abs(v.numpy() - d['value']).max() < 1e-6
    w.save_checkpoint(str(tmp_path), 'again.pth.tar')
    w2 = NNetWrapper(Game, device='cpu', backend='torch')
    w2.load_checkpoint(str(tmp_path), 'again.pth.tar')
    p2, v2 = w2.process(torch.from_numpy(d['obs']))
    assert torch.equal(p, p2) and torch.equal(v, v2)
    with pytest.raises(FileNotFoundError):
        w2.load_checkpoint(str(tmp_path), 'missing.pth.tar')


def test_load_checkpoint_does_not_execute_foreign_globals(tmp_path):
    """A checkpoint whose pickle names an arbitrary callable (here os.system via __reduce__) must not run it: the default loader
    resolves only an allow-list of globals and turns everything else into inert placeholders; the weights still load."""
    import pickle
    import torch
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import NNetWrapper, DEFAULT_NET_ARGS
    marker = tmp_path / 'pwned'

    class Evil:
        def __reduce__(self):
            import os as _os
            return (_os.system, ('touch %s' % marker,))

    w = NNetWrapper(Game, device='cpu', backend='torch')
    torch.save({'state_dict': w.nnet.state_dict(), 'args': dict(DEFAULT_NET_ARGS), 'opt_state': Evil()},
               str(tmp_path / 'evil.pth.tar'), pickle_protocol=pickle.HIGHEST_PROTOCOL)
    w2 = NNetWrapper(Game, device='cpu', backend='torch')
    w2.load_checkpoint(str(tmp_path), 'evil.pth.tar')
    assert not marker.exists()
    for k, v in w.nnet.state_dict().items():
        assert torch.equal(v, w2.nnet.state_dict()[k])
