import sys, importlib, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import net_pins
from alphazero_general_amd import nnet as N
for key, env, argname, sizes in net_pins.CONFIGS:
    Game = importlib.import_module('alphazero_general_amd.envs.' + env).Game
    net = N.NNetWrapper(Game, getattr(N, argname), device='cuda:0', dtype=torch.float16)
    net.adopt(net_pins.det_fill(net.nnet.state_dict())); net.refresh()
    C, H, W = Game.observation_size()
    x = torch.from_numpy(net_pins.planes((64, C, H, W), 7)).to('cuda:0')
    p, v = net.process(x)
    p, v = p.float().cpu().numpy(), v.float().cpu().numpy()
    ent = -(p * np.log(p + 1e-30)).sum(1)
    print(key, 'policy max mean %.3f, entropy mean %.3f of %.3f, distinct rows %d, value mean %s' % (p.max(1).mean(), ent.mean(), np.log(p.shape[1]), len({r.tobytes() for r in p}), v.mean(0).round(3)))
