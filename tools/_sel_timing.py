import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_general_amd.engine import DeviceEngine
B = 2048
eng = DeviceEngine(0, B, seed=0, cpuct=4.0, fpu_reduction=0.4, add_root_noise=True, add_root_temp=True, example_capacity=400000, sims_hint=100)
obs = torch.zeros((B, 42, 8), dtype=torch.float16, device=eng.device)
g = torch.Generator(device='cpu'); g.manual_seed(0)
pol = torch.rand((B, 7), generator=g) + 0.05; pol = (pol / pol.sum(1, keepdim=True)).to(eng.device)
val = torch.rand((B, 3), generator=g) + 0.05; val = (val / val.sum(1, keepdim=True)).to(eng.device)
for move in range(12):
    ts = []
    for s in range(100):
        e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e0.record(); eng.select(obs); e1.record(); eng.backup(pol, val); e2.record()
        ts.append((e0, e1, e2))
    eng.advance(True)
    torch.cuda.synchronize()
    sel = [a.elapsed_time(b) * 1e3 for a, b, c in ts]; bak = [b.elapsed_time(c) * 1e3 for a, b, c in ts]
    d = [eng.tree_info(i)['depth'] for i in range(0, B, 64)]
    print('move %2d select us: sim0 %.1f sim1 %.1f sim10 %.1f sim50 %.1f sim99 %.1f | backup sim50 %.1f | depth(last) mean %.1f max %d maxdepth %d' % (
        move, sel[0], sel[1], sel[10], sel[50], sel[99], bak[50], sum(d) / len(d), max(d), max(eng.tree_info(i)['max_depth'] for i in range(0, B, 64))))
