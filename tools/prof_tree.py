"""Steady-state tree kernels only (select/backup/advance at 2048 slots, uniform evaluator) -- target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_general_amd.engine import DeviceEngine

B, sims, moves = 2048, 100, 3
eng = DeviceEngine(0, B, seed=0, cpuct=4.0, fpu_reduction=0.4, add_root_noise=True, add_root_temp=True, example_capacity=400000, sims_hint=sims)
obs = torch.zeros((B, 42, 8), dtype=torch.float16, device=eng.device)
g = torch.Generator(device='cpu'); g.manual_seed(0)
pol = torch.rand((B, 7), generator=g) + 0.05; pol = (pol / pol.sum(1, keepdim=True)).to(eng.device)
val = torch.rand((B, 3), generator=g) + 0.05; val = (val / val.sum(1, keepdim=True)).to(eng.device)
for m in range(moves):
    for s in range(sims):
        eng.select(obs); eng.backup(pol, val)
    eng.advance(True)
print(eng.counters())
