"""k_select / k_backup time for brandubh at 512 slots (uniform evaluator), via the engine's HIP-event profile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_general_amd.engine import DeviceEngine
game = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, sims = (512, 200) if game == 1 else (2048, 100)
e = DeviceEngine(game, B, cpuct=1.25, fpu_reduction=0.2, seed=0, games_per_iteration=1 << 30, sims_hint=sims, add_root_noise=True, add_root_temp=True)
pol = torch.full((B, e.A), 1.0 / e.A, device=e.device); val = torch.full((B, e.NV), 1.0 / e.NV, device=e.device)
obs = torch.zeros((B, e.gi.obs_h * e.gi.obs_w, 8), dtype=torch.float16, device=e.device)
for mv in range(12):
    if mv == 8: e.profile(True)
    for s in range(sims):
        e.select(obs); e.backup(pol, val)
    e.advance(True)
p = e.profile_read()
print('game', game, 'select us %.2f backup us %.2f advance us %.1f' % (p['select_ms'] * 1e3 / p['select_n'], p['backup_ms'] * 1e3 / p['backup_n'], p['advance_ms'] * 1e3 / p['advance_n']))
