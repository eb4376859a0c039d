"""Arena tower launch (BASELINE config 4 at its per-GPU size: 256 leaf rows split between two 128ch x 8 models, ONE multi-model launch,
azg_resnet_policy_value_multi_f16) by tile shape: boards per workgroup x pixel groups.  Needs the tuning build
(tools/build_timing.sh tuning; AZG_LIB_PATH=alphazero_general_amd/lib/libazg_tuning.so), which reads AZG_TOWER_BOARDS /
AZG_TOWER_PSPLIT.  Run without arguments: re-executes itself once per shape and prints one line each."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) == 1:
    lib = os.path.join(ROOT, 'alphazero_general_amd', 'lib', 'libazg_tuning.so')
    for boards, psplit in ((1, 1), (1, 2), (2, 1), (2, 2)):
        env = dict(os.environ, AZG_LIB_PATH=lib, AZG_TOWER_BOARDS=str(boards), AZG_TOWER_PSPLIT=str(psplit))
        subprocess.run([sys.executable, os.path.abspath(__file__), 'run'], env=env)
    sys.exit(0)

import torch
from alphazero_general_amd import nnet as N
from alphazero_general_amd.envs.connect4 import Game
nets = []
for sd in (0, 1):
    torch.manual_seed(sd)
    n = N.NNetWrapper(Game, N.CONNECT4_NET_ARGS, device='cuda:0'); n.refresh(); nets.append(n._hip)
out = []
for split in ((128, 128), (96, 160), (256, 0)):
    B = sum(split)
    x = torch.zeros((B, 42, 8), dtype=torch.float16, device='cuda:0'); x[:, :, :3] = (torch.rand(B, 42, 3, device='cuda:0') > 0.6).half()
    pol = torch.zeros((B, 7), device='cuda:0'); val = torch.zeros((B, 3), device='cuda:0')
    rpm = torch.tensor(split, dtype=torch.int32, device='cuda:0')
    for _ in range(5):
        N.HipResNet.forward_models(nets, x, pol, val, rpm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        N.HipResNet.forward_models(nets, x, pol, val, rpm)
    e1.record(); torch.cuda.synchronize()
    out.append('%d+%d rows: %.1f us' % (split[0], split[1], e0.elapsed_time(e1) * 10))
print('boards/workgroup %s, pixel groups %s: %s' % (os.environ.get('AZG_TOWER_BOARDS'), os.environ.get('AZG_TOWER_PSPLIT'), '; '.join(out)), flush=True)
