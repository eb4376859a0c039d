"""Time of the sparse-heads launch by itself (azg_leaf_heads_sparse_f16: one wavefront per slot computes the logits of its leaf's valid
actions) at several slot counts -- brandubh, leaves a few plies deep.  usage: ubench_sparse_heads.py [brandubh|trimok]"""
import importlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_general_amd import nnet as N
from alphazero_general_amd.engine import DeviceEngine
game = sys.argv[1] if len(sys.argv) > 1 else 'brandubh'
Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
torch.manual_seed(0)
net = N.NNetWrapper(Game, N.BRANDUBH_NET_ARGS if game == 'brandubh' else N.DEFAULT_NET_ARGS, device='cuda:0'); net.refresh()
hip = net._hip
hw = Game.observation_size()[1] * Game.observation_size()[2]
for B in (32, 256, 512, 2048):
    e = DeviceEngine(Game.AZG_GAME_ID, B, seed=1, sims_hint=8)
    obs = torch.zeros((B, hw, 8), dtype=torch.float16, device=e.device)
    e.select(obs)
    for s in range(5):
        e.backup_select_features(hip.forward_features_nhwc8(obs), hip.head_rows, hip.head2_b, obs, select=True)
    feat = hip.forward_features_nhwc8(obs)
    out = e.leaf_heads_sparse(feat, hip.head_rows, hip.head2_b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        e.leaf_heads_sparse(feat, hip.head_rows, hip.head2_b, out=out)
    e1.record(); torch.cuda.synchronize()
    k = sum(len(e.root_children(i)['a']) for i in range(min(B, 16))) / min(B, 16)
    print('%s: %5d slots  %.2f us per launch (root children ~%.0f)' % (game, B, e0.elapsed_time(e1) * 5, k), flush=True)
    e.close()
