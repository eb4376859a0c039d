"""The NN boundary of the hot path: a policy/value ResNet with the reference's architecture and checkpoint keys
(alphazero/NNetArchitecture.py:69-120) and an NNetWrapper-shaped object exposing the two calls the search uses
(alphazero/NNetWrapper.py:207-232): predict(board) -> (p, v) numpy and process(batch) -> (P, V) device tensors of
probabilities.  The network stays PyTorch-ROCm (MIOpen / hipBLASLt -> MFMA); what is new is the inference path:
eval-mode BatchNorm folded into the convolutions, channels-last fp16, softmax epilogue in fp32, and optional
hipGraph capture at a fixed batch size.  Training is out of scope.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .utils import dotdict

DEFAULT_NET_ARGS = dotdict(num_channels=32, depth=4, value_head_channels=16, policy_head_channels=16,
                           value_dense_layers=[512, 64], policy_dense_layers=[512, 256])        # Coach.py:108-116
CONNECT4_NET_ARGS = dotdict(num_channels=128, depth=8, value_head_channels=32, policy_head_channels=32,
                            value_dense_layers=[1024, 256], policy_dense_layers=[1024])           # envs/connect4/train.py:45-50
BRANDUBH_NET_ARGS = dotdict(num_channels=64, depth=4, value_head_channels=16, policy_head_channels=16,
                            value_dense_layers=[1024, 128], policy_dense_layers=[1024])           # envs/hnefatafl/train_brandubh.py:50-55


def _mlp(sizes):
    """Linear chain with Identity between layers (the reference passes activation=nn.Identity, :88-93,97-102);
    module indices 0,2,4.. keep the reference's Sequential numbering so state_dicts interchange."""
    layers = []
    for i in range(len(sizes) - 1):
        layers += [nn.Linear(sizes[i], sizes[i + 1]), nn.Identity()]
    return nn.Sequential(*layers)


class ResidualBlock(nn.Module):
    """Pre-activation block: bn1-relu-conv1-bn2-relu-conv2 + x (NNetArchitecture.py:36-66, stride 1)."""

    def __init__(self, ch):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(ch)
        self.conv1 = nn.Conv2d(ch, ch, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(ch)
        self.conv2 = nn.Conv2d(ch, ch, 3, padding=1, bias=False)

    def forward(self, x):
        out = self.conv1(F.relu(self.bn1(x)))
        out = self.conv2(F.relu(self.bn2(out)))
        return out + x


class ResNet(nn.Module):
    def __init__(self, obs_size, action_size, value_size, args):
        super().__init__()
        self.channels, self.board_x, self.board_y = obs_size
        self.action_size = action_size
        ch = args.num_channels
        self.conv1 = nn.Conv2d(self.channels, ch, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(ch)
        self.resnet = nn.Sequential(*[ResidualBlock(ch) for _ in range(args.depth)])
        self.v_conv = nn.Conv2d(ch, args.value_head_channels, 1, bias=False)
        self.v_bn = nn.BatchNorm2d(args.value_head_channels)
        self.v_fc = _mlp([self.board_x * self.board_y * args.value_head_channels] + list(args.value_dense_layers) + [value_size])
        self.pi_conv = nn.Conv2d(ch, args.policy_head_channels, 1, bias=False)
        self.pi_bn = nn.BatchNorm2d(args.policy_head_channels)
        self.pi_fc = _mlp([self.board_x * self.board_y * args.policy_head_channels] + list(args.policy_dense_layers) + [action_size])

    def trunk(self, s):
        s = s.view(-1, self.channels, self.board_x, self.board_y)
        return self.resnet(F.relu(self.bn1(self.conv1(s))))

    def forward(self, s):
        s = self.trunk(s)
        v = self.v_fc(torch.flatten(self.v_bn(self.v_conv(s)), 1))
        pi = self.pi_fc(torch.flatten(self.pi_bn(self.pi_conv(s)), 1))
        return F.log_softmax(pi, dim=1), F.log_softmax(v, dim=1)


# ------------------------------------------------------------------------------------------------ inference path
def _fold(conv_w, bn):
    """conv followed by eval-mode BN -> (weight, bias)."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return conv_w * scale.view(-1, 1, 1, 1), bn.bias - bn.running_mean * scale


class FoldedResNet(nn.Module):
    """Eval-only restatement of ResNet.forward with every conv->BN pair folded; the pre-activation BN at the head of
    a block cannot fold (the residual stream also bypasses it) and stays as a fused affine+relu."""

    def __init__(self, net: ResNet):
        super().__init__()
        with torch.no_grad():
            self.shape = (net.channels, net.board_x, net.board_y)
            w, b = _fold(net.conv1.weight, net.bn1)
            self.stem_w, self.stem_b = nn.Parameter(w.clone()), nn.Parameter(b.clone())
            self.pre_scale, self.pre_shift = nn.ParameterList(), nn.ParameterList()
            self.w1, self.b1, self.w2 = nn.ParameterList(), nn.ParameterList(), nn.ParameterList()
            for blk in net.resnet:
                sc = blk.bn1.weight / torch.sqrt(blk.bn1.running_var + blk.bn1.eps)
                self.pre_scale.append(nn.Parameter(sc.view(1, -1, 1, 1).clone()))
                self.pre_shift.append(nn.Parameter((blk.bn1.bias - blk.bn1.running_mean * sc).view(1, -1, 1, 1).clone()))
                w, b = _fold(blk.conv1.weight, blk.bn2)
                self.w1.append(nn.Parameter(w.clone())); self.b1.append(nn.Parameter(b.clone()))
                self.w2.append(nn.Parameter(blk.conv2.weight.clone()))
            w, b = _fold(net.v_conv.weight, net.v_bn)
            wp, bp = _fold(net.pi_conv.weight, net.pi_bn)
            self.head_w = nn.Parameter(torch.cat([w, wp]).clone()); self.head_b = nn.Parameter(torch.cat([b, bp]).clone())
            self.vc = w.shape[0]
            self.v_fc, self.pi_fc = self._chain(net.v_fc), self._chain(net.pi_fc)
        for p in self.parameters():
            p.requires_grad_(False)

    @staticmethod
    def _chain(seq):
        """A chain of Linear layers with Identity in between is one affine map: collapse it."""
        W, b = None, None
        for m in seq:
            if isinstance(m, nn.Linear):
                if W is None:
                    W, b = m.weight.clone(), m.bias.clone()
                else:
                    W, b = m.weight @ W, m.weight @ b + m.bias
        lin = nn.Linear(W.shape[1], W.shape[0])
        lin.weight.copy_(W); lin.bias.copy_(b)
        return lin

    def forward(self, s):
        s = F.relu(F.conv2d(s, self.stem_w, self.stem_b, padding=1))
        for i in range(len(self.w1)):
            t = F.relu(s * self.pre_scale[i] + self.pre_shift[i])
            t = F.relu(F.conv2d(t, self.w1[i], self.b1[i], padding=1))
            s = F.conv2d(t, self.w2[i], None, padding=1) + s
        h = F.conv2d(s, self.head_w, self.head_b)
        v = self.v_fc(torch.flatten(h[:, :self.vc], 1))
        pi = self.pi_fc(torch.flatten(h[:, self.vc:], 1))
        return F.softmax(pi.float(), dim=1), F.softmax(v.float(), dim=1)     # == exp(log_softmax) (NNetWrapper.py:231)


# ------------------------------------------------------------------------------------- hand-written MFMA tower
def pack_conv_weight(w, ks):
    """[128, Cin, 3, 3] float -> fp16 MFMA A-fragment order [9 taps][ks][8 cout-subtiles][64 lanes][8] that
    csrc/azg_conv.h reads: lane = g*16 + i holds W[cout = ms*16 + i, cin = ks*32 + g*8 + j, tap]."""
    cout, cin = w.shape[0], w.shape[1]
    assert cout % 32 == 0 and cin <= ks * 32
    wp = torch.zeros((cout, ks * 32, 3, 3), dtype=torch.float32, device=w.device)
    wp[:, :cin] = w.float()
    t = wp.permute(2, 3, 0, 1).reshape(9, cout // 16, 16, ks, 4, 8)   # [tap, ms, i, ks, g, j]
    return t.permute(0, 3, 1, 4, 2, 5).contiguous().reshape(-1).to(torch.float16)


def pack_stem_weight(w):
    """Stem [C, Cin <= 8, 3, 3] float -> fp16 A fragments [3 k-steps][C/16 cout-subtiles][64 lanes][8]: the stem's input is one
    8-channel chunk per pixel, so a k-step of 32 holds FOUR taps -- lane = g*16 + i holds W[cout = ms*16 + i, cin = j,
    tap = 4*ks + g] (zero for tap > 8): csrc/azg_conv.h conv_stem."""
    cout, cin = w.shape[0], w.shape[1]
    assert cout % 32 == 0 and cin <= 8
    wk = torch.zeros((cout, 12, 8), dtype=torch.float32, device=w.device)                 # [cout, tap slot, channel]
    wk[:, :9, :cin] = w.float().reshape(cout, cin, 9).permute(0, 2, 1)
    t = wk.reshape(cout // 16, 16, 3, 4, 8)                                               # [ms, i, ks, g, j]
    return t.permute(2, 0, 3, 1, 4).contiguous().reshape(-1).to(torch.float16)


def _reference_pickle(trusted=False):
    """A pickle module for checkpoints the REFERENCE wrote: alphazero.utils.dotdict -> utils.dotdict.  By default only an
    allow-list of globals is resolved (tensor rebuild functions, storages, OrderedDict, dotdict); every other global becomes an
    inert placeholder -- only 'state_dict' and the architecture keys of 'args' are used -- so loading a foreign file cannot run
    code.  trusted=True resolves every importable global like the reference's own torch.load (NNetWrapper.py:259)."""
    import pickle
    import types

    class _Missing:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __setstate__(self, state):
            pass

    def allowed(module, name):
        if module == 'collections' and name == 'OrderedDict':
            return True
        if module == 'torch._utils' and name.startswith('_rebuild_'):
            return True
        if module == 'torch' and (name.endswith('Storage') or name in ('Size', 'device', 'dtype')):
            return True
        if module in ('torch.serialization',) and name == '_get_layout':
            return True
        if module == 'numpy.core.multiarray' or module == 'numpy._core.multiarray':
            return name in ('scalar', '_reconstruct')
        if module == 'numpy' and name in ('dtype', 'ndarray'):
            return True
        return False

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module.startswith('alphazero') and name == 'dotdict':
                return dotdict
            if not trusted and not allowed(module, name):
                return _Missing
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return _Missing

    mod = types.ModuleType('azg_reference_pickle')
    for k in dir(pickle):
        if not k.startswith('__'):
            setattr(mod, k, getattr(pickle, k))
    mod.Unpickler = Unpickler
    return mod


class HipResNet:
    """The eval-mode network on the hand-written gfx950 MFMA kernels of csrc/azg_conv.h: the residual tower is one persistent
    launch (azg_resnet_tower_f16) and the two heads -- 1x1 conv + BN + flatten + Linear chain, all linear in the reference
    (NNetArchitecture.py:88-102,112-118) -- are collapsed into ONE [H*W*C, A + P+1] GEMM followed by the two softmaxes: fused
    behind the tower in the same launch when A + P+1 <= 16 at 128 channels (azg_resnet_policy_value_f16), else in the wide-head
    kernel (azg_policy_value_heads_f16).  The input is the engine's obs_dtype 2 format [B, H*W, 8] fp16."""

    @staticmethod
    def profile(on=True):
        """HIP-event timing of every network launch of this process, recorded by the library on the launch stream
        (azg_profile_net_enable); read with profile_read()."""
        from . import _abi
        _abi.check(_abi.lib().azg_profile_net_enable(int(on)))

    @staticmethod
    def profile_read():
        import ctypes as C
        from . import _abi
        ms = (C.c_double * 3)(); n = (C.c_int64 * 3)()
        _abi.check(_abi.lib().azg_profile_net_read(ms, n))
        return dict(tower_ms=ms[0], heads_ms=ms[1], search_ms=ms[2], tower_n=n[0], heads_n=n[1], search_n=n[2])

    def __init__(self, folded: FoldedResNet, game_id, device):
        from . import _abi
        self.L, self.game, self.device = _abi.lib(), int(game_id), torch.device(device)
        self._check = _abi.check
        C, H, W = folded.shape
        self.C, self.HW = C, H * W
        self.CH = CH = int(folded.stem_w.shape[0])                      # tower width
        assert CH in (32, 64, 128) and C <= 8, 'the MFMA tower is built for 32, 64 or 128 channels'
        f32 = dict(dtype=torch.float32, device=self.device)
        with torch.no_grad():
            self.stem_w = pack_stem_weight(folded.stem_w.float()).to(self.device)
            self.stem_b = folded.stem_b.float().to(**f32).contiguous()
            self.blocks = []
            for i in range(len(folded.w1)):
                self.blocks.append(dict(
                    ps=folded.pre_scale[i].float().reshape(-1).to(**f32).contiguous(),
                    pt=folded.pre_shift[i].float().reshape(-1).to(**f32).contiguous(),
                    w1=pack_conv_weight(folded.w1[i].float(), CH // 32).to(self.device), b1=folded.b1[i].float().to(**f32).contiguous(),
                    w2=pack_conv_weight(folded.w2[i].float(), CH // 32).to(self.device)))
            self.zero_b = torch.zeros(CH, **f32)
            # the same parameters laid out for the fused persistent tower (azg_resnet_tower_f16)
            # (+ the readable slack the weight prefetch ring runs into behind the last layer: include/azg.h AZG_TOWER_W_SLACK_KSTEPS;
            #  the library states the total it may read, azg_tower_weights_size)
            layers = [self.stem_w] + [t for b in self.blocks for t in (b['w1'], b['w2'])]
            need = int(self.L.azg_tower_weights_size(CH, len(self.blocks)))
            have = sum(int(t.numel()) for t in layers)
            assert 0 < need - have <= 32 * CH * 32, (need, have)
            self.tower_w = torch.cat(layers + [torch.zeros(need - have, dtype=torch.float16, device=self.device)]).contiguous()
            self.tower_b = torch.stack([self.stem_b] + [t for b in self.blocks for t in (b['b1'], self.zero_b)]).contiguous()
            nb = len(self.blocks)
            self.tower_ps = torch.stack([b['ps'] for b in self.blocks]).contiguous() if nb else torch.zeros((1, CH), **f32)
            self.tower_pt = torch.stack([b['pt'] for b in self.blocks]).contiguous() if nb else torch.zeros((1, CH), **f32)
            # heads: logits[b, o] = sum_{pos,k} s[b,pos,k] * Wfull[pos*128+k, o] + bfull[o]
            hw, hb = folded.head_w.float().reshape(-1, CH), folded.head_b.float()            # [vc+pc, CH], [vc+pc]
            vc = folded.vc
            Wv, bv = folded.v_fc.weight.float(), folded.v_fc.bias.float()                     # [NV, vc*HW] (index c*HW+pos)
            Wp, bp = folded.pi_fc.weight.float(), folded.pi_fc.bias.float()
            NV, A, HW = Wv.shape[0], Wp.shape[0], self.HW
            fv = torch.einsum('ocp,ck->pko', Wv.reshape(NV, vc, HW), hw[:vc])                 # [HW, 128, NV]
            fp = torch.einsum('ocp,ck->pko', Wp.reshape(A, -1, HW), hw[vc:])                  # [HW, 128, A]
            self.A, self.NV = A, NV
            bfv = bv + torch.einsum('ocp,c->o', Wv.reshape(NV, vc, HW), hb[:vc])
            bfp = bp + torch.einsum('ocp,c->o', Wp.reshape(A, -1, HW), hb[vc:])
            self.head_b = torch.cat([bfp, bfv]).to(**f32).contiguous()
            self.fused_head = (A + NV) <= 16 and CH == 128
            if self.fused_head:                                  # MFMA fragment order for the in-tower heads
                wf = torch.zeros((HW, 128, 16), dtype=torch.float32, device=fp.device)
                wf[:, :, :A + NV] = torch.cat([fp, fv], dim=2)
                # [p, k=ks*32+g*8+j, out i] -> [p][ks][g][i][j]
                self.head_w_packed = wf.reshape(HW, 4, 4, 8, 16).permute(0, 1, 2, 4, 3).contiguous().reshape(-1) \
                                       .to(self.device, torch.float16).contiguous()
                self.head_b16 = torch.zeros(16, **f32)
                self.head_b16[:A + NV] = self.head_b
            # the wide-head kernel's layout (azg_policy_value_heads_f16): [k/32][OS][64 lanes][8]
            OS, KS = (A + NV + 15) // 16, HW * CH // 32
            wf = torch.zeros((HW * CH, OS * 16), dtype=torch.float32, device=fp.device)
            wf[:, :A + NV] = torch.cat([fp, fv], dim=2).reshape(HW * CH, A + NV)
            # [k = ks*32 + g*8 + j, out = sub*16 + i] -> [ks][sub][g][i][j]
            self.head_w_wide = wf.reshape(KS, 4, 8, OS, 16).permute(0, 3, 1, 4, 2).contiguous().reshape(-1) \
                                 .to(self.device, torch.float16).contiguous()
            self.head_b_wide = torch.zeros(OS * 16, **f32)
            self.head_b_wide[:A + NV] = self.head_b
            self.head_opad = OS * 16
            # the heads FACTORISED like the reference computes them (1x1 head convs in the tower launch, then the collapsed
            # Linear chains on 16 + 16 head channels per pixel): a quarter of the collapsed matrix's weight traffic at 64 channels
            pc = hw.shape[0] - vc
            self.fact_head = (not self.fused_head) and vc == 16 and pc == 16
            if self.fact_head:
                FK = (HW * 16 + 31) // 32 * 32
                self.feat_k = FK
                w1 = torch.cat([hw[vc:], hw[:vc]])                                            # [32, CH]: policy rows, then value rows
                # A fragments [ks][ms][lane g*16+i][j] = W1[ms*16 + i, ks*32 + g*8 + j]
                self.head1_w = w1.reshape(2, 16, CH // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous().reshape(-1) \
                                 .to(self.device, torch.float16).contiguous()
                self.head1_b = torch.cat([hb[vc:], hb[:vc]]).to(**f32).contiguous()
                OSP = (A + 15) // 16
                w2p = torch.zeros((FK, OSP * 16), dtype=torch.float32, device=Wp.device)      # feature pos*16 + c  <-  flatten index c*HW + pos
                w2p[:HW * 16, :A] = Wp.reshape(A, pc, HW).permute(2, 1, 0).reshape(HW * pc, A)
                w2v = torch.zeros((FK, 16), dtype=torch.float32, device=Wv.device)
                w2v[:HW * 16, :NV] = Wv.reshape(NV, vc, HW).permute(2, 1, 0).reshape(HW * vc, NV)
                frag = lambda w, osub: w.reshape(FK // 32, 4, 8, osub, 16).permute(0, 3, 1, 4, 2).contiguous().reshape(-1) \
                                        .to(self.device, torch.float16).contiguous()
                self.head2_wp, self.head2_wv = frag(w2p, OSP), frag(w2v, 1)
                # (the same policy fragments subtile-major, [OSP][FK/32][64 lanes][8]: what the persistent exact launch streams)
                self.head2_wps = self.head2_wp.reshape(FK // 32, OSP, 64 * 8).permute(1, 0, 2).contiguous().reshape(-1)
                self.head2_b = torch.zeros(OS * 16, **f32)
                self.head2_b[:A] = bp; self.head2_b[A:A + NV] = bv
                # the same chains one ROW per output (policy outputs over the policy features, then the value outputs over the
                # value features): what the tree launch reads when it computes only the logits it needs (sparse heads)
                self.head_rows = torch.cat([w2p[:, :A].t(), w2v[:, :NV].t()]).to(self.device, torch.float16).contiguous()
        self._bufs = {}

    @property
    def wide_head(self):
        """heads in their own launch (the wide-head kernel) instead of fused behind the tower."""
        return not self.fused_head

    def _scratch(self, kind, key, B, make):
        """Scratch tensors of a (kind, key) user, allocated once at the largest row count seen and handed out as views [:B]:
        a caller whose batch size changes every call (the arena's host-split path) keeps ONE set instead of one per size.
        A captured graph uses a fixed B under its own key, so its addresses never move."""
        ent = self._bufs.get((kind, key))
        if ent is None or ent[0] < B:
            ent = (B, make(B))
            self._bufs[(kind, key)] = ent
        return ent[1]

    def _buffers(self, B, key=0):
        mk = lambda n: (torch.empty((n * self.HW, self.CH), dtype=torch.float16, device=self.device),)
        return self._scratch('act', key, B, mk)[0][:B * self.HW]

    def forward_logits_nhwc8(self, x, key=0):
        """Wide-head networks only: x [B, H*W, 8] fp16 -> logits [B, OS*16] float32 (A policy logits, then P+1 value logits per
        row), for DeviceEngine.backup_logits / backup_select_logits, which run the softmaxes inside the tree launch."""
        assert self.wide_head
        return self.forward_nhwc8(x, key, logits_only=True)

    def forward_nhwc8(self, x, key=0, logits_only=False):
        """x: [B, H*W, 8] fp16 -> (policy [B, A], value [B, P+1]) float32 probabilities.  `key` selects a private set
        of activation buffers (one per captured graph, so that graphs on different streams never share scratch)."""
        B = x.shape[0]
        if self.fused_head and not logits_only:                  # tower + heads + softmax in ONE launch
            import ctypes as C
            vp = lambda q: C.c_void_p(q.data_ptr())
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            pol, val = [t[:B] for t in self._scratch('pv', key, B, lambda n: (
                torch.empty((n, self.A), dtype=torch.float32, device=self.device),
                torch.empty((n, self.NV), dtype=torch.float32, device=self.device)))]
            self._check(self.L.azg_resnet_policy_value_f16(st, self.game, vp(x), vp(self.tower_w), vp(self.tower_b), vp(self.tower_ps),
                                                           vp(self.tower_pt), int(B), len(self.blocks), vp(self.head_w_packed),
                                                           vp(self.head_b16), int(self.A), int(self.NV), vp(pol), vp(val)))
            return pol, val
        import ctypes as C
        vp = lambda q: C.c_void_p(q.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        pol, val, ws = [t[:B] for t in self._scratch('pvw', key, B, lambda n: tuple(
            torch.empty((n, w), dtype=torch.float32, device=self.device) for w in (self.A, self.NV, self.head_opad)))]
        null = C.c_void_p(0)
        if self.fact_head:                                       # tower + 1x1 head convs: one launch; collapsed dense chains: one more
            feat = self._scratch('feat', key, B, lambda n: (torch.zeros((n, 2 * self.feat_k), dtype=torch.float16, device=self.device),))[0][:B]
            self._check(self.L.azg_resnet_tower_features_f16(st, self.game, vp(x), vp(self.tower_w), vp(self.tower_b), vp(self.tower_ps),
                                                             vp(self.tower_pt), int(B), len(self.blocks), int(self.CH), vp(self.head1_w),
                                                             vp(self.head1_b), vp(feat), int(self.feat_k)))
            self._check(self.L.azg_policy_value_heads_fact_f16(st, vp(feat), vp(self.head2_wp), vp(self.head2_wv), vp(self.head2_b), int(B),
                                                               int(self.feat_k), int(self.A), int(self.NV), vp(ws),
                                                               null if logits_only else vp(pol), null if logits_only else vp(val)))
            return ws if logits_only else (pol, val)
        s = self._buffers(B, key)                                # tower: one persistent launch, activations resident in LDS
        self._check(self.L.azg_resnet_tower_f16(st, self.game, vp(x), vp(self.tower_w), vp(self.tower_b), vp(self.tower_ps),
                                                vp(self.tower_pt), vp(s), int(B), len(self.blocks), int(self.CH)))
        # fully collapsed heads GEMM + both softmaxes: two launches (or one, stopping at the logits)
        self._check(self.L.azg_policy_value_heads_f16(st, vp(s), vp(self.head_w_wide), vp(self.head_b_wide), int(B), self.HW * self.CH,
                                                      int(self.A), int(self.NV), vp(ws), null if logits_only else vp(pol),
                                                      null if logits_only else vp(val)))
        return ws if logits_only else (pol, val)

    def forward_features_nhwc8(self, x, key=0):
        """Factorised-heads networks only: x [B, H*W, 8] fp16 -> head features [B, 2 * feat_k] fp16 (tower + the two 1x1 head
        convolutions, ONE launch), for DeviceEngine.backup_select_features, which applies the collapsed Linear chains itself --
        only to the valid actions of every leaf."""
        assert self.fact_head
        import ctypes as C
        vp = lambda q: C.c_void_p(q.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        B = x.shape[0]
        feat = self._scratch('feat', key, B, lambda n: (torch.zeros((n, 2 * self.feat_k), dtype=torch.float16, device=self.device),))[0][:B]
        self._check(self.L.azg_resnet_tower_features_f16(st, self.game, vp(x), vp(self.tower_w), vp(self.tower_b), vp(self.tower_ps),
                                                         vp(self.tower_pt), int(B), len(self.blocks), int(self.CH), vp(self.head1_w),
                                                         vp(self.head1_b), vp(feat), int(self.feat_k)))
        return feat

    @property
    def can_search(self):
        """a persistent search launch exists for this network: connect4 x 128 channels with fused heads (azg_search_f16), or
        factorised heads on brandubh x 64 / the 3-player env x 32 / connect4 x {32, 64} channels -- the reference's default net
        (Coach.py:108-116) on connect4 is the 32-channel one -- (azg_search_wide_exact_f16 / azg_search_wide_f16)."""
        return (self.fused_head and self.game == 0 and self.CH == 128) or (self.fact_head and (self.game, self.CH) in ((1, 64), (2, 32), (0, 32), (0, 64)))

    def search(self, engine, sims, exact=False):
        """`sims` whole simulations (select -> this network -> backup) on every slot of `engine` in one persistent launch: the
        trees, the leaf planes and the logits / probabilities never leave the GPU's LDS / HBM and nothing is launched per
        simulation.  Self-play engines only.  Factorised heads: exact=True evaluates ALL A + P+1 logits inside the launch (the
        bits NNetWrapper.process returns; softmax over all A, mask, renormalise: MCTS.pyx:239-245), exact=False only the logits
        of each leaf's valid actions (sparse heads: equal to rounding).  Fused heads (connect4) are always exact."""
        if not self.can_search:
            raise NotImplementedError('no persistent search launch for this (game, network) -- use select / network / backup')
        import ctypes as C
        vp = lambda q: C.c_void_p(q.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if self.fused_head:
            self._check(self.L.azg_search_f16(engine.h, st, vp(self.tower_w), vp(self.tower_b), vp(self.tower_ps), vp(self.tower_pt),
                                              len(self.blocks), vp(self.head_w_packed), vp(self.head_b16), int(sims)))
        elif exact:
            self._check(self.L.azg_search_wide_exact_f16(engine.h, st, vp(self.tower_w), vp(self.tower_b), vp(self.tower_ps), vp(self.tower_pt),
                                                         len(self.blocks), int(self.CH), vp(self.head1_w), vp(self.head1_b), vp(self.head2_wps),
                                                         vp(self.head2_wv), vp(self.head2_b), int(self.feat_k), int(sims)))
        else:
            self._check(self.L.azg_search_wide_f16(engine.h, st, vp(self.tower_w), vp(self.tower_b), vp(self.tower_ps), vp(self.tower_pt),
                                                   len(self.blocks), int(self.CH), vp(self.head1_w), vp(self.head1_b), vp(self.head_rows),
                                                   vp(self.head2_b), int(self.feat_k), int(sims)))

    def search_tile(self, engine, exact=False):
        """the tile the persistent wide-head launch runs `engine` with (azg_search_wide_tile_info): dict(games_per_workgroup, workgroups,
        workgroups_per_cu, cus, source = 'model' | 'measured' | 'forced', trial_us = set-up measurement per tile shape); None for
        fused-head networks (connect4 x 128: the tile follows the engine's size alone) or before the launch's first call."""
        if not (self.fact_head and self.can_search):
            return None
        import ctypes as C
        info = (C.c_int32 * 12)()
        if self.L.azg_search_wide_tile_info(engine.h, int(self.CH), len(self.blocks), int(bool(exact)), info) != 0:
            return None
        return dict(games_per_workgroup=int(info[0]), workgroups=int(info[1]), workgroups_per_cu=int(info[2]), cus=int(info[3]),
                    source=('model', 'measured', 'forced')[int(info[4])], trial_sims=int(info[5]), trial_us=[round(info[8 + t] / 1e3, 1) for t in range(4)])

    def stream_bytes(self, exact=True, kbar=None):
        """bytes of network parameters ONE workgroup of a persistent launch streams from L2 per simulation (every workgroup re-reads them:
        they do not fit LDS beside the image): all tower fragments + the head operands -- fused heads: the collapsed [H*W*128, 16] matrix;
        factorised exact: the 1x1 head fragments + every policy subtile's and the value chain's fragments; sparse: the 1x1 head fragments
        + kbar gathered rows of the collapsed matrix.  (bench.py: the operand-stream bound of the small-shard launches.)"""
        tower = sum(int(t.numel()) for t in [self.stem_w] + [t for b in self.blocks for t in (b['w1'], b['w2'])]) * 2
        if self.fused_head:
            head = int(self.head_w_packed.numel()) * 2
        elif self.fact_head and exact:
            head = (int(self.head1_w.numel()) + int(self.head2_wps.numel()) + int(self.head2_wv.numel())) * 2
        elif self.fact_head:
            head = int(self.head1_w.numel()) * 2 + int((kbar or 0) + self.NV) * self.feat_k * 2
        else:
            head = int(self.head_w_wide.numel()) * 2
        return dict(tower=tower, heads=head, total=tower + head)

    @staticmethod
    def forward_models(nets, x_all, policy_all, value_all, rows_per_model):
        """Arena evaluation in ONE launch without a host read of the batch split: nets[m] (HipResNets of one architecture)
        evaluates rows [sum(rows_per_model[:m]), + rows_per_model[m]) of x_all [B, H*W, 8] into policy_all / value_all
        (rows_per_model: int32 device tensor from DeviceEngine.arena_rows).  Needs the fused tower + heads kernel."""
        import ctypes as C
        n0 = nets[0]
        for n in nets:
            if not n.fused_head:
                raise NotImplementedError('multi-model launches need the fused tower + heads kernel (128 channels, A + NV <= 16)')
            assert (n.game, n.CH, len(n.blocks), n.A, n.NV) == (n0.game, n0.CH, len(n0.blocks), n0.A, n0.NV)
        arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        vp = lambda q: C.c_void_p(q.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        n0._check(n0.L.azg_resnet_policy_value_multi_f16(
            st, n0.game, vp(x_all), len(nets), arr([n.tower_w for n in nets]), arr([n.tower_b for n in nets]),
            arr([n.tower_ps for n in nets]), arr([n.tower_pt for n in nets]), int(x_all.shape[0]), len(n0.blocks),
            arr([n.head_w_packed for n in nets]), arr([n.head_b16 for n in nets]), int(n0.A), int(n0.NV), vp(policy_all), vp(value_all),
            vp(rows_per_model)))

    @staticmethod
    def search_arena(nets, engine, sims, player_to_index=None, slot_seats=None):
        """a whole arena move in ONE persistent launch (azg_search_arena_f16): every game of `engine` (an arena engine) is searched
        `sims` simulations on its mover's tree with its mover's model -- nets[player_to_index[mover]], or the slot's own seating
        (slot_seats: int32 device tensor, 4 bits per player).  Needs the fused tower + heads kernel on every model."""
        import ctypes as C
        n0 = nets[0]
        for n in nets:
            if not n.fused_head or n.game != 0 or n.CH != 128:
                raise NotImplementedError('the persistent arena launch needs connect4 models with the fused tower + heads kernel (128 channels)')
            assert (len(n.blocks), n.A, n.NV) == (len(n0.blocks), n0.A, n0.NV)
        arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p2i = None if player_to_index is None else (C.c_int32 * len(player_to_index))(*[int(x) for x in player_to_index])
        seats = None if slot_seats is None else C.c_void_p(slot_seats.data_ptr())
        n0._check(n0.L.azg_search_arena_f16(engine.h, st, len(nets), arr([n.tower_w for n in nets]), arr([n.tower_b for n in nets]),
                                            arr([n.tower_ps for n in nets]), arr([n.tower_pt for n in nets]), len(n0.blocks),
                                            arr([n.head_w_packed for n in nets]), arr([n.head_b16 for n in nets]), p2i, seats, int(sims)))

    def to_nhwc8(self, batch):
        """[B, C, H, W] (any float dtype) -> [B, H*W, 8] fp16."""
        B, C = batch.shape[0], batch.shape[1]
        if batch.dtype == torch.float32 and batch.is_cuda and batch.is_contiguous() and B > 0:      # one launch (azg_obs_to_nhwc8_f16)
            import ctypes as Ct
            x = torch.empty((B, self.HW, 8), dtype=torch.float16, device=self.device)
            self._check(self.L.azg_obs_to_nhwc8_f16(Ct.c_void_p(torch.cuda.current_stream().cuda_stream), Ct.c_void_p(batch.data_ptr()), int(B), int(C),
                                                    int(self.HW), Ct.c_void_p(x.data_ptr())))
            return x
        x = torch.zeros((B, self.HW, 8), dtype=torch.float16, device=self.device)
        x[:, :, :C] = batch.to(self.device).reshape(B, C, self.HW).permute(0, 2, 1)
        return x


class CapturedNet:
    """One hipGraph-captured fixed-batch evaluation: static input x, static outputs policy / value."""

    def __init__(self, graph, x, policy, value, run=None):
        self.graph, self.x, self.policy, self.value = graph, x, policy, value
        self.run_logits = None
        self.run_features = None        # factorised heads: -> (features, head rows, head bias) for backup_select_features
        self.run = run                   # the same evaluation as plain launches on the current stream -> (policy, value); lets a
                                         # caller capture it inside a larger graph (selfplay: a whole round of simulations)

    def replay(self):
        self.graph.replay()


_PINNED = {}                                                   # data_ptr -> [nbytes, registered?, weakref to the tensor object, registrations]


def _unpin(key):
    ent = _PINNED.pop(key, None)
    if ent is not None and ent[1] == 'registered':
        if ent[3] >= 8:                                        # (a registration that served many calls was worth it: an address the allocator
            _PIN_COUNT.pop(key, None)                          #  hands out again -- Coach's tensors of the next iteration -- starts afresh)
        try:
            # (the tensor object can be collected right behind a non_blocking H2D copy of its memory -- a temporary view, a tensor from an mp
            #  queue as in Arena.pyx:275: the DMA must have finished before the pages are unlocked)
            torch.cuda.synchronize()
            torch.cuda.cudart().cudaHostUnregister(key)
        except Exception:                                      # noqa: BLE001 (interpreter shutdown)
            pass


def pin_shared(t):
    """Page-lock a caller-owned SHARED-MEMORY CPU tensor in place (hipHostRegister) for as long as the tensor object lives: Coach /
    Arena hand the same shared input tensors to nnet.process for a whole iteration (Coach.py:294-314 -- their own pin_memory() calls
    discard the result, SURVEY.md Q17, so the tensors arrive pageable), and a pageable host -> device copy is staged synchronously.
    Registered memory is DMA'd directly.  Only tensors that say is_shared() are touched; the registration is dropped when the tensor
    object is collected (a weakref callback, before its memory is unmapped); a caller that hands a NEW tensor object over the same
    memory every call is left on the pageable path after a few short-lived registrations; a failure is remembered.  True if the tensor is pinned."""
    import weakref
    if t.device.type != 'cpu' or not torch.cuda.is_available():
        return False
    if not t.is_shared():                                      # (a private tensor is never registered, nor remembered)
        return t.is_pinned()
    key, n = t.untyped_storage().data_ptr(), t.untyped_storage().nbytes()
    ent = _PINNED.get(key)
    if ent is not None and ent[0] >= n:
        ent[3] += 1
        _PIN_STATS['dma' if ent[1] != 'no' else 'staged'] += 1
        return ent[1] != 'no'
    if ent is not None:                                        # a larger storage at the same address: drop the old (shorter) registration first
        _unpin(key)
    ok = 'no'
    try:
        if t.is_pinned():
            ok = 'pinned'
        elif t.is_shared() and n >= 4096 and _PIN_COUNT.get(key, 0) < 4:
            if int(torch.cuda.cudart().cudaHostRegister(key, n, 0)) == 0:
                ok = 'registered'
                _PIN_COUNT[key] = _PIN_COUNT.get(key, 0) + 1
    except Exception:                                          # noqa: BLE001 (a runtime without host registration: the pageable path stays)
        ok = 'no'
    _PINNED[key] = [n, ok, weakref.ref(t, lambda _r, key=key: _unpin(key)), 1]
    _PIN_STATS['dma' if ok != 'no' else 'staged'] += 1
    if ok == 'registered':
        _PIN_STATS['registrations'] += 1
    elif ok == 'no' and not _PIN_STATS['warned']:              # the fall-back is never silent: say once why this tensor stays pageable
        _PIN_STATS['warned'] = 1
        import warnings
        warnings.warn('alphazero_general_amd: a shared batch tensor (%d bytes) is copied through the pageable (staged) path: %s -- pin_stats() counts them'
                      % (n, 'smaller than a page' if n < 4096 else 'its address was registered and dropped %d times (a new tensor object per call?)'
                         % _PIN_COUNT.get(key, 0) if _PIN_COUNT.get(key, 0) >= 4 else 'hipHostRegister refused it'), RuntimeWarning, stacklevel=3)
    return ok != 'no'


_PIN_COUNT = {}
_PIN_STATS = {'dma': 0, 'staged': 0, 'registrations': 0, 'warned': 0}


def pin_stats(reset=False):
    """How the shared CPU batches handed to NNetWrapper.process travelled: 'dma' (page-locked in place: registered or already pinned),
    'staged' (pageable: the runtime stages the copy synchronously), 'registrations' (hipHostRegister calls that succeeded)."""
    out = {k: _PIN_STATS[k] for k in ('dma', 'staged', 'registrations')}
    if reset:
        for k in out:
            _PIN_STATS[k] = 0
    return out


class NNetWrapper:
    """The slice of alphazero/NNetWrapper.py the search path calls: predict (:207-223), process (:225-232),
    __call__ (:35-36), plus state_dict-compatible save/load of the network (:240-274)."""

    def __init__(self, game_cls, args=None, *, device=None, dtype=torch.float16, fast=True, backend='auto'):
        self.game_cls = game_cls
        self.args = dotdict(DEFAULT_NET_ARGS.copy() if args is None else args)
        for k, v in DEFAULT_NET_ARGS.items():
            self.args.setdefault(k, v)
        obs = tuple(game_cls.observation_size())
        self.device = torch.device(device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu'))
        self.nnet = ResNet(obs, game_cls.action_size(), game_cls.num_players() + game_cls.has_draw(), self.args).to(self.device)
        self.nnet.eval()
        self.dtype = dtype if self.device.type == 'cuda' else torch.float32
        self.fast = fast
        # backend: 'torch' = folded PyTorch modules (MIOpen); 'hip' = hand-written MFMA tower (csrc/azg_conv.h);
        # 'auto' = hip when the kernel is instantiated for (game geometry, tower width) and a GPU is present, else torch
        self.backend = backend
        self._infer = None
        self._hip = None
        self._graph = None

    def __call__(self, board):
        return self.predict(board)

    def refresh(self):
        """Rebuild the folded inference network after the weights changed."""
        net = FoldedResNet(self.nnet).to(self.device).to(self.dtype)
        if self.device.type == 'cuda':
            net = net.to(memory_format=torch.channels_last)
        self._infer, self._graph, self._hip = net.eval(), None, None
        use_hip = self.backend == 'hip' or (self.backend == 'auto' and self.device.type == 'cuda'
                                            and (getattr(self.game_cls, 'AZG_GAME_ID', None), self.args.num_channels) in
                                            ((0, 32), (0, 64), (0, 128), (1, 64), (1, 128), (2, 32)))
        if use_hip:
            self._hip = HipResNet(FoldedResNet(self.nnet).to(self.device), self.game_cls.AZG_GAME_ID, self.device)
        return self

    @torch.no_grad()
    def process(self, batch):
        """batch [B,C,H,W] (any float dtype, any device) -> (policy [B,A], value [B,P+1]) float32 probabilities on
        the network's device."""
        if not self.fast:
            pi, v = self.nnet(batch.to(self.device, torch.float32))
            return torch.exp(pi), torch.exp(v)
        if self._infer is None:
            self.refresh()
        if self._hip is not None:
            if batch.device.type == 'cpu' and pin_shared(batch):     # (compat mode: the caller's shared batch tensor, DMA'd instead of staged)
                batch = batch.to(self.device, non_blocking=True)
            return self._hip.forward_nhwc8(self._hip.to_nhwc8(batch))
        x = batch.to(self.device, self.dtype)
        if self.device.type == 'cuda':
            x = x.contiguous(memory_format=torch.channels_last)
        return self._infer(x)

    def predict(self, board):
        b = torch.as_tensor(np.asarray(board, dtype=np.float32))[None]
        p, v = self.process(b)
        return p[0].cpu().numpy(), v[0].cpu().numpy()

    # ---- hipGraph-captured fixed-batch evaluation: static input/output tensors the engine reads and writes
    def capture(self, batch_size, in_dtype=None):
        """Capture one fixed-batch evaluation into a hipGraph.  Returns (static input, policy, value); with the MFMA
        backend the input is the engine's obs_dtype 2 tensor [B, H*W, 8] fp16, otherwise [B, C, H, W].  Every call makes
        an independent graph with its own buffers (CapturedNet), so several can run on different streams."""
        cap = self.capture_net(batch_size, in_dtype)
        self._graph = (cap.graph, cap.x, cap.policy, cap.value)
        return cap.x, cap.policy, cap.value

    def capture_net(self, batch_size, in_dtype=None):
        assert self.device.type == 'cuda'
        if self._infer is None:
            self.refresh()
        in_dtype = in_dtype or self.dtype
        C, H, W = self.nnet.channels, self.nnet.board_x, self.nnet.board_y
        self._ncap = getattr(self, '_ncap', 0) + 1
        key = self._ncap
        if self._hip is not None:
            x = torch.zeros((batch_size, H * W, 8), dtype=torch.float16, device=self.device)
            run = lambda: self._hip.forward_nhwc8(x, key)
        else:
            x = torch.zeros((batch_size, C, H, W), dtype=in_dtype, device=self.device)
            run = lambda: self.process(x)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(3):
                run()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), torch.no_grad():
            p, v = run()
        cn = CapturedNet(g, x, p, v, run)
        if self._hip is not None and self._hip.wide_head:
            cn.run_logits = lambda: self._hip.forward_logits_nhwc8(x, key)   # stops at the logits (softmax inside the tree launch)
            if self._hip.fact_head:                                  # stops at the head features (sparse heads inside the tree launch)
                cn.run_features = lambda: (self._hip.forward_features_nhwc8(x, key), self._hip.head_rows, self._hip.head2_b)
        return cn

    @property
    def input_is_nhwc8(self):
        return self._hip is not None

    def replay(self):
        self._graph[0].replay()

    def adopt(self, source, args=None):
        """Take over the weights of a LIVE network without a checkpoint file: `source` is a state_dict, a torch module, or a wrapper
        that holds one as `.nnet` -- the reference's own NNetWrapper (NNetWrapper.py:118-126; its ResNet's state_dict keys are this
        ResNet's, the same compatibility load_checkpoint relies on).  `args` (default: source.args when it has them) names the
        architecture; when its keys differ from this wrapper's the network is rebuilt first, as load_checkpoint(use_saved_args=True)
        does.  The folded inference copies are dropped, so the next process() / search evaluates the adopted weights.  Returns self."""
        if args is None:
            args = getattr(source, 'args', None)
        sd = source
        if not isinstance(sd, dict):
            mod = getattr(source, 'nnet', source)
            mod = getattr(mod, 'module', mod)                        # (torch.nn.DataParallel)
            sd = mod.state_dict()
        if args is not None:
            arch = {k: args[k] for k in DEFAULT_NET_ARGS if k in args}
            if any(self.args.get(k) != v for k, v in arch.items()):
                self.args.update(arch)
                obs = tuple(self.game_cls.observation_size())
                self.nnet = ResNet(obs, self.game_cls.action_size(), self.game_cls.num_players() + self.game_cls.has_draw(),
                                   self.args).to(self.device)
                self.nnet.eval()
        self.nnet.load_state_dict({k: v.detach() for k, v in sd.items()})      # (copies onto this wrapper's device; strict: a key mismatch raises)
        self._infer = self._hip = self._graph = None
        return self

    def save_checkpoint(self, folder='checkpoint', filename='checkpoint.pth.tar', make_dirs=True):
        """NNetWrapper.save_checkpoint (NNetWrapper.py:239-250): {'state_dict', 'args'} under folder/filename (no optimizer /
        scheduler state: this wrapper only evaluates); the reference's load_checkpoint reads it."""
        import os
        import pickle
        if make_dirs and not os.path.exists(folder):
            os.makedirs(folder)
        torch.save({'state_dict': self.nnet.state_dict(), 'args': self.args}, os.path.join(folder, filename),
                   pickle_protocol=pickle.HIGHEST_PROTOCOL)

    def load_checkpoint(self, folder='checkpoint', filename='checkpoint.pth.tar', use_saved_args=True, trusted=False):
        """NNetWrapper.load_checkpoint (NNetWrapper.py:252-276), also for files the REFERENCE wrote: their pickles name
        alphazero.utils.dotdict (mapped to this package's dotdict when the reference is not importable); opt_state / sch_state
        are ignored.  With use_saved_args the network is rebuilt from the saved architecture keys first.  Returns the saved
        args (or None).  The file is read with torch's weights_only loader when it can be, else with an allow-list unpickler
        (unknown globals become placeholders); trusted=True unpickles everything, like the reference does."""
        import os
        path = os.path.join(folder, filename)
        if not os.path.exists(path):
            raise FileNotFoundError('No model in path {}'.format(path))
        try:
            with torch.serialization.safe_globals([dotdict]):
                ck = torch.load(path, map_location='cpu', weights_only=True)
        except Exception:
            ck = torch.load(path, map_location='cpu', weights_only=False, pickle_module=_reference_pickle(trusted))
        saved = ck.get('args')
        if use_saved_args and saved is not None:
            arch = {k: saved[k] for k in DEFAULT_NET_ARGS if k in saved}
            if any(self.args.get(k) != v for k, v in arch.items()):
                self.args.update(arch)
                obs = tuple(self.game_cls.observation_size())
                self.nnet = ResNet(obs, self.game_cls.action_size(), self.game_cls.num_players() + self.game_cls.has_draw(),
                                   self.args).to(self.device)
                self.nnet.eval()
        self.nnet.load_state_dict(ck['state_dict'])
        self._infer = self._hip = self._graph = None
        return dotdict(saved) if saved is not None else None
