// azg_conv.h -- the policy/value ResNet of alphazero/NNetArchitecture.py:36-120 (eval mode) on gfx950 MFMA.
//
// PyTorch issues, per residual-tower layer, a MIOpen implicit-GEMM kernel + 4 elementwise launches (bias, ReLU, pre-activation
// BN affine, residual add: profiles/r01_bench_kernel_stats_baseline.csv, 47 + ~38 us per layer).  Here the whole tower -- and,
// for narrow action spaces, both heads and their softmaxes -- is ONE persistent launch (k_tower2) with the activations
// resident in LDS; wide heads (brandubh: 588 + 3 outputs) run in k_heads.
//
// Formulation of a convolution: Y^T[cout, pixel] = sum over 9 taps of W_tap[cout, cin] . X^T[cin, pixel + shift(tap)]
//   * a workgroup owns BOARDS whole boards; its conv input lives in an LDS image (see TowerGeom) that all 9 taps read at
//     constant offsets; taps that fall off the board hit zero padding;
//   * wave w computes couts [32w, 32w+32) for all pixels: v_mfma_f32_16x16x32_f16 with A = weights (16 couts x 32 cin,
//     streamed from L2 in a pre-packed fragment order: each wave needs only its own cout slice, so LDS sharing would buy
//     nothing) and B = activations (32 cin x 16 pixels, one ds_read_b128 per lane);
//     D[cout = 4*(lane/16)+r][pixel = lane%16] puts 4 consecutive channels of one pixel in a lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace azg {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ================================================================================================ fused tower
// t = relu(bn(s)) -> u = relu(conv1(t)+b) -> s' = conv2(u) + s for every block, activations never leaving the CU between the
// input planes and the final stream: the only per-layer traffic is the weight fragments, shared by all CUs out of L2.
struct TowerParams {
    const void *x;            // [boards*H*W, 8] fp16 (NHWC, channels padded to 8) -- the engine's obs_dtype 2
    const void *w;            // packed fragments: stem [9][1][8][64] then per block conv1, conv2 [9][4][8][64], 16 B each
    const float *bias;        // [1 + 2*nblocks][128]
    const float *pre_scale;   // [nblocks][128]
    const float *pre_shift;   // [nblocks][128]
    void *y;                  // [boards*H*W, 128] fp16: final residual stream (written when head_w == null)
    int boards, nblocks;
    // optional fused heads (A + NV <= 16): logits = s_final . head_w + head_b, two softmaxes, written as f32 probabilities
    const void *head_w;       // packed fragments [H*W][4][64] x 16 B: lane g*16+i holds Wfull[p*128 + ks*32 + g*8 + j][out i]
    const float *head_b;      // [16]
    float *policy, *value;    // [boards, A], [boards, NV]
    int A, NV;
    unsigned long long *dbg;  // AZG_TOWER_TIMING builds only: s_memtime stamps of workgroup 0 [layer][wave][5]
    // optional factorised heads, first stage (head1_w != null, head_w == null): instead of the final stream the launch writes the
    // 32 head channels of every pixel -- the two 1x1 head convolutions with their BatchNorms folded (NNetArchitecture.py:88-89,
    // 97-98), 16 policy channels then 16 value channels -- as feature rows feat[board][2][feat_k] fp16, feature index pos * 16 + c
    // (feat_k = H*W*16 rounded up to 32; the padding is never written: the buffer must start zeroed)
    const void *head1_w;      // packed fragments [C/32][2][64] x 16 B: lane g*16+i holds W1[out = ms*16 + i][cin = ks*32 + g*8 + j]
    const float *head1_b;     // [32]
    void *feat;
    int feat_k;
    // optional multi-model launch (the arena: every model evaluates its own contiguous slice of the leaf batch and the split
    // is only known on the device): model m owns boards [sum(rows_per_model[0..m)), + rows_per_model[m]) of x / policy /
    // value / y and brings its own parameters (model 0: the fields above, model m > 0: alt[m-1]); tiles never straddle
    // models; `boards` is then only the upper bound the grid is sized for (k_tower2 only)
    const int32_t *rows_per_model;
    int nmodels;
    struct Model { const void *w; const float *bias, *pre_scale, *pre_shift; const void *head_w; const float *head_b; } alt[3];
};

// LDS image of the tower: every board is stored with one pad line above and two pad columns to the right
// ((H+1) lines of W+2 positions, LEAD pad rows in front), so that ALL nine taps of every real pixel are plain
// in-bounds reads that return zeros off the board -- no per-tap masking and no address arithmetic: the address of
// pixel p's fragment for (tap, k-step) is   lane_base[p] + const(tap, ks)   and the constant folds into the ds_read
// offset field (the main loop is nothing but [2 MFMA, 1 ds_read offset:imm] groups).
// Rows are 288 B apart (256 B of channels + 32 B pad): chunk c of padded row q starts at 16-byte bank slot
// (c + 2q) mod 16 -- the conflict-free mapping derived above, obtained from the pad instead of a swizzle.  What it
// needs is that the 8 lanes {0-3,12-15} and the 8 lanes {4-11} of a fragment each cover all residues q mod 8; padded
// rows are not consecutive, so the pixel -> (subtile, lane) assignment is a host-built table (tower_pixmap) that
// deals every 8-lane set one pixel of each residue class.  The board stride (== 2 mod 8) makes the classes equal.
template <int H, int W, int BOARDS, int C = 128>
struct TowerGeom {
    static constexpr int HW = H * W, ROWS = BOARDS * HW, NSUB = (ROWS + 15) / 16;
    static constexpr int PW = W + 2, LEAD = PW + 1;
    static constexpr int BS0 = (H + 1) * PW, BSTRIDE = BS0 + ((2 - BS0 % 8) + 8) % 8;     // == 2 (mod 8)
    static constexpr int TROWS = LEAD + (BOARDS - 1) * BSTRIDE + BS0 + PW + 1;
    // row stride = channels * 2 B + 32 B pad: RSTRIDE/16 == 2 (mod 4) puts chunk c of row q at slot (c + k q) mod 16 with k in
    // {2,6,10,14} -- 288 B for 128 channels, 160 B for 64 (see the conflict-freeness argument above)
    static constexpr int CH = C, KSC = C / 32, NW = C / 32, WSTEP = C * 4;   // k-steps per tap, waves, fragments (16 B) per k-step
    static constexpr int RSTRIDE = 2 * C + 32, TILE = TROWS * RSTRIDE;
    static constexpr int BIAS = (PW + 1) * RSTRIDE;                          // makes every tap offset non-negative
    __host__ __device__ static __forceinline__ int qrow(int p) {             // padded row of pixel p (tile-local)
        const int b = p / HW, pos = p - b * HW, y = pos / W, x = pos - y * W;
        return LEAD + b * BSTRIDE + (y + 1) * PW + x;
    }
    // Border classes.  A pixel of the top row reads the zero pad row for the three dy = -1 taps (likewise bottom / left /
    // right), so a subtile made ONLY of top-row pixels can drop those taps' MFMAs: every product in them is an exact zero.
    // The subtiles are therefore filled class by class -- CLASSES == 5: interior, top row, bottom row, left column, right
    // column (rows own the corners); CLASSES == 3: middle rows, top row, bottom row -- whenever that needs no more subtiles
    // than the unclassed tiling (6x7 x 4 boards: 5+2+2+1+1 = 11 subtiles, 81 of 99 subtile-taps remain; 6x7 x 2, 7x7 x 2 and
    // 5x5 x 2 fit the three-class form).  CLASSES == 1 otherwise.
    static constexpr int sub16(int n) { return (n + 15) / 16; }
    static constexpr int S5I = sub16(BOARDS * (H - 2) * (W - 2)), S5R = sub16(BOARDS * W), S5C = sub16(BOARDS * (H - 2));
    static constexpr int S3M = sub16(BOARDS * (H - 2) * W);
    static constexpr int CLASSES = (H > 2 && W > 2 && S5I + 2 * S5R + 2 * S5C == NSUB) ? 5 : (H > 2 && S3M + 2 * S5R == NSUB) ? 3 : 1;
    static constexpr int SMAIN = CLASSES == 5 ? S5I : S3M;                   // subtiles of class 0
    __host__ __device__ static constexpr int pixel_class(int p) {
        const int pos = p % HW, y = pos / W, x = pos % W;
        if (CLASSES == 1) return 0;
        if (y == 0) return 1;
        if (y == H - 1) return 2;
        if (CLASSES == 3) return 0;
        return x == 0 ? 3 : x == W - 1 ? 4 : 0;
    }
    __host__ __device__ static constexpr int class_first(int c) {           // first subtile of class c (class_first(CLASSES) == NSUB)
        if (CLASSES == 1) return c == 0 ? 0 : NSUB;
        return c == 0 ? 0 : c == 1 ? SMAIN : c == 2 ? SMAIN + S5R : c == 3 ? SMAIN + 2 * S5R : c == 4 ? SMAIN + 2 * S5R + S5C : NSUB;
    }
    __host__ __device__ static constexpr int subtile_class(int gs) {
        int c = 0;
        while (c + 1 < CLASSES && gs >= class_first(c + 1)) c++;
        return c;
    }
    __host__ __device__ static constexpr bool tap_active(int tap, int gs) {  // false: every pixel of subtile gs is off-board for the tap
        const int c = subtile_class(gs), dy = tap / 3 - 1, dx = tap % 3 - 1;
        return !((c == 1 && dy < 0) || (c == 2 && dy > 0) || (c == 3 && dx < 0) || (c == 4 && dx > 0));
    }
};

// host: pixel index for (subtile, lane&15), -1 = spare lane.  Within a border class (see TowerGeom), 8-lane set k = k-th
// pixel of every residue class.
template <class GEO>
static bool tower_pixmap(int16_t *map /*[NSUB*16]*/) {          // false: a class did not fit its subtiles (cannot happen: CLASSES checks the counts)
    static const int setA[8] = {0, 1, 2, 3, 12, 13, 14, 15}, setB[8] = {4, 5, 6, 7, 8, 9, 10, 11};
    for (int i = 0; i < GEO::NSUB * 16; i++) map[i] = -1;
    for (int c = 0; c < GEO::CLASSES; c++) {
        const int s0 = GEO::class_first(c), s1 = GEO::class_first(c + 1);
        int cls[8][GEO::ROWS], cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = 0; p < GEO::ROWS; p++) if (GEO::pixel_class(p) == c) { int r = GEO::qrow(p) & 7; cls[r][cnt[r]++] = p; }
        // deal the residue classes round-robin: 8-lane set k takes, for every residue r, the k-th pixel of class r if it
        // exists; leftovers (residue classes of unequal size) go to the free lanes of the class's subtiles
        int nsets = (s1 - s0) * 2, used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < nsets; k++) {
            const int *lanes = (k & 1) ? setB : setA; int ps = s0 + (k >> 1), li = 0;
            for (int r = 0; r < 8; r++) if (used[r] < cnt[r] && used[r] <= k) map[ps * 16 + lanes[li++]] = (int16_t)cls[r][used[r]++];
        }
        for (int r = 0; r < 8; r++)
            while (used[r] < cnt[r]) {
                bool placed = false;
                for (int i = s0 * 16; i < s1 * 16 && used[r] < cnt[r]; i++) if (map[i] < 0) { map[i] = (int16_t)cls[r][used[r]++]; placed = true; }
                if (!placed) return false;
            }
    }
    return true;
}

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// ---- collapsed heads for large action spaces (brandubh: A + NV = 591) ------------------------------------------------------
// logits[b, o] = sum_k y[b, k] * Wh[k, o] + bias[o] over the tower's final stream y [boards, K = H*W*C] (fp16 rows), then the
// two softmaxes of NNetArchitecture.py:112-118.  Too wide to fuse behind the tower (every tile would stream the whole 3.7 MB
// matrix), so it is its own launch: workgroup = (16 boards) x (HEAD_NS output subtiles of 16); its eight waves split K, each
// streaming activation fragments (A operand: 16 boards x 32 k) and pre-packed weight fragments (B operand: 32 k x 16 outputs,
// [k-step][subtile][64 lanes][8 halves]) straight from L2 -- no reuse inside a workgroup, so no LDS staging.  With
// blockIdx = group * nchunks + chunk and 8 chunks, the workgroups sharing a weight chunk sit on one XCD (blockIdx mod 8).
// The job is L2->CU bandwidth bound: (HEAD_NS*16 + 16) * K * 2 bytes per workgroup.
constexpr int HEAD_NS = 5, HEAD_WAVES = 8, HEAD_U = 4;   // (batches of 2-4 k-steps measured best; 7 is 2-6 % slower)

// One workgroup of the heads GEMM: 16 boards x nsub <= HEAD_NS output subtiles over `ksteps` k-steps of 32.  yrow: this lane's A
// operand stream (board i16, k offset g * 8); wl: this lane's B fragments of subtile 0 of the chunk, `wstride` fragments (64 lanes
// each) from one k-step to the next.  The eight waves split K and work in batches of HEAD_U k-steps: all the fragment loads of a
// batch are issued back to back (the job is L2 latency and bandwidth, not MFMA), branch-free: subtiles past the end re-read the
// last real one (their accumulators are never stored), k-steps past the end re-read the last one with the A fragment zeroed.
// logits[board][out0 + s * 16 + i] for s < nsub, columns below out_lim only.
__device__ __forceinline__ void heads_chunk(float (*red)[HEAD_NS * 256], const half8 *yrow, const half8 *wl, size_t wstride, int ksteps, int nsub,
                                            const float *bias, float *logits, int opad, int b0, int boards, int out0, int out_lim) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    floatx4 acc[HEAD_NS];
#pragma unroll
    for (int s = 0; s < HEAD_NS; s++) acc[s] = (floatx4){0.f, 0.f, 0.f, 0.f};
    size_t soff[HEAD_NS];
#pragma unroll
    for (int s = 0; s < HEAD_NS; s++) soff[s] = (size_t)min(s, nsub - 1) * 64;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k0 = wave; k0 < ksteps; k0 += HEAD_WAVES * HEAD_U) {
        half8 a[HEAD_U], b[HEAD_U][HEAD_NS];
#pragma unroll
        for (int u = 0; u < HEAD_U; u++) {
            const int ks = k0 + u * HEAD_WAVES, kc = min(ks, ksteps - 1);
            a[u] = yrow[(size_t)kc * 4];
            if (ks >= ksteps) a[u] = zero8;
#pragma unroll
            for (int s = 0; s < HEAD_NS; s++) b[u][s] = wl[(size_t)kc * wstride + soff[s]];
        }
        __builtin_amdgcn_sched_barrier(0);                      // (left alone hipcc sinks every load next to its MFMA and waits)
#pragma unroll
        for (int u = 0; u < HEAD_U; u++)
#pragma unroll
            for (int s = 0; s < HEAD_NS; s++) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b[u][s], acc[s], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < HEAD_NS; s++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave][(s * 4 + r) * 64 + lane] = acc[s][r];
    __syncthreads();
    for (int e = tid; e < HEAD_NS * 256; e += HEAD_WAVES * 64) {              // D[m = board g*4 + r][n = output i16]
        const int s = e >> 8, r = (e >> 6) & 3, ln = e & 63, board = b0 + (ln >> 4) * 4 + r, out = out0 + s * 16 + (ln & 15);
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < HEAD_WAVES; w++) sum += red[w][e];
        if (board < boards && s < nsub && out < out_lim) logits[(size_t)board * opad + out] = sum + bias[out];
    }
}

__global__ __launch_bounds__(HEAD_WAVES * 64) void k_heads(const _Float16 *y, const half8 *wp, const float *bias, float *logits, int boards,
                                                          int ksteps, int osub) {
    __shared__ float red[HEAD_WAVES][HEAD_NS * 256];
    const int lane = threadIdx.x & 63, g = lane >> 4, i16 = lane & 15;
    const int nchunks = (osub + HEAD_NS - 1) / HEAD_NS, grp = blockIdx.x / nchunks, chunk = blockIdx.x - grp * nchunks;
    const int b0 = grp * 16, s0 = chunk * HEAD_NS;
    const half8 *yrow = reinterpret_cast<const half8 *>(y) + (size_t)min(b0 + i16, boards - 1) * ((size_t)ksteps * 4) + g;
    heads_chunk(red, yrow, wp + (size_t)s0 * 64 + lane, (size_t)osub * 64, ksteps, min(HEAD_NS, osub - s0), bias, logits, osub * 16, b0, boards,
                s0 * 16, osub * 16);
}

// Second stage of the factorised heads (NNetArchitecture.py:90-93,99-102, the Linear chains collapsed: they have no activation):
// policy logits from the 16 policy channels of every pixel, value logits from the 16 value channels.  One output subtile (16
// outputs) of 16 boards = FOUR MFMA accumulation chains, one per contiguous quarter of the k-steps (one wavefront each),
// summed as (q0 + q1) + (q2 + q3).  This is the full-width path behind NNetWrapper.process; the search loop computes only the
// logits it uses (sparse heads, azg_kernels.h).  `afrag(ks)` delivers the A operand (16 boards x 32 features from the global
// feature rows), wl this lane's weight fragments (`wstride` half8 from one k-step to the next).  The loads of a chain are issued
// in batches of HEADF_U k-steps, branch-free (k-steps past the end re-read the last one with the A fragment zeroed).
constexpr int HEADF_U = 7, HEADF_Q = 4, HEADF_NS = 5;            // k-steps per load batch, K quarters, subtiles per wavefront
// NS chains at once (they share the A fragments): acc[s] += sum over k-steps [k_begin, k_end) of afrag(ks) x wl[ks * wstride + soff[s]]
template <int NS, class AF>
__device__ __forceinline__ void heads_fact_chains(AF &&afrag, const half8 *wl, size_t wstride, const size_t (&soff)[NS], int k_begin, int k_end,
                                                  floatx4 (&acc)[NS]) {
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k0 = k_begin; k0 < k_end; k0 += HEADF_U) {
        half8 a[HEADF_U], b[HEADF_U][NS];
#pragma unroll
        for (int u = 0; u < HEADF_U; u++) {
            const int ks = k0 + u, kc = min(ks, k_end - 1);
            a[u] = afrag(kc);
            if (ks >= k_end) a[u] = zero8;
#pragma unroll
            for (int s = 0; s < NS; s++) b[u][s] = wl[(size_t)kc * wstride + soff[s]];
        }
        __builtin_amdgcn_sched_barrier(0);                      // (left alone hipcc sinks every load next to its MFMA and waits)
#pragma unroll
        for (int u = 0; u < HEADF_U; u++)
#pragma unroll
            for (int s = 0; s < NS; s++) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b[u][s], acc[s], 0, 0, 0);
    }
}
// the factorised heads' parameters: subtile s < osp = policy outputs s*16.. from the policy half of the features (weights wp
// [fk/32][osp][64]), subtile osp = the value outputs from the value half (weights wv [fk/32][64])
struct HeadsFact { const half8 *wp, *wv; const float *bias; int fk, osp, A, NV; };
// (the same chains for the persistent exact launch, heads_full_lds: the policy fragments subtile-major)
struct HeadsFull { const half8 *wps, *wv; const float *bias; };
__device__ __forceinline__ int heads_fact_kq(int ksteps) { return (ksteps + HEADF_Q - 1) / HEADF_Q; }      // k-steps per quarter

// workgroup = 16 boards x one chunk of HEADF_NS policy subtiles (or the value subtile), wave = K quarter
__global__ __launch_bounds__(HEADF_Q * 64) void k_heads_fact(const _Float16 *feat, HeadsFact hf, float *logits, int boards, int opad) {
    __shared__ floatx4 red[HEADF_Q][HEADF_NS][64];
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6, g = lane >> 4, i16 = lane & 15;
    const int ncp = (hf.osp + HEADF_NS - 1) / HEADF_NS, nchunks = ncp + 1, grp = blockIdx.x / nchunks, chunk = blockIdx.x - grp * nchunks;
    const int b0 = grp * 16, ksteps = hf.fk / 32, KQ = heads_fact_kq(ksteps);
    const bool is_v = chunk == ncp;
    const int s0 = is_v ? hf.osp : chunk * HEADF_NS, nsub = is_v ? 1 : min(HEADF_NS, hf.osp - s0);
    const half8 *frow = reinterpret_cast<const half8 *>(feat) + (size_t)min(b0 + i16, boards - 1) * ((size_t)hf.fk / 4) + g + (is_v ? hf.fk / 8 : 0);
    size_t soff[HEADF_NS];
#pragma unroll
    for (int s = 0; s < HEADF_NS; s++) soff[s] = (size_t)min(s, nsub - 1) * 64;     // (subtiles past the end re-read the last real one)
    floatx4 acc[HEADF_NS];
#pragma unroll
    for (int s = 0; s < HEADF_NS; s++) acc[s] = (floatx4){0.f, 0.f, 0.f, 0.f};
    heads_fact_chains<HEADF_NS>([&](int ks) { return frow[(size_t)ks * 4]; }, is_v ? hf.wv + lane : hf.wp + (size_t)s0 * 64 + lane,
                                is_v ? (size_t)64 : (size_t)hf.osp * 64, soff, min(kq * KQ, ksteps), min((kq + 1) * KQ, ksteps), acc);
#pragma unroll
    for (int s = 0; s < HEADF_NS; s++) red[kq][s][lane] = acc[s];
    __syncthreads();
    for (int s = kq; s < nsub; s += HEADF_Q) {                              // D[m = board g*4 + r][n = output i16]
        const floatx4 sum = (red[0][s][lane] + red[1][s][lane]) + (red[2][s][lane] + red[3][s][lane]);
        const int out = is_v ? hf.A + i16 : (s0 + s) * 16 + i16, lim = is_v ? hf.A + hf.NV : hf.A;
        if (out < lim) {
            const float bo = hf.bias[out];
#pragma unroll
            for (int r = 0; r < 4; r++) { const int board = b0 + g * 4 + r; if (board < boards) logits[(size_t)board * opad + out] = sum[r] + bo; }
        }
    }
}

// one wave per board: softmax over the A policy logits (held in registers: up to 16 per lane, A <= 1024) and the NV value logits
__global__ __launch_bounds__(256) void k_heads_softmax(const float *logits, float *policy, float *value, int boards, int opad, int A, int NV) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= boards) return;
    heads_softmax_row(logits + (size_t)b * opad, lane, A, NV, policy + (size_t)b * A, value + (size_t)b * NV);
}


// ------------------------------------------------------------------------------------------------ k_tower2
// Two workgroups per CU.  With the residual stream in registers only ONE LDS image is needed (t / u / final s take
// turns in it), 81 KB per workgroup, so two workgroups (two tiles) are resident per CU and one runs its epilogue /
// barriers while the other keeps the MFMA pipe busy.  To fit 2 waves per SIMD (<= 256 registers per wave) the B
// fragments are a short rolling window (PF reads ahead of their MFMAs) instead of a double-buffered k-step.
template <class GEO, int KS, int NSUB>
struct FragOff {                                             // LDS immediate of fragment t = kk * NSUB + ps
    static constexpr int get(int t) {
        const int kk = t / NSUB, tap = kk / KS, ks = kk % KS;
        return GEO::BIAS + ((tap / 3 - 1) * GEO::PW + (tap % 3 - 1)) * GEO::RSTRIDE + ks * 64;
    }
};

// weight ring depth: a k-step of a small tile (NSUB <= 4: 6-8 MFMAs) is far shorter than an L2 round trip, so the small
// shapes fetch 8 k-steps ahead; the big ones (22 MFMAs per k-step, registers scarce) one or two.  Every layer after the stem
// must advance the ring by whole turns (9 * KS k-steps), so that it always starts at slot STEM_KSTEPS % N (the stem's 3
// k-steps): N = 9 for small tiles, 2 for even KS, 3 for odd KS (32 channels).
template <int NSUB, int KS> struct WeightRing { static constexpr int N = NSUB <= 4 ? 9 : (KS % 2 ? 3 : 2); };

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {            // f(integral_constant<int, I>) for I in [0, N): indices usable as
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }   // immediates and in if constexpr
}

// The stem convolution.  Its input has 8 channels per pixel -- ONE 16-byte chunk of the row -- so a k-step of 32 holds FOUR taps:
// lane group g reads chunk 0 of the row of tap 4 * kk + g (weights packed to match, nnet.pack_stem_weight; the three slots past
// tap 8 carry zero weights and re-read the centre).  3 k-steps instead of 9 one-eighth-full ones.  The weight ring runs on as in
// conv_main2 (slot kk, prefetch WR - 1 k-steps ahead into the next layer).
constexpr int STEM_KSTEPS = 3;
// (k-split towers: a wave's main stream starts at wmain and advances KSTR k-steps of the packed order per step of its own; KSTR = 1,
//  wmain = wfrag + STEM_KSTEPS * WSTEP is the plain continuous stream)
template <class GEO, int KSTR>
__device__ __forceinline__ const half8 *stream_frag(const half8 *wfrag, const half8 *wmain, int step) {
    return step < STEM_KSTEPS ? wfrag + (size_t)step * GEO::WSTEP : wmain + (size_t)(step - STEM_KSTEPS) * (KSTR * GEO::WSTEP);
}
template <class GEO, int NSUB, int WR, int KSTR = 1>
__device__ __forceinline__ void conv_stem(const char *in, const unsigned (&lb)[NSUB], int g, const half8 *wfrag, const half8 *wmain,
                                          half8 (&a)[WR][2], floatx4 (&acc)[2][NSUB]) {
    unsigned soff[STEM_KSTEPS];
#pragma unroll
    for (int kk = 0; kk < STEM_KSTEPS; kk++) {
        const int t = 4 * kk + g, tap = t < 9 ? t : 4;
        soff[kk] = (unsigned)(GEO::BIAS + ((tap / 3 - 1) * GEO::PW + (tap % 3 - 1)) * GEO::RSTRIDE - g * 16);   // (lb carries + g * 16)
    }
    if constexpr (NSUB <= 4) {                                  // small tiles (one wave per SIMD, registers to spare): every read first
        half8 b[STEM_KSTEPS][NSUB];
#pragma unroll
        for (int kk = 0; kk < STEM_KSTEPS; kk++)
#pragma unroll
            for (int ps = 0; ps < NSUB; ps++) b[kk][ps] = *reinterpret_cast<const half8 *>(in + lb[ps] + soff[kk]);
#pragma unroll
        for (int kk = 0; kk < STEM_KSTEPS; kk++) {
            const int an = (kk + WR - 1) % WR, ac = kk % WR;
            { const half8 *f = stream_frag<GEO, KSTR>(wfrag, wmain, kk + WR - 1); a[an][0] = f[0]; a[an][1] = f[64]; }
#pragma unroll
            for (int ps = 0; ps < NSUB; ps++) {
                acc[0][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ac][0], b[kk][ps], acc[0][ps], 0, 0, 0);
                acc[1][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ac][1], b[kk][ps], acc[1][ps], 0, 0, 0);
            }
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < STEM_KSTEPS; kk++) {
            const int an = (kk + WR - 1) % WR, ac = kk % WR;
            { const half8 *f = stream_frag<GEO, KSTR>(wfrag, wmain, kk + WR - 1); a[an][0] = f[0]; a[an][1] = f[64]; }
            half8 b[NSUB];
#pragma unroll
            for (int ps = 0; ps < NSUB; ps++) b[ps] = *reinterpret_cast<const half8 *>(in + lb[ps] + soff[kk]);
#pragma unroll
            for (int ps = 0; ps < NSUB; ps++) {
                acc[0][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ac][0], b[ps], acc[0][ps], 0, 0, 0);
                acc[1][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ac][1], b[ps], acc[1][ps], 0, 0, 0);
            }
        }
    }
}

template <class GEO, int KS, int NSUB, bool SKIP>
struct TapPlan {                                             // which fragments t = kk * NSUB + ps the main loop touches
    static constexpr int NSTEP = 9 * KS, TOT = NSTEP * NSUB;
    static constexpr bool act(int t) { return t < TOT && (!SKIP || GEO::tap_active((t / NSUB) / KS, t % NSUB)); }
    static constexpr int mfmas(int kk) { int n = 0; for (int ps = 0; ps < NSUB; ps++) n += act(kk * NSUB + ps); return n; }
    static constexpr int reads(int kk, int pf) { int n = 0; for (int ps = 0; ps < NSUB; ps++) n += act(kk * NSUB + ps + pf); return n; }
};

// KSTR: see stream_frag.  DUAL: the k-steps of the second half of every tap's channels accumulate into accB (the caller adds the two
// sets at the end): the summation order of the k-split form, so that a board's outputs do not depend on the tile shape.
template <class GEO, int KS, int NSUB, int RB, int WR, bool SKIP, int KSTR = 1, bool DUAL = false>
__device__ __forceinline__ void conv_main2(const char *in, const unsigned (&lb)[NSUB], const half8 *wfrag, half8 (&a)[WR][2],
                                           floatx4 (&acc)[2][NSUB], floatx4 (&accB)[2][NSUB]) {
    // fragment t = (kk, ps) is read PF fragments ahead into a register ring.  Large tiles index the ring by subtile (slot
    // ps % RING: collision-free with PF = 4 for NSUB = 11 at RING = 6, checked case by case; one slot per subtile otherwise);
    // small tiles (NSUB <= 4, the low-latency shapes for small batches) by fragment number, RING = PF + 1.
    // SKIP: fragments whose subtile is wholly off-board for the tap (TowerGeom::tap_active) are neither read nor multiplied
    using TP = TapPlan<GEO, KS, NSUB, SKIP>;
    constexpr int NSTEP = 9 * KS, PF = NSUB <= 4 ? 6 : 4;                          // (small tiles run one wave per SIMD: nothing else
    constexpr bool TRING = NSUB <= 4;                                             //  hides the LDS latency; measured 78 -> 72 us at 256 boards)
    constexpr int RING = TRING ? PF + 1 : NSUB == 11 ? 6 : NSUB;
    using FO = FragOff<GEO, KS, NSUB>;
    half8 bb[RING];
    static_for<0, PF>([&](auto ti) __attribute__((always_inline)) {
        constexpr int t = decltype(ti)::value;
        if constexpr (TP::act(t)) bb[(TRING ? t : t % NSUB) % RING] = *reinterpret_cast<const half8 *>(in + lb[t % NSUB] + FO::get(t));
    });
    static_for<0, NSTEP>([&](auto ki) __attribute__((always_inline)) {
        constexpr int kk = decltype(ki)::value;
        constexpr int an = (RB + kk + WR - 1) % WR, ac = (RB + kk) % WR;
        a[an][0] = wfrag[(size_t)(kk + WR - 1) * (KSTR * GEO::WSTEP)]; a[an][1] = wfrag[(size_t)(kk + WR - 1) * (KSTR * GEO::WSTEP) + 64];
        static_for<0, NSUB>([&](auto pi) __attribute__((always_inline)) {
            constexpr int ps = decltype(pi)::value;
            constexpr int t = kk * NSUB + ps, psn = (ps + PF) % NSUB;
            constexpr int cur = (TRING ? t : ps) % RING, nxt = (TRING ? t + PF : psn) % RING;
            if constexpr (TP::act(t + PF)) bb[nxt] = *reinterpret_cast<const half8 *>(in + lb[psn] + FO::get(t + PF));
            if constexpr (TP::act(t)) {
                if constexpr (DUAL && (kk % KS) >= KS / 2) {
                    accB[0][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ac][0], bb[cur], accB[0][ps], 0, 0, 0);
                    accB[1][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ac][1], bb[cur], accB[1][ps], 0, 0, 0);
                } else {
                    acc[0][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ac][0], bb[cur], acc[0][ps], 0, 0, 0);
                    acc[1][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ac][1], bb[cur], acc[1][ps], 0, 0, 0);
                }
            }
        });
        constexpr int NR = TP::reads(kk, PF), NM = TP::mfmas(kk);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        static_for<0, NSUB>([&](auto pi) __attribute__((always_inline)) {
            constexpr int i = decltype(pi)::value;
            if constexpr (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if constexpr (i < NM) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
    });
}

// PSPLIT > 1 (narrow towers at small batches, where a workgroup of C/32 waves leaves SIMDs empty): the tile's pixel subtiles
// are dealt to PSPLIT wave groups, wave = (cout group cg, pixel group ph); every wave still streams its own cout slice.
//
// SEARCH = SearchArgs<G> turns the kernel into the whole simulation loop of its games (no launch, no HBM round trip per
// simulation): a workgroup owns BOARDS = C/32 games, wave w runs find_leaf for game w and writes the leaf observation straight
// into the LDS image, the workgroup evaluates its boards, wave w backs game w up from the probabilities left in LDS -- `sims`
// times.  Workgroups never synchronise with each other.  (Measured: the same throughput as three launches per simulation --
// the tower is bound by LDS-read issue and power, not by launch gaps; staggering the two workgroups of a CU changed nothing.)
struct NoSearch {};
template <class G> struct SearchArgs { View ev; int sims; using Game = G; static constexpr bool WIDE = false; static constexpr bool ARENA = false; };
// The same for the batched Arena (Arena.pyx:208-328; SelfPlayAgent.pyx arena mode :62-73,117-132): one game per workgroup -- arena shards are
// small (256 games per GPU), a one-board tile per CU is what fills the chip --, wave 0 walks the MOVER's tree (tree_of_slot), the
// workgroup evaluates the leaf with the MOVER's model: model = player_to_index[mover] (seat.v, SelfPlayAgent.pyx:44-47) or the slot's own
// seating (seat_of_slot: 4 bits per player), parameters of model m > 0 in TowerParams::alt[m - 1].  The mover -- hence tree and model --
// is fixed for the length of the launch (one move).
template <class G> struct SearchArena { View ev; int sims; SeatMap seat; const uint32_t *seat_of_slot; using Game = G; static constexpr bool WIDE = false; static constexpr bool ARENA = true; };
// The same for networks with factorised heads (wide action spaces, any tower width): BOARDS games per workgroup, wave b walks game
// b, wave BOARDS + b prepares its priors and shuffle (the two-wave scheme of k_backup_select2), the head convolutions leave their
// features in LDS and the next tree phase computes the logits it needs from them (azg_kernels.h, sparse heads: the value
// logits by the walker, the policy logits of the leaf's valid actions by the helper) -- the code of the launch-per-phase path.
// EXACT_: the launch computes ALL A + P + 1 logits of its boards from the head features (hf: the collapsed Linear chains in
// k_heads_fact's fragment order, heads_full_lds below) -- what NNetWrapper.process returns (NNetWrapper.py:225-232) -- and the tree
// phase takes softmax over all A, masks and renormalises like MCTS.pyx:239-245: bit-identical to the launch-per-phase form
// k_tower2 -> k_heads_fact -> k_backup_select2<IN_LOGITS>.  Otherwise the sparse heads (hd: one row per output, azg_kernels.h).
template <class G, int MINB = 1, bool EXACT_ = false> struct SearchWide {
    View ev; int sims; HeadRows hd; HeadsFull hf;
    using Game = G; static constexpr bool WIDE = true; static constexpr int MIN_BLOCKS = MINB; static constexpr bool EXACT = EXACT_;
};

// OVERLAPPED one-game tile (exact heads, four wavefronts, a wide policy head: brandubh up to 512 games per GPU).  The heads phase is a
// parameter stream through the CU's L1 miss path, not arithmetic, and only ONE wavefront of the next tree phase -- the helper -- needs
// the policy logits; the walker needs the value row.  So AZG_OVL_NW wavefronts (waves 1 .. NW) stream the head matrix, value subtile
// first, while wave 0 walks (backup_path, descent, expansion) and wave 3 runs the shuffle masks and the rules of the walk; the helper
// (wave 1) turns to the priors when the last policy subtile is in LDS.  No workgroup barrier between the head convolutions and the next
// tower: the wavefronts meet through LDS generation flags and a snapshot of the header (WideScratch::SNAP).  Same arithmetic, same
// results as the phase-by-phase form.  0 = off (the heads phase and the tree phase take turns).
// MEASURED AND NOT ADOPTED (profiles/r06_heads_waves_ab.txt): bit-identical results, 9-10 % slower at 512 brandubh games -- a streaming
// wavefront moves one subtile (25 KB, all the registers hold) per ~2.3 k cycles whether two or four wavefronts stream, so the stream
// needs all four; build.py --variant overlap (-DAZG_OVL_NW=2) builds it.
#ifndef AZG_OVL_NW
#define AZG_OVL_NW 0
#endif
// per-game LDS scratch of the wide search mode (behind the image)
// COMPACT (solo tree phase, several games per workgroup: LDS is what limits the games a CU holds): the softmax output and the masked
// policy of np.sum take turns in the logits' place -- every stage reads its input into registers before the next one writes (one
// wavefront, LDS accesses of a wave complete in order), the value logits sit behind the A policy slots and are never overwritten
template <class G, int HW, bool COMPACT = false> struct WideScratch {
    static constexpr int A = G::A, NV = G::P + 1, OPAD = (A + NV + 15) / 16 * 16, FK = (HW * 16 + 31) / 32 * 32;
    static_assert(!COMPACT || A >= 8, "compact scratch: the masked policy needs A slots");
    // (COMPACT drops what only the multi-wavefront tree phase uses -- the shuffle masks, the walk's mailbox -- and leaves the path in HBM:
    //  its reads and writes are one coalesced access per simulation each, off the dependent chain)
    static constexpr int LG = 0, PI = COMPACT ? LG : LG + OPAD * 4, M = COMPACT ? LG : PI + A * 4, SCR = COMPACT ? LG + OPAD * 4 : M + (A < 8 ? 8 : A) * 4, ACT = SCR + 256,
                         LESS = (ACT + ((G::MAXK + 63) / 64) * 256 + 15) / 16 * 16, FLAGS = LESS + (COMPACT ? 0 : 512), NFLAGS = (COMPACT || AZG_OVL_NW == 0) ? 4 : 8, FEAT = FLAGS + NFLAGS * 4,
                         // the game's tree state for the length of the launch (the workgroup owns the game): header, last path, tape
                         // counter + the two per-slot tallies, root state -- read and written in LDS, copied from / to HBM once
                         HDR = (FEAT + 2 * FK * 2 + 63) / 64 * 64, PATH = HDR + 64, MAXD = G::MAX_TURNS + 2, CTR = PATH + (COMPACT ? 0 : MAXD * 16),
                         STATE = CTR + 32, MAIL = (STATE + (int)sizeof(azg_state) + 15) / 16 * 16,                 // (WalkMail: azg_kernels.h)
                         // SNAP: the header and the tape counter as the walker LEFT them at the end of its tree phase (overlapped one-game
                         // tile, wide_overlap_nw: the other wavefronts of the game enter the next tree phase long after the walker has --
                         // the helper only when the walker may have FINISHED it, so two copies take turns: phase s writes copy s & 1)
                         SNAP = (MAIL + (COMPACT ? 0 : (int)sizeof(WalkMail)) + 15) / 16 * 16, SNAP_BYTES = 80,
                         BYTES = (SNAP + ((COMPACT || AZG_OVL_NW == 0) ? 0 : 2 * SNAP_BYTES) + 15) / 16 * 16;
};
// all of the wide search mode's LDS behind the image: the per-game scratch, an error word, and the value head's P + 1 weight rows
// + biases (the same for every game and simulation: fetched once per launch instead of once per simulation by every walker;
// COMPACT: they stay in global memory)
template <class G> struct WideMailFits { static_assert(G::MAX_TURNS + 2 <= 128, "WalkMail::act holds one action per level of a find_leaf path"); };
template <class G, int HW, int BOARDS, bool COMPACT = false> struct WideLds : WideMailFits<G> {
    static constexpr int NV = G::P + 1, FK = WideScratch<G, HW, COMPACT>::FK;
    // (ZERO: a feature row of zeros -- the A-operand rows of heads_full_lds that no board stands behind; zeroed once per launch)
    static constexpr int ERR = BOARDS * WideScratch<G, HW, COMPACT>::BYTES, ZERO = (ERR + 16 + 15) / 16 * 16, VROWS = ZERO + FK * 2, VBIAS = VROWS + (COMPACT ? 0 : NV * FK * 2),
                         BYTES = (VBIAS + (COMPACT ? 0 : NV * 4) + 15) / 16 * 16;
};

// The full-width heads of a tile's boards inside the persistent launch (EXACT): logits[board][o] for all A policy outputs and the
// P + 1 value outputs, from the boards' head features in LDS.  Output subtile s (16 outputs; s == OSP: the value outputs over the
// value half) is ONE wavefront's job: A operand = the boards' features (row m = board m; the lanes of rows past BOARDS read a zero
// row), B operand = the subtile's weight fragments streamed from L2, FOUR accumulation chains, one per contiguous quarter of the
// k-steps, summed as (q0 + q1) + (q2 + q3) + bias -- exactly k_heads_fact's arithmetic for a board (an MFMA row depends on its own
// A row only), so the logits are bit-identical to NNetWrapper.process's.  The chains are issued interleaved (independent
// accumulators); the fragments AND the bias of the wave's NEXT subtile are requested as the current ones are consumed, in the order
// they will be used, and the subtile loop is fully unrolled -- vmcnt retires in order, so every MFMA then waits for exactly its own
// fragment instead of draining the stream (a loop back-edge or a bias load at the point of use both made the compiler wait for
// vmcnt(0): 36 k cycles per evaluation; 28 k now -- 0.94 MB at ~33 B/clk, what one workgroup gets through the CU's L1 miss path with
// 100 KB in flight; a second subtile in flight does not fit the registers: profiles/r05_wide_tile_sweep.txt).
//   wps: the policy chain's fragments SUBTILE-major, [OSP][k-step][64 lanes] x 16 B (k_heads_fact's wp is k-step-major: a wave here
//   streams one subtile's 25 KB contiguously); wv [k-step][64]; bias f32 [A + P + 1].
// the fragments and the bias of a wavefront's FIRST subtile: they do not depend on the evaluation, so the launch requests them before
// the head convolutions and the barrier behind them (one L2 round trip of the stream hidden per simulation)
// consumption order of a subtile's k-steps: the four accumulation chains (one per contiguous quarter of the k-steps) interleaved
template <int KS, int KQ> struct HeadsOrder {
    int ks[KS], q[KS];
    constexpr HeadsOrder() : ks{}, q{} {
        int n = 0;
        for (int j = 0; j < KQ; j++)
            for (int qq = 0; qq < HEADF_Q; qq++) { const int k = qq * KQ + j; if (k < KS && k < (qq + 1) * KQ) { ks[n] = k; q[n] = qq; n++; } }
    }
};
#ifndef HEADS_A_RING
#define HEADS_A_RING 4
#endif
template <class G, int HW> struct HeadsFirst { half8 b[WideScratch<G, HW>::FK / 32]; float bias; };
// VFIRST (the overlapped one-game tile): the wavefronts take ITEMS instead of subtiles -- item 0 is the value subtile, item j > 0
// the policy subtile j - 1 -- so that the first thing streaming wavefront 0 finishes is the value row the walker is waiting for
template <int OSP, bool VFIRST> __device__ __forceinline__ int heads_item_subtile(int j) { return VFIRST ? (j == 0 ? OSP : min(j - 1, OSP)) : min(j, OSP); }
template <class G, int HW, int NW, bool VFIRST = false>
__device__ __forceinline__ void heads_full_prefetch(const HeadsFull &hf, int wave, int lane, HeadsFirst<G, HW> &pf) {
    constexpr int A = G::A, NV = G::P + 1, KS = WideScratch<G, HW>::FK / 32, KQ = (KS + HEADF_Q - 1) / HEADF_Q, OSP = (A + 15) / 16;
    const int i16 = lane & 15, s0 = heads_item_subtile<OSP, VFIRST>(wave);
    const half8 *w0 = s0 == OSP ? hf.wv + lane : hf.wps + (size_t)s0 * (KS * 64) + lane;
    pf.bias = hf.bias[min(s0 == OSP ? A + i16 : s0 * 16 + i16, A + NV - 1)];
#pragma unroll
    for (int j = 0; j < KQ; j++)
#pragma unroll
        for (int q = 0; q < HEADF_Q; q++) { const int ks = q * KQ + j; if (ks < KS && ks < (q + 1) * KQ) pf.b[ks] = w0[(size_t)ks * 64]; }
    // (nothing may sink these requests towards their uses: where the heads loop follows at once -- the multi-game tiles -- the scheduler had
    //  moved every one of them down to just in front of its MFMA, into ONE register quad: the first subtile of every evaluation waited for 25
    //  L2 round trips one after the other, `s_waitcnt vmcnt(0)` in front of each MFMA -- about 20 k of the phase's 25 k cycles)
    __builtin_amdgcn_sched_barrier(0);
}
template <class G, int HW, int BOARDS, int NW, bool COMPACT, bool VFIRST = false>
__device__ __forceinline__ void heads_full_lds(char *scr0, const char *zero_row, const HeadsFull &hf, int wave, int lane, HeadsFirst<G, HW> &pf,
                                               [[maybe_unused]] int *vflag = nullptr, [[maybe_unused]] int vgen = 0) {
    using WS = WideScratch<G, HW, COMPACT>;
    constexpr int A = G::A, NV = G::P + 1, FK = WS::FK, KS = FK / 32, KQ = (KS + HEADF_Q - 1) / HEADF_Q, OSP = (A + 15) / 16;
    constexpr int NIT = (OSP + 1 + NW - 1) / NW;                             // subtiles per wavefront (the last one may be a repeat)
    static_assert(BOARDS <= 16 && HEADF_Q == 4, "the tile's boards are the D rows; four K quarters");
    const int g = lane >> 4, i16 = lane & 15;
    const bool live = i16 < BOARDS;
    const char *fb = live ? scr0 + i16 * WS::BYTES + WS::FEAT + g * 16 : zero_row + g * 16;
    const int vhalf = live ? FK * 2 : 0;
    auto frags = [&](int s_) { return s_ == OSP ? hf.wv + lane : hf.wps + (size_t)s_ * (KS * 64) + lane; };
    // consumption order of the k-steps: the four chains interleaved
    constexpr HeadsOrder<KS, KQ> ord;
    half8 (&b)[KS] = pf.b;                                                   // (heads_full_prefetch)
    float bias_cur = pf.bias, bias_nxt = 0.f;
    // The A operand (the boards' features) comes out of LDS through a ring of AR registers that runs AR steps ahead of the MFMAs,
    // across subtile boundaries.  Left to the register allocator every step's ds_read_b128 landed in ONE register quad, issued behind
    // the previous MFMA and waited for with lgkmcnt(0) in front of the next: 25 exposed LDS round trips per subtile (~2.3 k cycles,
    // the same whether two or four wavefronts streamed -- what looked like the pace of the fragment stream was this).
    constexpr int AR = HEADS_A_RING;
    static_assert(AR >= 1 && AR <= KS, "A ring depth");
    half8 ar[AR];
    {
        const int s0 = heads_item_subtile<OSP, VFIRST>(wave);
        const char *f0 = fb + (s0 == OSP ? vhalf : 0);
#pragma unroll
        for (int p = 0; p < AR; p++) ar[p] = *reinterpret_cast<const half8 *>(f0 + ord.ks[p] * 64);
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        int s_ = wave + it * NW;                                             // (scalar; VFIRST: the item number)
        // (opaque: what an iteration derives from its subtile number -- fragment and feature addresses, the bias index -- is computed in
        //  the iteration, not hoisted for all of them in front of the unrolled loop, where it sat in registers across the whole phase)
        asm volatile("" : "+s"(s_));
        const int sc = heads_item_subtile<OSP, VFIRST>(s_), sn = heads_item_subtile<OSP, VFIRST>(s_ + NW);   // (past the end: the value subtile once more, not stored)
        const bool is_v = sc == OSP;
        const half8 *wn = frags(sn);
        const char *f = fb + (is_v ? vhalf : 0), *fn = fb + (sn == OSP ? vhalf : 0);
        if (it + 1 < NIT) bias_nxt = hf.bias[min(sn == OSP ? A + i16 : sn * 16 + i16, A + NV - 1)];
        floatx4 acc[HEADF_Q];
#pragma unroll
        for (int q = 0; q < HEADF_Q; q++) acc[q] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < KS; p++) {
            const int ks = ord.ks[p], q = ord.q[p];
            const int slot = (it * KS + p) % AR;                             // (compile-time: both loops are unrolled)
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[slot], b[ks], acc[q], 0, 0, 0);
            if (it + 1 < NIT) b[ks] = wn[(size_t)ks * 64];
            const bool refill = p + AR < KS || it + 1 < NIT;
            if (p + AR < KS) ar[slot] = *reinterpret_cast<const half8 *>(f + ord.ks[p + AR] * 64);
            else if (it + 1 < NIT) ar[slot] = *reinterpret_cast<const half8 *>(fn + ord.ks[p + AR - KS] * 64);
            // (pin the interleave: one fragment request and one A read behind every MFMA, in consumption order -- left alone the
            //  scheduler bunches the requests at the end of the subtile and the next one starts by draining them)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (it + 1 < NIT) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (refill) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const floatx4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        const int out = is_v ? A + i16 : sc * 16 + i16, lim = is_v ? A + NV : A;
        if (s_ <= OSP && 4 * g < BOARDS && out < lim) {                      // D[m = board 4 g + r][n = output i16]
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (4 * g + r < BOARDS) *reinterpret_cast<float *>(scr0 + (4 * g + r) * WS::BYTES + WS::LG + out * 4) = sum[r] + bias_cur;
        }
        if constexpr (VFIRST) {
            // the value row is complete: tell the walker.  A RELAXED store behind s_waitcnt lgkmcnt(0) (the row's ds_writes have executed;
            // the flag may go out as a flat store, which is not ordered with them otherwise) -- a release here would be s_waitcnt
            // vmcnt(0) too, i.e. drain the fragment stream this loop keeps in flight
            if (it == 0 && wave == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(vflag, vgen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
            }
        }
        bias_cur = bias_nxt;
    }
}

// (the wide search mode keeps one workgroup per CU busy for a whole move and mixes three phases with different register needs:
//  one wave per SIMD, the whole register file)
// SOLO tree phase (fewer than two wavefronts per game: several games per workgroup at large batches): wave b does ALL of game b's
// process_results and find_leaf itself -- policy logits, softmax, priors, value, path, walk with the rules, shuffle ranks --, the
// one-wave functions of k_select / k_backup in the order of k_backup_select2; waves past BOARDS only take part in the tower
template <int C, int PSPLIT, int KSPLIT, int BOARDS> constexpr bool wide_solo() { return (C / 32) * PSPLIT * KSPLIT < 2 * BOARDS; }
template <class SEARCH, int BOARDS, int NWAVES> constexpr int wide_overlap_nw() {
    if constexpr (__is_same(SEARCH, NoSearch)) return 0;
    else if constexpr (!SEARCH::WIDE) return 0;
    else return (SEARCH::EXACT && BOARDS == 1 && NWAVES == 4 && SEARCH::Game::A > 64) ? AZG_OVL_NW : 0;
}
template <class SEARCH> __device__ __forceinline__ int sa_error_word(const SEARCH &sa) {
    if constexpr (__is_same(SEARCH, NoSearch)) return 0; else return sa.ev.gcount[GC_ERROR];
}
template <class SEARCH, class = void> struct WideGameOf { using type = C4; };
template <class SEARCH> struct WideGameOf<SEARCH, std::void_t<typename SEARCH::Game>> { using type = typename SEARCH::Game; };
// (-DAZG_HEADLINE_ONE_WG, build.py --variant headline1: the experiment that settles "spill into the unused AGPRs" for the headline kernel --
//  on gfx950 a wavefront's VGPRs and AGPRs come out of ONE 512-entry file per SIMD lane, so at two wavefronts per SIMD a wave owns 256
//  registers IN TOTAL: "AGPR 0" in the resource table means all 256 are architectural, not that 256 more lie idle.  More registers per wave
//  = one workgroup per CU; profiles/r06_headline_one_wg_ab.txt has what that costs.)
#ifdef AZG_HEADLINE_ONE_WG
#define AZG_SEARCH_MIN_BLOCKS 1
#else
#define AZG_SEARCH_MIN_BLOCKS 2
#endif
template <class SEARCH> constexpr int tower_min_blocks() { if constexpr (__is_same(SEARCH, NoSearch)) return 2; else { if constexpr (SEARCH::WIDE) return SEARCH::MIN_BLOCKS; else return AZG_SEARCH_MIN_BLOCKS; } }

// KSPLIT = 2 (64-channel towers of one board): the two 32-channel k-steps of every tap go to two wave groups -- wave = (cout group
// cg, pixel group ph, k group kg), every wave runs 9 k-steps over ALL its pixel group's subtiles and finishes half of them: the
// partial sums of the other half go to the partner through LDS.  No weight fragment is fetched twice (unlike a pixel split), every
// SIMD gets a wave of each of the CU's two workgroups, and a wave's load issue hides under the other's MFMAs.
// KHALF order (7x7 x 64 channels, every tile shape): out = (bias + sum over the first channel half) + (sum over the second half).
template <int H, int W, int C> constexpr bool tower_khalf_order() { return H == 7 && W == 7 && C == 64; }
// One-board tiles keep the layers' biases and the blocks' pre-activation affines in LDS (behind everything else): a tile with one wave
// per SIMD has nothing to cover the L2 round trip of a parameter fetch (~270 cycles, in every layer, and the persistent launches would
// repeat it in every simulation).  fp32 biases [2 * nblocks + 1][C], then the affine as the fp16 values the epilogue uses, scale
// [nblocks][C] and shift [nblocks][C]: sized by the launch (dynamic LDS), (12 * nblocks + 8) * C bytes.
template <int C, int BOARDS> constexpr int tower_layer_param_bytes(int nblocks) { return ((2 * nblocks + 1) * C * 4 + 2 * nblocks * C * 2 + C * 4 + 15) / 16 * 16; }   // (+ one row of slack: the bias prefetch of the last layer)
// (+ the 1x1 head convolutions of the factorised heads: C / 32 k-steps x 2 fragments of 64 lanes x 16 bytes, then their 32 biases)
// (tiles of several games of the persistent wide launch, SOLO: the layers' parameters only -- what LDS the games leave holds them but not
//  the head fragments too; measured neutral while a neighbour workgroup covers the fetches, kept for the workgroup that is alone on its CU)
template <int C, int BOARDS, bool SOLO = false> constexpr int tower_param_bytes(int nblocks) {
    return BOARDS == 1 ? tower_layer_param_bytes<C, BOARDS>(nblocks) + (C / 32) * 2 * 1024 + 128 : SOLO ? tower_layer_param_bytes<C, BOARDS>(nblocks) : 0;
}
template <int H, int W, int BOARDS, int C, int PSPLIT = 1, class SEARCH = NoSearch, int KSPLIT = 1>
__global__ __launch_bounds__(C * 2 * PSPLIT * KSPLIT, tower_min_blocks<SEARCH>()) void k_tower2(TowerParams Pin, const int16_t *pixmap, SEARCH sa) {
    using GEO = TowerGeom<H, W, BOARDS, C>;
    constexpr bool IS_SEARCH = !__is_same(SEARCH, NoSearch);
    constexpr bool IS_WIDE = []() { if constexpr (IS_SEARCH) return SEARCH::WIDE; else return false; }();
    static_assert(!IS_SEARCH || IS_WIDE || (PSPLIT == 1 && C == 128 && BOARDS <= C / 32), "search mode: one wave per game, fused heads");
    constexpr bool IS_ARENA = []() { if constexpr (IS_SEARCH) { if constexpr (!SEARCH::WIDE) return SEARCH::ARENA; } return false; }();
    static_assert(!IS_ARENA || BOARDS == 1, "arena search: one game (one mover, one model) per workgroup");
    static_assert(!IS_WIDE || (C / 32) * PSPLIT * KSPLIT >= BOARDS, "wide search mode: at least one wavefront per game");
    constexpr bool SOLO = IS_WIDE && wide_solo<C, PSPLIT, KSPLIT, BOARDS>();
    constexpr bool EXACT = []() { if constexpr (IS_WIDE) return SEARCH::EXACT; else return false; }();
    constexpr int OVL_NW = wide_overlap_nw<SEARCH, BOARDS, C * 2 * PSPLIT * KSPLIT / 64>();
    constexpr bool OVL = OVL_NW > 0;
    static_assert(OVL_NW == 0 || OVL_NW == 2, "overlapped tile: waves 1 and 2 stream, wave 3 follows the walk");
    TowerParams P = Pin;
    constexpr int NT = C * 2 * PSPLIT * KSPLIT, KS = C / 32, CPR = C / 8;   // threads, k-steps per tap, 16-B chunks per row
    constexpr bool KHALF = tower_khalf_order<H, W, C>();
    constexpr int XCHG_OFF = []() {                              // the k-split exchange area: behind the image and the wide search mode's scratch
        if constexpr (IS_WIDE) return GEO::TILE + WideLds<typename SEARCH::Game, GEO::HW, BOARDS, wide_solo<C, PSPLIT, KSPLIT, BOARDS>()>::BYTES; else return GEO::TILE;
    }();
    static_assert(KSPLIT == 1 || (KSPLIT == 2 && KS == 2 && KHALF), "k-split: 64-channel towers in k-half order");
    constexpr int NSUBT = GEO::NSUB, NSUB = (NSUBT + PSPLIT - 1) / PSPLIT;   // subtiles of the tile / of one wave
    constexpr bool PLDS = BOARDS == 1 || SOLO;                   // the layers' parameters in LDS (see tower_param_bytes)
    constexpr bool PLDS_H1 = BOARDS == 1;                        // ... and the 1x1 head convolutions' fragments
    constexpr int PARAM_OFF = XCHG_OFF + (KSPLIT == 2 ? (C / 32 * PSPLIT * 2) * NSUB * 1024 : 0);
    constexpr bool TAPSKIP = PSPLIT == 1 && GEO::CLASSES > 1;   // (a wave's subtile numbers must be compile-time constants)
    constexpr int HW = GEO::HW, ROWS = GEO::ROWS, TILE = GEO::TILE, RS = GEO::RSTRIDE;
    // LDS scratch in the zero rows above board 0 (restored to zero after use): [0, 4096) heads reduction, then 256 B of
    // probabilities + a flag word (search mode) -- all inside the pad line, which ends at (LEAD + PW) * RS
    constexpr int SCRATCH_PV = 4096;
    static_assert(!IS_SEARCH || IS_WIDE || SCRATCH_PV + 1024 <= (GEO::LEAD + GEO::PW) * RS, "scratch must stay inside the pad rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *img = smem;
    // (one-board tiles, one or two waves per SIMD and issue bound: the wave number in an SGPR -- everything derived from it, the cout /
    //  pixel / k group and the wave's slice of the weight stream, is then scalar: fewer VALU instructions and registers around the MFMAs)
    const int tid = threadIdx.x, lane = tid & 63, wave = (BOARDS == 1 || IS_WIDE) ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6, g = lane >> 4, i16 = lane & 15, lane_k = lane;
    const int cg = wave % (C / 32), ph = (wave / (C / 32)) % PSPLIT, kg = wave / ((C / 32) * PSPLIT);
    int ntiles = (Pin.boards + BOARDS - 1) / BOARDS;
    if (Pin.rows_per_model) {
        ntiles = 0;
        for (int m = 0; m < Pin.nmodels; m++) ntiles += (min(Pin.rows_per_model[m], Pin.boards) + BOARDS - 1) / BOARDS;
    }
    for (int c = tid; c < TILE / 16; c += NT) reinterpret_cast<uint4 *>(smem)[c] = make_uint4(0, 0, 0, 0);   // pads stay 0
    if constexpr (IS_WIDE) {                                     // (the feature rows' padding must read as zero)
        using WL = WideLds<typename SEARCH::Game, HW, BOARDS, SOLO>;
        for (int c = tid; c < WL::VROWS / 16; c += NT) reinterpret_cast<uint4 *>(smem + TILE)[c] = make_uint4(0, 0, 0, 0);
        constexpr int A_ = SEARCH::Game::A;
        if constexpr (!SOLO && !EXACT) {
            const uint4 *vsrc = reinterpret_cast<const uint4 *>(sa.hd.rows + (size_t)A_ * sa.hd.fk);      // rows A .. A + P of the head matrix
            for (int c = tid; c < WL::NV * WL::FK * 2 / 16; c += NT) reinterpret_cast<uint4 *>(smem + TILE + WL::VROWS)[c] = vsrc[c];
            if (tid < WL::NV) reinterpret_cast<float *>(smem + TILE + WL::VBIAS)[tid] = sa.hd.bias[A_ + tid];
        }
    }
    unsigned lb[NSUB];
    unsigned livemask = 0;
    const int ecol = (g & 1) ? (2 * cg + 1) * 16 + (g - 1) * 4 : (2 * cg) * 16 + g * 4;
#pragma unroll
    for (int ps = 0; ps < NSUB; ps++) {
        const int gs = ph * NSUB + ps;                          // the tile's subtile this wave's ps-th one is
        const int p = gs < NSUBT ? pixmap[min(gs, NSUBT - 1) * 16 + i16] : -1;
        if (p >= 0) livemask |= 1u << ps;
        const int q = p >= 0 ? GEO::qrow(p) : GEO::LEAD;
        lb[ps] = (unsigned)(q * RS + g * 16 - GEO::BIAS);
    }
    const unsigned edelta = (unsigned)(GEO::BIAS - g * 16 + ecol * 2);      // epilogue cell of a pixel = fragment base + edelta
    // (the four-wave search launch: two waves per SIMD cover each other's weight latency, and its registers are the scarce resource --
    //  a ring of 3 k-steps instead of 9: 82 -> 48 spilled registers, launch 5.47 -> 5.38 ms)
    // (the same for the 2-game tiles, whose small pixel groups would otherwise take the deep ring of the one-wave-per-SIMD shapes: 154 -> 56
    //  spilled registers, brandubh 1024 games 11.2 -> 9.95 ms per move; the 3- and 4-game tiles keep their ring of 2: a third slot costs the
    //  sparse 4-game launch 33 %)
    constexpr int WR = (IS_WIDE && (KSPLIT == 2 || (BOARDS == 2 && tower_min_blocks<SEARCH>() == 2))) ? 3 : WeightRing<NSUB, C / 32>::N;
    static_assert((9 * (C / 32)) % WR == 0, "a layer must advance the weight ring by whole turns");
    const unsigned slot_role = __builtin_amdgcn_s_getreg(63492) & 1;       // HW_ID.wave_id parity: the two waves of a SIMD differ
    half8 a[WR][2];
    floatx4 acc[2][NSUB];
    constexpr int NOWN = NSUB / KSPLIT;                          // subtiles whose sums this wave finishes (k-split: ps in [kg * NOWN, + NOWN))
    static_assert(NSUB % KSPLIT == 0, "k-split: an even number of subtiles per wave");
    half2v sreg[NOWN][4];
    unsigned lbo[NOWN];                                          // their image rows, their live bits
    unsigned liveown = 0;
#pragma unroll
    for (int j = 0; j < NOWN; j++) {
        lbo[j] = lb[j];
        liveown |= ((livemask >> j) & 1u) << j;
        if constexpr (KSPLIT == 2) { if (kg) { lbo[j] = lb[j + NOWN]; liveown = (liveown & ~(1u << j)) | (((livemask >> (j + NOWN)) & 1u) << j); } }
    }
    __syncthreads();

#ifdef AZG_TOWER_TIMING
    if (P.dbg && tid == 0) {
        unsigned long long *d = P.dbg + 2048 + (size_t)blockIdx.x * 8;
        d[0] = __builtin_amdgcn_s_memtime(); d[2] = __builtin_amdgcn_s_getreg(63492); d[3] = __builtin_amdgcn_s_getreg(63508);
    }
#endif
    for (int gtile = blockIdx.x; gtile < ntiles; gtile += gridDim.x) {
        int tile = gtile;
        if (Pin.rows_per_model) {                                // which model's slice, which tile of it
            int m = 0, first = 0;
            for (; m + 1 < Pin.nmodels; m++) {
                const int n = min(Pin.rows_per_model[m], Pin.boards), t = (n + BOARDS - 1) / BOARDS;
                if (tile < t) break;
                tile -= t; first += n;
            }
            P.boards = min(Pin.rows_per_model[m], Pin.boards);
            P.x = reinterpret_cast<const uint4 *>(Pin.x) + (size_t)first * HW;
            if (Pin.y) P.y = reinterpret_cast<char *>(Pin.y) + (size_t)first * HW * C * 2;
            if (Pin.policy) { P.policy = Pin.policy + (size_t)first * Pin.A; P.value = Pin.value + (size_t)first * Pin.NV; }
            if (m > 0) {
                const TowerParams::Model &M = Pin.alt[m - 1];
                P.w = M.w; P.bias = M.bias; P.pre_scale = M.pre_scale; P.pre_shift = M.pre_shift; P.head_w = M.head_w; P.head_b = M.head_b;
            } else {
                P.w = Pin.w; P.bias = Pin.bias; P.pre_scale = Pin.pre_scale; P.pre_shift = Pin.pre_shift; P.head_w = Pin.head_w; P.head_b = Pin.head_b;
            }
        }
        if constexpr (IS_ARENA) {                                // the mover's model evaluates this game for the whole move
            const int sl = min(gtile, sa.ev.B - 1);
            const int mover = __builtin_amdgcn_readfirstlane((int)sa.ev.states[sl].player);
            const int m = __builtin_amdgcn_readfirstlane(sa.seat_of_slot ? (int)((sa.seat_of_slot[sl] >> (4 * mover)) & 15u) : sa.seat.v[mover & 7]);
            if (m > 0 && m < Pin.nmodels) {
                const TowerParams::Model &M = Pin.alt[m - 1];
                P.w = M.w; P.bias = M.bias; P.pre_scale = M.pre_scale; P.pre_shift = M.pre_shift; P.head_w = M.head_w; P.head_b = M.head_b;
            } else {
                P.w = Pin.w; P.bias = Pin.bias; P.pre_scale = Pin.pre_scale; P.pre_shift = Pin.pre_shift; P.head_w = Pin.head_w; P.head_b = Pin.head_b;
            }
        }
        const half8 *wl = reinterpret_cast<const half8 *>(P.w) + (size_t)(2 * cg) * 64 + lane;
        const half8 *wm0 = wl + (size_t)(STEM_KSTEPS + (KSPLIT == 2 ? kg : 0)) * GEO::WSTEP;      // the wave's main stream (see stream_frag)
        const int row0 = tile * ROWS;
        const int rows_here = min(ROWS, P.boards * HW - row0);
        int nsims = 1;
        if constexpr (IS_SEARCH) nsims = sa.sims + (IS_WIDE ? 1 : 0);  // (wide: the last iteration is the last backup, no tower)
        [[maybe_unused]] const int sc_off = PARAM_OFF + (2 * P.nblocks + 1) * C * 4, sh_off = sc_off + P.nblocks * C * 2;
        [[maybe_unused]] const int h1_off = PARAM_OFF + tower_layer_param_bytes<C, BOARDS>(P.nblocks);
        if constexpr (PLDS) {                                    // this tile's model: biases and affines -> LDS (read after the barrier that precedes the layers)
            float *pb_ = reinterpret_cast<float *>(smem + PARAM_OFF);
            _Float16 *psc_ = reinterpret_cast<_Float16 *>(smem + sc_off), *psh_ = reinterpret_cast<_Float16 *>(smem + sh_off);
            for (int c = tid; c < (2 * P.nblocks + 1) * C; c += NT) pb_[c] = P.bias[c];
            for (int c = tid; c < P.nblocks * C; c += NT) { psc_[c] = (_Float16)P.pre_scale[c]; psh_[c] = (_Float16)P.pre_shift[c]; }
            if (PLDS_H1 && P.head_w == nullptr && P.head1_w != nullptr) {
                uint4 *h1_ = reinterpret_cast<uint4 *>(smem + h1_off);
                for (int c = tid; c < KS * 2 * 64; c += NT) h1_[c] = reinterpret_cast<const uint4 *>(P.head1_w)[c];
                if (tid < 32) reinterpret_cast<float *>(smem + h1_off + KS * 2 * 1024)[tid] = P.head1_b[tid];
            }
        }
        [[maybe_unused]] bool lds_live = false;                  // wide search: the games' tree state is in LDS (written back after the loop)
        for (int sim = 0; sim < nsims; sim++) {
        // (wide search mode: an opaque copy of the lane id, new in every simulation.  Nothing a phase derives from it is invariant in
        //  the simulation loop, so LLVM cannot hoist one phase's per-lane constants out of the loop, where they would stay live through
        //  the OTHER phases and add to their registers instead of sharing them -- the launch has to fit two waves per SIMD)
        int lane_sim = lane_k;
        if constexpr (IS_WIDE) asm volatile("" : "+v"(lane_sim));
        const int lane = lane_sim, tid = wave * 64 + lane, g = lane >> 4, i16 = lane & 15;
#ifdef AZG_TOWER_TIMING
        unsigned long long wt_[6] = {0, 0, 0, 0, 0, 0};
#define AZG_WPHASE(i) do { if (IS_WIDE) wt_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AZG_WPHASE(i) do { } while (0)
#endif
        AZG_WPHASE(0);
        if constexpr (IS_WIDE) {
            // ---- tree phase of the wide search mode: process_results of simulation sim - 1 and find_leaf of simulation sim of the
            // workgroup's games, two wavefronts per game exactly like k_backup_select2 (walker + helper, hand-offs through LDS
            // generation flags); the logits of simulation sim - 1 are in LDS, the leaf planes go straight into the image
            using G = typename SEARCH::Game;
            using WS = WideScratch<G, HW, SOLO>;
            static_assert(G::CELLS == HW, "game / tower geometry mismatch");
            constexpr int A = G::A, NV = G::P + 1;
            const int bd = wave % BOARDS, role = wave / BOARDS;      // 0: walks the tree of game bd, 1: its helper, else idle
            char *ws = smem + TILE + bd * WS::BYTES;
            float *lg = reinterpret_cast<float *>(ws + WS::LG);
            int *flags = reinterpret_cast<int *>(ws + WS::FLAGS);
            int *errw = reinterpret_cast<int *>(smem + TILE + WideLds<G, HW, BOARDS, SOLO>::ERR);
            if (sim == 0 && tid < WS::NFLAGS * BOARDS) reinterpret_cast<int *>(smem + TILE + (tid / WS::NFLAGS) * WS::BYTES + WS::FLAGS)[tid % WS::NFLAGS] = 0;
            // (overlapped tile: the wavefronts reach this point at different times -- later checks sit behind the head convolutions' barrier)
            if (OVL ? sim == 0 : (sim & 15) == 0) {                  // sticky device error: stop, uniformly over the workgroup
                if (tid == 0) *errw = sa.ev.gcount[GC_ERROR];
                __syncthreads();
                const int err = *errw;
                if (err) break;
            }
            int slot = tile * BOARDS + bd;
            asm volatile("" : "+v"(slot));                       // (opaque: nothing of the trees is hoisted out of the simulation loop)
            slot = __builtin_amdgcn_readfirstlane(slot);
            // (k-split workgroups have wavefronts to spare in the tree phase: the third one takes the shuffle masks off the helper)
            constexpr bool MASK_WAVE = NT / 64 >= 3 * BOARDS;
            // (and the fourth one runs the game rules one level behind the walk: WalkMail, azg_kernels.h)
            constexpr bool RULES_WAVE = NT / 64 >= 4 * BOARDS;
            const bool livegame = slot < sa.ev.B && role < (SOLO ? 1 : RULES_WAVE ? 4 : MASK_WAVE ? 3 : 2);
            [[maybe_unused]] WalkMail *mail = reinterpret_cast<WalkMail *>(ws + WS::MAIL);
            const int tree = slot;                               // (self-play engines only: one tree per slot)
            // The tree functions reach the header, the path, the tape counter, the tallies and the root state through the View's
            // pointers: here those point into the game's LDS scratch (offset so that [tree] / [slot] lands on it), so that every
            // simulation's header / path / counter reads are LDS reads instead of a chain of three or four L2 round trips (4 k
            // of the walker's 26 k cycles) and the tallies' read-modify-writes never wait for HBM.  Node blocks stay in HBM.
            View evl = sa.ev;
            evl.hdr = reinterpret_cast<TreeHdr *>(ws + WS::HDR) - tree;
            if constexpr (!SOLO) evl.path = reinterpret_cast<PathEnt *>(ws + WS::PATH) - (size_t)tree * sa.ev.maxd;
            evl.tape_ctr = reinterpret_cast<uint64_t *>(ws + WS::CTR) - slot;
            evl.slot_sims = reinterpret_cast<int64_t *>(ws + WS::CTR + 8) - slot;
            evl.slot_exp = reinterpret_cast<int64_t *>(ws + WS::CTR + 16) - slot;
            evl.states = reinterpret_cast<azg_state *>(ws + WS::STATE) - slot;
            if (sim == 0) {                                      // HBM -> LDS, once per launch
                if (RULES_WAVE && role == 0 && lane < 8) reinterpret_cast<int *>(mail)[lane] = 0;
                if (livegame && role == 0) {
                    if (lane < 4) reinterpret_cast<uint4 *>(ws + WS::HDR)[lane] = reinterpret_cast<const uint4 *>(sa.ev.hdr + tree)[lane];
                    if (lane < (int)sizeof(azg_state) / 16) reinterpret_cast<uint4 *>(ws + WS::STATE)[lane] = reinterpret_cast<const uint4 *>(sa.ev.states + slot)[lane];
                    if (lane == 0) {
                        *reinterpret_cast<uint64_t *>(ws + WS::CTR) = sa.ev.tape_ctr[slot];
                        *reinterpret_cast<int64_t *>(ws + WS::CTR + 8) = sa.ev.slot_sims[slot];
                        *reinterpret_cast<int64_t *>(ws + WS::CTR + 16) = sa.ev.slot_exp[slot];
                    }
                }
                lds_live = true;
                __syncthreads();
            }
            HdrR hr; uint64_t ctr0 = 0;
            if constexpr (OVL) {
                // the walker reads the live header (it is its only writer); the others the snapshot the walker took when its last tree
                // phase ended -- by the time they get here the walker may be several levels into THIS phase
                if (livegame) {
                    const bool snap = sim > 0 && role != 0;
                    const char *sp = ws + WS::SNAP + ((sim - 1) & 1) * WS::SNAP_BYTES;
                    load_hdr(snap ? reinterpret_cast<const TreeHdr *>(sp) : evl.hdr + tree, hr);
                    ctr0 = snap ? *reinterpret_cast<const uint64_t *>(sp + 64) : evl.tape_ctr[slot];
                }
                if (sim == 0) __syncthreads();
            } else {
            if (livegame) { load_hdr(evl.hdr + tree, hr); ctr0 = evl.tape_ctr[slot]; }
            __syncthreads();                                     // both wavefronts of a game hold the header as the last launch / phase left it
            }
            const bool has_policy = livegame && sim > 0 && !hr.leaf_e && hr.leaf_fc >= 0;
            const bool root_noise = has_policy && hr.leaf == LEAF_IS_ROOT && sa.ev.add_noise;
            auto sink = [&](const typename G::S &ls, int ln) {       // leaf observation -> the image rows of board bd (32 stem channels)
                if (ln < HW) {
                    char *row = img + GEO::qrow(bd * HW + ln) * RS;
                    *reinterpret_cast<half8 *>(row) = G::obs8(ls, ln);
                    const uint4 z = make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4 *>(row + 16) = z; *reinterpret_cast<uint4 *>(row + 32) = z; *reinterpret_cast<uint4 *>(row + 48) = z;
                }
            };
            int *act = reinterpret_cast<int *>(ws + WS::ACT);
            if (livegame && role == 0) {
                if (sim < sa.sims) AZG_TSTAMP(evl, slot, lane, 8);   // (tree-timing builds: tools/wide_walker_stamps.py)
                typename G::S st = G::load(&evl.states[slot], lane);
                if (sim == 0) {
                    select_tree<G>(evl, slot, tree, hr, st, ctr0, lane, act, sink, NoGate{}, NoRanks{});
                } else {
                    Node *nodes = tree_nodes(sa.ev, tree, hr.base);
                    if constexpr (SOLO) {                            // the helper's part first: the walk may enter the previous leaf
                        if (has_policy) {
                            float *pi = reinterpret_cast<float *>(ws + WS::PI);
                            if constexpr (!EXACT) {                  // (EXACT: heads_full_lds left all A logits there)
                                leaf_policy_logits<G, false>(sa.hd, nodes, hr.leaf_fc, hr.leaf_k, reinterpret_cast<const _Float16 *>(ws + WS::FEAT), lg, lane);
                                wave_sync();
                            }
                            policy_softmax_row<A>(lg, lane, A, pi);
                            wave_sync();
                            backup_policy<G>(evl, slot, hr, nodes, pi, reinterpret_cast<float *>(ws + WS::M), reinterpret_cast<float *>(ws + WS::SCR), lane);
                            wave_sync();
                        }
                    }
                    float val[NV];
                    float pv = 0.f;
                    if (!hr.leaf_e) {                                // (a terminal leaf backs its win state up, not the network)
                        using WL = WideLds<G, HW, BOARDS, SOLO>;
                        HeadRows hv = sa.hd;                         // the value rows and biases out of LDS (offset so that row A + r lands on them)
                        if constexpr (!SOLO) {
                            hv.rows = reinterpret_cast<const _Float16 *>(smem + TILE + WL::VROWS) - (size_t)A * sa.hd.fk;
                            hv.bias = reinterpret_cast<const float *>(smem + TILE + WL::VBIAS) - A;
                        }
                        if constexpr (!EXACT) {
                            leaf_value_logits<G>(hv, reinterpret_cast<const _Float16 *>(ws + WS::FEAT) + sa.hd.fk, lg + A, lane);
                            wave_sync();
                        }
                        if constexpr (OVL) flag_wait_gen(sa.ev, &flags[3], sim);     // (the streaming wavefront's first subtile)
                        pv = value_softmax(lg + A, lane, NV);
                    }
#pragma unroll
                    for (int j = 0; j < NV; j++) val[j] = rl(pv, j);
                    const int prev_leaf = hr.leaf;
                    if (sim < sa.sims) AZG_TSTAMP(evl, slot, lane, 1);
                    backup_path<G>(evl, slot, tree, hr, nodes, val, lane);
                    if (sim < sa.sims) AZG_TSTAMP(evl, slot, lane, 3);
                    if (sim < sa.sims) {
                        wave_sync();
                        bool waited = false;
                        const unsigned long long *less = reinterpret_cast<const unsigned long long *>(ws + WS::LESS);
                        if constexpr (SOLO) {                        // (the root noise of the backup above drew one tape number)
                            select_tree<G>(evl, slot, tree, hr, st, ctr0 + (root_noise ? 1 : 0), lane, act, sink, NoGate{}, NoRanks{});
                        } else
                        select_tree<G>(evl, slot, tree, hr, st, ctr0, lane, act, sink, [&](int node) {
                            if (waited || node != prev_leaf) return false;
                            AZG_TSTAMP(evl, slot, lane, 9);
                            flag_wait_gen(sa.ev, &flags[0], sim); waited = true;
                            AZG_TSTAMP(evl, slot, lane, 0);
                            return root_noise;
                        }, [&](int k, int ln, int &pos) {
                            if (k > 64 || sa.ev.perm_tape) return false;
                            AZG_TSTAMP(evl, slot, lane, 2);
                            flag_wait_gen(sa.ev, &flags[1], sim);
                            AZG_TSTAMP(evl, slot, lane, 15);
                            pos = __popcll(less[ln] & (k == 64 ? ~0ULL : ((1ULL << k) - 1ULL)));
                            return true;
                        }, RULES_WAVE ? mail : nullptr, sim);
                    }
                }
                if constexpr (OVL) {                                 // the header and the counter as this tree phase leaves them
                    wave_sync();
                    char *sp = ws + WS::SNAP + (sim & 1) * WS::SNAP_BYTES;
                    if (lane < 4) reinterpret_cast<uint4 *>(sp)[lane] = reinterpret_cast<const uint4 *>(ws + WS::HDR)[lane];
                    if (lane == 0) *reinterpret_cast<uint64_t *>(sp + 64) = *reinterpret_cast<const uint64_t *>(ws + WS::CTR);
                }
            } else if (RULES_WAVE && livegame && role == 3) {        // the rules of the walk: play_action per level, then the leaf
                if (sim > 0 && sim < sa.sims) {
                    if constexpr (OVL) {                             // (overlapped tile: the mask wavefront is streaming the policy head)
                        reinterpret_cast<unsigned long long *>(ws + WS::LESS)[lane] = shuffle_less_mask<(G::MAXK < 64 ? G::MAXK : 64)>(sa.ev, slot, ctr0 + (root_noise ? 1 : 0), lane);
                        flag_set_gen(&flags[1], sim, lane);
                    }
                    follow_tree<G>(evl, slot, G::load(&evl.states[slot], lane), mail, sim, lane, act, sink);
                }
            } else if (MASK_WAVE && livegame && role == 2) {         // the shuffle of the next expansion (every expansion waits for it)
                if (!OVL && sim > 0 && sim < sa.sims) {
                    reinterpret_cast<unsigned long long *>(ws + WS::LESS)[lane] = shuffle_less_mask<(G::MAXK < 64 ? G::MAXK : 64)>(sa.ev, slot, ctr0 + (root_noise ? 1 : 0), lane);
                    flag_set_gen(&flags[1], sim, lane);
                }
            } else if (livegame && sim > 0) {                        // the helper: shuffle of the next expansion, priors of the previous leaf
#ifdef AZG_TOWER_TIMING
                unsigned long long ht_[6] = {wt_[0], 0, 0, 0, 0, 0};
#define AZG_HSTAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ht_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AZG_HSTAMP(i) do { } while (0)
#endif
                AZG_HSTAMP(1);
                if (!MASK_WAVE && sim < sa.sims) {                   // (masks first: every expansion waits for them, see k_backup_select2)
                    reinterpret_cast<unsigned long long *>(ws + WS::LESS)[lane] = shuffle_less_mask<(G::MAXK < 64 ? G::MAXK : 64)>(sa.ev, slot, ctr0 + (root_noise ? 1 : 0), lane);
                    flag_set_gen(&flags[1], sim, lane);
                }
                AZG_HSTAMP(2);
                if (has_policy) {
                    Node *nodes = tree_nodes(sa.ev, tree, hr.base);
                    float *pi = reinterpret_cast<float *>(ws + WS::PI);
                    if constexpr (!EXACT) {
                        leaf_policy_logits<G, tower_min_blocks<SEARCH>() == 1>(sa.hd, nodes, hr.leaf_fc, hr.leaf_k, reinterpret_cast<const _Float16 *>(ws + WS::FEAT), lg, lane);
                        wave_sync();
                    }
                    if constexpr (OVL) flag_wait_gen(sa.ev, &flags[2], sim);         // (the other streaming wavefront's policy subtiles)
                    AZG_HSTAMP(3);
                    policy_softmax_row<A>(lg, lane, A, pi);
                    wave_sync();
                    AZG_HSTAMP(4);
                    backup_policy<G>(evl, slot, hr, nodes, pi, reinterpret_cast<float *>(ws + WS::M), reinterpret_cast<float *>(ws + WS::SCR), lane);
                    AZG_HSTAMP(5);
                }
                flag_set_gen(&flags[0], sim, lane);
#ifdef AZG_TOWER_TIMING
                if (P.dbg && wave == BOARDS && lane == 0 && blockIdx.x < 512 && sim >= 8 && has_policy)   // helper of board 0: header, masks, logits, softmax, priors
                    for (int i = 0; i < 5; i++) P.dbg[2048 + 4096 * 5 + (size_t)blockIdx.x * 8 + i] += ht_[i + 1] - ht_[i];
                if (P.dbg && wave == BOARDS && lane == 0 && blockIdx.x < 512 && sim >= 8 && has_policy) P.dbg[2048 + 4096 * 5 + (size_t)blockIdx.x * 8 + 5] += 1;
#endif
            } else if (role == 0 && lane < HW) {                     // no game behind this board: zero planes
                char *row = img + GEO::qrow(bd * HW + lane) * RS;
                const uint4 z = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4 *>(row) = z; *reinterpret_cast<uint4 *>(row + 16) = z;
                *reinterpret_cast<uint4 *>(row + 32) = z; *reinterpret_cast<uint4 *>(row + 48) = z;
            }
            if (sim == sa.sims) break;                           // the last backup is done: no evaluation follows
            AZG_WPHASE(1);
        } else if constexpr (IS_SEARCH) {
            using G = typename SEARCH::Game;
            static_assert(G::CELLS == HW && G::A < 8, "search mode needs a game whose tree functions use no LDS scratch");
            // a sticky device error (tree arena full) stops the trees; the decision must be uniform over the workgroup.  Looked at
            // every 16th simulation only (a global load and two barriers): the tree functions themselves never write past a full
            // arena, they just stop expanding
            if ((sim & 15) == 0) {
                int *flag = reinterpret_cast<int *>(img + SCRATCH_PV + 512);
                if (tid == 0) *flag = sa.ev.gcount[GC_ERROR];
                __syncthreads();
                const int err = *flag;
                __syncthreads();
                if (tid == 0) *flag = 0;
                if (err) break;
            }
            int slot = tile * BOARDS + wave;
            asm volatile("" : "+v"(slot));                       // (opaque: keeps the slot's tree addresses from being hoisted out of
                                                                 //  the simulation loop and spilled across the tower)
            slot = __builtin_amdgcn_readfirstlane(slot);
            if (wave >= BOARDS) {                                // (one-board tiles: waves 1.. only take part in the tower)
            } else if (slot < sa.ev.B) {
                select_slot<G>(sa.ev, slot, lane, nullptr, [&](const typename G::S &st, int ln) {
                    if (ln < HW) {                               // leaf observation -> the image rows of board `wave`, channels 8.. zero
                        char *row = img + GEO::qrow(wave * HW + ln) * RS;
                        *reinterpret_cast<half8 *>(row) = G::obs8(st, ln);
                        const uint4 z = make_uint4(0, 0, 0, 0);
                        *reinterpret_cast<uint4 *>(row + 16) = z; *reinterpret_cast<uint4 *>(row + 32) = z; *reinterpret_cast<uint4 *>(row + 48) = z;
                    }
                });
            } else if (lane < HW) {
                char *row = img + GEO::qrow(wave * HW + lane) * RS;
                const uint4 z = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4 *>(row) = z; *reinterpret_cast<uint4 *>(row + 16) = z;
                *reinterpret_cast<uint4 *>(row + 32) = z; *reinterpret_cast<uint4 *>(row + 48) = z;
            }
        } else {
            const uint4 *xg = reinterpret_cast<const uint4 *>(P.x) + (size_t)row0;
            for (int c = tid; c < ROWS * 4; c += NT) {
                const int p = c >> 2, chunk = c & 3;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (chunk == 0 && p < rows_here) v = xg[p];
                *reinterpret_cast<uint4 *>(img + GEO::qrow(p) * RS + chunk * 16) = v;
            }
        }
#ifdef AZG_TOWER_TIMING
        if (P.dbg && tid == 0) P.dbg[1024 + blockIdx.x * 2] = __builtin_amdgcn_s_memtime();
#endif
        const half8 *wt = wm0;
#pragma unroll
        for (int j = 0; j < WR - 1; j++) { const half8 *f = stream_frag<GEO, KSPLIT>(wl, wm0, j); a[j][0] = f[0]; a[j][1] = f[64]; }
#ifdef AZG_TOWER_TIMING
        asm volatile("s_waitcnt vmcnt(0)");
        if (P.dbg && tid == 0) P.dbg[1024 + blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef AZG_TOWER_TIMING
#define AZG_STAMP2(i) do { if (P.dbg && blockIdx.x == 0 && tile == 0 && lane == 0) P.dbg[(layer * 4 + wave) * 5 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define AZG_WGSTAMP(i) do { if (P.dbg && tid == 0) P.dbg[2048 + (size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AZG_STAMP2(i) do { } while (0)
#define AZG_WGSTAMP(i) do { } while (0)
#endif
        AZG_WGSTAMP(4);
        // (one-board tiles: the biases come out of LDS, a layer's are read before the barrier that ends the previous layer)
        [[maybe_unused]] floatx4 bnext[2];
        if constexpr (PLDS) {
#pragma unroll
            for (int m = 0; m < 2; m++) bnext[m] = *reinterpret_cast<const floatx4 *>(smem + PARAM_OFF + ((2 * cg + m) * 16 + g * 4) * 4);
        }
        for (int layer = 0; layer <= 2 * P.nblocks; layer++) {
            AZG_STAMP2(0);
            // The co-resident workgroups of a CU take turns at issue priority, layer by layer.  Left alone the arbiter favours
            // the older wave of each SIMD throughout: that workgroup runs at solo speed, the other one in its gaps, and the
            // kernel ends with the second one alone on the CU for the last quarter (s_memtime: 441k vs 610k cycles).  Alternating
            // makes both finish together: -5 % wall.
            if ((layer ^ slot_role) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
            const float *bias = P.bias + (size_t)layer * C;
            const bool is_s = (layer & 1) == 0;
            const int nb = layer >> 1;
            const bool has_next = nb < P.nblocks;
            const half2v zero2 = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int c0 = (2 * cg + m) * 16 + g * 4;
                floatx4 bv;
                if constexpr (PLDS) bv = bnext[m];
                else bv = (floatx4){bias[c0], bias[c0 + 1], bias[c0 + 2], bias[c0 + 3]};
                if constexpr (KSPLIT == 2) { if (kg && layer) bv = (floatx4){0.f, 0.f, 0.f, 0.f}; }   // (the stem is not split: both k groups run it whole)
#pragma unroll
                for (int ps = 0; ps < NSUB; ps++) acc[m][ps] = bv;
            }
            // the next block's pre-activation affine: big tiles fetch it after the main loop (registers are the scarce resource at
            // 2 waves per SIMD, the co-resident wave hides the latency); small tiles run one wave per SIMD with registers to
            // spare, so they fetch it BEFORE the main loop (after it the load's latency would be fully exposed: ~1 k cycles per block)
            constexpr bool AFFINE_EARLY = NSUB <= 4;
            half2v sc[4], sh[4];
#pragma unroll
            for (int j = 0; j < 4; j++) sc[j] = sh[j] = zero2;
            auto fetch_affine = [&]() {
                if (is_s && has_next) {
                    if constexpr (PLDS) {
                        const char *ps_ = smem + sc_off + (nb * C + ecol) * 2, *pt_ = smem + sh_off + (nb * C + ecol) * 2;
#pragma unroll
                        for (int j = 0; j < 4; j++) { sc[j] = *reinterpret_cast<const half2v *>(ps_ + 4 * j); sh[j] = *reinterpret_cast<const half2v *>(pt_ + 4 * j); }
                    } else {
                        const float *ps_ = P.pre_scale + (size_t)nb * C + ecol, *pt_ = P.pre_shift + (size_t)nb * C + ecol;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            sc[j] = (half2v){(_Float16)ps_[2 * j], (_Float16)ps_[2 * j + 1]};
                            sh[j] = (half2v){(_Float16)pt_[2 * j], (_Float16)pt_[2 * j + 1]};
                        }
                    }
                }
            };
            if constexpr (AFFINE_EARLY) fetch_affine();
            if (layer == 0) conv_stem<GEO, NSUB, WR, KSPLIT>(smem, lb, g, wl, wm0, a, acc);
            else if constexpr (KSPLIT == 2) {                   // this wave's channel half: the image rows shifted by kg * 64 bytes, 9 k-steps
                conv_main2<GEO, 1, NSUB, STEM_KSTEPS % WR, WR, TAPSKIP, 2>(smem + kg * 64, lb, wt, a, acc, acc);
                wt += (size_t)9 * KS * GEO::WSTEP;
            } else if constexpr (KHALF) {
                floatx4 accB[2][NSUB];
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int ps = 0; ps < NSUB; ps++) accB[m][ps] = (floatx4){0.f, 0.f, 0.f, 0.f};
                conv_main2<GEO, KS, NSUB, STEM_KSTEPS % WR, WR, TAPSKIP, 1, true>(smem, lb, wt, a, acc, accB);
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int ps = 0; ps < NSUB; ps++) acc[m][ps] += accB[m][ps];
                wt += (size_t)9 * KS * GEO::WSTEP;
            } else { conv_main2<GEO, KS, NSUB, STEM_KSTEPS % WR, WR, TAPSKIP>(smem, lb, wt, a, acc, acc); wt += (size_t)9 * KS * GEO::WSTEP; }
            AZG_STAMP2(1);
            if constexpr (!AFFINE_EARLY) fetch_affine();
            // k-split: the partial sums of the subtiles the PARTNER finishes go to it through LDS (this wave's region, 1 KB per
            // accumulator tile), those of this wave's own subtiles come back after the barrier: out = first half + second half
            [[maybe_unused]] char *xown = nullptr, *xpar = nullptr;
            if constexpr (KSPLIT == 2) {
                char *xb = smem + XCHG_OFF + lane * 16;
                xown = xb + wave * (2 * NOWN * 1024);
                xpar = xb + (wave ^ ((C / 32) * PSPLIT)) * (2 * NOWN * 1024);
                if (layer) {
#pragma unroll
                    for (int m = 0; m < 2; m++)
#pragma unroll
                        for (int j = 0; j < NOWN; j++)
                            *reinterpret_cast<floatx4 *>(xown + (m * NOWN + j) * 1024) = kg ? acc[m][j] : acc[m][j + NOWN];
                }
            }
            __syncthreads();                                    // every wave is done reading the image
            AZG_STAMP2(2);
            floatx4 fin[2][NOWN];
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int j = 0; j < NOWN; j++) {
                    fin[m][j] = acc[m][j];
                    if constexpr (KSPLIT == 2) {
                        if (kg) fin[m][j] = acc[m][j + NOWN];
                        if (layer) {
                            const floatx4 o = *reinterpret_cast<const floatx4 *>(xpar + (m * NOWN + j) * 1024);
                            fin[m][j] = kg ? o + fin[m][j] : fin[m][j] + o;      // (first half + second half, whoever adds)
                        }
                    }
                }
            int oz = 0;                                         // opaque zero: the store offsets lb + edelta are loop invariants that
            asm volatile("" : "+s"(oz));                        // LLVM would otherwise precompute, keep live across the main loop and spill
#pragma unroll
            for (int ps = 0; ps < NOWN; ps++) {
                half2v v[4];
                {
                    union { half2v h; unsigned u; } a0, a1, b0, b1;
                    a0.h = (half2v){(_Float16)fin[0][ps][0], (_Float16)fin[0][ps][1]}; a1.h = (half2v){(_Float16)fin[0][ps][2], (_Float16)fin[0][ps][3]};
                    b0.h = (half2v){(_Float16)fin[1][ps][0], (_Float16)fin[1][ps][1]}; b1.h = (half2v){(_Float16)fin[1][ps][2], (_Float16)fin[1][ps][3]};
                    auto r0 = __builtin_amdgcn_permlane16_swap(a0.u, b0.u, false, false);
                    auto r1 = __builtin_amdgcn_permlane16_swap(a1.u, b1.u, false, false);
                    a0.u = r0[0]; b0.u = r0[1]; a1.u = r1[0]; b1.u = r1[1];
                    v[0] = a0.h; v[1] = a1.h; v[2] = b0.h; v[3] = b1.h;
                }
                const unsigned off = lbo[ps] + (edelta + (unsigned)oz);
                const bool lv = (liveown >> ps) & 1;
                if (!is_s) {
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = __builtin_elementwise_max(v[j], zero2);
                    if (lv) *reinterpret_cast<uint4 *>(img + off) = *reinterpret_cast<uint4 *>(v);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (layer == 0) v[j] = __builtin_elementwise_max(v[j], zero2); else v[j] += sreg[ps][j];
                        sreg[ps][j] = v[j];
                    }
                    if (lv) {
                        if (has_next) {
                            half2v t[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) t[j] = __builtin_elementwise_max(v[j] * sc[j] + sh[j], zero2);
                            *reinterpret_cast<uint4 *>(img + off) = *reinterpret_cast<uint4 *>(t);
                        } else {
                            *reinterpret_cast<uint4 *>(img + off) = *reinterpret_cast<uint4 *>(v);     // final stream for the heads
                        }
                    }
                }
            }
            if constexpr (PLDS) {                               // (past the last layer this reads the first affine row: in bounds, unused)
#pragma unroll
                for (int m = 0; m < 2; m++) bnext[m] = *reinterpret_cast<const floatx4 *>(smem + PARAM_OFF + ((layer + 1) * C + (2 * cg + m) * 16 + g * 4) * 4);
            }
            AZG_STAMP2(3);
            __syncthreads();
            AZG_STAMP2(4);
        }
        AZG_WGSTAMP(5);
        AZG_WPHASE(2);
        if (P.head_w == nullptr && P.head1_w != nullptr) {
            // first stage of the factorised heads: 32 head channels per pixel, centre tap only; the waves of cout group 0
            // compute them for their own pixel subtiles straight out of the image (the final stream)
            [[maybe_unused]] _Float16 *feat_lds = nullptr;       // wide search mode: the features stay in LDS
            if constexpr (IS_WIDE) feat_lds = reinterpret_cast<_Float16 *>(smem + TILE + WideScratch<typename SEARCH::Game, HW, SOLO>::FEAT);
            [[maybe_unused]] HeadsFirst<typename std::conditional<IS_WIDE, typename WideGameOf<SEARCH>::type, C4>::type, HW> hfirst;
            // (one-game tiles only: the 100 registers the fragments wait in cost the multi-game tiles more in spills than the round trip:
            //  brandubh 512 games 7.24 -> 7.03 ms per move, 2048 games 15.19 -> 15.61 with it, same box)
            constexpr bool HEADS_EARLY = EXACT && BOARDS == 1;
            if constexpr (OVL) {                                 // (the streaming wavefronts; items instead of subtiles: value first)
                if (wave >= 1 && wave <= OVL_NW) heads_full_prefetch<typename SEARCH::Game, HW, OVL_NW, true>(sa.hf, wave - 1, lane, hfirst);
                if (IS_WIDE && tid == 0 && (sim & 15) == 15) {   // (the sticky-error look, every 16th simulation: read behind the barrier below)
                    using WLE = WideLds<typename WideGameOf<SEARCH>::type, HW, BOARDS, SOLO>;
                    *reinterpret_cast<int *>(smem + TILE + WLE::ERR) = sa_error_word(sa);
                }
            } else
            if constexpr (HEADS_EARLY) heads_full_prefetch<typename SEARCH::Game, HW, NT / 64>(sa.hf, wave, lane, hfirst);
            if (cg == 0) {
                int opaque = 0;
                asm volatile("" : "+s"(opaque));                 // (keeps these loop invariants from being hoisted across the layers)
                const half8 *hw1 = reinterpret_cast<const half8 *>(P.head1_w) + lane + opaque;
                floatx4 hacc[2][NOWN];                          // (k-split: each k group does the subtiles it finished)
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    const int c0 = m * 16 + g * 4;
                    floatx4 bv;
                    if constexpr (PLDS_H1) bv = *reinterpret_cast<const floatx4 *>(smem + h1_off + KS * 2 * 1024 + c0 * 4 + opaque);
                    else bv = (floatx4){P.head1_b[c0], P.head1_b[c0 + 1], P.head1_b[c0 + 2], P.head1_b[c0 + 3]};
#pragma unroll
                    for (int ps = 0; ps < NOWN; ps++) hacc[m][ps] = bv;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    half8 a0, a1;
                    if constexpr (PLDS_H1) {
                        a0 = *reinterpret_cast<const half8 *>(smem + h1_off + ((ks * 2) * 64 + lane) * 16 + opaque);
                        a1 = *reinterpret_cast<const half8 *>(smem + h1_off + ((ks * 2 + 1) * 64 + lane) * 16 + opaque);
                    } else { a0 = hw1[(size_t)(ks * 2) * 64]; a1 = hw1[(size_t)(ks * 2 + 1) * 64]; }
#pragma unroll
                    for (int ps = 0; ps < NOWN; ps++) {
                        const half8 b = *reinterpret_cast<const half8 *>(img + lbo[ps] + (GEO::BIAS + ks * 64));
                        hacc[0][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b, hacc[0][ps], 0, 0, 0);
                        hacc[1][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b, hacc[1][ps], 0, 0, 0);
                    }
                }
                _Float16 *fg = reinterpret_cast<_Float16 *>(P.feat);
#pragma unroll
                for (int ps = 0; ps < NOWN; ps++) {
                    const int gs = ph * NSUB + ps + (KSPLIT == 2 ? kg * NOWN : 0);
                    const int p = gs < NSUBT ? pixmap[min(gs, NSUBT - 1) * 16 + i16] : -1;
                    if (p >= 0 && p < rows_here) {
                        const int bd = p / HW, pos = p - bd * HW;
                        _Float16 *dst = fg + (size_t)(tile * BOARDS + bd) * 2 * P.feat_k + pos * 16 + g * 4;
                        if constexpr (IS_WIDE)
                            dst = reinterpret_cast<_Float16 *>(reinterpret_cast<char *>(feat_lds) + bd * WideScratch<typename SEARCH::Game, HW, SOLO>::BYTES) + pos * 16 + g * 4;
#pragma unroll
                        for (int m = 0; m < 2; m++) {
                            const half4 h = {(_Float16)hacc[m][ps][0], (_Float16)hacc[m][ps][1], (_Float16)hacc[m][ps][2], (_Float16)hacc[m][ps][3]};
                            *reinterpret_cast<half4 *>(dst + m * P.feat_k) = h;
                        }
                    }
                }
            }
            if constexpr (IS_WIDE) {
                // (second stage: the next tree phase turns the features into the logits it needs -- sparse heads, azg_kernels.h)
                __syncthreads();                                 // the features of every board are in LDS
                AZG_WPHASE(3);
                if constexpr (OVL) {
                    // no barrier from here to the next tower: wave 0 goes straight on to walk (it waits for the value row by flag), wave 3
                    // to the masks and the rules, waves 1 .. NW stream the head matrix; wave 2 then reports its policy subtiles
                    using WSO = WideScratch<typename SEARCH::Game, HW, SOLO>;
                    using WLO = WideLds<typename SEARCH::Game, HW, BOARDS, SOLO>;
                    if ((sim & 15) == 15 && *reinterpret_cast<const int *>(smem + TILE + WLO::ERR)) break;   // (uniform: every wavefront reads the word behind the barrier)
                    int *flags_ = reinterpret_cast<int *>(smem + TILE + WSO::FLAGS);
                    if (wave >= 1 && wave <= OVL_NW) {
                        heads_full_lds<typename SEARCH::Game, HW, BOARDS, OVL_NW, SOLO, true>(smem + TILE, smem + TILE + WLO::ZERO, sa.hf, wave - 1, lane, hfirst, &flags_[3], sim + 1);
                        if (wave == 2) flag_set_gen(&flags_[2], sim + 1, lane);
                    }
                } else if constexpr (HEADS_EARLY) {
                    heads_full_lds<typename SEARCH::Game, HW, BOARDS, NT / 64, SOLO>(smem + TILE, smem + TILE + WideLds<typename SEARCH::Game, HW, BOARDS, SOLO>::ZERO, sa.hf, wave, lane, hfirst);
                } else if constexpr (EXACT) {
                    HeadsFirst<typename SEARCH::Game, HW> hl;
                    heads_full_prefetch<typename SEARCH::Game, HW, NT / 64>(sa.hf, wave, lane, hl);
                    heads_full_lds<typename SEARCH::Game, HW, BOARDS, NT / 64, SOLO>(smem + TILE, smem + TILE + WideLds<typename SEARCH::Game, HW, BOARDS, SOLO>::ZERO, sa.hf, wave, lane, hl);
                }
                AZG_WPHASE(4);
#ifdef AZG_TOWER_TIMING
#ifndef AZG_PHASE_TID
#define AZG_PHASE_TID 0                                              // (the thread whose stamps are summed: 64 = wavefront 1, which streams policy subtiles in every heads layout)
#endif
                if (P.dbg && tid == AZG_PHASE_TID && blockIdx.x < 512 && sim >= 8)      // tree (incl. the sparse heads), tower, head conv (cycles, summed over simulations)
                    for (int i = 0; i < 4; i++) P.dbg[2048 + 4096 * 4 + (size_t)blockIdx.x * 4 + i] += wt_[i + 1] - wt_[i];
#endif
            }
        } else if (P.head_w == nullptr) {
            uint4 *yg = reinterpret_cast<uint4 *>(P.y) + (size_t)row0 * CPR;
            for (int c = tid; c < rows_here * CPR; c += NT) {
                const int p = c / CPR, chunk = c - p * CPR;
                yg[c] = *reinterpret_cast<const uint4 *>(img + GEO::qrow(p) * RS + chunk * 16);
            }
        } else if constexpr (C == 128) {
            // logits[board, o] = sum over (pixel, channel) of s_final * Wfull: one MFMA per (pixel, 32-channel step), A = head
            // weights from L2, B = the final stream in the LDS image (column n = board).  The waves split the pixels; each
            // batch issues its 16 weight-fragment loads back to back (branch-free: pixels past the end re-read the last one
            // with a zeroed B fragment), four independent accumulators (one per channel step) avoid a dependent MFMA chain.
            constexpr int NPW = (HW + 3) / 4, JB = 4;
            // (an opaque zero keeps LLVM from hoisting this block's per-pixel offsets and weight pointers -- loop invariants -- out
            //  of the tile loop, where they would sit in registers across all the layers and be spilled: 36 MB of stores per launch)
            int opaque = 0;
            asm volatile("" : "+s"(opaque));
            floatx4 hacc4[4];
#pragma unroll
            for (int ks = 0; ks < 4; ks++) hacc4[ks] = (floatx4){0.f, 0.f, 0.f, 0.f};
            const bool bvalid = i16 < BOARDS;
            const unsigned bbase = (unsigned)((GEO::LEAD + (bvalid ? i16 : 0) * GEO::BSTRIDE) * RS + g * 16);
            const half8 *hw = reinterpret_cast<const half8 *>(P.head_w) + lane;
            const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
            // (with pixel-split workgroups only the first four waves -- one per cout group -- take part: the same pixel split and
            //  the same summation order in every tile shape, so a board's probabilities do not depend on the shape)
            if (wave < 4)
#pragma unroll
            for (int j0 = 0; j0 < NPW; j0 += JB) {
                half8 af[JB][4], bf[JB][4];
#pragma unroll
                for (int j = 0; j < JB; j++) {
                    if (j0 + j < NPW) {
                        const int p = wave + opaque + 4 * (j0 + j), pc = min(p, HW - 1);
                        const int y = pc / W, x = pc - y * W;
                        const unsigned poff = (unsigned)(((y + 1) * GEO::PW + x) * RS);
#pragma unroll
                        for (int ks = 0; ks < 4; ks++) {
                            af[j][ks] = hw[(size_t)(pc * 4 + ks) * 64];
                            bf[j][ks] = *reinterpret_cast<const half8 *>(img + bbase + poff + ks * 64);
                            if (!bvalid || p >= HW) bf[j][ks] = zero8;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < JB; j++)
                    if (j0 + j < NPW) {
#pragma unroll
                        for (int ks = 0; ks < 4; ks++) hacc4[ks] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j][ks], bf[j][ks], hacc4[ks], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            const floatx4 hacc = (hacc4[0] + hacc4[1]) + (hacc4[2] + hacc4[3]);
            // the reduction scratch lives in the pad rows at the top of the image (zeroed again below) -- but real data of
            // board 0 must not be clobbered before every wave has finished reading it
            AZG_WGSTAMP(6);
            __syncthreads();
            float *red = reinterpret_cast<float *>(img);
            if (wave < 4) {
#pragma unroll
                for (int r = 0; r < 4; r++) red[(wave * 16 + g * 4 + r) * 16 + i16] = hacc[r];
            }
            __syncthreads();
            const int nb_here = min(BOARDS, P.boards - tile * BOARDS);
            if (tid < 64) {                                      // lane = (board, output): both softmaxes inside 16-lane groups
                const int bd = tid >> 4, o = tid & 15, A = P.A, NV = P.NV;
                const int bc = bd < BOARDS ? bd : 0;
                const float lgt = ((red[(0 * 16 + o) * 16 + bc] + red[(1 * 16 + o) * 16 + bc]) + (red[(2 * 16 + o) * 16 + bc] + red[(3 * 16 + o) * 16 + bc]))
                                  + P.head_b[o];
                const bool isp = o < A, isv = o >= A && o < A + NV;
                float mp = isp ? lgt : -INFINITY, mv = isv ? lgt : -INFINITY;
#pragma unroll
                for (int d = 8; d; d >>= 1) { mp = fmaxf(mp, __shfl_xor(mp, d, 16)); mv = fmaxf(mv, __shfl_xor(mv, d, 16)); }
                const float e = isp ? __expf(lgt - mp) : isv ? __expf(lgt - mv) : 0.f;
                float sp = isp ? e : 0.f, sv = isv ? e : 0.f;
#pragma unroll
                for (int d = 8; d; d >>= 1) { sp += __shfl_xor(sp, d, 16); sv += __shfl_xor(sv, d, 16); }
                if constexpr (IS_SEARCH) {                       // probabilities stay in LDS for the backup waves: [board][16]
                    reinterpret_cast<float *>(img + SCRATCH_PV)[bd * 16 + o] = isp ? e / sp : e / sv;
                } else if (bd < nb_here) {
                    if (isp) P.policy[(size_t)(tile * BOARDS + bd) * A + o] = e / sp;
                    if (isv) P.value[(size_t)(tile * BOARDS + bd) * NV + (o - A)] = e / sv;
                }
            }
            if constexpr (IS_SEARCH) {
                using G = typename SEARCH::Game;
                __syncthreads();
                int slot = tile * BOARDS + wave;
                asm volatile("" : "+v"(slot));
                slot = __builtin_amdgcn_readfirstlane(slot);
                const float *pv = reinterpret_cast<const float *>(img + SCRATCH_PV) + wave * 16;
                if (wave < BOARDS && slot < sa.ev.B) backup_slot<G>(sa.ev, slot, lane, pv, pv + P.A, nullptr, nullptr);
            }
            AZG_WGSTAMP(7);
            __syncthreads();
            if (tid < 256) reinterpret_cast<uint4 *>(img)[tid] = make_uint4(0, 0, 0, 0);
            if constexpr (IS_SEARCH) { if (tid < 16) reinterpret_cast<uint4 *>(img + SCRATCH_PV)[tid] = make_uint4(0, 0, 0, 0); }
        }
        if constexpr (!OVL) __syncthreads();                    // (overlapped tile: the next barrier is the one in front of the next tower)
        }                                                        // sims
        if constexpr (IS_WIDE) {
            // LDS -> HBM: header, tape counter, tallies and the last path of every game, as the launch-per-phase path leaves them
            using G = typename SEARCH::Game;
            using WS = WideScratch<G, HW, SOLO>;
            __syncthreads();
            const int bd = wave % BOARDS, role = wave / BOARDS, slot = tile * BOARDS + bd;
            if (lds_live && role == 0 && slot < sa.ev.B) {
                const char *ws = smem + TILE + bd * WS::BYTES;
                if (lane < 4) reinterpret_cast<uint4 *>(sa.ev.hdr + slot)[lane] = reinterpret_cast<const uint4 *>(ws + WS::HDR)[lane];
                static_assert(sizeof(TreeHdr) == 4 * sizeof(uint4) && sizeof(azg_state) % 16 == 0 && sizeof(PathEnt) == sizeof(uint4),
                              "the LDS mirror moves the header, the root state and the path as whole 16-byte chunks");
                if constexpr (!SOLO) {
                    const int depth = *reinterpret_cast<const int *>(ws + WS::HDR + offsetof(TreeHdr, depth));
                    for (int j = lane; j < depth && j < sa.ev.maxd; j += 64)
                        reinterpret_cast<uint4 *>(sa.ev.path + (size_t)slot * sa.ev.maxd)[j] = reinterpret_cast<const uint4 *>(ws + WS::PATH)[j];
                }
                if (lane == 0) {
                    sa.ev.tape_ctr[slot] = *reinterpret_cast<const uint64_t *>(ws + WS::CTR);
                    sa.ev.slot_sims[slot] = *reinterpret_cast<const int64_t *>(ws + WS::CTR + 8);
                    sa.ev.slot_exp[slot] = *reinterpret_cast<const int64_t *>(ws + WS::CTR + 16);
                }
            }
        }
    }
#ifdef AZG_TOWER_TIMING
    if (P.dbg && tid == 0) P.dbg[2048 + (size_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();
#endif
}

// leaf observation planes [B, C, H, W] (any of the engine's obs dtypes) are written by k_select directly as the stem's
// NHWC8 fp16 rows when obs_dtype == 2 (see G::write_obs_nhwc8).

}  // namespace azg
