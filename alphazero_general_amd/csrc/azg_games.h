// azg_games.h -- device-side game rules ("rule kernels" registered per game; Game plugin API alphazero/Game.py:7-113).
//
// A game policy G provides, for one wavefront working on one game:
//   G::S                      wave-uniform register state
//   load(azg_state*, lane)    HBM -> registers         store(S, azg_state*, lane)  registers -> HBM
//   play(S&, action)          GameState.play_action    win_bits(S)   GameState.win_state as bit flags
//   valid_list(S, lane, act_lds, k)   GameState.valid_moves: lane i < k holds the i-th valid action (ascending)
//   write_obs<OT>(S, out, lane)       GameState.observation
//   symmetry(S, k, lane) / sym_action(a, k)   GameState.symmetries
#pragma once
#include "azg_device.h"

namespace azg {

// ================================================================================================ connect4
// alphazero/envs/connect4/connect4.pyx + Connect4Logic.pyx.  The reference scans an int board cell by cell;
// here the board is two 48-bit bitboards (bit r*8+c, column 7 = padding so shifts never wrap) held in SGPRs,
// built from the ABI's int8 cells with two wave ballots.
struct C4 {
    static constexpr int ID = AZG_GAME_CONNECT4;
    static constexpr int A = 7, H = 6, W = 7, CELLS = 42, P = 2, HAS_DRAW = 1, MAX_TURNS = 42, NSYM = 2;
    static constexpr int OBS_C = 4, OBS = OBS_C * CELLS, MAXK = 7;
    struct S { uint64_t b0, b1; int player, turns; };

    static AZG_DEV S load(const azg_state *st, int lane) {
        int r = lane >> 3, c = lane & 7;
        int8_t v = (lane < 48 && c < 7) ? st->cells[r * 7 + c] : (int8_t)0;
        S s;
        s.b0 = __ballot(v == 1);                    // stone of player 0 = 1   (connect4.pyx:65)
        s.b1 = __ballot(v == -1);
        s.player = __builtin_amdgcn_readfirstlane(st->player);
        s.turns = __builtin_amdgcn_readfirstlane(st->turns);
        return s;
    }
    static AZG_DEV void init(S &s) { s.b0 = s.b1 = 0; s.player = 0; s.turns = 0; }
    static AZG_DEV int cell(const S &s, int i) {    // i = r*7+c
        int r = i / 7, c = i - r * 7; int b = r * 8 + c;
        return (int)((s.b0 >> b) & 1) - (int)((s.b1 >> b) & 1);
    }
    static AZG_DEV void store(const S &s, azg_state *st, int lane) {
        st->cells[lane] = lane < CELLS ? (int8_t)cell(s, lane) : (int8_t)0;
        if (lane == 0) { st->player = s.player; st->turns = s.turns; st->aux[0] = 0; st->aux[1] = 0; }
    }
    // Connect4Logic.pyx:40-47 add_stone (lowest empty row of the column) + Game.py:76-79 _update_turn
    static AZG_DEV void play(S &s, int a) {
        uint64_t occ = s.b0 | s.b1;
        int cnt = __popcll((occ >> a) & 0x0101010101010101ULL);
        uint64_t bit = 1ULL << ((5 - cnt) * 8 + a);
        if (s.player == 0) s.b0 |= bit; else s.b1 |= bit;
        s.player ^= 1; s.turns += 1;
    }
    static AZG_DEV uint64_t valid_mask(const S &s) { return ~(s.b0 | s.b1) & 0x7FULL; }   // :49-57 top row empty
    static AZG_DEV bool has4(uint64_t b) {
        uint64_t m;
        m = b & (b >> 1); if (m & (m >> 2)) return true;      // rows        :64-72
        m = b & (b >> 8); if (m & (m >> 16)) return true;     // columns     :74-82
        m = b & (b >> 9); if (m & (m >> 18)) return true;     // diagonal    :86-92
        m = b & (b >> 7); if (m & (m >> 14)) return true;     // anti-diag   :93-99
        return false;
    }
    // Connect4Logic.pyx:59-110 + connect4.pyx:68-81: bit0 = player 0 won, bit1 = player 1 won, bit2 = draw
    static AZG_DEV int win_bits(const S &s) {
        if (has4(s.b0)) return 1;
        if (has4(s.b1)) return 2;
        if (valid_mask(s) == 0) return 4;
        return 0;
    }
    // lane i < k receives the i-th valid action in ascending order
    static AZG_DEV int valid_list(const S &s, int lane, int *act_lds, int (&my_a)[1]) {
        uint64_t vm = valid_mask(s);
        int k = __popcll(vm);
        // i-th set bit of vm: A <= 7, unrolled select
        int a = -1, cnt = 0;
#pragma unroll
        for (int c = 0; c < A; c++) { if ((vm >> c) & 1) { if (cnt == lane) a = c; cnt++; } }
        my_a[0] = a; (void)act_lds;
        return k;
    }
    // connect4.pyx:83-91: planes [pieces==1, pieces==-1, player, turns/42]
    template <typename OT> static AZG_DEV void write_obs(const S &s, OT *out, int lane) {
        float turn = (float)((double)s.turns / 42.0);
#pragma unroll
        for (int e0 = 0; e0 < OBS; e0 += 64) {
            int e = e0 + lane;
            if (e < OBS) {
                int plane = e / CELLS, i = e - plane * CELLS;
                int c = cell(s, i);
                float v = plane == 0 ? (c == 1 ? 1.f : 0.f) : plane == 1 ? (c == -1 ? 1.f : 0.f) : plane == 2 ? (float)s.player : turn;
                out[e] = (OT)v;
            }
        }
    }
    // the same four planes as NHWC fp16 rows with the channel dim padded to 8 (input format of the MFMA stem conv)
    static AZG_DEV void write_obs_nhwc8(const S &s, _Float16 *out, int lane) {
        if (lane < CELLS) {
            const int c = cell(s, lane);
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            h8 v = {(_Float16)(c == 1 ? 1.f : 0.f), (_Float16)(c == -1 ? 1.f : 0.f), (_Float16)(float)s.player,
                    (_Float16)(float)((double)s.turns / 42.0), (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            *reinterpret_cast<h8 *>(out + lane * 8) = v;
        }
    }
    // connect4.pyx:96-99: k = 1 mirrors the columns, pi -> pi[::-1]
    static AZG_DEV S symmetry(const S &s, int k) {
        if (k == 0) return s;
        S t = s; t.b0 = t.b1 = 0;
#pragma unroll
        for (int c = 0; c < 7; c++) {
            uint64_t col0 = (s.b0 >> c) & 0x0101010101010101ULL, col1 = (s.b1 >> c) & 0x0101010101010101ULL;
            t.b0 |= col0 << (6 - c); t.b1 |= col1 << (6 - c);
        }
        return t;
    }
    static AZG_DEV int sym_action(int a, int k) { return k == 0 ? a : 6 - a; }
};

}  // namespace azg
