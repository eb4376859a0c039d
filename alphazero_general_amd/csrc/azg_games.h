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
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    static AZG_DEV h8 obs8(const S &s, int lane) {               // the four planes of cell `lane` (< CELLS), channels 4..7 zero
        const int c = cell(s, lane);
        return (h8){(_Float16)(c == 1 ? 1.f : 0.f), (_Float16)(c == -1 ? 1.f : 0.f), (_Float16)(float)s.player,
                    (_Float16)(float)((double)s.turns / 42.0), (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    }
    static AZG_DEV void write_obs_nhwc8(const S &s, _Float16 *out, int lane) {
        if (lane < CELLS) *reinterpret_cast<h8 *>(out + lane * 8) = obs8(s, lane);
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


// ================================================================================================ brandubh
// 7x7 tafl: alphazero/envs/brandubh/fastafl.pyx (Game) + fastafl/cengine.pyx (Board) + boardgame/board.pyx, board
// options of variants.brandubh_args (king_two_sided_capture, move_over_throne, king cannot re-enter the throne).
// The reference walks Python lists of Square objects; here lane i < 49 owns cell i = y*7+x (the Board._state value,
// cengine.pyx:24-32) and everything is expressed on 49-bit wave ballots: sliding moves by per-lane ray walks over
// the passable mask, custodian captures by uniform readlanes around the moved piece, the recursive group-surround
// check (cengine.pyx:204-247) as a bitboard flood fill ("no member of the connected enemy group touches an empty
// square"), win tests by neighbour masks.
struct BR {
    static constexpr int ID = AZG_GAME_BRANDUBH;
    static constexpr int A = 588, H = 7, W = 7, CELLS = 49, P = 2, HAS_DRAW = 1, MAX_TURNS = 100, NSYM = 8;
    static constexpr int OBS_C = 5, OBS = OBS_C * CELLS, MAXK = 96;      // 8 attackers x 12 destinations bounds the move list
    static constexpr uint64_t ALL = (1ULL << 49) - 1;
    static constexpr uint64_t COL0 = 0x0040810204081ULL, COL6 = COL0 << 6;
    struct S { int cell; int player, turns, kc; };

    static AZG_DEV uint64_t nbr(uint64_t m) {       // squares orthogonally adjacent to the set m (board edges: nothing)
        return ((m << 7) | (m >> 7) | ((m & ~COL6) << 1) | ((m & ~COL0) >> 1)) & ALL;
    }
    static AZG_DEV uint64_t mask_eq(const S &s, int lane, int v) { return __ballot(lane < CELLS && s.cell == v); }
    static AZG_DEV bool is_king(int v) { return v == 3 || v == 7 || v == 8; }
    static AZG_DEV bool is_att(int v) { return v == 1 || is_king(v); }                       // ATTACKERS cengine.pyx:42
    static AZG_DEV int rd(const S &s, int idx) { return __builtin_amdgcn_readlane(s.cell, __builtin_amdgcn_readfirstlane(idx)); }

    static AZG_DEV S load(const azg_state *st, int lane) {
        S s;
        s.cell = lane < CELLS ? (int)st->cells[lane] : 0;
        s.player = __builtin_amdgcn_readfirstlane(st->player);
        s.turns = __builtin_amdgcn_readfirstlane(st->turns);
        s.kc = __builtin_amdgcn_readfirstlane(st->aux[0]);
        return s;
    }
    static AZG_DEV void init(S &s) {                // fastafl/variants.py:13-19
        const int lane = threadIdx.x & 63;
        const int y = lane / 7, x = lane - y * 7;
        int v = 0;
        if (lane < CELLS) {
            const bool corner = (x == 0 || x == 6) && (y == 0 || y == 6);
            if (corner) v = 5;
            else if (x == 3 && y == 3) v = 7;
            else if ((x == 3 && (y == 2 || y == 4)) || (y == 3 && (x == 2 || x == 4))) v = 1;
            else if ((x == 3 && (y <= 1 || y >= 5)) || (y == 3 && (x <= 1 || x >= 5))) v = 2;
        }
        s.cell = v; s.player = 0; s.turns = 0; s.kc = 0;
    }
    static AZG_DEV void store(const S &s, azg_state *st, int lane) {
        st->cells[lane] = lane < CELLS ? (int8_t)s.cell : (int8_t)0;
        if (lane == 0) { st->player = s.player; st->turns = s.turns; st->aux[0] = s.kc; st->aux[1] = 0; }
    }
    static AZG_DEV void decode(int a, int &src, int &dst) {          // fastafl.pyx:48-63 get_move
        const int mt = a % 12; src = a / 12;
        const int sx = src % 7, sy = src / 7;
        int nx, ny;
        if (mt < 6) { nx = sx; ny = mt + (mt >= sy ? 1 : 0); }
        else { nx = mt - 6; nx += (nx >= sx ? 1 : 0); ny = sy; }
        dst = ny * 7 + nx;
    }
    static AZG_DEV int encode(int x, int y, int nx, int ny) {        // fastafl.pyx:66-79 get_action
        const int mt = x == nx ? (ny < y ? ny : ny - 1) : (nx < x ? 6 + nx : 6 + nx - 1);
        return 12 * (x + y * 7) + mt;
    }
    // Board.move (cengine.pyx:249-272, no validity / win checks) + _check_capture (:172-197) + _check_surround (:228-247)
    static AZG_DEV void play(S &s, int a) {
        const int lane = threadIdx.x & 63;
        a = __builtin_amdgcn_readfirstlane(a);
        int src, dst; decode(a, src, dst);
        const int srcv = rd(s, src), dstv = rd(s, dst);
        const int piece = (srcv == 7 || srcv == 8) ? 3 : srcv;                               // remove_piece :311-326
        const int newsrc = srcv == 7 ? 4 : srcv == 8 ? 5 : 0;
        const int newdst = (dstv == 4 || dstv == 5) ? piece + dstv : piece;                  // add_piece :293-309
        if (lane == src) s.cell = newsrc;
        if (lane == dst) s.cell = newdst;
        // every capture below starts from a square next to the moved piece that holds an enemy or -- two-sided king capture fires
        // for any mover (Q20) -- the king: most moves have none, and are done here
        {
            const bool black = newdst == 2;                                                  // (the mover; a king on the throne / an escape counts as an attacker for black)
            const uint64_t relevant = __ballot(lane < CELLS && (black ? is_att(s.cell) : (s.cell == 2 || s.cell == 3)));
            if ((nbr(1ULL << dst) & relevant) == 0) { s.turns += 1; s.player ^= 1; return; }
        }
        const int dx4[4] = {0, 1, 0, -1}, dy4[4] = {1, 0, -1, 0};                            // DIRECTIONS :46
        const int mx = dst % 7, my = dst / 7;
        // ---- custodian capture around the moved piece ----
        const bool friendly_att = is_att(newdst);
        const int enemy = newdst != 3 ? 3 - newdst : 2;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int ex = mx + dx4[d], ey = my + dy4[d];
            if (ex < 0 || ex > 6 || ey < 0 || ey > 6) continue;
            const int ev = rd(s, ey * 7 + ex);
            const bool do_capture = ev == 3;                                                 // two-sided king capture (Q20)
            if (ev == enemy || do_capture) {
                const int fx = ex + dx4[d], fy = ey + dy4[d];
                if (fx < 0 || fx > 6 || fy < 0 || fy > 6) continue;
                const int fv = rd(s, fy * 7 + fx);
                const bool friendly = friendly_att ? is_att(fv) : fv == newdst;
                if (friendly || fv == 4 || fv == 5) {
                    if (do_capture) s.kc = 1;
                    else if (lane == ey * 7 + ex) s.cell = 0;
                }
            }
        }
        // ---- group surround: start squares in DIRECTIONS order, removals visible to the later ones ----
        const bool enemy_is_att = rd(s, dst) == 2;
        uint64_t checked = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int ex = mx + dx4[d], ey = my + dy4[d];
            if (ex < 0 || ex > 6 || ey < 0 || ey > 6) continue;
            const int e = ey * 7 + ex;
            const uint64_t enemyM = enemy_is_att ? __ballot(lane < CELLS && is_att(s.cell)) : mask_eq(s, lane, 2);
            if (!((enemyM >> e) & 1) || ((checked >> e) & 1)) continue;
            uint64_t g = 1ULL << e;
            for (;;) { const uint64_t g2 = g | (nbr(g) & enemyM); if (g2 == g) break; g = g2; }
            checked |= g;
            const uint64_t empty = mask_eq(s, lane, 0);
            if ((nbr(g) & empty) == 0) {                                                     // every member fully blocked
                const uint64_t kings = __ballot(lane < CELLS && is_king(s.cell));
                if (g & kings) s.kc = 1;                                                     // a king is never lifted (:243-244)
                if (lane < CELLS && ((g >> lane) & 1) && !is_king(s.cell)) s.cell = 0;
            }
        }
        s.turns += 1; s.player ^= 1;
    }
    // Game.win_state (fastafl.pyx:186-199) + Board.get_winner (cengine.pyx:163-169) + _has_legals_check (:134-141)
    static AZG_DEV int win_bits(const S &s) {
        const int lane = threadIdx.x & 63;
        if (s.turns >= 100) return 4;
        const uint64_t esc = mask_eq(s, lane, 8);
        const uint64_t adjE = nbr(mask_eq(s, lane, 0)), adjX = nbr(mask_eq(s, lane, 5));
        const uint64_t kings = __ballot(lane < CELLS && is_king(s.cell));
        const bool def_has = (mask_eq(s, lane, 2) & adjE) != 0;
        const bool att_has = ((mask_eq(s, lane, 1) & adjE) | (kings & (adjE | adjX))) != 0;
        if (esc != 0 || !def_has) return 2;           // attackers (player 1) win: result[2 - 1]
        if (s.kc || !att_has) return 1;               // defenders (player 0) win: result[2 - 2]
        return 0;
    }
    // cells of a 7-cell line (bit p = the piece's own position) a sliding piece at p reaches: the runs of passable cells on
    // both sides of p (Board.legal_moves walks them square by square, cengine.pyx:109-132)
    static AZG_DEV unsigned reach7(unsigned line, int p) {
        const unsigned up = line >> (p + 1);                                  // cells above p, bit 0 = p + 1
        const unsigned run = (unsigned)__builtin_ctz(~up);                    // trailing ones (a 7-bit line: ~up is never 0)
        const unsigned reach_up = ((1u << run) - 1u) << (p + 1);
        const unsigned below = (1u << p) - 1u, blk = ~line & below;           // blockers below p
        const unsigned cut = blk ? (2u << (31 - __builtin_clz(blk))) : 1u;    // first bit above the highest blocker
        return reach_up | (line & below & ~(cut - 1u));
    }
    // Game.valid_moves (fastafl.pyx:171-178) + Board.legal_moves (cengine.pyx:109-132): ascending action list in LDS.
    // Branch-free per lane: the row of the passable mask and (through the transposed board) its column are 7-bit lines, the
    // reachable runs come from count-trailing-ones / count-leading-zeros, the move types from two shifts; the list offsets
    // from a DPP row scan of the per-lane move counts.
    static AZG_DEV int valid_list(const S &s, int lane, int *act_lds, int (&my_a)[2]) {
        const int team = 2 - (s.turns & 1);                                                  // Board.to_play :330-331
        const bool mine = lane < CELLS && (team == 1 ? is_att(s.cell) : s.cell == 2);
        const bool king = is_king(s.cell);
        const int x = lane % 7, y = lane / 7;
        const int cellT = __shfl(s.cell, lane < CELLS ? x * 7 + y : 0);                      // the transposed board
        const uint64_t E = mask_eq(s, lane, 0), T = mask_eq(s, lane, 4), X = mask_eq(s, lane, 5);
        const uint64_t Et = __ballot(lane < CELLS && cellT == 0), Tt = __ballot(lane < CELLS && cellT == 4), Xt = __ballot(lane < CELLS && cellT == 5);
        const uint64_t pass = E | T | (king ? X : 0ULL), passT = Et | Tt | (king ? Xt : 0ULL);     // the ray continues over these
        const unsigned row = (unsigned)(pass >> (7 * y)) & 0x7Fu, col = (unsigned)(passT >> (7 * x)) & 0x7Fu;
        const unsigned trow = (unsigned)(T >> (7 * y)) & 0x7Fu, tcol = (unsigned)(Tt >> (7 * x)) & 0x7Fu;
        const unsigned rx = reach7(row, x) & ~trow, ry = reach7(col, y) & ~tcol;             // nobody lands on the empty throne
        // move_type (fastafl.pyx:66-79): vertical ny -> ny or ny - 1, horizontal nx -> 6 + nx or 6 + nx - 1
        const unsigned mvy = (ry & ((1u << y) - 1u)) | ((ry >> (y + 1)) << y), mvx = (rx & ((1u << x) - 1u)) | ((rx >> (x + 1)) << x);
        const unsigned mv = mine ? (mvy | (mvx << 6)) : 0u;                                  // bit = move_type (0..11)
        // list offset of the lane = moves of the lanes below it: an inclusive scan of the per-lane counts inside each row of 16
        // (four DPP row shifts) + the totals of the rows below; then one predicated store per move type instead of a per-lane loop
        const int cnt = __popc(mv);
        int inc = cnt;
        inc += dpp_i<0x111>(inc); inc += dpp_i<0x112>(inc); inc += dpp_i<0x114>(inc); inc += dpp_i<0x118>(inc);   // row_shr:1, 2, 4, 8
        const int r0 = rl(inc, 15), r1 = rl(inc, 31), r2 = rl(inc, 47), r3 = rl(inc, 63);
        const int lrow = lane >> 4;
        const int off = inc - cnt + (lrow == 0 ? 0 : lrow == 1 ? r0 : lrow == 2 ? r0 + r1 : r0 + r1 + r2);
        const int k = r0 + r1 + r2 + r3;
#pragma unroll
        for (int b = 0; b < 12; b++) {
            const int pos = off + __popc(mv & ((1u << b) - 1u));
            if (((mv >> b) & 1u) && pos < MAXK) act_lds[pos] = 12 * lane + b;
        }
        wave_sync();
        my_a[0] = lane < k ? act_lds[lane] : -1;
        my_a[1] = 64 + lane < k ? act_lds[64 + lane] : -1;
        wave_sync();
        return k < MAXK ? k : MAXK;
    }
    // Game.observation (fastafl.pyx:84-121,205-211): planes [black(2), white(1), king, to-move colour, 0] (Q18)
    template <typename OT> static AZG_DEV void write_obs(const S &s, OT *out, int lane) {
        const float colour = (float)(s.turns & 1), turn_no = (float)(s.turns / 100);
        for (int e0 = 0; e0 < OBS; e0 += 64) {
            const int e = e0 + lane;
            const int plane = e / CELLS, i = e - plane * CELLS;
            const int c = __shfl(s.cell, i < CELLS ? i : 0);
            if (e < OBS) {
                const float v = plane == 0 ? (c == 2 ? 1.f : 0.f) : plane == 1 ? (c == 1 ? 1.f : 0.f)
                              : plane == 2 ? (is_king(c) ? 1.f : 0.f) : plane == 3 ? colour : turn_no;
                out[e] = (OT)v;
            }
        }
    }
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    static AZG_DEV h8 obs8(const S &s, int lane) {               // the five planes of cell `lane` (< CELLS: the lane's own cell)
        const int c = s.cell; (void)lane;
        return (h8){(_Float16)(c == 2 ? 1.f : 0.f), (_Float16)(c == 1 ? 1.f : 0.f), (_Float16)(is_king(c) ? 1.f : 0.f),
                    (_Float16)(float)(s.turns & 1), (_Float16)(float)(s.turns / 100), (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    }
    static AZG_DEV void write_obs_nhwc8(const S &s, _Float16 *out, int lane) {
        if (lane < CELLS) *reinterpret_cast<h8 *>(out + lane * 8) = obs8(s, lane);
    }
    // Game.symmetries (fastafl.pyx:213-256): k = (i-1)*2 + flip; state = fliplr^flip(rot90^i(state)); the policy index is
    // permuted by the reference's own coordinate loop (sym_action)
    static AZG_DEV S symmetry(const S &s, int k) {
        const int lane = threadIdx.x & 63;
        const int i = k / 2 + 1, flip = k & 1;
        int r = lane / 7, c = lane - (lane / 7) * 7;
        if (flip) c = 6 - c;                              // final[r][c] = rotated[r][6-c]
        for (int t = 0; t < i; t++) { const int nr = c, nc = 6 - r; r = nr; c = nc; }       // rot90: new[r][c] = old[c][6-r]
        S o = s;
        const int v = __shfl(s.cell, lane < CELLS ? r * 7 + c : 0);
        o.cell = lane < CELLS ? v : 0;
        return o;
    }
    static AZG_DEV int sym_action(int a, int k) {
        const int i = k / 2 + 1, flip = k & 1;
        int src, dst; decode(a, src, dst);
        int x = src % 7, y = src / 7, nx = dst % 7, ny = dst / 7;
        for (int t = 0; t < i; t++) { const int tx = x, tnx = nx; x = 6 - y; nx = 6 - ny; y = tx; ny = tnx; }   // :241-246
        if (flip) { x = 6 - x; nx = 6 - nx; }
        return encode(x, y, nx, ny);
    }
};


// ================================================================================================ trimok (3 players)
// Build-defined N-player env (BASELINE config 5; rules in alphazero_general_amd/envs/trimok.py): 5x5 board, three players
// place stones in turn, three in a row wins, full board draws.  One 25-bit bitboard per player in SGPRs.
struct TM {
    static constexpr int ID = AZG_GAME_TRIMOK;
    static constexpr int A = 25, H = 5, W = 5, CELLS = 25, P = 3, HAS_DRAW = 1, MAX_TURNS = 25, NSYM = 1;
    static constexpr int OBS_C = 5, OBS = OBS_C * CELLS, MAXK = 25;
    struct S { uint32_t b[3]; int player, turns; };
    static AZG_DEV S load(const azg_state *st, int lane) {
        const int8_t v = lane < CELLS ? st->cells[lane] : (int8_t)0;
        S s;
        s.b[0] = (uint32_t)__ballot(v == 1); s.b[1] = (uint32_t)__ballot(v == 2); s.b[2] = (uint32_t)__ballot(v == 3);
        s.player = __builtin_amdgcn_readfirstlane(st->player);
        s.turns = __builtin_amdgcn_readfirstlane(st->turns);
        return s;
    }
    static AZG_DEV void init(S &s) { s.b[0] = s.b[1] = s.b[2] = 0; s.player = 0; s.turns = 0; }
    static AZG_DEV int cell(const S &s, int i) { return (int)((s.b[0] >> i) & 1) + 2 * (int)((s.b[1] >> i) & 1) + 3 * (int)((s.b[2] >> i) & 1); }
    static AZG_DEV void store(const S &s, azg_state *st, int lane) {
        st->cells[lane] = lane < CELLS ? (int8_t)cell(s, lane) : (int8_t)0;
        if (lane == 0) { st->player = s.player; st->turns = s.turns; st->aux[0] = 0; st->aux[1] = 0; }
    }
    static AZG_DEV void play(S &s, int a) {
        const uint32_t bit = 1u << a;
        if (s.player == 0) s.b[0] |= bit; else if (s.player == 1) s.b[1] |= bit; else s.b[2] |= bit;
        s.player = s.player == 2 ? 0 : s.player + 1; s.turns += 1;          // Game.py:73-79 (player + 1) % num_players
    }
    static AZG_DEV bool has3(uint32_t m) {
        constexpr uint32_t XLE2 = 0x00739CE7u;                               // cells with x <= 2 (line start, going right)
        constexpr uint32_t XGE2 = 0x01CE739Cu;                               // cells with x >= 2 (anti-diagonal start)
        if (m & (m >> 1) & (m >> 2) & XLE2) return true;                     // horizontal
        if (m & (m >> 5) & (m >> 10)) return true;                           // vertical
        if (m & (m >> 6) & (m >> 12) & XLE2) return true;                    // diagonal  (+1, +1)
        if (m & (m >> 4) & (m >> 8) & XGE2) return true;                     // diagonal  (-1, +1)
        return false;
    }
    static AZG_DEV int win_bits(const S &s) {
        if (has3(s.b[0])) return 1;
        if (has3(s.b[1])) return 2;
        if (has3(s.b[2])) return 4;
        if ((s.b[0] | s.b[1] | s.b[2]) == 0x1FFFFFFu) return 8;
        return 0;
    }
    static AZG_DEV int valid_list(const S &s, int lane, int *act_lds, int (&my_a)[1]) {
        const uint32_t vm = ~(s.b[0] | s.b[1] | s.b[2]) & 0x1FFFFFFu;
        const int k = __popc(vm);
        // lane i takes the i-th set bit: the lane of an empty cell pushes its index to the lane of its rank (one ds_permute; the
        // other lanes push to lane 63, which nobody reads: k <= 25)
        const bool empty = ((vm >> (lane & 31)) & 1u) != 0 && lane < CELLS;
        const int rank = __popc(vm & ((1u << (lane & 31)) - 1u));
        const int a = __builtin_amdgcn_ds_permute((empty ? rank : 63) << 2, lane);
        my_a[0] = lane < k ? a : -1; (void)act_lds;
        return k;
    }
    template <typename OT> static AZG_DEV void write_obs(const S &s, OT *out, int lane) {
        const float turn = (float)((double)s.turns / 25.0);
#pragma unroll
        for (int e0 = 0; e0 < OBS; e0 += 64) {
            const int e = e0 + lane;
            if (e < OBS) {
                const int plane = e / CELLS, i = e - plane * CELLS;
                const float v = plane < 3 ? (float)((s.b[plane == 0 ? 0 : plane == 1 ? 1 : 2] >> i) & 1) : plane == 3 ? (float)s.player : turn;
                out[e] = (OT)v;
            }
        }
    }
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    static AZG_DEV h8 obs8(const S &s, int lane) {               // the five planes of cell `lane` (< CELLS)
        return (h8){(_Float16)(float)((s.b[0] >> lane) & 1), (_Float16)(float)((s.b[1] >> lane) & 1), (_Float16)(float)((s.b[2] >> lane) & 1),
                    (_Float16)(float)s.player, (_Float16)(float)((double)s.turns / 25.0), (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    }
    static AZG_DEV void write_obs_nhwc8(const S &s, _Float16 *out, int lane) {
        if (lane < CELLS) *reinterpret_cast<h8 *>(out + lane * 8) = obs8(s, lane);
    }
    static AZG_DEV S symmetry(const S &s, int k) { (void)k; return s; }
    static AZG_DEV int sym_action(int a, int k) { (void)k; return a; }
};

}  // namespace azg
