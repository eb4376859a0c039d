/*
 * azg_brandubh_ref.c -- ORACLE rules of the 7x7 tafl env (TEST INFRASTRUCTURE ONLY; see azg_oracle.h).
 *
 * Literal restatement of alphazero/envs/brandubh/fastafl.pyx (Game), fastafl/cengine.pyx (Board) and
 * boardgame/board.pyx (BaseBoard), with the reference's own control flow: piece lists in row-major order,
 * direction tuples in the reference's order, the recursive surround check with its list-based bookkeeping.
 * Board options of variants.brandubh_args: king_two_sided_capture=True, move_over_throne=True,
 * king_can_enter_throne=False (fastafl/variants.py:22, cengine.pyx:54).
 *
 * azo_state: cells[y*7+x] = Board._state[y, x]; player/turns = Game._player/_turns (== Board.num_turns);
 * aux[0] = Board._king_captured.
 */
#include "azg_oracle.h"
#include <string.h>

enum { P_ATT = 1, P_DEF = 2, P_KING = 3, P_KING_THRONE = 7, P_KING_ESCAPE = 8, T_NORMAL = 0, T_THRONE = 4, T_ESCAPE = 5 };  /* cengine.pyx:24-32 */
#define BW 7
#define BH 7
#define DRAW_MOVE_COUNT 100                     /* fastafl.pyx:43 */
static const int DIRS[4][2] = { {0, 1}, {1, 0}, {0, -1}, {-1, 0} };   /* cengine.pyx:46 (dx, dy) */
#define AT(s, x, y) ((s)->cells[(y) * BW + (x)])

static int in_bounds(int x, int y) { return x >= 0 && x <= BW - 1 && y >= 0 && y <= BH - 1; }   /* board.pyx:171-172 */
static int is_king_val(int v) { return v == P_KING || v == P_KING_THRONE || v == P_KING_ESCAPE; }
static int in_attackers(int v) { return v == P_ATT || is_king_val(v); }                       /* ATTACKERS cengine.pyx:42 */

void azo_br_init(azo_state *s) {                 /* variants.py:13-19 */
    static const char *rows[7] = { "5002005", "0002000", "0001000", "2217122", "0001000", "0002000", "5002005" };
    memset(s, 0, sizeof(*s));
    for (int y = 0; y < 7; y++) for (int x = 0; x < 7; x++) AT(s, x, y) = (int8_t)(rows[y][x] - '0');
}

static int to_play(const azo_state *s) { return 2 - (s->turns % 2); }      /* cengine.pyx:330-331 + board.pyx:343-344 */

/* Board._is_valid cengine.pyx:90-107 */
static int is_valid(const azo_state *s, int x, int y, int is_king) {
    if (!in_bounds(x, y)) return 0;
    int v = AT(s, x, y);
    if (v == T_NORMAL) return 1;
    if (v == T_ESCAPE) return is_king;
    return 0;                                   /* king_can_enter_throne is False; falls off the end -> False */
}

/* fastafl.pyx:66-79 get_action */
static int get_action(int x, int y, int nx, int ny) {
    int move_type;
    if (x - nx == 0) move_type = ny < y ? ny : ny - 1;
    else { move_type = BH + nx - 1; if (nx >= x) move_type -= 1; }
    return (BW + BH - 2) * (x + y * BW) + move_type;
}
/* fastafl.pyx:48-63 get_move */
static void get_move(int action, int *sx, int *sy, int *nx, int *ny) {
    int size = BW + BH - 2, move_type = action % size, a = action / size;
    *sx = a % BW; *sy = a / BW;
    if (move_type < BH - 1) { *nx = *sx; *ny = move_type; if (move_type >= *sy) *ny += 1; }
    else { *nx = move_type - BH + 1; if (*nx >= *sx) *nx += 1; *ny = *sy; }
}

/* Game.valid_moves fastafl.pyx:171-178 + Board.legal_moves cengine.pyx:109-132 + BaseBoard._iter_pieces/get_squares */
void azo_br_valid_moves(const azo_state *s, uint8_t *valid) {
    memset(valid, 0, 588);
    int team = to_play(s);
    for (int y = 0; y < BH; y++) for (int x = 0; x < BW; x++) {          /* np.where row-major */
        int v = AT(s, x, y);
        int mine = team == P_ATT ? in_attackers(v) : v == P_DEF;         /* _get_team cengine.pyx:276-283 */
        if (!mine) continue;
        int is_king = is_king_val(v);
        for (int d = 0; d < 4; d++) {
            int cx = x + DIRS[d][0], cy = y + DIRS[d][1];
            int is_throne = in_bounds(cx, cy) && AT(s, cx, cy) == T_THRONE;          /* move_over_throne */
            while (is_throne || is_valid(s, cx, cy, is_king)) {
                if (!is_throne) valid[get_action(x, y, cx, cy)] = 1;                /* :125-127 */
                cx += DIRS[d][0]; cy += DIRS[d][1];
                is_throne = in_bounds(cx, cy) && AT(s, cx, cy) == T_THRONE;
            }
        }
    }
}

/* Board.remove_piece cengine.pyx:311-326 (raise_no_piece=False) */
static int remove_piece(azo_state *s, int x, int y) {
    int dest = AT(s, x, y), nv = T_NORMAL, piece = dest;
    if (dest == P_KING_THRONE) { nv = T_THRONE; piece = P_KING; }
    else if (dest == P_KING_ESCAPE) { nv = T_ESCAPE; piece = P_KING; }
    AT(s, x, y) = (int8_t)nv;
    return piece;
}
/* Board.add_piece cengine.pyx:293-309 (_check_valid=False) */
static int add_piece(azo_state *s, int x, int y, int piece) {
    int dest = AT(s, x, y);
    if (dest == T_ESCAPE || dest == T_THRONE) {
        if (piece == P_KING) { AT(s, x, y) = (int8_t)(piece + dest); return 0; }
        return -1;                              /* PositionError */
    }
    AT(s, x, y) = (int8_t)piece;
    return 0;
}

/* Board._check_capture cengine.pyx:172-197 */
static void check_capture(azo_state *s, int mx, int my) {
    int piece_val = AT(s, mx, my);
    int friendly_is_attackers = in_attackers(piece_val);
    int enemy = piece_val != P_KING ? 3 - piece_val : P_DEF;
    for (int d = 0; d < 4; d++) {
        int ex = mx + DIRS[d][0], ey = my + DIRS[d][1];
        if (!in_bounds(ex, ey)) continue;
        int value = AT(s, ex, ey);
        int do_capture = value == P_KING;       /* king_two_sided_capture and value == piece_king */
        if (value == enemy || do_capture) {
            int fx = ex + DIRS[d][0], fy = ey + DIRS[d][1];
            if (!in_bounds(fx, fy)) continue;
            value = AT(s, fx, fy);
            int friendly = friendly_is_attackers ? in_attackers(value) : value == piece_val;
            if (friendly || value == T_THRONE || value == T_ESCAPE) {
                if (do_capture) s->aux[0] = 1;  /* _king_captured */
                else AT(s, ex, ey) = T_NORMAL;
            }
        }
    }
}

typedef struct { int x[256], y[256], n; } sqlist;
static int in_list(const sqlist *l, int x, int y) { for (int i = 0; i < l->n; i++) if (l->x[i] == x && l->y[i] == y) return 1; return 0; }
static void push(sqlist *l, int x, int y) { if (l->n < 256) { l->x[l->n] = x; l->y[l->n] = y; l->n++; } }

static int in_enemy(int v, int enemy_is_attackers) { return enemy_is_attackers ? in_attackers(v) : v == P_DEF; }

/* Board.__recurse_check cengine.pyx:204-226; returns is_captured, *exit_recurse */
static int recurse_check(const azo_state *s, int x, int y, sqlist *checked, int enemy_is_attackers, int *exit_recurse) {
    push(checked, x, y);
    int sx[4], sy[4], ns = 0;
    for (int d = 0; d < 4; d++) { int nx = x + DIRS[d][0], ny = y + DIRS[d][1]; if (in_bounds(nx, ny)) { sx[ns] = nx; sy[ns] = ny; ns++; } }
    int all_blocked = 1;
    for (int i = 0; i < ns; i++) if (AT(s, sx[i], sy[i]) == T_NORMAL) all_blocked = 0;
    if (!all_blocked) { *exit_recurse = 1; return 0; }
    int lx[4], ly[4], nl = 0;                   /* the comprehension is evaluated once, before the loop */
    for (int i = 0; i < ns; i++) if (in_enemy(AT(s, sx[i], sy[i]), enemy_is_attackers) && !in_list(checked, sx[i], sy[i])) { lx[nl] = sx[i]; ly[nl] = sy[i]; nl++; }
    int all_captured = 1; *exit_recurse = 0;
    for (int i = 0; i < nl; i++) {
        int ex = 0;
        int cap = recurse_check(s, lx[i], ly[i], checked, enemy_is_attackers, &ex);
        if (!cap) all_captured = 0;
        *exit_recurse = ex;
        if (ex) break;
    }
    return all_captured;
}

/* Board._check_surround cengine.pyx:228-247 */
static void check_surround(azo_state *s, int mx, int my) {
    int enemy_is_attackers = AT(s, mx, my) == P_DEF;    /* _get_team(piece, enemy=True) cengine.pyx:276-283 */
    int stx[4], sty[4], nst = 0;
    for (int d = 0; d < 4; d++) {
        int nx = mx + DIRS[d][0], ny = my + DIRS[d][1];
        if (in_bounds(nx, ny) && in_enemy(AT(s, nx, ny), enemy_is_attackers)) { stx[nst] = nx; sty[nst] = ny; nst++; }
    }
    if (!nst) return;
    sqlist checked_squares; checked_squares.n = 0;
    for (int i = 0; i < nst; i++) {
        if (in_list(&checked_squares, stx[i], sty[i])) continue;
        sqlist to_capture; to_capture.n = 0;
        int ex = 0;
        if (recurse_check(s, stx[i], sty[i], &to_capture, enemy_is_attackers, &ex)) {
            for (int j = 0; j < to_capture.n; j++) {
                if (is_king_val(AT(s, to_capture.x[j], to_capture.y[j]))) s->aux[0] = 1;
                else remove_piece(s, to_capture.x[j], to_capture.y[j]);
            }
        }
        for (int j = 0; j < to_capture.n; j++) push(&checked_squares, to_capture.x[j], to_capture.y[j]);
    }
}

/* Game.play_action fastafl.pyx:180-184 + Board.move cengine.pyx:249-272 (no validity / win checks) */
int azo_br_play(azo_state *s, int action) {
    int sx, sy, nx, ny;
    get_move(action, &sx, &sy, &nx, &ny);
    if (add_piece(s, nx, ny, remove_piece(s, sx, sy)) != 0) return -1;
    check_capture(s, nx, ny);
    check_surround(s, nx, ny);
    s->turns += 1;                              /* Board.num_turns and Game._turns advance together */
    s->player = (s->player + 1) % 2;
    return 0;
}

/* Board._has_legals_check cengine.pyx:134-141 */
static int has_legals_check(const azo_state *s, int x, int y) {
    int is_king = is_king_val(AT(s, x, y));
    for (int d = 0; d < 4; d++) if (is_valid(s, x + DIRS[d][0], y + DIRS[d][1], is_king)) return 1;
    return 0;
}
/* BaseBoard.has_legal_moves board.pyx:197-221 (pieces=(), piece_type given) */
static int has_legal_moves(const azo_state *s, int team) {
    static const int att[4] = { P_ATT, P_KING, P_KING_THRONE, P_KING_ESCAPE };
    static const int def[1] = { P_DEF };
    const int *types = team == P_ATT ? att : def; int nt = team == P_ATT ? 4 : 1;
    for (int t = 0; t < nt; t++)
        for (int y = 0; y < BH; y++) for (int x = 0; x < BW; x++)
            if (AT(s, x, y) == types[t] && has_legals_check(s, x, y)) return 1;
    return 0;
}

/* Game.win_state fastafl.pyx:186-199 + Board.get_winner cengine.pyx:163-169 */
void azo_br_win_state(const azo_state *s, uint8_t *ws) {
    ws[0] = ws[1] = ws[2] = 0;
    if (s->turns >= DRAW_MOVE_COUNT) { ws[2] = 1; return; }
    int king_escaped = 0;
    for (int i = 0; i < 49; i++) if (s->cells[i] == P_KING_ESCAPE) king_escaped = 1;      /* :144-147 */
    int king_captured = s->aux[0] != 0;                                                     /* :152-160 two-sided rule */
    int winner = 0;
    if (king_escaped || !has_legal_moves(s, P_DEF)) winner = P_ATT;
    else if (king_captured || !has_legal_moves(s, P_ATT)) winner = P_DEF;
    if (winner != 0) ws[2 - winner] = 1;
}

/* Game.observation fastafl.pyx:84-121,205-211 (integer divisions under cdivision: SURVEY.md Q18) */
void azo_br_observation(const azo_state *s, float *obs) {
    float colour = (float)(2 - to_play(s) / (2 - 1));
    float turn_no = (float)(s->turns / DRAW_MOVE_COUNT);
    for (int i = 0; i < 49; i++) {
        int v = s->cells[i];
        obs[0 * 49 + i] = v == 2 ? 1.f : 0.f;
        obs[1 * 49 + i] = v == 1 ? 1.f : 0.f;
        obs[2 * 49 + i] = (v == 3 || v == 7 || v == 8) ? 1.f : 0.f;
        obs[3 * 49 + i] = colour;
        obs[4 * 49 + i] = turn_no;
    }
}

/* Game.symmetries fastafl.pyx:213-256: index k = (i-1)*2 + flip, i = 1..4 rotations (np.rot90), flip = np.fliplr;
 * the policy is permuted with the reference's own coordinate loop. */
void azo_br_symmetry(const azo_state *s, const float *pi, int k, azo_state *so, float *pio) {
    int i = k / 2 + 1, flip = k % 2;
    *so = *s;
    int8_t cur[49], nxt[49];
    memcpy(cur, s->cells, 49);
    for (int r = 0; r < i; r++) {               /* np.rot90 (counter-clockwise): new[row][col] = old[col][W-1-row] */
        for (int row = 0; row < 7; row++) for (int col = 0; col < 7; col++) nxt[row * 7 + col] = cur[col * 7 + (6 - row)];
        memcpy(cur, nxt, 49);
    }
    if (flip) { for (int row = 0; row < 7; row++) for (int col = 0; col < 7; col++) nxt[row * 7 + col] = cur[row * 7 + (6 - col)]; memcpy(cur, nxt, 49); }
    memcpy(so->cells, cur, 49);
    for (int a = 0; a < 588; a++) pio[a] = 0.f;
    for (int a = 0; a < 588; a++) {
        int x, y, nx, ny;
        get_move(a, &x, &y, &nx, &ny);
        for (int r = 0; r < i; r++) {           /* :241-246 */
            int tx = x, tnx = nx;
            x = BW - 1 - y; nx = BW - 1 - ny; y = tx; ny = tnx;
        }
        if (flip) { x = BW - 1 - x; nx = BW - 1 - nx; }
        pio[get_action(x, y, nx, ny)] = pi[a];
    }
}
