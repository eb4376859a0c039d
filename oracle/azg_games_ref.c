/*
 * azg_games_ref.c -- ORACLE game rules (TEST INFRASTRUCTURE ONLY; see azg_oracle.h).
 *
 * connect4 follows alphazero/envs/connect4/connect4.pyx and Connect4Logic.pyx literally: an int board of
 * 1 / -1 / 0, scanned cell by cell exactly in the reference's order.  (The product uses bitboards instead.)
 * brandubh lives in azg_brandubh_ref.c, the 3-player env in azg_trimok_ref.c.
 */
#include "azg_oracle.h"
#include <string.h>

/* implemented in the per-game files */
void azo_br_init(azo_state *s);
int  azo_br_play(azo_state *s, int action);
void azo_br_valid_moves(const azo_state *s, uint8_t *valid);
void azo_br_win_state(const azo_state *s, uint8_t *ws);
void azo_br_observation(const azo_state *s, float *obs);
void azo_br_symmetry(const azo_state *s, const float *pi, int k, azo_state *so, float *pio);
void azo_tm_init(azo_state *s);
int  azo_tm_play(azo_state *s, int action);
void azo_tm_valid_moves(const azo_state *s, uint8_t *valid);
void azo_tm_win_state(const azo_state *s, uint8_t *ws);
void azo_tm_observation(const azo_state *s, float *obs);
void azo_tm_symmetry(const azo_state *s, const float *pi, int k, azo_state *so, float *pio);

/* ---- connect4 --------------------------------------------------------------------------------- */
#define C4_H 6
#define C4_W 7
#define C4_WIN 4
#define C4_AT(s, r, c) ((s)->cells[(r) * C4_W + (c)])

static void c4_init(azo_state *s) { memset(s, 0, sizeof(*s)); }

/* Connect4Logic.pyx:40-47 add_stone + connect4.pyx:63-66 play_action */
static int c4_play(azo_state *s, int col) {
    int stone = s->player == 0 ? 1 : -1;            /* (1, -1)[self.player]  connect4.pyx:65 */
    for (int r = 0; r < C4_H; r++) {
        if (C4_AT(s, (C4_H - 1) - r, col) == 0) {
            C4_AT(s, (C4_H - 1) - r, col) = (int8_t)stone;
            s->player = (s->player + 1) % 2;       /* Game.py:73-79 _update_turn */
            s->turns += 1;
            return 0;
        }
    }
    return -1;                                      /* ValueError  Connect4Logic.pyx:47 */
}

/* Connect4Logic.pyx:49-57 */
static void c4_valid(const azo_state *s, uint8_t *valid) {
    for (int c = 0; c < C4_W; c++) valid[c] = C4_AT(s, 0, c) == 0;
}

/* Connect4Logic.pyx:59-110 get_win_state + connect4.pyx:68-81 win_state */
static void c4_win_state(const azo_state *s, uint8_t *ws) {
    ws[0] = ws[1] = ws[2] = 0;
    static const int players[2] = { 1, -1 };
    for (int pi = 0; pi < 2; pi++) {
        int player = players[pi], total;
        for (int r = 0; r < C4_H; r++) {            /* rows :64-72 */
            total = 0;
            for (int c = 0; c < C4_W; c++) {
                if (C4_AT(s, r, c) == player) total++; else total = 0;
                if (total == C4_WIN) goto won;
            }
        }
        for (int c = 0; c < C4_W; c++) {            /* columns :74-82 */
            total = 0;
            for (int r = 0; r < C4_H; r++) {
                if (C4_AT(s, r, c) == player) total++; else total = 0;
                if (total == C4_WIN) goto won;
            }
        }
        for (int r = 0; r < C4_H - C4_WIN + 1; r++) { /* diagonals :84-101 */
            for (int c = 0; c < C4_W - C4_WIN + 1; c++) {
                int good = 1;
                for (int x = 0; x < C4_WIN; x++) if (C4_AT(s, r + x, c + x) != player) { good = 0; break; }
                if (good) goto won;
            }
            for (int c = C4_WIN - 1; c < C4_W; c++) {
                int good = 1;
                for (int x = 0; x < C4_WIN; x++) if (C4_AT(s, r + x, c - x) != player) { good = 0; break; }
                if (good) goto won;
            }
        }
        continue;
    won:
        ws[player == 1 ? 0 : 1] = 1;                /* connect4.pyx:74-79 */
        return;
    }
    int nvalid = 0;                                 /* draw :104-105, index -1 -> slot 2 (connect4.pyx:73,79) */
    for (int c = 0; c < C4_W; c++) nvalid += C4_AT(s, 0, c) == 0;
    if (nvalid == 0) ws[2] = 1;
}

/* connect4.pyx:83-91 (MULTI_PLANE_OBSERVATION) */
static void c4_observation(const azo_state *s, float *obs) {
    float turn = (float)((double)s->turns / 42.0);  /* np.full_like(..., turns / MAX_TURNS, dtype=float32) */
    for (int i = 0; i < C4_H * C4_W; i++) {
        obs[0 * 42 + i] = s->cells[i] == 1 ? 1.f : 0.f;
        obs[1 * 42 + i] = s->cells[i] == -1 ? 1.f : 0.f;
        obs[2 * 42 + i] = (float)s->player;
        obs[3 * 42 + i] = turn;
    }
}

/* connect4.pyx:96-99: [(self, pi), (mirror columns, pi[::-1])] */
static void c4_symmetry(const azo_state *s, const float *pi, int k, azo_state *so, float *pio) {
    *so = *s;
    if (k == 0) { for (int a = 0; a < C4_W; a++) pio[a] = pi[a]; return; }
    for (int r = 0; r < C4_H; r++) for (int c = 0; c < C4_W; c++) C4_AT(so, r, c) = C4_AT(s, r, C4_W - 1 - c);
    for (int a = 0; a < C4_W; a++) pio[a] = pi[C4_W - 1 - a];
}

/* ---- dispatch ---------------------------------------------------------------------------------- */
int azo_game_info_get(int game, azo_game_info *o) {
    switch (game) {
    case AZO_GAME_CONNECT4: *o = (azo_game_info){ 7, 4, 6, 7, 2, 1, 42, 2, 42 }; return 0;     /* connect4.pyx:11-17 */
    case AZO_GAME_BRANDUBH: *o = (azo_game_info){ 588, 5, 7, 7, 2, 1, 100, 8, 49 }; return 0;  /* fastafl.pyx:34-41 + SURVEY Q19 */
    case AZO_GAME_TRIMOK:   *o = (azo_game_info){ 25, 5, 5, 5, 3, 1, 25, 1, 25 }; return 0;    /* build-defined 3-player env */
    }
    return -1;
}
void azo_game_init(int game, azo_state *s) {
    if (game == AZO_GAME_CONNECT4) c4_init(s); else if (game == AZO_GAME_BRANDUBH) azo_br_init(s); else azo_tm_init(s);
}
int azo_game_play(int game, azo_state *s, int a) {
    if (game == AZO_GAME_CONNECT4) return c4_play(s, a);
    if (game == AZO_GAME_BRANDUBH) return azo_br_play(s, a);
    return azo_tm_play(s, a);
}
void azo_game_valid_moves(int game, const azo_state *s, uint8_t *v) {
    if (game == AZO_GAME_CONNECT4) c4_valid(s, v); else if (game == AZO_GAME_BRANDUBH) azo_br_valid_moves(s, v); else azo_tm_valid_moves(s, v);
}
void azo_game_win_state(int game, const azo_state *s, uint8_t *ws) {
    if (game == AZO_GAME_CONNECT4) c4_win_state(s, ws); else if (game == AZO_GAME_BRANDUBH) azo_br_win_state(s, ws); else azo_tm_win_state(s, ws);
}
void azo_game_observation(int game, const azo_state *s, float *obs) {
    if (game == AZO_GAME_CONNECT4) c4_observation(s, obs); else if (game == AZO_GAME_BRANDUBH) azo_br_observation(s, obs); else azo_tm_observation(s, obs);
}
void azo_game_symmetry(int game, const azo_state *s, const float *pi, int k, azo_state *so, float *pio) {
    if (game == AZO_GAME_CONNECT4) c4_symmetry(s, pi, k, so, pio);
    else if (game == AZO_GAME_BRANDUBH) azo_br_symmetry(s, pi, k, so, pio);
    else azo_tm_symmetry(s, pi, k, so, pio);
}
