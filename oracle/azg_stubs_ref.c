/* temporary stubs until the brandubh / trimok oracle rules land (TEST INFRASTRUCTURE ONLY) */
#include "azg_oracle.h"
#include <string.h>
void azo_tm_init(azo_state *s) { memset(s, 0, sizeof(*s)); }
int  azo_tm_play(azo_state *s, int a) { (void)s; (void)a; return -1; }
void azo_tm_valid_moves(const azo_state *s, uint8_t *v) { (void)s; (void)v; }
void azo_tm_win_state(const azo_state *s, uint8_t *w) { (void)s; (void)w; }
void azo_tm_observation(const azo_state *s, float *o) { (void)s; (void)o; }
void azo_tm_symmetry(const azo_state *s, const float *pi, int k, azo_state *so, float *pio) { (void)s; (void)pi; (void)k; (void)so; (void)pio; }
