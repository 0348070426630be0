/*
 * meda_oracle.c -- CPU restatement of the reference MEDA environment (env/MEDA/meda.py).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against outputs of the real
 * `MEDAEnv` captured in this container by tools/oracle/gen_meda_golden.py
 * (tests/golden/meda_*.npz, replayed by tests/test_oracle_meda_golden.py).
 *
 * State reduction used here and in the kernels (all exact):
 *   - every droplet and destination is a (2r+1)x(2r+1) box with r = 2 (meda.py:150,208-211), so a
 *     box is its centre (cx, cy); x runs along `length`, y along `width` (meda.py:131-138);
 *   - `distances[i]` is always sqrt(dx^2+dy^2) of the current centre to the destination centre
 *     (meda.py:91-94,282,291) or 0 after the snap (:274-275), so comparisons are done on the
 *     integer d^2: `< 4` <=> d^2 < 16, `< 6` <=> d^2 < 36, `< 9` <=> d^2 < 81, `==`/`<` on sqrt
 *     <=> on d^2;
 *   - `fails` only matters through `fails == 0`; every term is <= 0 so it is zero iff no proximity
 *     punishment ever happened in the episode (meda.py:521-531).  The per-step float value
 *     info['constraints'] = np.sum(punish) is still produced exactly.
 *
 * RNG contract: Philox4x32-10 as in dmfb_oracle.c, streams 1 MOVE, 5 MEDA_TASK,
 * 3 DEGRADE.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MEDA_MAX_AGENTS 32
#define MEDA_R 2
#define MEDA_TASK_MAX_DRAWS (1u << 20)

#define MEDA_OK 0
#define MEDA_ERR_BAD_ARG (-1)
#define MEDA_ERR_TOO_MANY_DROPLETS (-3) /* RuntimeError("Too many droplets in the WxL MEDA array") meda.py:151-154 */
#define MEDA_ERR_BAD_SIZE (-4)          /* assert w > 0 and l > 0                                   meda.py:472 */
#define MEDA_ERR_NO_AGENTS (-5)         /* assert n_agents > 0                                      meda.py:473 */
#define MEDA_ERR_UNSUPPORTED (-6)

typedef struct {
    int cx[MEDA_MAX_AGENTS], cy[MEDA_MAX_AGENTS];   /* droplets[i].x_center/.y_center */
    int gx[MEDA_MAX_AGENTS], gy[MEDA_MAX_AGENTS];   /* destinations[i] */
    int sx[MEDA_MAX_AGENTS], sy[MEDA_MAX_AGENTS];   /* starts[i] */
    uint8_t status[MEDA_MAX_AGENTS];                /* RoutingTaskManager.status (meda.py:159) */
    double *health, *usage, *degrade;               /* (w, l) indexed [y][x]  meda.py:494-504 */
    int step_count;
    int failed;                                     /* fails != 0 */
    uint32_t env_id, rng_step, rng_ep;
} meda_env;

typedef struct {
    int W, L, n, fov, b_degrade, version; /* version 0: MEDAEnv.getOneObs, 2: MEDAEnv_v0_2.getOneObs */
    double per_degrade;
    int max_step;
    uint64_t seed;
    int E;
    meda_env *envs;
} meda_oracle;

static inline void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                 uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
enum { STREAM_MOVE = 1, STREAM_DEGRADE = 3, STREAM_MEDA_TASK = 5 };
static inline void env_philox(const meda_oracle *o, const meda_env *e, uint32_t c1, uint32_t c2, uint32_t stream,
                              uint32_t out[4]) {
    philox4x32_10((uint32_t)o->seed, (uint32_t)(o->seed >> 32), e->env_id, c1, c2, stream << 8, out);
}
static inline double u53(uint32_t hi, uint32_t lo) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}
static inline int below(uint32_t w, int n) { return (int)(((uint64_t)w * (uint32_t)n) >> 32); }
static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int d2(int ax, int ay, int bx, int by) { return (ax - bx) * (ax - bx) + (ay - by) * (ay - by); }

/* ---- task generation: RoutingTaskManager.addTask / _genLegalDroplet (meda.py:175-233) ---- */
typedef struct { uint32_t k; } draw_ctr;
/* getRandomYX (meda.py:224-227): y in [r, width-r-1], x in [r, length-r-1] */
static void random_yx(const meda_oracle *o, const meda_env *e, draw_ctr *c, int *y, int *x) {
    uint32_t w[4];
    env_philox(o, e, e->rng_ep, c->k, STREAM_MEDA_TASK, w);
    c->k++;
    *y = MEDA_R + below(w[0], o->W - 2 * MEDA_R);
    *x = MEDA_R + below(w[1], o->L - 2 * MEDA_R);
}
/* _genLegalDroplet: redraw while any droplet of the SAME list is closer than 1.5*(r+r+2) = 9 */
static void gen_legal(const meda_oracle *o, const meda_env *e, draw_ctr *c, const int *lx, const int *ly, int count,
                      int *ox, int *oy) {
    for (;;) {
        int y, x, ok = 1;
        random_yx(o, e, c, &y, &x);
        for (int j = 0; j < count; ++j)
            if (d2(x, y, lx[j], ly[j]) < 81) { ok = 0; break; }
        if (ok || c->k >= MEDA_TASK_MAX_DRAWS) { *ox = x; *oy = y; return; }
    }
}
static void gen_task(meda_oracle *o, meda_env *e) {
    draw_ctr c = {0};
    for (int i = 0; i < o->n; ++i) {
        gen_legal(o, e, &c, e->sx, e->sy, i, &e->sx[i], &e->sy[i]);
        gen_legal(o, e, &c, e->gx, e->gy, i, &e->gx[i], &e->gy[i]);
        /* while droplets[-1].isDropletOverlap(destinations[-1]) (meda.py:180-182): boxes overlap
         * <=> |dx| <= 2r and |dy| <= 2r */
        while (iabs(e->sx[i] - e->gx[i]) <= 2 * MEDA_R && iabs(e->sy[i] - e->gy[i]) <= 2 * MEDA_R &&
               c.k < MEDA_TASK_MAX_DRAWS)
            gen_legal(o, e, &c, e->gx, e->gy, i, &e->gx[i], &e->gy[i]);
    }
    e->rng_ep++;
}
/* RoutingTaskManager.restart (meda.py:170-173) */
static void restart_task(meda_oracle *o, meda_env *e) {
    for (int i = 0; i < o->n; ++i) { e->cx[i] = e->sx[i]; e->cy[i] = e->sy[i]; e->status[i] = 0; }
}
/* MEDAEnv.updateHealth (meda.py:600-605): only when b_degrade */
static void update_health(meda_oracle *o, meda_env *e) {
    if (!o->b_degrade || !e->health) return;
    for (int c = 0; c < o->W * o->L; ++c)
        if (e->usage[c] > 50.0) { e->health[c] = e->health[c] * e->degrade[c]; e->usage[c] = 0.0; }
}

int meda_oracle_check_cfg(int W, int L, int n) {
    if (W <= 0 || L <= 0) return MEDA_ERR_BAD_SIZE;
    if (n <= 0) return MEDA_ERR_NO_AGENTS;
    if (n > (W / 15) * (L / 15)) return MEDA_ERR_TOO_MANY_DROPLETS; /* meda.py:151-154 */
    if (n > MEDA_MAX_AGENTS || W > 128 || L > 128) return MEDA_ERR_UNSUPPORTED;
    return MEDA_OK;
}

int meda_oracle_create(int W, int L, int n, int fov, int b_degrade, double per_degrade, int with_maps, uint64_t seed,
                       int E, uint32_t env_id0, int version, meda_oracle **out) {
    int rc = meda_oracle_check_cfg(W, L, n);
    if (rc) return rc;
    if (E <= 0 || !out || fov < 1) return MEDA_ERR_BAD_ARG;
    meda_oracle *o = (meda_oracle *)calloc(1, sizeof(*o));
    o->W = W; o->L = L; o->n = n; o->fov = fov; o->b_degrade = b_degrade; o->per_degrade = per_degrade;
    o->max_step = W + L; /* meda.py:492 */
    o->seed = seed; o->E = E; o->version = version;
    o->envs = (meda_env *)calloc((size_t)E, sizeof(meda_env));
    for (int k = 0; k < E; ++k) {
        meda_env *e = &o->envs[k];
        e->env_id = env_id0 + (uint32_t)k;
        gen_task(o, e); /* RoutingTaskManager.__init__: addTask x n (meda.py:156-157) */
        restart_task(o, e);
        if (b_degrade || with_maps) {
            size_t cells = (size_t)W * L;
            e->health = (double *)malloc(cells * 8); e->usage = (double *)malloc(cells * 8); e->degrade = (double *)malloc(cells * 8);
            double per_healthy = 1.0 - per_degrade;
            for (size_t c = 0; c < cells; ++c) {
                e->health[c] = 1.0; e->usage[c] = 0.0; e->degrade[c] = 1.0;
                if (b_degrade) { /* meda.py:497-502 */
                    uint32_t w[4];
                    env_philox(o, e, 0, (uint32_t)c, STREAM_DEGRADE, w);
                    double v = u53(w[0], w[1]) * 0.4 + 0.6;
                    e->degrade[c] = (u53(w[2], w[3]) < per_healthy) ? 1.0 : v;
                }
            }
        }
    }
    *out = o;
    return MEDA_OK;
}

void meda_oracle_destroy(meda_oracle *o) {
    if (!o) return;
    for (int k = 0; k < o->E; ++k) { free(o->envs[k].health); free(o->envs[k].usage); free(o->envs[k].degrade); }
    free(o->envs); free(o);
}

/* MEDAEnv.reset (meda.py:541-550): counters, refresh (new task), [obs], updateHealth */
void meda_oracle_reset(meda_oracle *o, const uint8_t *mask) {
    for (int k = 0; k < o->E; ++k) {
        if (mask && !mask[k]) continue;
        meda_env *e = &o->envs[k];
        e->step_count = 0; e->failed = 0;
        gen_task(o, e);
        restart_task(o, e);
        update_health(o, e); /* after getObs in the reference; obs does not depend on health */
    }
}
/* MEDAEnv.restart (meda.py:552-561): note `fails` is NOT cleared */
void meda_oracle_restart(meda_oracle *o, const uint8_t *mask) {
    for (int k = 0; k < o->E; ++k) {
        if (mask && !mask[k]) continue;
        restart_task(o, &o->envs[k]);
        o->envs[k].step_count = 0;
    }
}
/* starts/destinations int32 [E][n][2] as (x_center, y_center); then restart + fails = 0 (fresh env) */
void meda_oracle_set_task(meda_oracle *o, const int32_t *starts, const int32_t *ends) {
    for (int k = 0; k < o->E; ++k) {
        meda_env *e = &o->envs[k];
        for (int i = 0; i < o->n; ++i) {
            e->sx[i] = starts[(k * o->n + i) * 2]; e->sy[i] = starts[(k * o->n + i) * 2 + 1];
            e->gx[i] = ends[(k * o->n + i) * 2];   e->gy[i] = ends[(k * o->n + i) * 2 + 1];
        }
        restart_task(o, e);
        e->step_count = 0; e->failed = 0;
    }
}
void meda_oracle_get_task(const meda_oracle *o, int32_t *starts, int32_t *ends) {
    for (int k = 0; k < o->E; ++k)
        for (int i = 0; i < o->n; ++i) {
            const meda_env *e = &o->envs[k];
            starts[(k * o->n + i) * 2] = e->sx[i]; starts[(k * o->n + i) * 2 + 1] = e->sy[i];
            ends[(k * o->n + i) * 2] = e->gx[i];   ends[(k * o->n + i) * 2 + 1] = e->gy[i];
        }
}
void meda_oracle_get_state(const meda_oracle *o, int32_t *pos, uint8_t *status, int32_t *step_count, uint8_t *failed) {
    for (int k = 0; k < o->E; ++k) {
        const meda_env *e = &o->envs[k];
        for (int i = 0; i < o->n; ++i) {
            if (pos) { pos[(k * o->n + i) * 2] = e->cx[i]; pos[(k * o->n + i) * 2 + 1] = e->cy[i]; }
            if (status) status[k * o->n + i] = e->status[i];
        }
        if (step_count) step_count[k] = e->step_count;
        if (failed) failed[k] = (uint8_t)e->failed;
    }
}
int meda_oracle_get_map(const meda_oracle *o, int which, double *buf) {
    size_t cells = (size_t)o->W * o->L;
    for (int k = 0; k < o->E; ++k) {
        const meda_env *e = &o->envs[k];
        const double *src = which == 0 ? e->health : which == 1 ? e->usage : e->degrade;
        if (!src) return MEDA_ERR_BAD_ARG;
        memcpy(buf + k * cells, src, cells * 8);
    }
    return MEDA_OK;
}
int meda_oracle_set_map(meda_oracle *o, int which, const double *buf) {
    size_t cells = (size_t)o->W * o->L;
    for (int k = 0; k < o->E; ++k) {
        meda_env *e = &o->envs[k];
        double *dst = which == 0 ? e->health : which == 1 ? e->usage : e->degrade;
        if (!dst) return MEDA_ERR_BAD_ARG;
        memcpy(dst, buf + k * cells, cells * 8);
    }
    return MEDA_OK;
}

/* Droplet.move (meda.py:106-138) on the centre */
static void move_center(int *cx, int *cy, int action, int W, int L) {
    const int r = 3;
    switch (action) {
    case 0: *cy -= r; break;                         /* N  */
    case 1: *cx += r; break;                         /* E  */
    case 2: *cy += r; break;                         /* S  */
    case 3: *cx -= r; break;                         /* W  */
    case 4: *cx += r - 1; *cy -= r - 1; break;       /* NE */
    case 5: *cx += r - 1; *cy += r - 1; break;       /* SE */
    case 6: *cx -= r - 1; *cy += r - 1; break;       /* SW */
    case 7: *cx -= r - 1; *cy -= r - 1; break;       /* NW */
    default: return;                                 /* STALL (8): returns before the clamps */
    }
    if (*cx + MEDA_R >= L) *cx = L - 1 - MEDA_R; else if (*cx - MEDA_R < 0) *cx = MEDA_R;
    if (*cy + MEDA_R >= W) *cy = W - 1 - MEDA_R; else if (*cy - MEDA_R < 0) *cy = MEDA_R;
}

/* getMoveProb (meda.py:302-309): mean health over the footprint, summed row-major y then x */
static double move_prob(const meda_oracle *o, const meda_env *e, int cx, int cy) {
    if (!e->health) return 1.0;
    double prob = 0.0;
    for (int y = cy - MEDA_R; y <= cy + MEDA_R; ++y)
        for (int x = cx - MEDA_R; x <= cx + MEDA_R; ++x) prob += e->health[y * o->L + x];
    return prob / 25.0;
}

/* MEDAEnv.step (meda.py:513-539) + moveDroplets/moveOneDroplet/calPunish (:241-330) + addUsage (:591-598) */
static void step_env(meda_oracle *o, meda_env *e, const int32_t *actions, const double *uniforms, double *rewards,
                     uint8_t *dones, double *fail_out, uint8_t *success_out) {
    int n = o->n;
    e->step_count += 1;
    for (int i = 0; i < n; ++i) {
        if (e->status[i]) { rewards[i] = 0.0; continue; }
        int old = d2(e->cx[i], e->cy[i], e->gx[i], e->gy[i]);
        if (old < 16) { /* already achieved goal: snap, status turns True one step after arrival */
            e->cx[i] = e->gx[i]; e->cy[i] = e->gy[i];
            rewards[i] = 0.0; e->status[i] = 1;
        } else {
            double prob = move_prob(o, e, e->cx[i], e->cy[i]);
            double u;
            if (uniforms) u = uniforms[i];
            else { uint32_t w[4]; env_philox(o, e, e->rng_step, (uint32_t)i, STREAM_MOVE, w); u = u53(w[0], w[1]); }
            if (u <= prob) move_center(&e->cx[i], &e->cy[i], actions[i], o->W, o->L);
            int nd = d2(e->cx[i], e->cy[i], e->gx[i], e->gy[i]);
            if (nd < 16) rewards[i] = 0.0;
            else if (nd == old && actions[i] == 8) rewards[i] = -0.2;
            else if (nd < old) rewards[i] = -0.08;
            else rewards[i] = -0.4;
        }
    }
    e->rng_step++;
    /* calPunish (meda.py:321-330): pairs with centre distance < 1.5*(2+2) = 6 */
    double punish[MEDA_MAX_AGENTS];
    int any = 0;
    for (int i = 0; i < n; ++i) punish[i] = 0.0;
    for (int i = 0; i < n - 1; ++i)
        for (int j = i + 1; j < n; ++j)
            if (d2(e->cx[i], e->cy[i], e->cx[j], e->cy[j]) < 36) { punish[i] -= 0.6; punish[j] -= 0.6; any = 1; }
    /* fail = np.sum(punish): numpy pairwise order */
    double fail;
    if (n < 8) { fail = 0.0; for (int i = 0; i < n; ++i) fail += punish[i]; }
    else {
        double q[8];
        for (int j = 0; j < 8; ++j) q[j] = punish[j];
        int m = n - (n % 8), i;
        for (i = 8; i < m; i += 8) for (int j = 0; j < 8; ++j) q[j] += punish[i + j];
        fail = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
        for (; i < n; ++i) fail += punish[i];
    }
    for (int i = 0; i < n; ++i) rewards[i] += punish[i];
    if (any) e->failed = 1;
    int all = 1;
    for (int i = 0; i < n; ++i) all &= e->status[i];
    if (all) {
        for (int i = 0; i < n; ++i) rewards[i] = rewards[i] + 3.0;
        if (!e->failed) for (int i = 0; i < n; ++i) rewards[i] = rewards[i] + 3.0;
    }
    uint8_t success = 0;
    if (e->step_count < o->max_step) {
        if (all && !e->failed) success = 1;
        for (int i = 0; i < n; ++i) dones[i] = e->status[i];
        if (e->usage) /* addUsage: footprint of every agent that is not done (meda.py:591-598) */
            for (int i = 0; i < n; ++i)
                if (!e->status[i])
                    for (int y = e->cy[i] - MEDA_R; y <= e->cy[i] + MEDA_R; ++y)
                        for (int x = e->cx[i] - MEDA_R; x <= e->cx[i] + MEDA_R; ++x) e->usage[y * o->L + x] += 1.0;
    } else {
        for (int i = 0; i < n; ++i) dones[i] = 1;
    }
    *fail_out = fail;
    *success_out = success;
}

void meda_oracle_step(meda_oracle *o, const int32_t *actions, const double *uniforms, double *rewards, uint8_t *dones,
                      double *fail, uint8_t *success) {
    int n = o->n;
    for (int k = 0; k < o->E; ++k)
        step_env(o, &o->envs[k], actions + (size_t)k * n, uniforms ? uniforms + (size_t)k * n : NULL,
                 rewards + (size_t)k * n, dones + (size_t)k * n, fail + k, success + k);
}

/* MEDAEnv.getOneObs (meda.py:613-674): 4 layers [c][y-oy][x-ox] + (dx, dy); values are small
 * integers (the reference returns float64), emitted as int8. */
static void put_box(signed char *layer, int fov, int cx, int cy, int ox, int oy, int val, int clip) {
    for (int y = cy - MEDA_R; y <= cy + MEDA_R; ++y)
        for (int x = cx - MEDA_R; x <= cx + MEDA_R; ++x) {
            int nx = x - ox, ny = y - oy;
            if (clip) {
                nx = nx < 0 ? 0 : (nx > fov - 1 ? fov - 1 : nx);
                ny = ny < 0 ? 0 : (ny > fov - 1 ? fov - 1 : ny);
                layer[ny * fov + nx] = (signed char)val;
            } else if (nx >= 0 && nx < fov && ny >= 0 && ny < fov) {
                layer[ny * fov + nx] = (signed char)val;
            }
        }
}
static void one_obs(const meda_oracle *o, const meda_env *e, int a, signed char *out) {
    int fov = o->fov, ff = fov * fov, n = o->n;
    memset(out, 0, (size_t)(4 * ff + 2));
    int ox = e->cx[a] - fov / 2, oy = e->cy[a] - fov / 2;
    put_box(out, fov, e->cx[a], e->cy[a], ox, oy, a + 1, 0);
    put_box(out + ff, fov, e->gx[a], e->gy[a], ox, oy, a + 1, 0);
    for (int j = 0; j < n; ++j) if (j != a) put_box(out + 2 * ff, fov, e->cx[j], e->cy[j], ox, oy, j + 1, 0);
    for (int j = 0; j < n; ++j) if (j != a) put_box(out + 3 * ff, fov, e->gx[j], e->gy[j], ox, oy, j + 1, 1);
    out[4 * ff] = (signed char)(e->gx[a] - e->cx[a]);
    out[4 * ff + 1] = (signed char)(e->gy[a] - e->cy[a]);
}
/* MEDAEnv_v0_2.getOneObs (meda.py:850-897): 3 int8 layers + zoomed direction.  Quirks kept: the
 * boundary layer indexes its FIRST axis with the x bound and tests x against `width` (meda.py:880-890). */
static void one_obs_v02(const meda_oracle *o, const meda_env *e, int a, signed char *out) {
    int fov = o->fov, ff = fov * fov, n = o->n, hf = fov / 2;
    memset(out, 0, (size_t)(3 * ff + 2));
    int cx = e->cx[a], cy = e->cy[a];
    int ox = cx - hf, oy = cy - hf;
    int observed[MEDA_MAX_AGENTS];
    for (int j = 0; j < n; ++j) {
        observed[j] = 0;
        for (int y = e->cy[j] - MEDA_R; y <= e->cy[j] + MEDA_R; ++y)
            for (int x = e->cx[j] - MEDA_R; x <= e->cx[j] + MEDA_R; ++x) {
                int nx = x - ox, ny = y - oy;
                if (nx >= 0 && nx < fov && ny >= 0 && ny < fov) { out[ny * fov + nx] = (signed char)(j + 1); observed[j] = 1; }
            }
    }
    /* `for idx in observed` iterates a CPython set of small ints (meda.py:867-872): slot order of the
     * hash table.  hash(i) = i; the table has 8 slots until the 5th distinct insertion, which resizes
     * it to 32 (setobject.c: fill*5 >= mask*3 -> resize to used*4), after which slot == value, i.e.
     * ascending order.  With at most 4 members the order is by slot (v & 7), collisions re-probed with
     * i = (5*i + 1) & 7 (no linear probes at mask 7, perturb = v >> 5 = 0).  Later writers overwrite
     * earlier ones where clipped goals coincide, so the order is observable. */
    int order[MEDA_MAX_AGENTS], cnt = 0, members = 0;
    for (int j = 0; j < n; ++j) members += observed[j];
    if (members >= 5) {
        for (int j = 0; j < n; ++j) if (observed[j]) order[cnt++] = j;
    } else {
        int slots[8];
        for (int k = 0; k < 8; ++k) slots[k] = -1;
        for (int j = 0; j < n; ++j)
            if (observed[j]) { int i = j & 7; while (slots[i] >= 0) i = (i * 5 + 1) & 7; slots[i] = j; }
        for (int k = 0; k < 8; ++k) if (slots[k] >= 0) order[cnt++] = slots[k];
    }
    for (int t = 0; t < cnt; ++t) {
        int j = order[t];
        if (j != a) put_box(out + ff, fov, e->gx[j], e->gy[j], ox, oy, j + 1, 1);
    }
    int left = hf - cx, right = hf - (o->W - 1 - cx);
    if (left > 0) { for (int r = 0; r < left && r < fov; ++r) for (int c = 0; c < fov; ++c) out[2 * ff + r * fov + c] = 1; }
    else if (right > 0) { for (int r = (fov - right < 0 ? 0 : fov - right); r < fov; ++r) for (int c = 0; c < fov; ++c) out[2 * ff + r * fov + c] = 1; }
    int up = hf - cy, down = hf - (o->L - 1 - cy);
    if (up > 0) { for (int r = 0; r < fov; ++r) for (int c = 0; c < up && c < fov; ++c) out[2 * ff + r * fov + c] = 1; }
    else if (down > 0) { for (int r = 0; r < fov; ++r) for (int c = (fov - down < 0 ? 0 : fov - down); c < fov; ++c) out[2 * ff + r * fov + c] = 1; }
    /* np.array([round(dy/(width/30)), round(dx/(length/30))], dtype=int8): python round = half to even */
    out[3 * ff] = (signed char)(int)rint((double)(e->gy[a] - cy) / ((double)o->W / 30.0));
    out[3 * ff + 1] = (signed char)(int)rint((double)(e->gx[a] - cx) / ((double)o->L / 30.0));
}

int meda_oracle_obs_len(const meda_oracle *o) { return (o->version == 2 ? 3 : 4) * o->fov * o->fov + 2; }

void meda_oracle_observe(const meda_oracle *o, signed char *obs) {
    int len = meda_oracle_obs_len(o);
    for (int k = 0; k < o->E; ++k)
        for (int i = 0; i < o->n; ++i) {
            if (o->version == 2) one_obs_v02(o, &o->envs[k], i, obs + ((size_t)k * o->n + i) * len);
            else one_obs(o, &o->envs[k], i, obs + ((size_t)k * o->n + i) * len);
        }
}
