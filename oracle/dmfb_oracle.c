/*
 * dmfb_oracle.c -- CPU restatement of the reference DMFB environment.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP kernels in
 * marl_dmfb_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path never does.
 *
 * Parity pin: the reference ships no tests, so this oracle is
 * pinned against outputs of the reference itself, captured in this container by
 * tools/oracle/gen_dmfb_golden.py and committed under tests/golden/dmfb_*.npz
 * (tests/test_oracle_dmfb_golden.py replays them).
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  The per-droplet loop structure of the reference is kept on
 * purpose: one env at a time, droplets moved in index order.
 *
 * Randomness: the reference draws from Python's global MT stream
 * (env/DMFB/dmfb.py:335) and numpy's legacy global stream (:209-210, :159-161).
 * Neither is reproducible for thousands of lock-step envs, so the build defines a
 * counter-based stream (Philox4x32-10, see DESIGN.md "RNG contract") that this
 * oracle and the HIP kernels implement independently.  For parity against the
 * reference every draw can be injected instead (uniforms argument, set_task,
 * set_health/...).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DMFB_MAX_AGENTS 32
#define DMFB_MAX_BLOCKS 64
#define DMFB_TASK_MAX_ATTEMPTS (1u << 22)

/* error codes (mirrors include/dmfb_vec.h) */
#define DMFB_OK 0
#define DMFB_ERR_BAD_ARG (-1)
#define DMFB_ERR_FOV_TOO_LARGE (-2)     /* RuntimeError('Fov is too large')        dmfb.py:139-140 */
#define DMFB_ERR_TOO_MANY_DROPLETS (-3) /* TypeError('Too many droplets for DMFB') dmfb.py:144-146 */
#define DMFB_ERR_CHIP_TOO_SMALL (-4)    /* assert width >= 5 and length >= 5       dmfb.py:489 */
#define DMFB_ERR_NO_AGENTS (-5)         /* assert n_agents > 0                     dmfb.py:490 */
#define DMFB_ERR_UNSUPPORTED (-6)
#define DMFB_ERR_BAD_ACTION (-7)        /* TypeError('action is illegal')          dmfb.py:116 */

typedef struct {
    int W, L, n, fov, n_blocks, stall, b_degrade;
    double per_degrade;
    int max_step;
    uint64_t seed;
} dmfb_cfg;

typedef struct {
    int x[DMFB_MAX_AGENTS], y[DMFB_MAX_AGENTS];       /* Droplet.x/.y              dmfb.py:76-77 */
    int gx[DMFB_MAX_AGENTS], gy[DMFB_MAX_AGENTS];     /* Droplet.des_x/.des_y      dmfb.py:78-79 */
    int sx[DMFB_MAX_AGENTS], sy[DMFB_MAX_AGENTS];     /* RoutingTaskManager.starts dmfb.py:135 */
    int dist[DMFB_MAX_AGENTS];                        /* .distances                dmfb.py:137 */
    int blk[DMFB_MAX_BLOCKS][4];                      /* x_min,x_max,y_min,y_max   dmfb.py:34-41 */
    int nblk;
    double *health, *usage, *degrade;                 /* (W,L) row-major [x][y]    dmfb.py:147-151 */
    int step_count;                                   /* DMFBenv.step_count        dmfb.py:510 */
    long long constraints;                            /* DMFBenv.constraints       dmfb.py:511 */
    uint32_t env_id, rng_step, rng_ep, rng_map;
} dmfb_env;

typedef struct {
    dmfb_cfg cfg;
    int E;
    dmfb_env *envs;
    signed char zoom_x[512], zoom_y[512]; /* direction-vector zoom LUT, index d+255 */
} dmfb_oracle;

/* ------------------------------------------------------------------ Philox4x32-10 */
static inline void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                 uint32_t c2, uint32_t c3, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void dmfb_oracle_philox(uint32_t k0, uint32_t k1, const uint32_t ctr[4], uint32_t out[4]) {
    philox4x32_10(k0, k1, ctr[0], ctr[1], ctr[2], ctr[3], out);
}

enum { STREAM_MOVE = 1, STREAM_TASK = 2, STREAM_DEGRADE = 3, STREAM_BLOCK = 4 };

static inline void env_philox(const dmfb_oracle *o, const dmfb_env *e, uint32_t c1, uint32_t c2,
                              uint32_t stream, uint32_t sub, uint32_t out[4]) {
    philox4x32_10((uint32_t)o->cfg.seed, (uint32_t)(o->cfg.seed >> 32), e->env_id, c1, c2,
                  (stream << 8) | sub, out);
}

/* 53-bit double in [0,1) from two words: same lattice as Python's random.random() */
static inline double u53(uint32_t hi, uint32_t lo) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}
static inline int below(uint32_t w, int n) { return (int)(((uint64_t)w * (uint32_t)n) >> 32); }

/* ------------------------------------------------------------------ helpers */
static inline int iabs(int v) { return v < 0 ? -v : v; }

/* Droplet.move: env/DMFB/dmfb.py:103-124 */
static int droplet_move(int *x, int *y, int action, int W, int L) {
    switch (action) {
    case 0: break;            /* STALL */
    case 1: *x += 1; break;   /* RIGHT */
    case 2: *x -= 1; break;   /* LEFT  */
    case 3: *y -= 1; break;   /* DOWN  */
    case 4: *y += 1; break;   /* UP    */
    default: return DMFB_ERR_BAD_ACTION;
    }
    if (*x > W - 1) *x = W - 1; else if (*x < 0) *x = 0;
    if (*y > L - 1) *y = L - 1; else if (*y < 0) *y = 0;
    return DMFB_OK;
}

/* RoutingTaskManager._isTouchingBlocks: dmfb.py:301-308 */
static int touching_blocks(const dmfb_env *e, int px, int py) {
    for (int b = 0; b < e->nblk; ++b)
        if (px >= e->blk[b][0] && px <= e->blk[b][1] && py >= e->blk[b][2] && py <= e->blk[b][3])
            return 1;
    return 0;
}

/* RoutingTaskManager._isinvalidaction: dmfb.py:310-323 -- true iff ANY two droplets
 * share a cell (min off-diagonal squared distance == 0). */
static int any_overlap(const dmfb_env *e, int n) {
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (e->x[i] == e->x[j] && e->y[i] == e->y[j]) return 1;
    return 0;
}

/* python round(): half-to-even on the double value == rint() in the default mode */
static int zoom_one(int d, int hf, int size) {
    /* dmfb.py:444-453 */
    if (iabs(d) > hf) {
        double scale = (double)(size - hf) / (double)(10 - hf);
        if (d > 0) return (int)rint((double)(d - hf) / scale) + hf;
        return (int)rint((double)(d + hf) / scale) - hf;
    }
    return d;
}

/* ------------------------------------------------------------------ maps */
/* RoutingTaskManager._random_health_statue: dmfb.py:157-166 */
static void gen_degrade(dmfb_oracle *o, dmfb_env *e) {
    int W = o->cfg.W, L = o->cfg.L;
    if (!e->degrade) return;
    for (int c = 0; c < W * L; ++c) e->degrade[c] = 1.0;
    if (!o->cfg.b_degrade) return;
    double per_healthy = 1.0 - o->cfg.per_degrade;
    for (int c = 0; c < W * L; ++c) {
        uint32_t w[4];
        env_philox(o, e, e->rng_map, (uint32_t)c, STREAM_DEGRADE, 0, w);
        double v = u53(w[0], w[1]) * 0.4 + 0.6;
        double sel = u53(w[2], w[3]);
        e->degrade[c] = (sel < per_healthy) ? 1.0 : v;
    }
    e->rng_map++;
}

/* RoutingTaskManager.updateHealth: dmfb.py:465-471 (runs even when b_degrade is False) */
static void update_health(dmfb_oracle *o, dmfb_env *e) {
    if (!e->health) return;
    int cells = o->cfg.W * o->cfg.L;
    for (int c = 0; c < cells; ++c)
        if (e->usage[c] > 50.0) {
            e->health[c] = e->health[c] * e->degrade[c];
            e->usage[c] = 0.0;
        }
}

/* ------------------------------------------------------------------ task generation */
static void recompute_dist(dmfb_env *e, int n) {
    for (int i = 0; i < n; ++i) e->dist[i] = iabs(e->x[i] - e->gx[i]) + iabs(e->y[i] - e->gy[i]);
}

/* RoutingTaskManager._Generate_Start_End: dmfb.py:207-226.  2n points are drawn together
 * and rejected until every pair (starts AND ends in one set) has squared distance > 2.
 * Draw order is the build's Philox contract, not numpy's. */
static void gen_start_end(dmfb_oracle *o, dmfb_env *e) {
    int n = o->cfg.n, W = o->cfg.W, L = o->cfg.L;
    int px[2 * DMFB_MAX_AGENTS], py[2 * DMFB_MAX_AGENTS];
    for (uint32_t attempt = 0;; ++attempt) {
        for (int b = 0; b < n; ++b) {
            uint32_t w[4];
            env_philox(o, e, e->rng_ep, attempt, STREAM_TASK, (uint32_t)b, w);
            py[2 * b] = below(w[0], L);     px[2 * b] = below(w[1], W);
            py[2 * b + 1] = below(w[2], L); px[2 * b + 1] = below(w[3], W);
        }
        int ok = 1;
        for (int i = 0; i < 2 * n && ok; ++i)
            for (int j = i + 1; j < 2 * n; ++j) {
                int dx = px[i] - px[j], dy = py[i] - py[j];
                if (dx * dx + dy * dy <= 2) { ok = 0; break; }
            }
        if (ok || attempt == DMFB_TASK_MAX_ATTEMPTS - 1) break; /* bounded like the kernels */
    }
    for (int i = 0; i < n; ++i) {
        e->sx[i] = px[i]; e->sy[i] = py[i];
        e->gx[i] = px[n + i]; e->gy[i] = py[n + i];
    }
}

/* RoutingTaskManager.GenRandomBlocks: dmfb.py:228-251 */
static void gen_blocks(dmfb_oracle *o, dmfb_env *e) {
    int n = o->cfg.n, W = o->cfg.W, L = o->cfg.L, nb = o->cfg.n_blocks;
    e->nblk = 0;
    if (W < 5 || L < 5) return;
    if ((double)(nb * 4) / (double)(W * L) > 0.2) return;
    for (int b = 0; b < nb; ++b) {
        for (uint32_t attempt = 0;; ++attempt) {
            uint32_t w[4];
            env_philox(o, e, e->rng_ep, attempt, STREAM_BLOCK, (uint32_t)b, w);
            int y0 = below(w[0], L - 3), x0 = below(w[1], W - 3);
            int bad = 0;
            for (int i = 0; i < n && !bad; ++i) {
                if (e->sx[i] >= x0 && e->sx[i] <= x0 + 1 && e->sy[i] >= y0 && e->sy[i] <= y0 + 1) bad = 1;
                if (e->gx[i] >= x0 && e->gx[i] <= x0 + 1 && e->gy[i] >= y0 && e->gy[i] <= y0 + 1) bad = 1;
            }
            for (int k = 0; k < e->nblk && !bad; ++k) /* Block.isBlockOverlap: dmfb.py:56-69 */
                if (!(x0 > e->blk[k][1] || e->blk[k][0] > x0 + 1) &&
                    !(y0 > e->blk[k][3] || e->blk[k][2] > y0 + 1)) bad = 1;
            if (!bad || attempt == DMFB_TASK_MAX_ATTEMPTS - 1) { /* bounded like the kernels */
                e->blk[e->nblk][0] = x0; e->blk[e->nblk][1] = x0 + 1;
                e->blk[e->nblk][2] = y0; e->blk[e->nblk][3] = y0 + 1;
                e->nblk++;
                break;
            }
        }
    }
}

/* RoutingTaskManager.restartforall: dmfb.py:185-190 */
static void restart_env(dmfb_oracle *o, dmfb_env *e) {
    int n = o->cfg.n;
    for (int i = 0; i < n; ++i) { e->x[i] = e->sx[i]; e->y[i] = e->sy[i]; }
    recompute_dist(e, n);
    e->step_count = 0;
    e->constraints = 0;
}

/* DMFBenv.reset + RoutingTaskManager.refresh: dmfb.py:589-597, 174-183 */
static void reset_env(dmfb_oracle *o, dmfb_env *e, int new_flag) {
    gen_start_end(o, e);
    gen_blocks(o, e);
    e->rng_ep++;
    restart_env(o, e);
    if (new_flag) {
        if (e->health) {
            int cells = o->cfg.W * o->cfg.L;
            for (int c = 0; c < cells; ++c) { e->health[c] = 1.0; e->usage[c] = 0.0; }
            gen_degrade(o, e);
        }
    } else {
        update_health(o, e);
    }
}

/* ------------------------------------------------------------------ public API */
int dmfb_oracle_check_cfg(int W, int L, int n, int n_blocks, int fov) {
    /* DMFBenv.__init__ asserts (dmfb.py:489-490), then RoutingTaskManager guards (:139-146) */
    if (W < 5 || L < 5) return DMFB_ERR_CHIP_TOO_SMALL;
    if (n <= 0) return DMFB_ERR_NO_AGENTS;
    if (fov > (W < L ? W : L)) return DMFB_ERR_FOV_TOO_LARGE;
    if (n > (int)((W + 1) * (L + 1) / 9)) return DMFB_ERR_TOO_MANY_DROPLETS;
    if (n > DMFB_MAX_AGENTS || n_blocks > DMFB_MAX_BLOCKS || W > 255 || L > 255 || fov < 1) return DMFB_ERR_UNSUPPORTED;
    return DMFB_OK;
}

int dmfb_oracle_create(int W, int L, int n, int n_blocks, int fov, int stall, int b_degrade,
                       double per_degrade, int with_maps, uint64_t seed, int E, uint32_t env_id0,
                       dmfb_oracle **out) {
    int rc = dmfb_oracle_check_cfg(W, L, n, n_blocks, fov);
    if (rc) return rc;
    if (E <= 0 || !out) return DMFB_ERR_BAD_ARG;
    dmfb_oracle *o = (dmfb_oracle *)calloc(1, sizeof(*o));
    o->cfg.W = W; o->cfg.L = L; o->cfg.n = n; o->cfg.fov = fov; o->cfg.n_blocks = n_blocks;
    o->cfg.stall = stall; o->cfg.b_degrade = b_degrade; o->cfg.per_degrade = per_degrade;
    o->cfg.max_step = (W + L) * 2; /* dmfb.py:508 */
    o->cfg.seed = seed;
    o->E = E;
    o->envs = (dmfb_env *)calloc((size_t)E, sizeof(dmfb_env));
    int hf = fov / 2;
    for (int d = -255; d <= 255; ++d) {
        int zx = d, zy = d;
        if (hf != 10) { zx = zoom_one(d, hf, W); zy = zoom_one(d, hf, L); }
        o->zoom_x[d + 255] = (signed char)zx;
        o->zoom_y[d + 255] = (signed char)zy;
    }
    for (int k = 0; k < E; ++k) {
        dmfb_env *e = &o->envs[k];
        e->env_id = env_id0 + (uint32_t)k;
        if (b_degrade || with_maps) {
            size_t cells = (size_t)W * L;
            e->health = (double *)malloc(cells * sizeof(double));
            e->usage = (double *)malloc(cells * sizeof(double));
            e->degrade = (double *)malloc(cells * sizeof(double));
            for (size_t c = 0; c < cells; ++c) { e->health[c] = 1.0; e->usage[c] = 0.0; }
            gen_degrade(o, e); /* RoutingTaskManager.__init__: dmfb.py:147-151 */
        }
        /* RoutingTaskManager.__init__ ends with Generate_task(): dmfb.py:155 */
        gen_start_end(o, e);
        gen_blocks(o, e);
        e->rng_ep++;
        restart_env(o, e);
    }
    *out = o;
    return DMFB_OK;
}

void dmfb_oracle_destroy(dmfb_oracle *o) {
    if (!o) return;
    for (int k = 0; k < o->E; ++k) { free(o->envs[k].health); free(o->envs[k].usage); free(o->envs[k].degrade); }
    free(o->envs);
    free(o);
}

int dmfb_oracle_max_step(const dmfb_oracle *o) { return o->cfg.max_step; }
int dmfb_oracle_obs_len(const dmfb_oracle *o) { return 3 * o->cfg.fov * o->cfg.fov + 2; }

/* reset envs whose mask byte is non-zero (mask NULL = all) */
void dmfb_oracle_reset(dmfb_oracle *o, const uint8_t *mask, int new_flag) {
    for (int k = 0; k < o->E; ++k)
        if (!mask || mask[k]) reset_env(o, &o->envs[k], new_flag);
}

/* DMFBenv.restart: dmfb.py:599-605 */
void dmfb_oracle_restart(dmfb_oracle *o, const uint8_t *mask) {
    for (int k = 0; k < o->E; ++k)
        if (!mask || mask[k]) restart_env(o, &o->envs[k]);
}

/* task injection: starts/ends int32 [E][n][2] (x,y); then restartforall semantics */
void dmfb_oracle_set_task(dmfb_oracle *o, const int32_t *starts, const int32_t *ends) {
    int n = o->cfg.n;
    for (int k = 0; k < o->E; ++k) {
        dmfb_env *e = &o->envs[k];
        for (int i = 0; i < n; ++i) {
            e->sx[i] = starts[(k * n + i) * 2]; e->sy[i] = starts[(k * n + i) * 2 + 1];
            e->gx[i] = ends[(k * n + i) * 2];   e->gy[i] = ends[(k * n + i) * 2 + 1];
        }
        restart_env(o, e);
    }
}

/* blocks injection: int32 [E][nb][4] = x_min,x_max,y_min,y_max */
int dmfb_oracle_set_blocks(dmfb_oracle *o, const int32_t *blocks, int nb) {
    if (nb > DMFB_MAX_BLOCKS) return DMFB_ERR_UNSUPPORTED;
    for (int k = 0; k < o->E; ++k) {
        dmfb_env *e = &o->envs[k];
        e->nblk = nb;
        for (int b = 0; b < nb; ++b)
            for (int c = 0; c < 4; ++c) e->blk[b][c] = blocks[(k * nb + b) * 4 + c];
    }
    return DMFB_OK;
}

/* blocks int32 [E][cap][4]; returns the per-env block count (identical for all envs) */
int dmfb_oracle_get_blocks(const dmfb_oracle *o, int32_t *blocks, int cap) {
    for (int k = 0; k < o->E; ++k)
        for (int b = 0; b < o->envs[k].nblk && b < cap; ++b)
            for (int c = 0; c < 4; ++c) blocks[((size_t)k * cap + b) * 4 + c] = o->envs[k].blk[b][c];
    return o->envs[0].nblk;
}

void dmfb_oracle_get_task(const dmfb_oracle *o, int32_t *starts, int32_t *ends) {
    int n = o->cfg.n;
    for (int k = 0; k < o->E; ++k)
        for (int i = 0; i < n; ++i) {
            const dmfb_env *e = &o->envs[k];
            starts[(k * n + i) * 2] = e->sx[i]; starts[(k * n + i) * 2 + 1] = e->sy[i];
            ends[(k * n + i) * 2] = e->gx[i];   ends[(k * n + i) * 2 + 1] = e->gy[i];
        }
}

void dmfb_oracle_get_state(const dmfb_oracle *o, int32_t *pos, int32_t *dist, int32_t *step_count,
                           int64_t *constraints) {
    int n = o->cfg.n;
    for (int k = 0; k < o->E; ++k) {
        const dmfb_env *e = &o->envs[k];
        for (int i = 0; i < n; ++i) {
            if (pos) { pos[(k * n + i) * 2] = e->x[i]; pos[(k * n + i) * 2 + 1] = e->y[i]; }
            if (dist) dist[k * n + i] = e->dist[i];
        }
        if (step_count) step_count[k] = e->step_count;
        if (constraints) constraints[k] = e->constraints;
    }
}

/* which: 0 health, 1 usage, 2 degrade; buf float64 [E][W][L] */
int dmfb_oracle_get_map(const dmfb_oracle *o, int which, double *buf) {
    size_t cells = (size_t)o->cfg.W * o->cfg.L;
    for (int k = 0; k < o->E; ++k) {
        const dmfb_env *e = &o->envs[k];
        const double *src = which == 0 ? e->health : which == 1 ? e->usage : e->degrade;
        if (!src) return DMFB_ERR_BAD_ARG;
        memcpy(buf + k * cells, src, cells * sizeof(double));
    }
    return DMFB_OK;
}
int dmfb_oracle_set_map(dmfb_oracle *o, int which, const double *buf) {
    size_t cells = (size_t)o->cfg.W * o->cfg.L;
    for (int k = 0; k < o->E; ++k) {
        dmfb_env *e = &o->envs[k];
        double *dst = which == 0 ? e->health : which == 1 ? e->usage : e->degrade;
        if (!dst) return DMFB_ERR_BAD_ARG;
        memcpy(dst, buf + k * cells, cells * sizeof(double));
    }
    return DMFB_OK;
}

/* RoutingTaskManager.moveOneDroplet: dmfb.py:325-359.  `u` is the draw this droplet
 * consumes if it is not finished (finished droplets under stall consume none). */
static int move_one(dmfb_oracle *o, dmfb_env *e, int i, int action, double u, double *reward,
                    int *past_x, int *past_y) {
    int W = o->cfg.W, L = o->cfg.L, n = o->cfg.n;
    int x = e->x[i], y = e->y[i];
    if (o->cfg.stall && e->dist[i] == 0) {
        *reward = 0.0;
    } else {
        double prob = e->health ? e->health[x * L + y] : 1.0; /* getMoveProb: dmfb.py:361-363 */
        if (u <= prob) {
            int rc = droplet_move(&e->x[i], &e->y[i], action, W, L);
            if (rc) return rc;
            if (touching_blocks(e, e->x[i], e->y[i])) { e->x[i] = x; e->y[i] = y; }
            if (any_overlap(e, n)) { e->x[i] = x; e->y[i] = y; }
        }
        int new_dist = iabs(e->x[i] - e->gx[i]) + iabs(e->y[i] - e->gy[i]);
        int old = e->dist[i];
        if (new_dist == old && old == 0) *reward = -0.1;
        else if (new_dist == old && action == 0) *reward = -0.25;
        else if (new_dist < old) *reward = -0.1;
        else *reward = -0.4;
        e->dist[i] = new_dist;
    }
    *past_x = x; *past_y = y;
    return DMFB_OK;
}

/* DMFBenv.step (dmfb.py:560-587) + RoutingTaskManager.moveDroplets (:253-299) + addUsage (:459-463).
 * actions int32[n]; uniforms float64[n] indexed by AGENT (entry i is the draw droplet i takes if
 * it draws) or NULL for the Philox stream; outputs rewards f64[n], dones u8[n]. */
static int step_env(dmfb_oracle *o, dmfb_env *e, const int32_t *actions, const double *uniforms,
                    int record, double *rewards, uint8_t *dones, int32_t *constraints_out,
                    uint8_t *success_out) {
    int n = o->cfg.n;
    int was_done[DMFB_MAX_AGENTS], pastx[DMFB_MAX_AGENTS], pasty[DMFB_MAX_AGENTS];
    int sta[DMFB_MAX_AGENTS], dyn[DMFB_MAX_AGENTS];
    e->step_count += 1;
    for (int i = 0; i < n; ++i) was_done[i] = (e->dist[i] == 0); /* getTaskStatus before moves: :278 */
    for (int i = 0; i < n; ++i) {
        double u;
        if (uniforms) u = uniforms[i];
        else {
            uint32_t w[4]; /* RNG contract: droplets 2p and 2p + 1 share one Philox block, words (0 1) and (2 3) */
            env_philox(o, e, e->rng_step, (uint32_t)(i >> 1), STREAM_MOVE, 0, w);
            u = (i & 1) ? u53(w[2], w[3]) : u53(w[0], w[1]);
        }
        int rc = move_one(o, e, i, actions[i], u, &rewards[i], &pastx[i], &pasty[i]);
        if (rc) return rc;
    }
    e->rng_step++;
    /* comflic_static: dmfb.py:254-261 ; norm < 2  <=>  dx*dx+dy*dy < 4 */
    for (int i = 0; i < n; ++i) { sta[i] = 0; dyn[i] = 0; }
    for (int i = 0; i < n - 1; ++i)
        for (int j = i + 1; j < n; ++j) {
            int dx = e->x[i] - e->x[j], dy = e->y[i] - e->y[j];
            if (dx * dx + dy * dy < 4) { sta[i]++; sta[j]++; }
        }
    /* comflic_dynamic: dmfb.py:263-271 */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (i != j) {
                int dx = pastx[i] - e->x[j], dy = pasty[i] - e->y[j];
                if (dx * dx + dy * dy < 4) { dyn[i]++; dyn[j]++; }
            }
    int constraints = 0;
    for (int i = 0; i < n; ++i) constraints += sta[i] + dyn[i];
    /* rewards = np.array(rewards) - 2*sta - 2*dy : dmfb.py:288 */
    for (int i = 0; i < n; ++i) rewards[i] = (rewards[i] - (double)(2 * sta[i])) - (double)(2 * dyn[i]);
    if (o->cfg.stall)
        for (int i = 0; i < n; ++i)
            if (was_done[i]) rewards[i] = 0.0;
    int all_done = 1;
    for (int i = 0; i < n; ++i) all_done &= (e->dist[i] == 0);
    if (all_done) {
        for (int i = 0; i < n; ++i) rewards[i] = rewards[i] + 10.0;
        if (constraints == 0)
            for (int i = 0; i < n; ++i) rewards[i] = rewards[i] + 10.0;
    }
    if (record && e->usage) /* addUsage: dmfb.py:459-463 */
        for (int i = 0; i < n; ++i)
            if (e->dist[i] != 0) e->usage[e->x[i] * o->cfg.L + e->y[i]] += 1.0;
    e->constraints += constraints;
    uint8_t success = 0;
    if (e->step_count < o->cfg.max_step) {
        if (all_done && e->constraints == 0) success = 1;
        for (int i = 0; i < n; ++i) dones[i] = (uint8_t)(e->dist[i] == 0);
    } else {
        for (int i = 0; i < n; ++i) dones[i] = 1;
    }
    *constraints_out = constraints;
    *success_out = success;
    return DMFB_OK;
}

int dmfb_oracle_step(dmfb_oracle *o, const int32_t *actions, const double *uniforms, int record,
                     double *rewards, uint8_t *dones, int32_t *constraints, uint8_t *success) {
    int n = o->cfg.n;
    for (int k = 0; k < o->E; ++k) {
        int rc = step_env(o, &o->envs[k], actions + (size_t)k * n, uniforms ? uniforms + (size_t)k * n : NULL,
                          record, rewards + (size_t)k * n, dones + (size_t)k * n, constraints + k, success + k);
        if (rc) return rc;
    }
    return DMFB_OK;
}

/* RoutingTaskManager.getOneObs (dmfb.py:395-457) + DMFBenv.getOneObs flatten (:614-620):
 * out int8[3*fov*fov + 2], index c*fov*fov + x*fov + y, then the 2-vector. */
static void one_obs(const dmfb_oracle *o, const dmfb_env *e, int agent, signed char *out) {
    int fov = o->cfg.fov, n = o->cfg.n, W = o->cfg.W, L = o->cfg.L;
    int hf = fov / 2, ff = fov * fov;
    memset(out, 0, (size_t)(3 * ff + 2));
    int cx = e->x[agent], cy = e->y[agent];
    int ox = cx - hf, oy = cy - hf;
    for (int j = 0; j < n; ++j) { /* layer 0: every droplet in the window */
        int x = e->x[j] - ox, y = e->y[j] - oy;
        if (x >= 0 && x < fov && y >= 0 && y < fov) out[x * fov + y] = (signed char)(j + 1);
    }
    for (int j = 0; j < n; ++j) { /* layer 1: goals of the OTHER visible droplets, clipped */
        /* abs(d.x-center_x) < fov/2 with float fov/2  <=>  2*|dx| < fov */
        if (j != agent && 2 * iabs(e->x[j] - cx) < fov && 2 * iabs(e->y[j] - cy) < fov) {
            int x = e->gx[j] - ox, y = e->gy[j] - oy;
            x = x < 0 ? 0 : (x > fov - 1 ? fov - 1 : x);
            y = y < 0 ? 0 : (y > fov - 1 ? fov - 1 : y);
            out[ff + x * fov + y] = (signed char)(j + 1);
        }
    }
    for (int b = 0; b < e->nblk; ++b) /* layer 2: blocks in GLOBAL coords (reference quirk :422-426) */
        for (int i = e->blk[b][0]; i <= e->blk[b][1]; ++i)
            for (int j = e->blk[b][2]; j <= e->blk[b][3]; ++j)
                if (i >= 0 && i < fov && j >= 0 && j < fov) out[2 * ff + i * fov + j] = 1;
    int left = hf - cx, right = hf - (W - 1 - cx);
    if (left > 0) {
        for (int x = 0; x < left && x < fov; ++x) for (int y = 0; y < fov; ++y) out[2 * ff + x * fov + y] = 1;
    } else if (right > 0) {
        for (int x = (fov - right < 0 ? 0 : fov - right); x < fov; ++x) for (int y = 0; y < fov; ++y) out[2 * ff + x * fov + y] = 1;
    }
    int up = hf - cy, down = hf - (L - 1 - cy);
    if (up > 0) {
        for (int x = 0; x < fov; ++x) for (int y = 0; y < up && y < fov; ++y) out[2 * ff + x * fov + y] = 1;
    } else if (down > 0) {
        for (int x = 0; x < fov; ++x) for (int y = (fov - down < 0 ? 0 : fov - down); y < fov; ++y) out[2 * ff + x * fov + y] = 1;
    }
    out[3 * ff] = o->zoom_x[e->gx[agent] - cx + 255];
    out[3 * ff + 1] = o->zoom_y[e->gy[agent] - cy + 255];
}

void dmfb_oracle_observe(const dmfb_oracle *o, signed char *obs) {
    int n = o->cfg.n, len = 3 * o->cfg.fov * o->cfg.fov + 2;
    for (int k = 0; k < o->E; ++k)
        for (int i = 0; i < n; ++i) one_obs(o, &o->envs[k], i, obs + ((size_t)k * n + i) * len);
}

/* copy of the zoom table for inspection: out int8[2][511] */
void dmfb_oracle_zoom_lut(const dmfb_oracle *o, signed char *out) {
    memcpy(out, o->zoom_x, 511);
    memcpy(out + 511, o->zoom_y, 511);
}
