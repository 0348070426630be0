/*
 * meda_vec.h -- C ABI of the MI355X-native vectorised MEDA droplet-routing environment.
 *
 * Mirrors the object protocol of the reference's `MEDAEnv` (env/MEDA/meda.py:457-681) for E
 * independent chips resident in HBM on one GPU.  Conventions are those of dmfb_vec.h: plain C
 * types, caller-owned DEVICE buffers (`d_*`), `stream` = hipStream_t as void*, asynchronous
 * calls, negative int error codes, thread-compatible.
 *
 * Geometry (meda.py:35-47,106-138,208-227): a droplet is a 5x5 box (r = 2) given by its centre
 * (x, y); x runs along `length` (second map axis), y along `width` (first map axis): maps are
 * indexed [y][x] (meda.py:307,598).
 */
#ifndef MEDA_VEC_H
#define MEDA_VEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEDA_OK 0
#define MEDA_ERR_BAD_ARG (-1)
#define MEDA_ERR_TOO_MANY_DROPLETS (-3) /* RuntimeError("Too many droplets in the WxL MEDA array")  meda.py:151-154 */
#define MEDA_ERR_BAD_SIZE (-4)          /* assert w > 0 and l > 0                                   meda.py:472 */
#define MEDA_ERR_NO_AGENTS (-5)         /* assert n_agents > 0                                      meda.py:473 */
#define MEDA_ERR_UNSUPPORTED (-6)       /* outside the build limits */
#define MEDA_ERR_NO_MAPS (-8)
#define MEDA_ERR_HIP (-100)

#define MEDA_MAX_AGENTS 16
#define MEDA_MAX_DIM 128 /* centres and direction components are emitted as int8 */

#define MEDA_STEP_AUTORESET 2u /* chips whose episode ended this step are reset() inside the call */
#define MEDA_ACT_I32 0u
#define MEDA_ACT_I8 16u
#define MEDA_ACT_I64 32u

#define MEDA_MAP_HEALTH 0  /* MEDAEnv.m_health  meda.py:494 */
#define MEDA_MAP_USAGE 1   /* MEDAEnv.m_usage   meda.py:495 */
#define MEDA_MAP_DEGRADE 2 /* MEDAEnv.m_degrade meda.py:498-504 */

typedef struct meda_vec meda_vec;

/* MEDAEnv(w, l, n_agents, n_blocks, fov, stall, b_degrade, per_degrade) (meda.py:469); n_blocks and
 * stall are accepted and ignored by the reference and are not part of this struct. */
typedef struct {
    int32_t width, length, n_agents, fov;
    int32_t b_degrade;
    int32_t with_maps;
    double per_degrade;
    int32_t n_envs;
    uint32_t env_id0;
    uint64_t seed;
    int32_t device;
    int32_t obs_version; /* 0: MEDAEnv.getOneObs (4 layers, meda.py:613-674); 2: MEDAEnv_v0_2.getOneObs (3 int8 layers +
                            direction zoomed to 30x30, meda.py:850-897; the CLI default version is '0.2') */
} meda_vec_config;

typedef struct {
    double *d_rewards;     /* [E][n] float64                                   meda.py:526-527 */
    uint8_t *d_dones;      /* [E][n]                                           meda.py:529-537 */
    double *d_fail;        /* [E]    info['constraints'] = np.sum(punish) <= 0 meda.py:256,538 */
    uint8_t *d_success;    /* [E]    info['success']                           meda.py:530-531 */
    int8_t *d_obs;         /* [E][n][obs_len] (version 0: the reference returns float64 holding small ints) */
    double *d_team_reward; /* [E]    np.sum(rewards)/n                         common/rollout.py:33 */
    uint8_t *d_terminated; /* [E]    all(dones)                                common/rollout.py:34-35 */
} meda_vec_step_out;

int meda_vec_check_config(const meda_vec_config *cfg);               /* meda.py:151-154,472-473 */
int meda_vec_create(const meda_vec_config *cfg, void *stream, meda_vec **out); /* MEDAEnv.__init__ meda.py:469-510 */
int meda_vec_destroy(meda_vec *h);
size_t meda_vec_state_bytes(const meda_vec *h);
int meda_vec_obs_len(const meda_vec *h);  /* 4*fov*fov+2 (obs_version 0, meda.py:676-681) or 3*fov*fov+2 (obs_version 2) */
int meda_vec_max_step(const meda_vec *h); /* width+length (meda.py:492) */
int meda_vec_n_envs(const meda_vec *h);
int meda_vec_n_agents(const meda_vec *h);

/* MEDAEnv.reset() (meda.py:541-550): new task, counters and `fails` zeroed, updateHealth when b_degrade. */
int meda_vec_reset(meda_vec *h, const uint8_t *d_mask, int8_t *d_obs, void *stream);
/* MEDAEnv.restart() (meda.py:552-561): droplets back to the starts; `fails` is NOT cleared (as in the reference). */
int meda_vec_restart(meda_vec *h, const uint8_t *d_mask, int8_t *d_obs, void *stream);
/* routing_manager.starts/destinations = ...; restart(); fails = 0.  int32[E][n][2] = (x_center, y_center). */
int meda_vec_set_task(meda_vec *h, const int32_t *d_starts, const int32_t *d_ends, void *stream);
int meda_vec_get_task(const meda_vec *h, int32_t *d_starts, int32_t *d_ends, void *stream);

/* MEDAEnv.step(actions) (meda.py:513-539); arguments as dmfb_vec_step. Actions: 0 N,1 E,2 S,3 W,4 NE,5 SE,
 * 6 SW,7 NW,8 STALL (meda.py:23-32). */
int meda_vec_step(meda_vec *h, const void *d_actions, const double *d_uniforms, const uint8_t *d_active,
                  uint32_t flags, const meda_vec_step_out *out, void *stream);
/* MEDAEnv.getObs() (meda.py:607-674) */
int meda_vec_observe(const meda_vec *h, const uint8_t *d_mask, int8_t *d_obs, void *stream);

/* centres int32[E][n][2], status uint8[E][n] (RoutingTaskManager.status), step_count int32[E],
 * failed uint8[E] (fails != 0). Any pointer may be NULL. */
int meda_vec_get_state(const meda_vec *h, int32_t *d_pos, uint8_t *d_status, int32_t *d_step_count, uint8_t *d_failed,
                       void *stream);
int meda_vec_get_map(const meda_vec *h, int which, double *d_buf, void *stream); /* float64[E][width][length] */
int meda_vec_set_map(meda_vec *h, int which, const double *d_buf, void *stream);

/* How the handle maps chips to workgroups (profiling aid; no reference counterpart): out[0] = chips per workgroup of
 * the transition kernel, out[1] = chips per tile of the observation kernel, out[2] = its threads per workgroup,
 * out[3] = its (persistent) workgroup count. */
int meda_vec_launch_shape(const meda_vec *h, int32_t out[4]);

/* Measurement aid (no reference counterpart), as dmfb_vec_observe_timing: while enabled, every launch of the observation
 * kernel carries a HIP event pair that receives the dispatch's own start and end time stamps -- the duration rocprofv3
 * --kernel-trace reports.  _read waits for the timed launches (host-synchronising), returns their summed duration in
 * microseconds and their count (at most 256 per read) and starts a new series. */
int meda_vec_observe_timing(meda_vec *h, int enable);
int meda_vec_observe_timing_read(meda_vec *h, double *total_us, int *launches);

const char *meda_vec_strerror(int code);
int meda_vec_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MEDA_VEC_H */
