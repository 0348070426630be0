/*
 * dmfb_vec.h -- C ABI of the MI355X-native vectorised DMFB droplet-routing environment.
 *
 * The reference (jesselasse/MARL-DMFB) is pure Python and has no FFI; its boundary for this
 * path is the Python object protocol of `DMFBenv` (env/DMFB/dmfb.py:474-640).  Each entry
 * point below names the reference interface it replaces.  A handle owns E independent chips
 * ("envs") resident in HBM on one GPU and advances all of them in lock-step.
 *
 * Conventions
 *   - plain C types only; every `d_*` pointer is a DEVICE pointer owned by the caller
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream)
 *   - every call is asynchronous on `stream`; no call synchronises the host, none allocates
 *     device memory after dmfb_vec_create
 *   - return value: 0 on success, negative DMFB_ERR_* otherwise (dmfb_vec_strerror)
 *   - thread-compatible, not thread-safe: one handle, one stream at a time
 *   - coordinates are (x, y) with x in [0,width) indexing the FIRST map axis, as in the
 *     reference (`m_health[x][y]`, dmfb.py:362)
 */
#ifndef DMFB_VEC_H
#define DMFB_VEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMFB_OK 0
#define DMFB_ERR_BAD_ARG (-1)
#define DMFB_ERR_FOV_TOO_LARGE (-2)     /* RuntimeError('Fov is too large')         dmfb.py:139-140 */
#define DMFB_ERR_TOO_MANY_DROPLETS (-3) /* TypeError('Too many droplets for DMFB')  dmfb.py:144-146 */
#define DMFB_ERR_CHIP_TOO_SMALL (-4)    /* assert width >= 5 and length >= 5        dmfb.py:489 */
#define DMFB_ERR_NO_AGENTS (-5)         /* assert n_agents > 0                      dmfb.py:490 */
#define DMFB_ERR_UNSUPPORTED (-6)       /* outside the build limits (see DMFB_MAX_*) */
#define DMFB_ERR_BAD_ACTION (-7)        /* TypeError('action is illegal')           dmfb.py:116 (host wrappers only) */
#define DMFB_ERR_NO_MAPS (-8)           /* map access on a handle created without maps */
#define DMFB_ERR_HIP (-100)             /* a HIP runtime call failed; see dmfb_vec_last_hip_error */

#define DMFB_MAX_AGENTS 16  /* droplets per chip the kernels are instantiated for */
#define DMFB_MAX_DIM 255    /* width/length limit (positions are packed in bytes) */
#define DMFB_MAX_BLOCKS 64  /* 2x2 obstacle blocks per chip (one wave lane holds one block during generation) */

/* dmfb_vec_step flags */
#define DMFB_STEP_RECORD 1u     /* DMFBenv.step(record=True): addUsage                dmfb.py:570-571 */
#define DMFB_STEP_AUTORESET 2u  /* envs whose episode ended this step get reset(new=False) inside the
                                   same launch; their obs row is the FIRST obs of the new episode */
#define DMFB_ACT_I32 0u         /* d_actions is int32[E][n] */
#define DMFB_ACT_I8 16u         /* d_actions is int8 [E][n] */
#define DMFB_ACT_I64 32u        /* d_actions is int64[E][n] (torch argmax output) */

/* map selector for dmfb_vec_get_map / dmfb_vec_set_map */
#define DMFB_MAP_HEALTH 0   /* RoutingTaskManager.m_health   dmfb.py:147 */
#define DMFB_MAP_USAGE 1    /* RoutingTaskManager.m_usage    dmfb.py:148 */
#define DMFB_MAP_DEGRADE 2  /* RoutingTaskManager.m_degrade  dmfb.py:151 */

typedef struct dmfb_vec dmfb_vec;

/* Constructor arguments of DMFBenv(width, length, n_agents, n_blocks, fov, stall, b_degrade,
 * per_degrade) (dmfb.py:487) plus what a batch of chips on a GPU needs. */
typedef struct {
    int32_t width, length, n_agents, n_blocks, fov;
    int32_t stall;       /* bool */
    int32_t b_degrade;   /* bool */
    int32_t with_maps;   /* bool: keep health/usage/degrade maps even when !b_degrade */
    double per_degrade;
    int32_t n_envs;      /* E: chips owned by this handle */
    uint32_t env_id0;    /* global index of env 0 (rank offset when the batch is sharded) */
    uint64_t seed;       /* Philox key (DESIGN.md "RNG contract") */
    int32_t device;      /* HIP device ordinal */
} dmfb_vec_config;

/* Outputs of one lock-step transition; any pointer may be NULL to skip that output. */
typedef struct {
    double *d_rewards;      /* [E][n] per-agent rewards, float64 as returned by step()   dmfb.py:573-574 */
    uint8_t *d_dones;       /* [E][n] per-agent dones                                    dmfb.py:577-585 */
    int32_t *d_constraints; /* [E]    info['constraints'] of THIS step                   dmfb.py:586 */
    uint8_t *d_success;     /* [E]    info['success']                                    dmfb.py:579-580 */
    int8_t *d_obs;          /* [E][n][3*fov*fov+2] observation after the step            dmfb.py:576 */
    double *d_team_reward;  /* [E]    np.sum(rewards)/n in numpy's pairwise order        common/rollout.py:33 */
    uint8_t *d_terminated;  /* [E]    all(dones)                                         common/rollout.py:34-35 */
    int8_t *d_obs_terminal; /* [E][n][3*fov*fov+2] or NULL.  With DMFB_STEP_AUTORESET the d_obs row of an env whose episode ended is
                               the FIRST observation of its next episode; the observation the reference returns for that step (the
                               terminal one, `o_next` of the episode's last transition, common/rollout.py:38,118) is written here --
                               rows of envs that did not end in this call are left untouched.  Fused launch only (n_envs below
                               dmfb_vec_launch_shape()[2]); DMFB_ERR_UNSUPPORTED otherwise. */
} dmfb_vec_step_out;

/* The guards of DMFBenv.__init__ / RoutingTaskManager.__init__ (dmfb.py:489-490, 139-146). */
int dmfb_vec_check_config(const dmfb_vec_config *cfg);

/* DMFBenv.__init__ (dmfb.py:487-511): allocates all state in HBM, draws the degradation map
 * (dmfb.py:157-166) and the first task (Generate_task, dmfb.py:155). */
int dmfb_vec_create(const dmfb_vec_config *cfg, void *stream, dmfb_vec **out);
int dmfb_vec_destroy(dmfb_vec *h);

size_t dmfb_vec_state_bytes(const dmfb_vec *h); /* HBM held by the handle */
int dmfb_vec_obs_len(const dmfb_vec *h);        /* 3*fov*fov+2, get_env_info()['obs_shape'][-1]  dmfb.py:633-640 */
int dmfb_vec_max_step(const dmfb_vec *h);       /* 2*(width+length), 'episode_limit'            dmfb.py:508 */
int dmfb_vec_n_envs(const dmfb_vec *h);
int dmfb_vec_n_agents(const dmfb_vec *h);

/* DMFBenv.reset(new) (dmfb.py:589-597) for the envs whose mask byte is non-zero (d_mask NULL =
 * all): new task, counters zeroed, then maps re-initialised (new) or updateHealth (dmfb.py:465-471).
 * If d_obs is non-NULL the rows of the reset envs are overwritten with their first observation. */
int dmfb_vec_reset(dmfb_vec *h, const uint8_t *d_mask, int new_flag, int8_t *d_obs, void *stream);

/* DMFBenv.restart() (dmfb.py:599-605): droplets back to the stored starts, same task. */
int dmfb_vec_restart(dmfb_vec *h, const uint8_t *d_mask, int8_t *d_obs, void *stream);

/* Task injection = assigning routing_manager.starts/ends then restart() (dmfb.py:185-190):
 * d_starts, d_ends int32[E][n][2] (x, y). */
int dmfb_vec_set_task(dmfb_vec *h, const int32_t *d_starts, const int32_t *d_ends, void *stream);
int dmfb_vec_get_task(const dmfb_vec *h, int32_t *d_starts, int32_t *d_ends, void *stream);

/* Obstacle injection = assigning routing_manager.blocks (dmfb.py:34-41,138): d_blocks int32[E][nb][4] =
 * (x_min, x_max, y_min, y_max), nb <= the n_blocks the handle was created with; every chip gets nb blocks.
 * get returns the current per-chip count through *nb_out (host int) and fills int32[E][n_blocks][4]. */
int dmfb_vec_set_blocks(dmfb_vec *h, const int32_t *d_blocks, int nb, void *stream);
int dmfb_vec_get_blocks(const dmfb_vec *h, int32_t *d_blocks, int *nb_out, void *stream);

/* DMFBenv.step(actions, record) (dmfb.py:560-587) for all E envs.
 * d_uniforms: float64[E][n], entry [e][i] is the random.random() draw droplet i of env e
 * takes if it draws (dmfb.py:335), or NULL to use the handle's Philox stream.
 * d_active: uint8[E] or NULL (= all).  Envs whose byte is 0 are NOT stepped (their episode is
 * over and the caller has not reset them yet, common/rollout.py:46): state untouched, outputs
 * rewards 0 / dones 1 / constraints 0 / success 0 / team_reward 0 / terminated 1, obs row rewritten
 * with the unchanged observation. */
int dmfb_vec_step(dmfb_vec *h, const void *d_actions, const double *d_uniforms, const uint8_t *d_active,
                  uint32_t flags, const dmfb_vec_step_out *out, void *stream);

/* DMFBenv.getObs() (dmfb.py:622-626): int8[E][n][3*fov*fov+2]; d_mask as in reset. */
int dmfb_vec_observe(const dmfb_vec *h, const uint8_t *d_mask, int8_t *d_obs, void *stream);

/* Introspection used by the parity tests and the single-env facade: positions int32[E][n][2],
 * distances int32[E][n] (RoutingTaskManager.distances), step_count int32[E],
 * cumulative constraints int64[E] (DMFBenv.constraints). Any pointer may be NULL. */
int dmfb_vec_get_state(const dmfb_vec *h, int32_t *d_pos, int32_t *d_dist, int32_t *d_step_count,
                       int64_t *d_constraints, void *stream);

/* routing_manager.m_health / m_usage / m_degrade as float64[E][width][length]. */
int dmfb_vec_get_map(const dmfb_vec *h, int which, double *d_buf, void *stream);
int dmfb_vec_set_map(dmfb_vec *h, int which, const double *d_buf, void *stream);

/* How the handle maps chips to workgroups (profiling aid; no reference counterpart): out[0] = chips per 256-thread
 * workgroup of the fused step+observe launch, out[1] = chips per tile of the observation kernel, out[2] = batch size
 * from which dmfb_vec_step issues a step-only launch followed by the observation kernel, out[3] = chips per workgroup of
 * that step-only launch, out[4] = workgroups of the (persistent) observation kernel, out[5] = its threads per workgroup. */
int dmfb_vec_launch_shape(const dmfb_vec *h, int32_t out[6]);

/* Measurement aid (no reference counterpart): while enabled, every launch of the observation kernel (standalone or as the
 * second launch of dmfb_vec_step at large batches) carries a HIP event pair that receives the dispatch's own start and end
 * time stamps -- the duration rocprofv3 --kernel-trace reports.  _read waits for the timed launches (host-synchronising),
 * returns their summed duration in microseconds and their count (at most 256 per read) and starts a new series. */
int dmfb_vec_observe_timing(dmfb_vec *h, int enable);
int dmfb_vec_observe_timing_read(dmfb_vec *h, double *total_us, int *launches);

/* Direction-vector zoom table (dmfb.py:444-453) the handle was built with: int8[2][511], host memory. */
int dmfb_vec_zoom_lut(const dmfb_vec *h, int8_t *host_out);

const char *dmfb_vec_strerror(int code);
int dmfb_vec_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DMFB_VEC_H */
