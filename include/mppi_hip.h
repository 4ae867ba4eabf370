/*
 * mppi_hip.h — C ABI of libmppi_hip.so: the MI355X (gfx950) implementation of the
 * pi_mpc.mppi.MPPI.forward() hot path of kohonda/mppi_playground.
 *
 * The reference is pure Python/PyTorch, so the "FFI" a maintainer would add is a ctypes
 * binding inside src/pi_mpc/mppi.py (shown in INTEGRATION.md).  Every entry point below
 * names the reference code it replaces (file:line relative to the reference repo).
 *
 * Conventions
 *   - plain C: opaque handle, POD structs, raw pointers and sizes; no torch/HIP types.
 *   - every function returns 0 on success or a negative MPPI_E_* code; the message of the
 *     last failure on a handle is available from mppi_last_error().  Nothing throws.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All device work
 *     is enqueued asynchronously on it; only the functions documented "synchronises" block.
 *   - pointers named *_dev are device pointers (HBM), *_host are host pointers.
 *   - all arithmetic is fp32 (reference dtype, mppi.py:45).
 *
 * Data layout in HBM (owned by the handle)
 *   noise   lane-major tiles: for tile b (samples 64b..64b+63) and float4 group r
 *           (flat horizon index 4r..4r+3 of the [T*dc] row) lane l's float4 lives at
 *           ((b*R + r)*64 + l), R = ceil(T*dc/4).  A wavefront therefore reads or writes one
 *           contiguous 1 KiB segment per instruction.  This is the storage of the reference's
 *           `_action_noises` [N,T,dc] (mppi.py:261-263); mppi_export_noise() converts.
 *   costs   float[N]                                    (`costs`, mppi.py:333-336)
 *   summary float[MPPI_SUMMARY_HEAD + T*dc] per shard   (see mppi_weights_reduce)
 */
#ifndef MPPI_HIP_H
#define MPPI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct MppiSolver* mppi_handle_t;

enum {
    MPPI_OK = 0,
    MPPI_E_INVALID = -1,   /* bad argument / unsupported configuration */
    MPPI_E_HIP = -2,       /* a HIP runtime call failed                */
    MPPI_E_STATE = -3,     /* call sequence error (e.g. no map uploaded) */
    MPPI_E_NODEVICE = -4   /* no gfx950 device visible                 */
};

/* Native model plugins = the five shipped dynamics/cost pairs and the two
 * remaining example models.
 * MPPI_MODEL_GENERIC: the callables are opaque to the library (any torch dynamics/cost): the host
 * evaluates them on the exported actions and hands the costs back with mppi_set_costs; sampling,
 * softmax, the weighted reduction and the warm start still run here.  mppi_rollout_cost, the
 * state output of mppi_finalize and the mppi_rollout_* entry points are unavailable
 * (MPPI_E_INVALID). */
enum {
    MPPI_MODEL_GENERIC = -1,
    MPPI_MODEL_PENDULUM = 0,    /* example/pendulum.py:17-47                                  */
    MPPI_MODEL_CARTPOLE = 1,    /* example/cartpole.py:17-81                                  */
    MPPI_MODEL_MOUNTAINCAR = 2, /* example/mountaincar.py:17-55                               */
    MPPI_MODEL_NAV2D = 3,       /* src/envs/navigation_2d.py:218-279                          */
    MPPI_MODEL_RACING = 4,      /* src/envs/racing_env.py:327-372 + example/racing.py:110-159 */
    MPPI_MODEL_MJCARTPOLE = 5,  /* example/mujoco_cartpole.py:20-81 (continuous force, masspole = 1) */
    MPPI_MODEL_GOALZONE = 6     /* src/envs/goal_in_danger_zone.py:113-156 (7-state unicycle + circle) */
};

/* Racing parameter vector (mppi_set_model_params), racing_env.py:37-42,341-370, racing.py:41-46 */
enum { MPPI_RP_AMIN = 0, MPPI_RP_AMAX, MPPI_RP_SMIN, MPPI_RP_SMAX, MPPI_RP_L, MPPI_RP_VMAX, MPPI_RP_DT,
       MPPI_RP_XLO, MPPI_RP_XHI, MPPI_RP_YLO, MPPI_RP_YHI, MPPI_RP_QC, MPPI_RP_QL, MPPI_RP_QV, MPPI_RP_QO,
       MPPI_RP_QIN, MPPI_RP_QDIN, MPPI_RP_COUNT };
/* Navigation2D parameter vector, navigation_2d.py:54-72,218-279 */
enum { MPPI_NP_VMIN = 0, MPPI_NP_VMAX, MPPI_NP_WMIN, MPPI_NP_WMAX, MPPI_NP_DT, MPPI_NP_XLO, MPPI_NP_XHI,
       MPPI_NP_YLO, MPPI_NP_YHI, MPPI_NP_GX, MPPI_NP_GY, MPPI_NP_QO, MPPI_NP_COUNT };

/* Goal-in-danger-zone parameter vector, goal_in_danger_zone.py:78-87,113-156 */
enum { MPPI_GP_VMIN = 0, MPPI_GP_VMAX, MPPI_GP_WMIN, MPPI_GP_WMAX, MPPI_GP_DT, MPPI_GP_GX, MPPI_GP_GY, MPPI_GP_CX,
       MPPI_GP_CY, MPPI_GP_RADIUS, MPPI_GP_PENALTY, MPPI_GP_COUNT };

#define MPPI_MAX_PARAMS 32
#define MPPI_MAX_DIM_STATE 8
#define MPPI_MAX_DIM_CONTROL 4          /* controls held by MppiConfig (all native models have 1 or 2)       */
#define MPPI_MAX_DIM_CONTROL_GENERIC 64 /* MPPI_MODEL_GENERIC: any dim_control up to this (mppi.py:96-98)    */
#define MPPI_SUMMARY_HEAD 4 /* {min cost, sum e, sum e^2, sum e*c} */

/* Constructor arguments that reach the device path (MPPI.__init__, mppi.py:24-47,109-121). */
typedef struct MppiConfig {
    int32_t model;          /* MPPI_MODEL_*                                                     */
    int32_t horizon;        /* T                                                                */
    int32_t dim_state;      /* must match the model (any value >= 1 for MPPI_MODEL_GENERIC)     */
    int32_t dim_control;    /* must match the model (1..MPPI_MAX_DIM_CONTROL_GENERIC for GENERIC) */
    int64_t num_samples;    /* N held by THIS handle (= the local shard when sharded)           */
    int64_t sample_offset;  /* global index of local sample 0 (0 when not sharded)              */
    int64_t inherit_count;  /* global threshold int(N_global*(1-exploration)), mppi.py:266      */
    float u_min[MPPI_MAX_DIM_CONTROL];
    float u_max[MPPI_MAX_DIM_CONTROL];
    float sigmas[MPPI_MAX_DIM_CONTROL];
    uint64_t seed;          /* Philox key (mppi.py:46,93 `seed`)                                */
    int32_t device;         /* HIP device ordinal                                               */
    int32_t reserved;
} MppiConfig;

/* Library / build information ("gfx950", version). */
const char* mppi_version(void);
/* Integer version of THIS header's function signatures; bindings compare it with the constant they were written
 * against and refuse a stale library (a changed argument list would otherwise be called with shifted arguments). */
#define MPPI_ABI_VERSION 10
int mppi_abi_version(void);
/* Number of visible HIP devices (0 => the product cannot run; callers must fail loudly). */
int mppi_device_count(void);
const char* mppi_last_error(mppi_handle_t h);

/* MPPI.__init__ buffers (mppi.py:143-180): allocates noise tiles, costs, warm start (zeroed,
 * mppi.py:157), partials.  Does NOT draw the constructor sample (mppi.py:146-148 is only an
 * RNG-stream side effect; callers that mirror the torch CPU stream draw it themselves). */
int mppi_create(const MppiConfig* cfg, mppi_handle_t* out);
int mppi_destroy(mppi_handle_t h);
/* `u_min`, `u_max`, `sigmas` constructor tensors (mppi.py:32-34,96-98,109-121) as host arrays of n = dim_control
 * values: replaces the bounds / noise scales taken from MppiConfig.  REQUIRED once, before the first sample, for
 * generic handles with dim_control > MPPI_MAX_DIM_CONTROL (the config arrays hold four); native models accept it
 * only before mppi_set_model_params.  Synchronises. */
int mppi_set_control_limits(mppi_handle_t h, const float* u_min_host, const float* u_max_host, const float* sigmas_host,
                            int n);

/* copy.deepcopy(solver) (the reference's MPPI is a plain nn.Module, src/pi_mpc/mppi.py:16: every tensor it holds is copied
 * with it).  `dst` must have been created from the same MppiConfig; after the call it continues exactly like `src`: warm
 * start (`_previous_action_seq`), noise identity, costs and minimum of the last solve (get_top_samples / weights work on
 * the copy), Savitzky-Golay taps and history, the temperature with every device-resident search / dual state (ESSPS warm
 * grid, MPO dual and Adam moments), model parameters, maps, reference window, centre path and path index, options.
 * A borrowed state (mppi_bind_state) becomes an owned copy; a pending lazily completed state sequence is completed on
 * both handles first.  Set-up path: synchronises the device. */
int mppi_clone_state(mppi_handle_t dst, mppi_handle_t src);
/* Model constants (closures' Python constants / env attributes).  params: MPPI_RP_* or MPPI_NP_*
 * layout; models without parameters accept n == 0. */
int mppi_set_model_params(mppi_handle_t h, const float* params_host, int n);
/* ObstacleMap.convert_to_torch / LaneMap._map_torch (obstacle_map_2d.py:164-166,
 * lane_map_2d.py:85-88): occupancy grid cells[nx][ny] (0/1), first index = x.
 * slot 0 = obstacle map (nav2d, racing), slot 1 = lane map (racing).  Synchronises. */
int mppi_upload_map(mppi_handle_t h, int slot, const uint8_t* cells_host, int nx, int ny, float cell_size,
                    float origin_x, float origin_y);
/* Map construction on the device, bit-exact with the reference's host loops.  Both build slot `slot` in place
 * (same geometry arguments as mppi_upload_map), run on `stream` and return after the grid is complete; the caller
 * orders them against solves in flight on OTHER streams.
 *
 * ObstacleMap.add_circle_obstacle / add_rectangle_obstacle (obstacle_map_2d.py:103-158) for a whole obstacle list:
 *   circles [n_circles][3] = (ci, cj, r): centre cell np.round(center/cell + origin) and radius ceil(radius/cell)
 *                            in cells; every disc cell i^2+j^2 <= r^2 is written at clip(ci+i), clip(cj+j) (:118-123);
 *   rects   [n_rects][4]   = (x0, x1, y0, y1): the clipped half-open slice map[x0:x1, y0:y1] = 1 (:146-158).
 * The float -> cell conversions stay with the caller (per obstacle, float64 numpy semantics). */
int mppi_build_obstacle_map(mppi_handle_t h, int slot, int nx, int ny, float cell_size, float origin_x, float origin_y,
                            const int32_t* circles_host, int n_circles, const int32_t* rects_host, int n_rects,
                            void* stream);
/* LaneMap.populate_map (lane_map_2d.py:68-88): seeds [n_seeds][2] = in-bounds centre-line cells (:71-77); a cell is
 * drivable (0) iff its Euclidean distance transform value is <= (lane_width/2)/cell (:80-82), evaluated as the
 * integer test  min_seeds(dx^2+dy^2) <= max_d2  with max_d2 = the largest integer whose float64 sqrt is <= that
 * bound (computed by the caller). */
int mppi_build_lane_map(mppi_handle_t h, int slot, int nx, int ny, float cell_size, float origin_x, float origin_y,
                        const int32_t* seeds_host, int n_seeds, int64_t max_d2, void* stream);
/* Read a slot's grid back (cells_host may be NULL to query nx, ny only).  Synchronises the device. */
int mppi_download_map(mppi_handle_t h, int slot, uint8_t* cells_host, int* nx, int* ny);
/* racing_controller.reference_path = calc_ref_trajectory(...) (example/racing.py:73-81):
 * ref [rows][4] = (x, y, yaw, v_target), rows >= T (the cost reads rows 0..T-1). */
int mppi_set_reference(mppi_handle_t h, const float* ref_host, int rows, void* stream);

/* The racing control tick without the host (example/racing.py:73-81,161-218,221-266).
 *   mppi_set_center_path  `env.racing_center_path` [n][3] = (x, y, yaw) and the constants of
 *                         calc_ref_trajectory(horizon, DL, lookahead_distance, reference_path_interval): dind_host [rows]
 *                         = int(round(travel_i / DL)) for the rows = T+1 window rows (:201-205; float64 arithmetic of
 *                         the caller), v_target = env.V_MAX (:209).  Resets nothing else; synchronises (set-up path).
 *   mppi_ref_window       calc_ref_trajectory(state, path, cind, ...) as one kernel on `stream`: nearest centre-line
 *                         point (first minimum of the fp32 hypot, :189-196), ind = max(cind, ind) (:198) with the index
 *                         kept in device memory, window rows gathered straight into the model's reference table — the
 *                         same values mppi_set_reference would have uploaded, bit for bit.  state_dev [>= 2] device
 *                         pointer, or NULL for the state bound to the handle.  No host synchronisation.
 *   mppi_set/get_path_index  `racing_controller.current_path_index` (both synchronise the stream).
 *   mppi_get_reference    the current window as `reference_path` [rows][4] (device or host copy; host copies synchronise). */
int mppi_set_center_path(mppi_handle_t h, const float* path_host, int n, const int32_t* dind_host, int rows,
                         float v_target);
int mppi_ref_window(mppi_handle_t h, const float* state_dev, void* stream);
int mppi_set_path_index(mppi_handle_t h, int32_t cind, void* stream);
int mppi_get_path_index(mppi_handle_t h, int32_t* cind_out_host, void* stream);
int mppi_get_reference(mppi_handle_t h, float* ref_out, int rows, int on_device, void* stream);
/* `env.step(u)` of the shipped environments (src/envs/racing_env.py:142-163, src/envs/navigation_2d.py step) as one
 * launch, no handle needed: next = dynamics(state, clamp(u, u_min, u_max)) for native model `model` with the library
 * math in the reference's operation order; params_host = the model's MPPI_*P_* vector (cost weights may be omitted);
 * u_min_host / u_max_host [dim_control] or NULL (no pre-clamp); state_dev and next_state_dev may alias.  With
 * reached_out_dev != NULL also the goal test: *reached = norm(next[:2] - goal_xy_host) < goal_threshold (one byte). */
int mppi_model_step(int model, const float* params_host, int n_params, const float* u_min_host, const float* u_max_host,
                    const float* state_dev, const float* action_dev, float* next_state_dev, const float* goal_xy_host,
                    float goal_threshold, uint8_t* reached_out_dev, void* stream);
/* `ObstacleMap.compute_cost` / `LaneMap.compute_cost` (src/envs/obstacle_map_2d.py:168-200, src/envs/lane_map_2d.py:90-122)
 * as one launch, no handle needed — what `env.collision_check` of the examples' loops and cost plugins on the generic path
 * call: out_dev[i] = map_dev[ix][iy] with (ix, iy) = round_half_even(xy / cell_size + origin) (fp32, the reference's
 * operation order) when that lies inside the nx x ny grid (row-major float map), else 1.  Point i is the two floats at
 * xy_dev + i * stride (stride >= 2: e.g. 3 for the x, y of [x, y, theta] state rows). */
int mppi_grid_lookup(const float* map_dev, int nx, int ny, float cell_size, float origin_x, float origin_y, const float* xy_dev,
                     int64_t n, int64_t stride, float* out_dev, void* stream);

/* `_previous_action_seq` (mppi.py:157,255,452).  on_device != 0: pointer is a device pointer. */
int mppi_set_mean(mppi_handle_t h, const float* mean, int on_device, void* stream);
int mppi_get_mean(mppi_handle_t h, float* mean_out, int on_device, void* stream);
/* forward(state) argument (mppi.py:247-253), dim_state floats (copied). */
int mppi_set_state(mppi_handle_t h, const float* x0, int on_device, void* stream);
/* Zero-copy variant: the kernels of THIS solve (mppi_rollout_cost, mppi_finalize) read the state from the caller's
 * device buffer, which must stay valid and unmodified until that enqueued work ran.  mppi_rollout_cost snapshots the
 * state into the handle, and every later re-roll of that solve's samples (mppi_rollout_samples, mppi_top_samples,
 * mppi_rollout_candidates) starts from the snapshot — like the reference, whose `_state_seq_batch` is stored
 * (mppi.py:280-286,481) — so the caller may reuse or overwrite its buffer once the solve's kernels ran. */
int mppi_bind_state(mppi_handle_t h, const float* x0_dev);

/* Step 1 — `_noise_distribution.rsample` (mppi.py:261-263): eps ~ N(0, diag(sigma^2)) from the
 * device Philox4x32-10 stream, counter = (global sample index, float4 group, solve_idx): results
 * do not depend on how num_samples is sharded.  With option "noise_regen" = 1 (default) this only
 * fixes the identity of the solve's noise: the rollout and reduction kernels regenerate it in
 * registers and it is written to HBM only when an entry point needs the tiles
 * (mppi_export_noise, mppi_rollout_samples); with "noise_regen" = 0 the tiles are written here and
 * read back by both consumers (same values bit for bit). */
int mppi_sample(mppi_handle_t h, uint32_t solve_idx, void* stream);
/* Parity mode: load externally drawn noise eps[N][T][dc] (reference layout, device pointer)
 * into the tiled buffer (replaces mppi_sample for that solve). */
int mppi_inject_noise(mppi_handle_t h, const float* eps_dev, void* stream);
/* `_action_noises` / `_perturbed_action_seqs` attributes (mppi.py:261-275) in the reference
 * layout [N][T][dc]; either output may be NULL. */
int mppi_export_noise(mppi_handle_t h, float* eps_out_dev, float* actions_out_dev, void* stream);

/* Steps 1b-3 — clamp(mean + eps) (mppi.py:266-275), the N x T dynamics rollout
 * (mppi.py:280-286) and stage + terminal costs (mppi.py:291-336), fused; writes costs[N] and the
 * shard's minimum cost. */
int mppi_rollout_cost(mppi_handle_t h, void* stream);
/* `costs` (mppi.py:333-336) -> dst[N] (device or host).  Host copies synchronise. */
int mppi_get_costs(mppi_handle_t h, float* dst, int on_device, void* stream);
/* Overwrite costs[N] (used by tests and by the generic-callable path). */
int mppi_set_costs(mppi_handle_t h, const float* src, int on_device, void* stream);

/* Steps 5-6 — softmax(-costs/lambda) and sum_i w_i U_i (mppi.py:376-384), un-normalised:
 * writes the shard summary {min c, sum e, sum e^2, sum e*c, A[T*dc] = sum e_i*U_i} with
 * e_i = exp((-c_i)/lambda - max_j (-c_j)/lambda) over THIS shard.  summary_out_dev may be NULL
 * (the handle keeps its own copy). */
int mppi_weights_reduce(mppi_handle_t h, float lambda /* > 0, or MPPI_LAMBDA_DEVICE */, float* summary_out_dev,
                        void* stream);
/* Combine `num_shards` summaries (device array [num_shards][MPPI_SUMMARY_HEAD + T*dc]; NULL = this
 * handle's own, num_shards = 1 — or, with option "exchange_p2p", the summaries of all ranks from the
 * peer-to-peer buffer), form action_seq = A / sum e (mppi.py:381-385), optionally store it
 * as the next warm start (mppi.py:452), and roll it out with batch 1 (mppi.py:448-449,508-524).
 * action_out_dev [T][dc], state_seq_out_dev [T+1][ds], stats_out_dev [4] = {min c, sum e, sum e^2,
 * sum e*c} (global); any output may be NULL. */
int mppi_finalize(mppi_handle_t h, const float* summaries_dev, int num_shards, float lambda, int store_mean,
                  float* action_out_dev, float* state_seq_out_dev, float* stats_out_dev, void* stream);
/* `lambda_` = "ESSPS" | "LBPS" | "MPO" (mppi.py:183-210): the rule mppi_solve runs ON THE DEVICE when it is called with
 * lambda = MPPI_LAMBDA_DEVICE.  param = essps_target_ess (ESSPS), lbps_delta (LBPS), unused (MPO: see mppi_mpo_reset);
 * [lam_min, lam_max] = `lambda_min`, `lambda_max` (ESSPS, LBPS). */
enum { MPPI_AUTO_NONE = 0, MPPI_AUTO_ESSPS = 1, MPPI_AUTO_LBPS = 2, MPPI_AUTO_MPO = 3 };
int mppi_set_auto_lambda(mppi_handle_t h, int rule, double param, double lam_min, double lam_max);
/* MPPI.forward() of a native model in one call (mppi.py:223-460) = mppi_bind_state (x0_dev != NULL; NULL keeps the state
 * already set) + mppi_sample(solve_idx) + mppi_rollout_cost + [lambda == MPPI_LAMBDA_DEVICE: the configured rule —
 * mppi_essps_lambda_device / mppi_lbps_lambda_device before the weights, mppi_mpo_step_device after mppi_finalize] +
 * mppi_weights_reduce(lambda) + mppi_finalize(own summary — or all shards' with option "exchange_comm" /
 * "exchange_p2p" — store_mean = 1).  Same kernels and results as the individual calls; one host -> library transition
 * per solve and no host synchronisation for any temperature rule. */
int mppi_solve(mppi_handle_t h, const float* x0_dev, uint32_t solve_idx, float lambda, float* action_out_dev,
               float* state_seq_out_dev, float* stats_out_dev, void* stream);
/* Lazily completed state sequences (mppi_set_option("lazy_state_seq", 1)).  The reference returns `state_seq` — the batch-1
 * rollout of the solution, mppi.py:448-449,508-524 — with the action sequence, but nothing on a control loop's critical
 * path needs it: the next solve samples around the mean, env.step applies a[0].  With the option set, mppi_finalize /
 * mppi_solve of a native model on the multi-kernel path leave those T dependent steps out of the solve's last kernel; the
 * rollout (same code, same bits) then rides in one extra block of the NEXT mppi_rollout_cost / mppi_solve launch on that
 * stream — hidden behind its N-sample rollout — or, when somebody wants the state sequence before that, runs as a one-wave
 * kernel of its own: a reader of state_seq_out_dev calls mppi_join_state_seq(h, serial, its stream) first (no-op when
 * nothing is pending; the single launch of small problems always rolls out itself).  state_seq_out_dev must stay valid
 * until then.  mppi_state_seq_serial: the number of the last mppi_finalize and whether its state sequence is pending;
 * handing that number to mppi_join_state_seq makes a reader of an OLDER solve's sequence launch nothing. */
int mppi_join_state_seq(mppi_handle_t h, uint32_t serial /* 0 = whatever is pending */, void* stream);
int mppi_state_seq_serial(mppi_handle_t h, uint32_t* serial_out, int* pending_out);
/* out2_host = {mean device time of the stand-alone state-sequence kernel [ms] (-1: none), launches} since the last call
 * (option "timing" = 1). */
int mppi_get_state_seq_timing(mppi_handle_t h, float* out2_host);
/* mppi_solve as a SINGLE LAUNCH.  For small problems (option "fused_solve" = 1, the default: num_samples <= 4096, the
 * sizes of the reference's examples; <= 16 384 under a device-resident ESSPS / LBPS search) whose noise is regenerated in
 * registers, with T*dim_control <= 128 and no sharding, mppi_solve runs ONE cooperative kernel instead of 3-9 dependent
 * launches: min(#CUs, ceil(N/64)) blocks of 512 threads (at most 32 of them up to 4096 samples) exchange the partial sums
 * of the temperature search and their partial weighted rows through 8-byte {value, solve number} cells in HBM (relaxed
 * agent-scope stores, polled); with up to 32 blocks every block runs the search's scalar step itself and no hop is spent
 * on the global minimum (a block's sums are relative to its own minimum and rescaled where they are added), beyond
 * block 0 does and broadcasts; block 0 runs the tail of the solve.  Costs and minimum are bit-identical to the
 * multi-kernel path; temperature, action and state sequences equal to rounding (another summation partition).
 * "fused_solve" = 0 keeps the multi-kernel path, = 2 takes the single launch whenever every block can be resident at once
 * (num_samples <= 512 x #CUs = 131 072; measured on par or slower than the multi-kernel path beyond the defaults above:
 * a cell round trip costs what a kernel boundary costs).  Co-residency of the blocks is checked against the kernel's own
 * occupancy before every launch configuration is used (hipOccupancyMaxActiveBlocksPerMultiprocessor x #CUs; a grid that
 * does not fit takes the multi-kernel path in the same call); what other work holds of the device at run time cannot be
 * known at launch, so a poll that cannot complete within 20 ms (option "fused_timeout_us": 100 us .. 60 s, for a GPU that is
 * shared or preempted for longer) gives up: that solve returns the PREVIOUS plan (the warm start it sampled around,
 * unchanged, and its rollout from the current state — never NaN, never a partial combine), the statistics of THAT solve
 * are NaN (the `stats_out` of that call: the stale plan is visible in the solve it happened in), the flag below is
 * raised and the handle stays on the multi-kernel path until mppi_set_option("fused_rearm", 1).  It never hangs. */
int mppi_fused_error(mppi_handle_t h);
/* Step 7 inside mppi_finalize (mppi.py:423-443,598-620): Savitzky-Golay smoothing of [history(T-1); a(T)] per control
 * dimension (symmetric-flip padding, valid cross-correlation, keep the last T), applied whenever mppi_finalize is
 * called with store_mean != 0; the smoothed sequence is what is returned, stored as the warm start and rolled out,
 * and its first row is shifted into the history.  coeffs_host [window] = first row of pinv(vander) (mppi.py:568-596);
 * history_host [T-1][dc] or NULL (keep / zeros).  window = 0 switches the filter off.  T*dc <= 1024. */
int mppi_set_sg_filter(mppi_handle_t h, const float* coeffs_host, int window, const float* history_host);
int mppi_get_sg_history(mppi_handle_t h, float* history_host);

/* Device half of the automatic temperature searches (`_compute_ess`, `_lbps_objective`,
 * `_essps_objective`, the MPO dual; mppi.py:341-370,387-398,526-566): softmax statistics of this
 * shard's costs for one lambda, out5_host = {min c, max c, sum e, sum e^2, sum e*c} with
 * e_i = exp((-c_i)/lambda - (-min c)/lambda).  ESS = (sum e)^2 / sum e^2, E_w[c] = sum e*c / sum e,
 * logsumexp(-c/lambda) = -min c/lambda + log(sum e).  The root-finders stay on the host.  Synchronises. */
int mppi_softmax_stats(mppi_handle_t h, float lambda, double* out5_host, void* stream);
/* The same for `count` (1..32) temperatures in ONE pass over the costs: out_host[count][3] =
 * {sum e, sum e^2, sum e*c} per lambda (each relative to this shard's min c).  Lets the ESSPS
 * bracketing search probe a whole grid of lambdas per round trip.  Synchronises. */
int mppi_softmax_stats_multi(mppi_handle_t h, const float* lambdas_host, int count, double* out_host, void* stream);
/* ESSPS (mppi.py:351-370): lambda in [lam_min, lam_max] with ESS(lambda) = target_ess (end-point rules of
 * mppi.py:361-364), searched on the host from two 32-temperature passes of mppi_softmax_stats_multi and an
 * inverse polynomial interpolation (within ~1e-7 relative of scipy's brentq on the same statistics).  From the second
 * search of a handle on, the first grid is clustered around the previous root (with the end points and a sparse cover
 * of the rest of the range); when the root is found well inside the cluster and the interpolation has visibly converged
 * there (polynomials of two orders agree to 1e-5) the search ends after ONE pass (~1e-6 relative), otherwise the
 * bracket is refined by the second grid as in a cold search.  Option "essps_cold" makes the next search cold.
 * Unsharded handles; synchronises once or twice. */
int mppi_essps_lambda(mppi_handle_t h, double target_ess, double lam_min, double lam_max, double* lambda_out_host,
                      void* stream);
/* The same search with NO host synchronisation: both statistics passes and both scalar steps (end-point rules /
 * refined grid, root interpolation) run as kernels on `stream` (the second pair returns at once when the first grid
 * was enough), the first grid of the NEXT search is left in device memory, and so is the temperature.  Pass
 * MPPI_LAMBDA_DEVICE as the `lambda` of mppi_weights_reduce / mppi_finalize to use it; mppi_get_lambda reads it back
 * (synchronises the stream).  Identical arithmetic to mppi_essps_lambda (both call csrc/host_search.hpp). */
#define MPPI_LAMBDA_DEVICE (-1.0f)
int mppi_essps_lambda_device(mppi_handle_t h, double target_ess, double lam_min, double lam_max, void* stream);
/* Passes over the costs the last device-resident search took (ESSPS: 1 or 2; LBPS: 2); 0 = none yet.  Synchronises. */
int mppi_search_passes(mppi_handle_t h, void* stream);
/* The temperature a device-resident rule left in HBM, and (lambda_used_out_host != NULL) the one the last solve's weights
 * used — the same value for ESSPS / LBPS, the previous one for MPO (mppi.py:387-398 updates it AFTER the weights).
 * Synchronises `stream` (pass the stream the solve was enqueued on). */
int mppi_get_lambda(mppi_handle_t h, double* lambda_out_host, double* lambda_used_out_host, void* stream);
/* LBPS (mppi.py:341-349,534-557) with NO host synchronisation: the reference's ~25 dependent Brent probes become two
 * 32-temperature grids (one pass over the costs each; the grid minimum is bracketed by the second, finer grid) and the
 * minimiser of the quartic through the five grid points around the minimum in log(lambda) — the float64 minimiser to 3e-7
 * on exact statistics; against the reference's own Brent (which stops 6e-5 .. 5e-3 away from it) within 1e-3 relative
 * wherever the objective is not flat to fp32 rounding.  The temperature stays in HBM (MPPI_LAMBDA_DEVICE). */
int mppi_lbps_lambda_device(mppi_handle_t h, double delta, double lam_min, double lam_max, void* stream);
/* LBPS exactly as the reference searches it (mppi.py:341-349: scipy.optimize.minimize_scalar(method="bounded"), ported
 * step for step in csrc/host_search.hpp) with NO host synchronisation: one launch (lbps_brent_kernel) runs all ~22-31
 * dependent probes; every probe evaluates the statistics of mppi_softmax_stats bit for bit (the same threads add the same
 * costs in the same order) and every block takes the same double-precision Brent step, so the temperature equals
 * mppi_lbps_lambda's TO THE BIT — at one hop through memory per probe instead of two launches and a read-back.  The
 * temperature stays in HBM (MPPI_LAMBDA_DEVICE; mppi_get_lambda reads it back, mppi_search_passes returns the number of
 * probes).  This is what mppi_solve runs for MPPI_AUTO_LBPS (option "lbps_search" = 1 selects the grid search above).
 * mppi_search_error: 1 once a poll of that kernel gave up (budget = option "fused_timeout_us"; a block never became
 * resident because the GPU is shared): the temperature of that solve is NaN; option "search_rearm" clears the flag. */
int mppi_lbps_brent_device(mppi_handle_t h, double delta, double lam_min, double lam_max, void* stream);
int mppi_search_error(mppi_handle_t h);
/* LBPS (mppi.py:341-349,534-557): argmin over [lam_min, lam_max] of -(E_w[-c] - (max c - min c) * sqrt((1-delta)/delta)
 * / sqrt(ESS)), searched on the host with Brent's bounded minimiser (scipy minimize_scalar(method="bounded"): xatol
 * 1e-5, at most 500 evaluations); every probe is one mppi_softmax_stats round trip.  Unsharded handles; synchronises. */
int mppi_lbps_lambda(mppi_handle_t h, double delta, double lam_min, double lam_max, double* lambda_out_host, void* stream);
/* MPO (mppi.py:191-200,387-398): the dual variable log T and its Adam moments live in the handle.
 *   mppi_mpo_reset  log T = log(lambda0), moments cleared; epsilon = the KL bound (0.1), lr = Adam step (0.2)
 *   mppi_mpo_step   one Adam step on loss = softplus(logT) * (epsilon + logsumexp(-c / softplus(logT))) over the LAST
 *                   solve's costs; *lambda_out_host = exp(logT), the temperature of the NEXT solve.  Synchronises.
 *   mppi_mpo_state  out4_host = {log T, first moment, second moment, step count}.
 *   mppi_mpo_set_state  the inverse (restoring a saved solver); the next solve's temperature becomes exp(log T). */
int mppi_mpo_reset(mppi_handle_t h, double lambda0, double epsilon, double lr);
int mppi_mpo_set_state(mppi_handle_t h, const double* in4_host);
int mppi_mpo_step(mppi_handle_t h, double* lambda_out_host, void* stream);
/* mppi_mpo_step without the read-back (no host synchronisation): the dual, its moments and the resulting temperature
 * stay in device memory; the NEXT solve's weights read it through MPPI_LAMBDA_DEVICE.  Call it after mppi_finalize. */
int mppi_mpo_step_device(mppi_handle_t h, void* stream);
int mppi_mpo_state(mppi_handle_t h, double* out4_host);
/* Device address of the dual's log T (fp32; read-only for the caller — valid until mppi_destroy): the reference keeps it as an
 * nn.Parameter (mppi.py:194-199); a binding can wrap this address instead of copying the value after every solve. */
int mppi_mpo_log_temperature_ptr(mppi_handle_t h, float** out_dev);

/* `_weights` (mppi.py:376) for this shard given the GLOBAL {min c, sum e}: w_out_dev[N]. */
int mppi_weights(mppi_handle_t h, float lambda, float cmin_global, float sum_e_global, float* w_out_dev,
                 void* stream);
/* `_states_prediction(state, action_seqs)` (mppi.py:508-524) for k action sequences actions_dev[k][T][dc] ->
 * states_out_dev[k][T+1][ds] (step 8 after host-side smoothing; get_samples_from_posterior).  x0_dev = the start
 * state [ds] (device), or NULL for the state of the current solve; an explicit state does not disturb the solver. */
int mppi_rollout_actions(mppi_handle_t h, const float* actions_dev, int k, const float* x0_dev, float* states_out_dev,
                         void* stream);
/* get_samples_from_posterior, sampling half (mppi.py:489-503): samples_out_dev[k][T][dc] = loc_dev[T][dc] + eps with
 * eps ~ N(0, diag(sigma^2)) (unclamped, like MultivariateNormal(loc).sample()) from the device Philox stream at the
 * RESERVED solve index `solve_idx` (counter = (sample, float4 group, solve_idx): the call consumes the solver's
 * stream the way the reference's draw consumes torch's generator; sharding does not change the samples). */
int mppi_sample_posterior(mppi_handle_t h, uint32_t solve_idx, const float* loc_dev, int k, float* samples_out_dev,
                          void* stream);
/* `_state_seq_batch[top_indices]` (mppi.py:481): re-roll the k local samples idx_dev[k] from the
 * resident noise instead of materialising S[N][T+1][ds] -> states_out_dev[k][T+1][ds]. */
int mppi_rollout_samples(mppi_handle_t h, const int64_t* idx_dev, int k, float* states_out_dev, void* stream);
/* get_top_samples (mppi.py:462-487) in one call: the k (any 1 <= k <= num_samples) samples of the last solve with the largest
 * weight = the smallest cost (radix select on the device), sorted by descending weight, their state
 * (k <= 1024: one block sorts the candidates in LDS; larger k: a multi-pass bitonic sort in HBM), their state
 * trajectories re-rolled around the mean that solve sampled (states_out_dev [k][T+1][ds]) and their softmax
 * weights softmax(-c/lambda)_i (weights_out_dev [k]).  lambda = the temperature of that solve, or
 * MPPI_LAMBDA_DEVICE = the one its finalize step left in device memory (no read-back: the call never waits for the
 * host).  Up to 4096 samples and k <= 1024 (the reference examples call this every tick with such sizes) the whole
 * query is ONE launch: a block builds the (cost, index) words of all samples, sorts them in LDS and re-rolls the first k.
 * On a shard this ranks the shard's own samples (weights still use the global normalisation); see the two calls below. */
int mppi_top_samples(mppi_handle_t h, int k, float lambda, float* states_out_dev, float* weights_out_dev, void* stream);
/* The two halves of mppi_top_samples for sharded solvers.  A candidate is (cost key << 32) | GLOBAL sample index; the
 * key is an order-preserving bijection of the fp32 cost, so candidates of all shards can be merged by sorting the
 * 64-bit words ascending, and — the device noise being a function of the global index — any rank can then weigh
 * and re-roll the k winners:
 *   mppi_top_candidates:     this shard's k smallest costs -> cand_out_dev[k] (unordered)
 *   mppi_rollout_candidates: k merged candidates -> states_out_dev[k][T+1][ds], weights_out_dev[k], sorted by
 *                            descending weight (needs regenerated noise: option "noise_regen" = 1, no injection). */
int mppi_top_candidates(mppi_handle_t h, int k, uint64_t* cand_out_dev, void* stream);
int mppi_rollout_candidates(mppi_handle_t h, const uint64_t* cand_dev, int k, float lambda, float* states_out_dev,
                            float* weights_out_dev, void* stream);

/* What RCCL reports for this handle's communicator (ncclCommCount, ncclCommUserRank): diagnostics for multi-GPU runs. */
int mppi_comm_info(mppi_handle_t h, int* count_out, int* rank_out);
/* In-library collective for sharded solves (variant A: one all_gather of the 4+T*dc-float shard summaries; the
 * reference has no counterpart, src/pi_mpc/mppi.py:102-105 is single-device).  One process per GPU; RCCL is dlopen()ed
 * (librccl.so.1) on first use, so unsharded callers do not need it.
 *   mppi_comm_unique_id  ncclGetUniqueId on ONE rank: id_out128 [128 bytes]; hand it to every rank by any host channel
 *   mppi_comm_init       ncclCommInitRank(world, rank, id) for this handle's device — collective over the job's ranks
 *   mppi_comm_exchange   one stand-alone all_gather (self-test): data_dev [4+T*dc] -> gathered_out_dev [world][4+T*dc];
 *                        synchronises
 *   option "exchange_comm" = 1: mppi_weights_reduce (summary_out_dev = NULL) ends with ncclAllGather on ITS OWN stream —
 *                        no process-group stream, no events — and mppi_finalize called with summaries_dev = NULL combines
 *                        all `world` shards; a sharded solve is then the same single call as an unsharded one
 *                        (mppi_solve).  Every rank must issue the same sequence of solves.
 *   mppi_comm_destroy    ncclCommDestroy (also done by mppi_destroy). */
int mppi_comm_unique_id(void* id_out128);
int mppi_comm_init(mppi_handle_t h, int world, int rank, const void* id128);
int mppi_comm_exchange(mppi_handle_t h, const float* data_dev, float* gathered_out_dev, void* stream);
int mppi_comm_destroy(mppi_handle_t h);

/* Peer-to-peer exchange of the shard summaries (one process per GPU, same node): instead of an all_gather between
 * mppi_weights_reduce and mppi_finalize, every rank stores its 4+T*dc summary straight into all peers' exchange
 * buffers (fine-grained device memory shared through HIP IPC, 8-byte {value, sequence} cells so that data and
 * readiness arrive in one store) and mppi_finalize polls its own buffer.
 *   mppi_p2p_alloc    allocate this rank's buffer, return its 64-byte IPC handle (exchange the handles of all
 *                     ranks with any host-side collective)
 *   mppi_p2p_connect  map the peers' buffers: handles_host [world][64] and the ranks' HIP device ordinals
 *                     peer_devices_host [world], both in rank order; refuses (MPPI_E_STATE) unless every peer device
 *                     is visible and hipDeviceCanAccessPeer says it is directly addressable
 *   mppi_p2p_exchange one stand-alone exchange (self-test): data_dev [4+T*dc] -> gathered_out_dev [world][4+T*dc];
 *                     MPPI_E_STATE if a poll timed out (~20 s).  Synchronises.
 *   option "exchange_p2p" = 1: mppi_weights_reduce publishes the summary to the peers, and mppi_finalize called with
 *                     summaries_dev = NULL combines all `world` shards from the buffer.  Every rank must issue the
 *                     same sequence of reduce / exchange calls.
 *   mppi_p2p_error    1 once any poll on this handle timed out (results of that solve are void). */
int mppi_p2p_alloc(mppi_handle_t h, int world, int rank, void* ipc_handle_out64);
int mppi_p2p_connect(mppi_handle_t h, const void* ipc_handles_host, const int32_t* peer_devices_host);
int mppi_p2p_exchange(mppi_handle_t h, const float* data_dev, float* gathered_out_dev, void* stream);
int mppi_p2p_error(mppi_handle_t h);

/* Tuning knobs (not in the reference): "math" 0 = library sin/cos/tan/fmod/div, 1 = range-checked polynomial
 * fast paths, 2 = 1 + the hardware sin/cos for model-bounded arguments (default); "fused_solve", "fused_timeout_us", "fused_rearm" (see mppi_fused_error);
 * "lazy_state_seq" (see mppi_join_state_seq; off by default);
 * "essps_cold" (any value): the next ESSPS search of this handle starts from the geometric grid instead of the one
 * clustered around its last root; "noise_regen" (see mppi_sample); "mapping" 0 = lane per trajectory (default), 1 =
 * the north star's literal wavefront-per-trajectory rollout (comparison only, ~20x slower); "reduce_blocks" grid of the weighted
 * reduction; "fold_path" who sums the reduction's partial rows: 0 = by the live-row count of earlier solves (default),
 * 1 = inside mppi_finalize whenever the rows fit its LDS, 2 = always the separate summarize kernel (both use the same
 * summation tree: results are bit-identical); "essps_merge0" 1 = round 0 of mppi_essps_lambda_device as one launch too
 * (statistics pass + select step; round 1 always is: it returns at once when round 0 finished the search) — same
 * temperature to the bit, measured no faster (default 0); "timing" (see mppi_get_timing). */
int mppi_set_option(mppi_handle_t h, const char* key, int64_t value);
/* Device time per stage from HIP event pairs recorded on the caller's stream around every stage call
 * since the last drain (no host synchronisation while recording): out[0..3] = mean ms of
 * {sample, rollout_cost, weights_reduce, finalize}, out[4..7] = number of calls averaged.
 * Enabled by mppi_set_option(h, "timing", 1).  Synchronises and clears the recorded pairs. */
int mppi_get_timing(mppi_handle_t h, float* out_ms8);

#ifdef __cplusplus
}
#endif
#endif /* MPPI_HIP_H */
