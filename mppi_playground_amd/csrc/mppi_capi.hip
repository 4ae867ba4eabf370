// mppi_capi.hip — C ABI (include/mppi_hip.h) over the gfx950 kernels in mppi_kernels.hpp.
// Host-side state only: buffers, model context, launch geometry.  No torch, no exceptions across
// the boundary.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is dlopen()ed when a communicator is asked for

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mppi_hip.h"
#include "host_search.hpp"
#include "mppi_kernels.hpp"

using namespace mppi;

// mppi_finalize folds the partial rows itself while the reductions publish at most this many (sparse softmax);
// beyond it the multi-block summarize_kernel is cheaper than one block walking the rows.
static constexpr int FOLD_IN_FINALIZE_MAX_ROWS = 64;

struct MppiSolver {
    MppiConfig cfg{};
    Dims d{};
    int ds = 0, dc = 0;
    // device buffers
    float4* noise = nullptr;
    float* costs = nullptr;
    unsigned* min_key = nullptr;   // two slots, toggled per rollout (no memset between solves)
    int min_slot = 0;
    float* x0 = nullptr;           // owned copy of the state ...
    const float* x0_cur = nullptr; // ... or a borrowed device pointer (mppi_bind_state)
    float* x0_used = nullptr;      // the state the last rollout started from (snapshot taken by the rollout kernel)
    // generic handles whose dim_control is not 1, 2 or 4: per-column {sigma, lo, hi}[4R] table (see gen_noise4)
    float* coltab = nullptr;
    bool wide = false, limits_set = true;
    mppi::host::MpoState* mpo_dev = nullptr;  // MPO temperature dual + Adam moments, resident on the device (mppi_mpo_*)
    float* mpo_temp_dev = nullptr;            // softplus(log T): the temperature of the dual's next statistics pass
    LbpsDev* lbps_dev = nullptr;              // grids of the device-resident LBPS search
    float* stats_max = nullptr;               // [STATS_BLOCKS] per-block maximum cost (LBPS: the cost range)
    double lbps_lo = 0.0, lbps_hi = 0.0;      // [lam_min, lam_max] the preset round-0 grid was built for
    // device-resident Brent search of LBPS (lbps_brent_kernel): tagged cells, probe-tag base, error flag, the rule's variant
    unsigned long long* brent_cells = nullptr;  // [2][BRENT_LANES][BRENT_CELLS]
    unsigned brent_seq = 0;
    int* search_error = nullptr;               // mapped pinned: a poll of lbps_brent_kernel timed out
    int* search_error_dev = nullptr;
    int lbps_grid = 0;                         // option "lbps_search": 0 = Brent on the device (default), 1 = the two-grid search
    int brent_drop_block = 0;                  // test hook (option "search_test_drop_block"): launch one block too few
    int lds_max = 65536;                       // hipDeviceAttributeMaxSharedMemoryPerBlock
    // single-launch solve (solve_fused_kernel): cells the blocks exchange through, its error flag, the solve counter
    unsigned long long* fused_cells = nullptr;
    int* fused_error = nullptr;        // mapped pinned
    int* fused_error_dev = nullptr;
    unsigned fused_seq = 0;
    uint64_t fused_occ_key = 0;        // (math level, LDS bytes) the cached occupancy below belongs to
    int fused_occ_blocks = 0;          // resident blocks of solve_fused_kernel per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor)
    int fused_mode = 1;                // option "fused_solve": 0 = never, 1 = small problems (default), 2 = whenever resident
    long long fused_timeout_ticks = mppi::FUSED_TIMEOUT_TICKS;  // option "fused_timeout_us" (100 MHz ticks)
    int cu_count = 0;
    double* grid0_dev = nullptr;       // [STATS_L] round-0 grid of the fused LBPS search (ESSPS: essps_dev->grid0)
    double grid0_lo = 0.0, grid0_hi = 0.0;
    // the temperature rule mppi_solve applies when called with MPPI_LAMBDA_DEVICE (mppi_set_auto_lambda)
    int auto_rule = 0;
    double auto_param = 0.0, auto_lo = 0.0, auto_hi = 0.0;
    // pinned staging ring for small host -> device uploads without a stream synchronisation
    static constexpr int RING = 8;
    float* stage[RING] = {};
    hipEvent_t stage_ev[RING] = {};
    size_t stage_floats = 0;
    int stage_next = 0;
    // noise identity of the current solve and whether the tiles hold it
    GenCtx gen{};
    int last_reduce_blocks = 0;    // grid of the last weights_reduce (finalize folds its partials)
    int mapping = 0;               // 0: lane per trajectory (default); 1: wavefront per trajectory (comparison)
    float* noise_std = nullptr;    // [N][T][dc] copy of the noise for the wavefront-per-trajectory variant
    int noise_regen = 1;           // 1: Philox noise is regenerated in the kernels, never stored
    bool tiles_valid = false;      // the noise tiles hold the current solve's noise
    bool injected = false;         // ... because it was injected (cannot be regenerated)
    float* mean = nullptr;
    float* mean_used = nullptr;      // the mean the last rollout sampled around (snapshot taken by the rollout kernel)
    float* solve_stats = nullptr;    // [8]: {min c, sum e, sum e^2, sum e*c, lambda used} over all shards of the last finalize
    unsigned* topk_hist = nullptr;   // [3][TOPK_BINS] + 2 counters, kept zeroed between calls
    TopkSel* topk_sel = nullptr;     // [3]
    unsigned long long* topk_cand = nullptr;  // [topk_cap] (a power of two >= TOPK_MAX: the large-k sort pads to it)
    size_t topk_cap = 0;
    // peer-to-peer exchange of the shard summaries (mppi_p2p_*): off unless connected and enabled
    unsigned long long* p2p_local = nullptr;       // this rank's exchange buffer (fine-grained, IPC-exported)
    unsigned long long** p2p_peers_dev = nullptr;  // device array [world] of every rank's buffer as mapped here
    std::vector<void*> p2p_opened;                 // peer mappings to close
    int* p2p_error = nullptr;                      // mapped pinned flag raised by a timed-out poll
    int* p2p_error_dev = nullptr;
    int p2p_world = 0, p2p_rank = 0, p2p_lenp = 0;
    unsigned p2p_seq = 0;
    bool p2p_connected = false, p2p_enabled = false;
    // in-library collective (mppi_comm_*): one ncclAllGather of the shard summaries on the solve's own stream
    ncclComm_t comm = nullptr;
    int comm_world = 0, comm_rank = 0;
    float* comm_send = nullptr;      // [4 + T*dc] this shard's summary (written by summarize_kernel)
    float* comm_recv = nullptr;      // [world][4 + T*dc]
    bool comm_enabled = false;
    float* sg_coeffs = nullptr;      // Savitzky-Golay taps (device), window sg_window (0 = filter off)
    float* sg_history = nullptr;     // [T-1][dc] `_actions_history_for_sg` (mppi.py:160-166,441-443)
    int sg_window = 0;
    float* ref = nullptr;
    int ref_cap = 0;
    // device-resident reference window (mppi_set_center_path / mppi_ref_window)
    float* center8 = nullptr;        // [n][8] centre line with sin/cos of the yaw
    int32_t* win_dind = nullptr;     // [rows] index offsets of the window rows
    int32_t* path_index = nullptr;   // `current_path_index`, kept on the device
    int center_n = 0, win_rows = 0;
    float win_v = 0.0f;
    float* partials = nullptr;
    float* heads = nullptr;
    float* summary = nullptr;
    float* stats_part = nullptr;     // [STATS_BLOCKS][max(4, STATS_L*3)]
    unsigned long long* round1_cells = nullptr;  // [STATS_BLOCKS][STATS_L*3] {value, launch number}: essps_round1_kernel
    unsigned round1_seq = 0;
    bool essps_merge0 = false;  // round 0 as one launch too (option "essps_merge0": measured on par at 65 536 samples and
                                // 3.8 us SLOWER at 262 144 — profiles/r04_experiments.md; round 1 is merged for its skip case)
    double* stats_host_dev = nullptr;  // the device's view of stats_host
    double* stats_host = nullptr;    // mapped pinned [8 + STATS_L*3 + 3]: single-lambda stats, grid stats, device-searched lambda (next, used), its passes
    float* lams_dev = nullptr;       // [3][STATS_L]: caller's grid, ESSPS round-0 grid (preset), ESSPS round-1 grid (device-written)
    EsspsDev* essps_dev = nullptr;   // state of the device-resident ESSPS search
    float* lambda_dev = nullptr;     // the temperature that search left on the device (MPPI_LAMBDA_DEVICE)
    double essps_lo = 0.0, essps_hi = 0.0;  // [lam_min, lam_max] the device search's first grid was built for
    mppi::host::EsspsRange essps_range{};   // ... with its logs
    mppi::host::EsspsRoot essps_prev_host{0.0, 0.0, false};  // mppi_essps_lambda: last root (warm start of the next search) ...
    double essps_prev_lo = 0.0, essps_prev_hi = 0.0;          // ... and the range it was searched in
    bool lambda_dev_valid = false;
    uint8_t* map_cells[2] = {nullptr, nullptr};
    uint8_t* map_pad = nullptr;      // padded (and, for racing, summed) grid of the FAST lookup
    size_t map_bytes[2] = {0, 0}, map_pad_bytes = 0;
    bool params_set = false;
    ModelCtx ctx{};
    // options
    int math_fast = 2;
    int reduce_blocks = 512;
    int reduce_chains = 0;             // option "reduce_chains": 0 = by the grid size, 2 / 4 = pinned (A/B)
    int timing = 0;
    std::vector<hipEvent_t> ev_pool[5];  // per stage (4 = the deferred state sequence): start0, stop0, start1, stop1, ...
    size_t ev_used[5] = {0, 0, 0, 0, 0};
    int GPW = 8, nchunks = 1, colsp = 128;  // reduce: float4 groups per wave, column chunks, padded row
    bool summary_valid = false;             // summarize_kernel ran after the last reduce
    int* live_hint = nullptr;               // mapped pinned: partial rows the last fold saw (host-side hint)
    int* live_hint_dev = nullptr;
    int fold_mode = 0;                      // 0: choose by the hint; 1: fold inside finalize when it fits; 2: always summarize
    // lazily completed state sequence (option "lazy_state_seq"): finalize_kernel leaves {action, start state} in `b1` and
    // the batch-1 rollout of the solution rides in ONE EXTRA BLOCK of the next rollout kernel on the same stream
    // (mppi_rollout_cost) — or runs as its own one-wave kernel when somebody asks for it first (mppi_join_state_seq)
    int lazy_state = 0;
    float* b1 = nullptr;                    // [row + MPPI_MAX_DIM_STATE]
    float* pending_state_out = nullptr;     // where the not-yet-rolled-out state sequence of the last solve goes (or null)
    uint32_t pending_serial = 0;            // which solve that is (mppi_join_state_seq)
    hipStream_t pending_stream = nullptr;   // the stream its finalize ran on: a completion on ANOTHER stream waits for it
    hipEvent_t lazy_ev = nullptr;           // (created on first use: orders such a completion behind finalize's write of b1)
    uint32_t finalize_serial = 0;
    std::string err;
};

namespace {

int fail(mppi_handle_t h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}
#define HIP_TRY(h, expr)                                                                              \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fail(h, MPPI_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)

struct ModelDims { int ds, dc; };
bool model_dims(int model, ModelDims& md) {
    switch (model) {
    case MPPI_MODEL_PENDULUM: md = {2, 1}; return true;
    case MPPI_MODEL_CARTPOLE: md = {4, 1}; return true;
    case MPPI_MODEL_MOUNTAINCAR: md = {2, 1}; return true;
    case MPPI_MODEL_NAV2D: md = {3, 2}; return true;
    case MPPI_MODEL_RACING: md = {4, 2}; return true;
    case MPPI_MODEL_MJCARTPOLE: md = {4, 1}; return true;
    case MPPI_MODEL_GOALZONE: md = {7, 2}; return true;
    }
    return false;
}

// Brackets one stage with a pair of HIP events on the caller's stream (no host synchronisation);
// pairs accumulate until mppi_get_timing() drains them.
struct StageTimer {
    mppi_handle_t h; int stage; hipStream_t s; hipEvent_t stop = nullptr;
    static hipEvent_t next(mppi_handle_t h, int stage) {
        auto& pool = h->ev_pool[stage];
        if (h->ev_used[stage] == pool.size()) {
            if (pool.size() >= 16384) return nullptr;
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[h->ev_used[stage]++];
    }
    StageTimer(mppi_handle_t h_, int stage_, hipStream_t s_) : h(h_), stage(stage_), s(s_) {
        if (!h->timing || (h->timing == 2 && stage != 1)) return;  // timing = 2: rollout_cost stage only
        hipEvent_t start = next(h, stage);
        stop = start ? next(h, stage) : nullptr;
        if (start && stop) (void)hipEventRecord(start, s);
        else if (start) { --h->ev_used[stage]; }
    }
    ~StageTimer() {
        if (stop) (void)hipEventRecord(stop, s);
    }
};

int math_level(mppi_handle_t h);
// dispatch on (model, math level); level 2 exists for the models whose trigonometric arguments are bounded by the model
// itself (wrapped headings, clamped pole angle / position): all but the pendulum, whose angle is free, and the
// MuJoCo-style cart-pole, whose open-loop instability amplifies the hardware sin/cos error 20-fold
#define MPPI_DISPATCH_HW(MODEL_, CALL)                                                                \
        case MODEL_: if (ml_ == 2) { CALL(MODEL_, 2); } else if (ml_ == 1) { CALL(MODEL_, 1); } else { CALL(MODEL_, 0); } break;
#define MPPI_DISPATCH_NOHW(MODEL_, CALL)                                                              \
        case MODEL_: if (ml_) { CALL(MODEL_, 1); } else { CALL(MODEL_, 0); } break;
#define MPPI_DISPATCH(h, CALL)                                                                        \
    do {                                                                                              \
        const int ml_ = math_level(h);                                                                \
        switch ((h)->cfg.model) {                                                                     \
        case MPPI_MODEL_GENERIC: /* only reached by mppi_finalize without a state output */          \
        MPPI_DISPATCH_NOHW(MPPI_MODEL_PENDULUM, CALL)                                                 \
        MPPI_DISPATCH_HW(MPPI_MODEL_CARTPOLE, CALL)                                                   \
        MPPI_DISPATCH_HW(MPPI_MODEL_MOUNTAINCAR, CALL)                                                \
        MPPI_DISPATCH_HW(MPPI_MODEL_NAV2D, CALL)                                                      \
        MPPI_DISPATCH_HW(MPPI_MODEL_RACING, CALL)                                                     \
        MPPI_DISPATCH_NOHW(MPPI_MODEL_MJCARTPOLE, CALL)                                               \
        MPPI_DISPATCH_HW(MPPI_MODEL_GOALZONE, CALL)                                                   \
        }                                                                                             \
    } while (0)

// FAST kernels assume launch-uniform preconditions (see mppi_models.hpp); otherwise use FAST=false.
bool use_fast(mppi_handle_t h) {
    if (!h->math_fast) return false;
    const int m = h->cfg.model;
    if (m == MPPI_MODEL_NAV2D)
        return h->ctx.maps[0].inv_cell != 0.0f && h->ctx.pad != nullptr && h->ctx.wrap_safe != 0 && h->ctx.u_in_bounds != 0;
    if (m == MPPI_MODEL_GOALZONE) return h->ctx.wrap_safe != 0 && h->ctx.u_in_bounds != 0;
    if (m == MPPI_MODEL_RACING)
        return h->ctx.wrap_safe != 0 && h->ctx.u_in_bounds != 0 && h->ctx.maps[0].inv_cell != 0.0f && h->ctx.pad != nullptr && h->ctx.tan_small != 0 && h->ctx.inv_L != 0.0f;
    return true;
}

// 0 = library math; 1 = polynomial fast paths; 2 = 1 + hardware sin/cos of the wrapped headings (option "math")
int math_level(mppi_handle_t h) { return use_fast(h) ? (h->math_fast >= 2 ? 2 : 1) : 0; }

int check_ready(mppi_handle_t h) {
    const int m = h->cfg.model;
    if (m == MPPI_MODEL_GENERIC)
        return fail(h, MPPI_E_INVALID, "generic model: dynamics/cost are host callables, this entry point is unavailable");
    if ((m == MPPI_MODEL_NAV2D || m == MPPI_MODEL_RACING) && !h->map_cells[0])
        return fail(h, MPPI_E_STATE, "obstacle map (slot 0) not uploaded");
    if (m == MPPI_MODEL_RACING && !h->map_cells[1]) return fail(h, MPPI_E_STATE, "lane map (slot 1) not uploaded");
    if (m == MPPI_MODEL_RACING && (!h->ctx.ref || h->ctx.ref_rows < h->d.T))
        return fail(h, MPPI_E_STATE, "reference path not set or shorter than the horizon");
    return MPPI_OK;
}

// (Re)build the padded grid of the FAST lookup when the maps and the model parameters allow it; otherwise
// ctx.pad stays null and the FAST=false kernels (bounds-tested lookups) are dispatched.
void refresh_pad(mppi_handle_t h, hipStream_t s) {
    h->ctx.pad = nullptr;
    h->ctx.pad_stride = 0;
    const int model = h->cfg.model;
    const bool racing = model == MPPI_MODEL_RACING;
    if (!racing && model != MPPI_MODEL_NAV2D) return;
    if (!h->params_set || !h->map_cells[0] || (racing && !h->map_cells[1])) return;
    const MapView &a = h->ctx.maps[0], &b = h->ctx.maps[1];
    if (racing && (a.nx != b.nx || a.ny != b.ny || a.cell != b.cell || a.ox != b.ox || a.oy != b.oy)) return;
    const float* P = h->ctx.P;
    const float xlo = P[racing ? MPPI_RP_XLO : MPPI_NP_XLO], xhi = P[racing ? MPPI_RP_XHI : MPPI_NP_XHI];
    const float ylo = P[racing ? MPPI_RP_YLO : MPPI_NP_YLO], yhi = P[racing ? MPPI_RP_YHI : MPPI_NP_YHI];
    uint32_t koff = 0;
    if (!pad_map_plan(a, xlo, xhi, ylo, yhi, koff)) return;
    const size_t n = (size_t)(a.nx + 1) * (a.ny + 1);
    if (h->map_pad_bytes < n) {
        if (h->map_pad) (void)hipFree(h->map_pad);
        h->map_pad = nullptr; h->map_pad_bytes = 0;
        if (hipMalloc(&h->map_pad, n) != hipSuccess) return;
        h->map_pad_bytes = n;
    }
    hipLaunchKernelGGL(pad_map_kernel, dim3((unsigned)((a.ny + 1 + BLOCK - 1) / BLOCK), (unsigned)(a.nx + 1)), dim3(BLOCK), 0, s,
                       h->map_cells[0], racing ? h->map_cells[1] : (const uint8_t*)nullptr, a.nx, a.ny,
                       (uint8_t)(racing ? 2 : 1), h->map_pad);
    if (hipGetLastError() != hipSuccess) return;
    h->ctx.pad = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(h->map_pad) - (uintptr_t)koff);
    h->ctx.pad_stride = a.ny + 1;
}

// (re)allocate the grid of `slot` and fill in its geometry
int prepare_map(mppi_handle_t h, int slot, int nx, int ny, float cell, float ox, float oy) {
    if (!h || slot < 0 || slot > 1 || nx < 1 || ny < 1 || !(cell > 0.0f)) return fail(h, MPPI_E_INVALID, "bad map");
    const size_t n = (size_t)nx * ny;
    if (h->map_bytes[slot] < n) {
        if (h->map_cells[slot]) { (void)hipFree(h->map_cells[slot]); h->map_cells[slot] = nullptr; h->map_bytes[slot] = 0; }
        HIP_TRY(h, hipMalloc(&h->map_cells[slot], n));
        h->map_bytes[slot] = n;
    }
    MapView& m = h->ctx.maps[slot];
    m.cells = h->map_cells[slot];
    m.nx = nx; m.ny = ny; m.cell = cell; m.ox = ox; m.oy = oy;
    // Markstein division needs y = RN(1/cell) and a significand of cell that is not all ones.
    uint32_t bits; std::memcpy(&bits, &cell, 4);
    const bool all_ones = (bits & 0x7fffffu) == 0x7fffffu;
    m.inv_cell = all_ones ? 0.0f : 1.0f / cell;  // host IEEE division: correctly rounded
    return MPPI_OK;
}

// small integer table host -> device (map recipes); blocking, setup path only
int upload_ints(mppi_handle_t h, const int32_t* src, size_t count, int32_t** dst) {
    *dst = nullptr;
    if (!count) return MPPI_OK;
    HIP_TRY(h, hipMalloc(dst, sizeof(int32_t) * count));
    HIP_TRY(h, hipMemcpy(*dst, src, sizeof(int32_t) * count, hipMemcpyHostToDevice));
    return MPPI_OK;
}

// dynamic LDS of finalize_kernel: [row] action, [W][4 + row] summaries, the Savitzky-Golay staging and — when the
// kernel folds the partial rows itself — [64][row + 3] group sums (see finalize_kernel)
size_t finalize_lds_floats(mppi_handle_t h, int world, int sg_window, bool fold) {
    return (size_t)h->d.row + (size_t)world * (h->d.row + MPPI_SUMMARY_HEAD) +
           (sg_window ? (size_t)(2 * h->d.T - 1 + 2 * (sg_window / 2)) * h->dc : 0) +
           (fold ? (size_t)(SUM_BLOCK / SUM_COLS) * (h->d.row + 3) : 0);
}
// Short rows fold inside finalize_kernel (sparse softmax: no summarize launch); rows whose group sums do not fit the
// 64 KiB of LDS always take summarize_kernel.  A static property of the handle: the choice never depends on timing.
bool fold_fits(mppi_handle_t h) {
    return finalize_lds_floats(h, 1, 255 /* widest filter */, true) * sizeof(float) <= 64 * 1024;
}

// RCCL through dlopen: the library stays loadable (and every unsharded path usable) on a host without RCCL.  In a
// process that already holds a librccl.so.1 (PyTorch bundles one) the loader hands back that copy.
struct RcclApi {
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;        // optional (diagnostics: mppi_comm_info)
    decltype(&ncclCommUserRank) comm_user_rank = nullptr;
    bool ok = false;
};
const RcclApi& rccl() {
    static const RcclApi api = [] {
        RcclApi a;
        void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return a;
        a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(lib, "ncclGetUniqueId"));
        a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(lib, "ncclCommInitRank"));
        a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
        a.all_gather = reinterpret_cast<decltype(a.all_gather)>(dlsym(lib, "ncclAllGather"));
        a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(lib, "ncclGetErrorString"));
        a.comm_count = reinterpret_cast<decltype(a.comm_count)>(dlsym(lib, "ncclCommCount"));
        a.comm_user_rank = reinterpret_cast<decltype(a.comm_user_rank)>(dlsym(lib, "ncclCommUserRank"));
        a.ok = a.get_unique_id && a.comm_init_rank && a.comm_destroy && a.all_gather && a.error_string;
        return a;
    }();
    return api;
}
#define RCCL_TRY(h, expr)                                                                             \
    do {                                                                                              \
        ncclResult_t _r = (expr);                                                                     \
        if (_r != ncclSuccess)                                                                        \
            return fail(h, MPPI_E_HIP, std::string(#expr) + ": " + rccl().error_string(_r));          \
    } while (0)

P2pCtx p2p_ctx(mppi_handle_t h) {
    return P2pCtx{h->p2p_peers_dev, h->p2p_local, h->p2p_error_dev, h->p2p_world, h->p2p_rank, h->p2p_lenp, h->p2p_seq};
}

}  // namespace

static int mpo_upload(mppi_handle_t h, double lambda0, double epsilon, double lr, bool lambda_too);
static int flush_state_seq(mppi_handle_t h, hipStream_t s);
static int settle_state_seq(mppi_handle_t h);
static int order_behind_pending(mppi_handle_t h, hipStream_t s);

extern "C" {

const char* mppi_version(void) { return "mppi_hip 0.3.0 (gfx950, wave64, lane-per-trajectory)"; }
int mppi_abi_version(void) { return MPPI_ABI_VERSION; }

int mppi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* mppi_last_error(mppi_handle_t h) { return h ? h->err.c_str() : "null handle"; }

int mppi_create(const MppiConfig* cfg, mppi_handle_t* out) {
    if (!cfg || !out) return MPPI_E_INVALID;
    *out = nullptr;
    ModelDims md{};
    if (cfg->model == MPPI_MODEL_GENERIC) {
        // any control dimension: 1, 2 and 4 index the launch constants, every other value the per-column table
        if (cfg->dim_state < 1 || cfg->dim_control < 1 || cfg->dim_control > MPPI_MAX_DIM_CONTROL_GENERIC)
            return MPPI_E_INVALID;
        md = {cfg->dim_state, cfg->dim_control};
    } else {
        if (!model_dims(cfg->model, md)) return MPPI_E_INVALID;
        if (cfg->dim_state != md.ds || cfg->dim_control != md.dc) return MPPI_E_INVALID;
    }
    if (cfg->horizon < 1 || cfg->num_samples < 1) return MPPI_E_INVALID;
    if (mppi_device_count() <= 0) return MPPI_E_NODEVICE;
    MppiSolver* h = new (std::nothrow) MppiSolver();
    if (!h) return MPPI_E_INVALID;
    h->cfg = *cfg;
    h->ds = md.ds; h->dc = md.dc;
    Dims& d = h->d;
    d.N = cfg->num_samples;
    d.tiles = (d.N + 63) / 64;
    d.sample_offset = cfg->sample_offset;
    d.inherit_count = cfg->inherit_count;
    d.T = cfg->horizon;
    d.row = d.T * md.dc;
    d.R = (d.row + 3) / 4;
    for (int k = 0; k < MPPI_MAX_DIM_CONTROL; ++k) {
        d.u_min[k] = cfg->u_min[k]; d.u_max[k] = cfg->u_max[k]; d.sigma[k] = cfg->sigmas[k];
    }
    h->wide = cfg->model == MPPI_MODEL_GENERIC && md.dc != 1 && md.dc != 2 && md.dc != 4;
    // Every wave owns 8 float4 groups of a 32-group column chunk; longer rows (T*dim_control > 128) take more chunks
    // (grid.y), each regenerating / reading only its own groups.  (Rounds 1-3 gave such rows 32 groups per wave: 128
    // accumulators per lane, 163-231 VGPRs and 15-66 SGPR spills; the chunked form computes the same sums — a column is
    // owned by one wave either way and accumulates its tiles in the same order — without that kernel.)
    h->GPW = 8;
    const int chg = h->GPW * (BLOCK / WAVE);  // float4 groups per column chunk
    h->nchunks = (d.R + chg - 1) / chg;
    h->colsp = h->nchunks * chg * 4;
    *out = h;  // so that the caller can read the error and destroy on failure
    HIP_TRY(h, hipSetDevice(cfg->device));
    const size_t noise_bytes = (size_t)d.tiles * d.R * 64 * sizeof(float4);
    HIP_TRY(h, hipMalloc(&h->noise, noise_bytes));
    HIP_TRY(h, hipMemset(h->noise, 0, noise_bytes));
    HIP_TRY(h, hipMalloc(&h->costs, sizeof(float) * (size_t)d.N));
    HIP_TRY(h, hipMalloc(&h->min_key, 2 * sizeof(unsigned)));
    HIP_TRY(h, hipMemset(h->min_key, 0xFF, 2 * sizeof(unsigned)));
    const size_t x0_floats = (size_t)std::max(md.ds, MPPI_MAX_DIM_STATE);
    HIP_TRY(h, hipMalloc(&h->x0, sizeof(float) * x0_floats));
    HIP_TRY(h, hipMemset(h->x0, 0, sizeof(float) * x0_floats));
    h->x0_cur = h->x0;
    HIP_TRY(h, hipMalloc(&h->x0_used, sizeof(float) * x0_floats));
    HIP_TRY(h, hipMemset(h->x0_used, 0, sizeof(float) * x0_floats));
    h->gen = GenCtx{(uint32_t)cfg->seed, (uint32_t)(cfg->seed >> 32), 0u};
    d.dc = md.dc;
    HIP_TRY(h, hipMalloc(&h->mean, sizeof(float) * (size_t)d.row));
    HIP_TRY(h, hipMemset(h->mean, 0, sizeof(float) * (size_t)d.row));  // mppi.py:157
    HIP_TRY(h, hipMalloc(&h->mean_used, sizeof(float) * (size_t)d.row));
    HIP_TRY(h, hipMemset(h->mean_used, 0, sizeof(float) * (size_t)d.row));
    HIP_TRY(h, hipMalloc(&h->solve_stats, sizeof(float) * 8));
    HIP_TRY(h, hipMemset(h->solve_stats, 0, sizeof(float) * 8));
    HIP_TRY(h, hipMalloc(&h->topk_hist, sizeof(unsigned) * (3 * TOPK_BINS + 2)));
    HIP_TRY(h, hipMemset(h->topk_hist, 0, sizeof(unsigned) * (3 * TOPK_BINS + 2)));
    HIP_TRY(h, hipMalloc(&h->topk_sel, sizeof(TopkSel) * 3));
    HIP_TRY(h, hipMalloc(&h->topk_cand, sizeof(unsigned long long) * TOPK_MAX));
    h->topk_cap = TOPK_MAX;
    const int max_blocks = 2048;
    HIP_TRY(h, hipMalloc(&h->partials, sizeof(float) * (size_t)max_blocks * h->colsp));
    HIP_TRY(h, hipMalloc(&h->heads, sizeof(float) * (size_t)max_blocks * 4));
    HIP_TRY(h, hipMalloc(&h->summary, sizeof(float) * (size_t)(MPPI_SUMMARY_HEAD + d.row)));
    HIP_TRY(h, hipMalloc(&h->stats_part, sizeof(float) * STATS_L * 3 * STATS_BLOCKS));
    HIP_TRY(h, hipMalloc(&h->round1_cells, sizeof(unsigned long long) * STATS_L * 3 * STATS_BLOCKS));
    HIP_TRY(h, hipMemset(h->round1_cells, 0, sizeof(unsigned long long) * STATS_L * 3 * STATS_BLOCKS));
    HIP_TRY(h, hipHostMalloc((void**)&h->stats_host, sizeof(double) * (8 + STATS_L * 3 + 3), hipHostMallocMapped));
    HIP_TRY(h, hipHostGetDevicePointer((void**)&h->stats_host_dev, h->stats_host, 0));  // (looked up once: an API call per solve otherwise)
    HIP_TRY(h, hipMalloc(&h->mpo_dev, sizeof(mppi::host::MpoState)));
    HIP_TRY(h, hipMalloc(&h->mpo_temp_dev, sizeof(float)));
    HIP_TRY(h, hipMalloc(&h->lbps_dev, sizeof(LbpsDev)));
    HIP_TRY(h, hipMalloc(&h->stats_max, sizeof(float) * STATS_BLOCKS));
    HIP_TRY(h, hipMalloc(&h->brent_cells, sizeof(unsigned long long) * 2 * BRENT_LANES * BRENT_CELLS));
    HIP_TRY(h, hipMemset(h->brent_cells, 0, sizeof(unsigned long long) * 2 * BRENT_LANES * BRENT_CELLS));
    HIP_TRY(h, hipHostMalloc((void**)&h->search_error, sizeof(int) * 16, hipHostMallocMapped));
    *h->search_error = 0;
    HIP_TRY(h, hipHostGetDevicePointer((void**)&h->search_error_dev, h->search_error, 0));
    HIP_TRY(h, hipDeviceGetAttribute(&h->lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device));
    HIP_TRY(h, hipMalloc(&h->lams_dev, sizeof(float) * 3 * STATS_L));
    HIP_TRY(h, hipMalloc(&h->essps_dev, sizeof(EsspsDev)));
    HIP_TRY(h, hipMalloc(&h->lambda_dev, sizeof(float)));
    HIP_TRY(h, hipHostMalloc((void**)&h->live_hint, sizeof(int), hipHostMallocMapped));
    *h->live_hint = 0;
    HIP_TRY(h, hipHostGetDevicePointer((void**)&h->live_hint_dev, h->live_hint, 0));
    std::memset(&h->ctx, 0, sizeof(h->ctx));
    if (h->wide) {
        HIP_TRY(h, hipMalloc(&h->coltab, sizeof(float) * 12 * (size_t)d.R));
        h->limits_set = false;
        if (md.dc <= MPPI_MAX_DIM_CONTROL)  // the config arrays hold all of it (dim_control = 3)
            if (int rc = mppi_set_control_limits(h, cfg->u_min, cfg->u_max, cfg->sigmas, md.dc)) return rc;
    }
    if (int rc = mpo_upload(h, 1.0, 0.1, 0.2, false)) return rc;  // mppi.py:191-200
    HIP_TRY(h, hipDeviceGetAttribute(&h->cu_count, hipDeviceAttributeMultiprocessorCount, cfg->device));
    HIP_TRY(h, hipDeviceSynchronize());
    return MPPI_OK;
}

// u_min / u_max / sigmas of all dim_control controls (host arrays).  Required once for generic handles with more
// than MPPI_MAX_DIM_CONTROL controls (the config arrays hold four); replaces the bounds of any handle.  Synchronises.
int mppi_set_control_limits(mppi_handle_t h, const float* u_min, const float* u_max, const float* sigmas, int n) {
    if (!h || !u_min || !u_max || !sigmas || n != h->dc) return fail(h, MPPI_E_INVALID, "control limits: need dim_control values each");
    for (int k = 0; k < n; ++k)
        if (!(u_min[k] <= u_max[k]) || !(sigmas[k] >= 0.0f)) return fail(h, MPPI_E_INVALID, "control limits: need u_min <= u_max, sigma >= 0");
    if (h->cfg.model != MPPI_MODEL_GENERIC && h->params_set)  // the fast-path preconditions were derived from the old bounds
        return fail(h, MPPI_E_STATE, "control limits of a native model must be set before its parameters");
    Dims& d = h->d;
    for (int k = 0; k < std::min(n, (int)MPPI_MAX_DIM_CONTROL); ++k) {
        d.u_min[k] = h->cfg.u_min[k] = u_min[k];
        d.u_max[k] = h->cfg.u_max[k] = u_max[k];
        d.sigma[k] = h->cfg.sigmas[k] = sigmas[k];
    }
    if (h->wide) {
        const size_t C = 4 * (size_t)d.R;
        std::vector<float> tab(3 * C, 0.0f);
        for (int f = 0; f < d.row; ++f) {
            tab[f] = sigmas[f % n]; tab[C + f] = u_min[f % n]; tab[2 * C + f] = u_max[f % n];
        }
        HIP_TRY(h, hipDeviceSynchronize());
        HIP_TRY(h, hipMemcpy(h->coltab, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice));
        h->limits_set = true;
        h->tiles_valid = h->tiles_valid && h->injected;
    }
    return MPPI_OK;
}

int mppi_destroy(mppi_handle_t h) {
    if (!h) return MPPI_E_INVALID;
    (void)hipFree(h->noise); (void)hipFree(h->costs); (void)hipFree(h->min_key); (void)hipFree(h->x0);
    (void)hipFree(h->x0_used); (void)hipFree(h->coltab); (void)hipFree(h->lams_dev); (void)hipFree(h->essps_dev);
    (void)hipFree(h->lambda_dev); (void)hipFree(h->mpo_dev); (void)hipFree(h->mpo_temp_dev); (void)hipFree(h->lbps_dev);
    (void)hipFree(h->stats_max); (void)hipFree(h->brent_cells);
    if (h->search_error) (void)hipHostFree(h->search_error);
    (void)hipFree(h->mean); (void)hipFree(h->mean_used); (void)hipFree(h->solve_stats); (void)hipFree(h->topk_hist);
    (void)hipFree(h->topk_sel); (void)hipFree(h->topk_cand); (void)hipFree(h->sg_coeffs); (void)hipFree(h->sg_history);
    (void)hipFree(h->ref); (void)hipFree(h->partials); (void)hipFree(h->heads);
    (void)hipFree(h->center8); (void)hipFree(h->win_dind); (void)hipFree(h->path_index);
    (void)hipFree(h->summary); (void)hipFree(h->map_cells[0]); (void)hipFree(h->map_cells[1]);
    (void)hipFree(h->map_pad); (void)hipFree(h->stats_part); (void)hipFree(h->round1_cells); (void)hipFree(h->noise_std);
    if (h->stats_host) (void)hipHostFree(h->stats_host);
    if (h->live_hint) (void)hipHostFree(h->live_hint);
    (void)hipFree(h->fused_cells); (void)hipFree(h->grid0_dev);
    if (h->fused_error) (void)hipHostFree(h->fused_error);
    if (h->comm) (void)rccl().comm_destroy(h->comm);
    (void)hipFree(h->comm_send); (void)hipFree(h->comm_recv);
    for (void* pm : h->p2p_opened) (void)hipIpcCloseMemHandle(pm);
    (void)hipFree(h->p2p_local); (void)hipFree(h->p2p_peers_dev);
    if (h->p2p_error) (void)hipHostFree(h->p2p_error);
    for (int i = 0; i < MppiSolver::RING; ++i) {
        if (h->stage[i]) (void)hipHostFree(h->stage[i]);
        if (h->stage_ev[i]) (void)hipEventDestroy(h->stage_ev[i]);
    }
    for (auto& pool : h->ev_pool) for (auto& e : pool) if (e) (void)hipEventDestroy(e);
    if (h->lazy_ev) (void)hipEventDestroy(h->lazy_ev);
    (void)hipFree(h->b1);
    delete h;
    return MPPI_OK;
}

// copy.deepcopy(solver) (the reference is a plain nn.Module, mppi.py:16: every tensor it holds is copied with it): make `dst`
// — a handle created from the same MppiConfig — continue exactly like `src` from here: warm start, noise identity, costs and
// minimum of the last solve (queries), Savitzky-Golay history, the temperature and every device-resident search / dual
// state, model parameters, maps, reference window and path index, options.  Set-up path: synchronises the device.
static int reserve_ref(mppi_handle_t h, int rows);
static int clone_buf(mppi_handle_t h, void* dst, const void* src, size_t bytes) {
    if (!bytes || !src || !dst) return MPPI_OK;
    HIP_TRY(h, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    return MPPI_OK;
}
int mppi_clone_state(mppi_handle_t dst, mppi_handle_t src) {
    if (!dst || !src || dst == src) return fail(dst, MPPI_E_INVALID, "clone_state: two distinct handles");
    const MppiConfig &a = dst->cfg, &b = src->cfg;
    if (a.model != b.model || a.horizon != b.horizon || a.dim_state != b.dim_state || a.dim_control != b.dim_control ||
        a.num_samples != b.num_samples || a.sample_offset != b.sample_offset || a.inherit_count != b.inherit_count ||
        a.device != b.device)
        return fail(dst, MPPI_E_INVALID, "clone_state: the handles were created from different configurations");
    if (int rc = settle_state_seq(src)) return rc;
    if (int rc = settle_state_seq(dst)) return rc;
    HIP_TRY(dst, hipDeviceSynchronize());
    const Dims& d = src->d;
    dst->cfg = src->cfg; dst->d = src->d; dst->wide = src->wide; dst->limits_set = src->limits_set;
    const size_t x0_floats = (size_t)std::max(src->ds, (int)MPPI_MAX_DIM_STATE);
#define CLONE(field, bytes) do { if (int rc = clone_buf(dst, dst->field, src->field, (bytes))) return rc; } while (0)
    if (src->tiles_valid) CLONE(noise, (size_t)d.tiles * d.R * 64 * sizeof(float4));
    CLONE(costs, sizeof(float) * (size_t)d.N);
    CLONE(min_key, 2 * sizeof(unsigned));
    if (int rc = clone_buf(dst, dst->x0, src->x0_cur, sizeof(float) * (size_t)src->ds)) return rc;  // (a borrowed state becomes an owned copy)
    dst->x0_cur = dst->x0;
    CLONE(x0_used, sizeof(float) * x0_floats);
    if (src->coltab && dst->coltab) CLONE(coltab, sizeof(float) * 12 * (size_t)d.R);
    CLONE(mean, sizeof(float) * (size_t)d.row);
    CLONE(mean_used, sizeof(float) * (size_t)d.row);
    CLONE(solve_stats, sizeof(float) * 8);
    CLONE(summary, sizeof(float) * (size_t)(MPPI_SUMMARY_HEAD + d.row));
    CLONE(mpo_dev, sizeof(mppi::host::MpoState));
    CLONE(mpo_temp_dev, sizeof(float));
    CLONE(lbps_dev, sizeof(LbpsDev));
    CLONE(lams_dev, sizeof(float) * 3 * STATS_L);
    CLONE(essps_dev, sizeof(EsspsDev));
    CLONE(lambda_dev, sizeof(float));
    if (src->grid0_dev) {
        if (!dst->grid0_dev) HIP_TRY(dst, hipMalloc(&dst->grid0_dev, sizeof(double) * STATS_L));
        CLONE(grid0_dev, sizeof(double) * STATS_L);
    }
#undef CLONE
    dst->min_slot = src->min_slot; dst->gen = src->gen; dst->tiles_valid = src->tiles_valid; dst->injected = src->injected;
    dst->noise_regen = src->noise_regen; dst->mapping = src->mapping; dst->math_fast = src->math_fast;
    dst->reduce_blocks = src->reduce_blocks; dst->reduce_chains = src->reduce_chains; dst->fold_mode = src->fold_mode;
    dst->fused_mode = src->fused_mode; dst->fused_timeout_ticks = src->fused_timeout_ticks; dst->lbps_grid = src->lbps_grid;
    dst->essps_merge0 = src->essps_merge0;
    dst->auto_rule = src->auto_rule; dst->auto_param = src->auto_param; dst->auto_lo = src->auto_lo; dst->auto_hi = src->auto_hi;
    dst->lbps_lo = src->lbps_lo; dst->lbps_hi = src->lbps_hi; dst->grid0_lo = src->grid0_lo; dst->grid0_hi = src->grid0_hi;
    dst->essps_lo = src->essps_lo; dst->essps_hi = src->essps_hi; dst->essps_range = src->essps_range;
    dst->essps_prev_host = src->essps_prev_host; dst->essps_prev_lo = src->essps_prev_lo; dst->essps_prev_hi = src->essps_prev_hi;
    dst->lambda_dev_valid = src->lambda_dev_valid;
    for (int i = 0; i < 3; ++i) dst->stats_host[8 + STATS_L * 3 + i] = src->stats_host[8 + STATS_L * 3 + i];  // the temperature's host mirror
    dst->last_reduce_blocks = 0;            // the partial rows of src's last reduction are not copied ...
    dst->summary_valid = src->summary_valid || src->last_reduce_blocks > 0;
    if (!src->summary_valid && src->last_reduce_blocks > 0) {  // ... so a finalize on dst alone would find nothing: copy them after all
        if (int rc = clone_buf(dst, dst->partials, src->partials, sizeof(float) * (size_t)2048 * src->colsp)) return rc;
        if (int rc = clone_buf(dst, dst->heads, src->heads, sizeof(float) * (size_t)2048 * 4)) return rc;
        dst->last_reduce_blocks = src->last_reduce_blocks; dst->summary_valid = false;
    }
    // Savitzky-Golay filter
    dst->sg_window = 0;
    if (src->sg_coeffs && src->sg_history) {
        const size_t hist_floats = (size_t)std::max(d.T - 1, 1) * src->dc;
        if (!dst->sg_coeffs) HIP_TRY(dst, hipMalloc(&dst->sg_coeffs, sizeof(float) * 256));
        if (!dst->sg_history) HIP_TRY(dst, hipMalloc(&dst->sg_history, sizeof(float) * hist_floats));
        if (int rc = clone_buf(dst, dst->sg_coeffs, src->sg_coeffs, sizeof(float) * 256)) return rc;
        if (int rc = clone_buf(dst, dst->sg_history, src->sg_history, sizeof(float) * hist_floats)) return rc;
        dst->sg_window = src->sg_window;
    }
    // lazily completed state sequences: the option, not a pending rollout (both were settled above)
    if (src->lazy_state && !dst->b1) HIP_TRY(dst, hipMalloc(&dst->b1, sizeof(float) * ((size_t)d.row + MPPI_MAX_DIM_STATE)));
    dst->lazy_state = src->lazy_state;
    // model context: parameters and flags by value, every pointer re-aimed at dst's own copy
    const ModelCtx old = dst->ctx;
    dst->ctx = src->ctx;
    dst->params_set = src->params_set;
    for (int slot = 0; slot < 2; ++slot) {
        dst->ctx.maps[slot].cells = old.maps[slot].cells;
        if (!src->map_cells[slot]) { dst->ctx.maps[slot] = old.maps[slot]; continue; }
        const MapView& m = src->ctx.maps[slot];
        if (int rc = prepare_map(dst, slot, m.nx, m.ny, m.cell, m.ox, m.oy)) return rc;
        if (int rc = clone_buf(dst, dst->map_cells[slot], src->map_cells[slot], (size_t)m.nx * m.ny)) return rc;
    }
    dst->ctx.ref = nullptr; dst->ctx.ref_rows = 0;
    if (src->ref && src->ref_cap > 0) {
        if (int rc = reserve_ref(dst, src->ref_cap)) return rc;
        if (int rc = clone_buf(dst, dst->ref, src->ref, sizeof(float) * 8 * (size_t)src->ref_cap)) return rc;
        if (src->ctx.ref) { dst->ctx.ref = dst->ref; dst->ctx.ref_rows = src->ctx.ref_rows; }
    }
    if (src->center_n) {
        (void)hipFree(dst->center8); (void)hipFree(dst->win_dind);
        dst->center8 = nullptr; dst->win_dind = nullptr;
        HIP_TRY(dst, hipMalloc(&dst->center8, sizeof(float) * 8 * (size_t)src->center_n));
        HIP_TRY(dst, hipMalloc(&dst->win_dind, sizeof(int32_t) * (size_t)src->win_rows));
        if (!dst->path_index) HIP_TRY(dst, hipMalloc(&dst->path_index, sizeof(int32_t)));
        if (int rc = clone_buf(dst, dst->center8, src->center8, sizeof(float) * 8 * (size_t)src->center_n)) return rc;
        if (int rc = clone_buf(dst, dst->win_dind, src->win_dind, sizeof(int32_t) * (size_t)src->win_rows)) return rc;
        if (int rc = clone_buf(dst, dst->path_index, src->path_index, sizeof(int32_t))) return rc;
        dst->center_n = src->center_n; dst->win_rows = src->win_rows; dst->win_v = src->win_v;
    }
    refresh_pad(dst, nullptr);  // the padded grid of the fast lookups, rebuilt from dst's own maps
    HIP_TRY(dst, hipDeviceSynchronize());
    return MPPI_OK;
}

int mppi_set_model_params(mppi_handle_t h, const float* p, int n) {
    if (!h || n < 0 || n > MPPI_MAX_PARAMS || (n > 0 && !p)) return fail(h, MPPI_E_INVALID, "bad params");
    const int need = h->cfg.model == MPPI_MODEL_RACING ? MPPI_RP_COUNT : h->cfg.model == MPPI_MODEL_NAV2D ? MPPI_NP_COUNT
                     : h->cfg.model == MPPI_MODEL_GOALZONE ? MPPI_GP_COUNT : 0;
    if (n != need) return fail(h, MPPI_E_INVALID, "parameter count does not match the model");
    if (int rc = settle_state_seq(h)) return rc;  // a lazily completed state sequence belongs to the OLD constants: roll it out first
    for (int i = 0; i < n; ++i) h->ctx.P[i] = p[i];
    const float* um = h->cfg.u_min; const float* uM = h->cfg.u_max;
    if (h->cfg.model == MPPI_MODEL_GOALZONE) {
        h->ctx.u_in_bounds = (um[0] >= p[MPPI_GP_VMIN] && uM[0] <= p[MPPI_GP_VMAX] && um[1] >= p[MPPI_GP_WMIN] &&
                              uM[1] <= p[MPPI_GP_WMAX]) ? 1 : 0;
        const float w = std::fmax(std::fabs(p[MPPI_GP_WMIN]), std::fabs(p[MPPI_GP_WMAX]));
        h->ctx.wrap_safe = (w * std::fabs(p[MPPI_GP_DT]) < 3.0f) ? 1 : 0;
    }
    if (h->cfg.model == MPPI_MODEL_NAV2D) {
        h->ctx.u_in_bounds = (um[0] >= p[MPPI_NP_VMIN] && uM[0] <= p[MPPI_NP_VMAX] && um[1] >= p[MPPI_NP_WMIN] &&
                              uM[1] <= p[MPPI_NP_WMAX]) ? 1 : 0;
        const float w = std::fmax(std::fabs(p[MPPI_NP_WMIN]), std::fabs(p[MPPI_NP_WMAX]));
        h->ctx.wrap_safe = (w * std::fabs(p[MPPI_NP_DT]) < 3.0f) ? 1 : 0;
    }
    if (h->cfg.model == MPPI_MODEL_RACING) {
        h->ctx.u_in_bounds = (um[0] >= p[MPPI_RP_AMIN] && uM[0] <= p[MPPI_RP_AMAX] && um[1] >= p[MPPI_RP_SMIN] &&
                              uM[1] <= p[MPPI_RP_SMAX]) ? 1 : 0;
        const float sm = std::fmax(std::fabs(p[MPPI_RP_SMIN]), std::fabs(p[MPPI_RP_SMAX]));
        const float dth = std::fabs(p[MPPI_RP_VMAX]) * std::tan(std::fmin(sm, 1.5f)) / std::fabs(p[MPPI_RP_L]) *
                          std::fabs(p[MPPI_RP_DT]);
        h->ctx.wrap_safe = (sm < 1.5f && dth < 3.0f) ? 1 : 0;
        h->ctx.tan_small = (std::fabs(p[MPPI_RP_SMIN]) <= 0.25f && std::fabs(p[MPPI_RP_SMAX]) <= 0.25f) ? 1 : 0;
        const float L = p[MPPI_RP_L];
        uint32_t bits; std::memcpy(&bits, &L, 4);
        h->ctx.inv_L = (L > 0.0f && (bits & 0x7fffffu) != 0x7fffffu) ? 1.0f / L : 0.0f;
        h->ctx.unit_L = L == 1.0f ? 1 : 0;
    }
    h->params_set = true;
    refresh_pad(h, nullptr);  // the padded grid depends on the position clamp limits
    if (hipStreamSynchronize(nullptr) != hipSuccess) return fail(h, MPPI_E_HIP, "padded grid construction failed");
    return MPPI_OK;
}

int mppi_upload_map(mppi_handle_t h, int slot, const uint8_t* cells, int nx, int ny, float cell, float ox, float oy) {
    if (!h || !cells) return fail(h, MPPI_E_INVALID, "bad map");
    const size_t n = (size_t)(nx > 0 ? nx : 0) * (ny > 0 ? ny : 0);
    for (size_t i = 0; i < n; ++i)
        if (cells[i] > 1) return fail(h, MPPI_E_INVALID, "map cells must be 0/1 occupancy");
    if (int rc = settle_state_seq(h)) return rc;  // (a pending state sequence keeps the kernel variant of ITS solve)
    if (int rc = prepare_map(h, slot, nx, ny, cell, ox, oy)) return rc;
    HIP_TRY(h, hipMemcpy(h->map_cells[slot], cells, n, hipMemcpyHostToDevice));
    refresh_pad(h, nullptr);
    HIP_TRY(h, hipStreamSynchronize(nullptr));
    return MPPI_OK;
}

int mppi_build_obstacle_map(mppi_handle_t h, int slot, int nx, int ny, float cell, float ox, float oy,
                            const int32_t* circles, int n_circles, const int32_t* rects, int n_rects, void* stream) {
    if (!h || n_circles < 0 || n_rects < 0 || (n_circles && !circles) || (n_rects && !rects))
        return fail(h, MPPI_E_INVALID, "bad obstacle list");
    for (int c = 0; c < n_circles; ++c)
        if (circles[3 * c + 2] < 0) return fail(h, MPPI_E_INVALID, "circle radius must be >= 0 cells");
    if (int rc = settle_state_seq(h)) return rc;  // (a pending state sequence keeps the kernel variant of ITS solve)
    if (int rc = prepare_map(h, slot, nx, ny, cell, ox, oy)) return rc;
    hipStream_t s = (hipStream_t)stream;
    int32_t *dc = nullptr, *dr = nullptr;
    if (int rc = upload_ints(h, circles, (size_t)3 * n_circles, &dc)) return rc;
    if (int rc = upload_ints(h, rects, (size_t)4 * n_rects, &dr)) { (void)hipFree(dc); return rc; }
    hipLaunchKernelGGL(mppi::raster_obstacles_kernel, dim3((ny + mppi::BLOCK - 1) / mppi::BLOCK, nx), dim3(mppi::BLOCK), 0,
                       s, h->map_cells[slot], nx, ny, dc, n_circles, dr, n_rects);
    const hipError_t e = hipGetLastError();
    refresh_pad(h, s);
    const hipError_t e2 = hipStreamSynchronize(s);  // the recipe tables are freed below
    (void)hipFree(dc); (void)hipFree(dr);
    HIP_TRY(h, e);
    HIP_TRY(h, e2);
    return MPPI_OK;
}

int mppi_build_lane_map(mppi_handle_t h, int slot, int nx, int ny, float cell, float ox, float oy,
                        const int32_t* seeds, int n_seeds, int64_t max_d2, void* stream) {
    if (!h || n_seeds < 1 || !seeds || max_d2 < 0) return fail(h, MPPI_E_INVALID, "bad lane seeds");
    if (int rc = settle_state_seq(h)) return rc;  // (a pending state sequence keeps the kernel variant of ITS solve)
    if (int rc = prepare_map(h, slot, nx, ny, cell, ox, oy)) return rc;
    hipStream_t s = (hipStream_t)stream;
    int32_t* ds = nullptr;
    if (int rc = upload_ints(h, seeds, (size_t)2 * n_seeds, &ds)) return rc;
    hipLaunchKernelGGL(mppi::lane_map_kernel, dim3((ny + mppi::BLOCK - 1) / mppi::BLOCK, nx), dim3(mppi::BLOCK), 0, s,
                       h->map_cells[slot], nx, ny, ds, n_seeds, max_d2);
    const hipError_t e = hipGetLastError();
    refresh_pad(h, s);
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(ds);
    HIP_TRY(h, e);
    HIP_TRY(h, e2);
    return MPPI_OK;
}

int mppi_download_map(mppi_handle_t h, int slot, uint8_t* cells_host, int* nx, int* ny) {
    if (!h || slot < 0 || slot > 1) return fail(h, MPPI_E_INVALID, "bad slot");
    if (!h->map_cells[slot]) return fail(h, MPPI_E_STATE, "map slot is empty");
    const MapView& m = h->ctx.maps[slot];
    if (nx) *nx = m.nx;
    if (ny) *ny = m.ny;
    if (cells_host) {
        HIP_TRY(h, hipDeviceSynchronize());
        HIP_TRY(h, hipMemcpy(cells_host, h->map_cells[slot], (size_t)m.nx * m.ny, hipMemcpyDeviceToHost));
    }
    return MPPI_OK;
}

// Pinned staging slot of at least `floats` floats; waits (rarely) for the slot's previous upload.
static int stage_slot(mppi_handle_t h, size_t floats, float** out, hipEvent_t* ev) {
    if (floats > h->stage_floats) {
        for (int i = 0; i < MppiSolver::RING; ++i) {
            if (h->stage_ev[i]) HIP_TRY(h, hipEventSynchronize(h->stage_ev[i]));
            if (h->stage[i]) { (void)hipHostFree(h->stage[i]); h->stage[i] = nullptr; }
            HIP_TRY(h, hipHostMalloc((void**)&h->stage[i], sizeof(float) * floats, hipHostMallocDefault));
            if (!h->stage_ev[i]) HIP_TRY(h, hipEventCreateWithFlags(&h->stage_ev[i], hipEventDisableTiming));
        }
        h->stage_floats = floats;
    }
    const int i = h->stage_next;
    h->stage_next = (i + 1) % MppiSolver::RING;
    HIP_TRY(h, hipEventSynchronize(h->stage_ev[i]));  // no-op unless 8 uploads are still in flight
    *out = h->stage[i];
    *ev = h->stage_ev[i];
    return MPPI_OK;
}

// host -> device upload of a few floats through the pinned ring: asynchronous, no host wait
static int upload_small(mppi_handle_t h, float* dst_dev, const float* src_host, size_t floats, hipStream_t s) {
    float* st = nullptr; hipEvent_t ev = nullptr;
    if (int rc = stage_slot(h, std::max<size_t>(floats, 64), &st, &ev)) return rc;
    std::memcpy(st, src_host, sizeof(float) * floats);
    HIP_TRY(h, hipMemcpyAsync(dst_dev, st, sizeof(float) * floats, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipEventRecord(ev, s));
    return MPPI_OK;
}

static int reserve_ref(mppi_handle_t h, int rows);

int mppi_set_reference(mppi_handle_t h, const float* ref, int rows, void* stream) {
    if (!h || !ref || rows < 1) return fail(h, MPPI_E_INVALID, "bad reference");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = reserve_ref(h, rows)) return rc;
    float* st = nullptr; hipEvent_t ev = nullptr;
    if (int rc = stage_slot(h, (size_t)rows * 8, &st, &ev)) return rc;
    for (int i = 0; i < rows; ++i) {
        float* o = st + (size_t)i * 8;
        o[0] = ref[4 * i]; o[1] = ref[4 * i + 1]; o[2] = ref[4 * i + 2]; o[3] = ref[4 * i + 3];
        o[4] = sinf(o[2]); o[5] = cosf(o[2]);  // torch.sin/cos of the fp32 scalar, racing.py:127-139
        o[6] = o[7] = 0.0f;
    }
    HIP_TRY(h, hipMemcpyAsync(h->ref, st, sizeof(float) * 8 * (size_t)rows, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipEventRecord(ev, s));
    h->ctx.ref = h->ref;
    h->ctx.ref_rows = rows;
    return MPPI_OK;
}

// make sure h->ref holds `rows` rows (blocking reallocation: set-up path)
static int reserve_ref(mppi_handle_t h, int rows) {
    if (rows <= h->ref_cap) return MPPI_OK;
    if (h->ref) { HIP_TRY(h, hipDeviceSynchronize()); (void)hipFree(h->ref); }
    h->ref = nullptr; h->ref_cap = 0;
    HIP_TRY(h, hipMalloc(&h->ref, sizeof(float) * 8 * (size_t)rows));
    h->ref_cap = rows;
    return MPPI_OK;
}

int mppi_set_center_path(mppi_handle_t h, const float* path_host, int n, const int32_t* dind_host, int rows,
                         float v_target) {
    if (!h || !path_host || !dind_host || n < 1 || rows < 1) return fail(h, MPPI_E_INVALID, "bad centre path");
    if (h->cfg.model != MPPI_MODEL_RACING) return fail(h, MPPI_E_INVALID, "the reference window belongs to the racing model");
    if (rows < h->d.T) return fail(h, MPPI_E_INVALID, "window shorter than the horizon");
    for (int i = 0; i < rows; ++i)
        if (dind_host[i] < 0 || (i && dind_host[i] < dind_host[i - 1])) return fail(h, MPPI_E_INVALID, "window offsets must be >= 0 and non-decreasing");
    std::vector<float> tab((size_t)n * 8, 0.0f);
    for (int i = 0; i < n; ++i) {
        float* o = tab.data() + (size_t)i * 8;
        o[0] = path_host[3 * i]; o[1] = path_host[3 * i + 1]; o[2] = path_host[3 * i + 2];
        o[4] = sinf(o[2]); o[5] = cosf(o[2]);  // the calls mppi_set_reference makes per window row
    }
    HIP_TRY(h, hipDeviceSynchronize());
    (void)hipFree(h->center8); (void)hipFree(h->win_dind);
    h->center8 = nullptr; h->win_dind = nullptr; h->center_n = 0;
    HIP_TRY(h, hipMalloc(&h->center8, sizeof(float) * tab.size()));
    HIP_TRY(h, hipMemcpy(h->center8, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc(&h->win_dind, sizeof(int32_t) * (size_t)rows));
    HIP_TRY(h, hipMemcpy(h->win_dind, dind_host, sizeof(int32_t) * (size_t)rows, hipMemcpyHostToDevice));
    if (!h->path_index) {
        HIP_TRY(h, hipMalloc(&h->path_index, sizeof(int32_t)));
        HIP_TRY(h, hipMemset(h->path_index, 0, sizeof(int32_t)));
    }
    if (int rc = reserve_ref(h, rows)) return rc;
    h->center_n = n; h->win_rows = rows; h->win_v = v_target;
    return MPPI_OK;
}

int mppi_ref_window(mppi_handle_t h, const float* state_dev, void* stream) {
    if (!h) return MPPI_E_INVALID;
    if (!h->center_n) return fail(h, MPPI_E_STATE, "mppi_ref_window before mppi_set_center_path");
    const RefWindowCtx w{h->center8, h->win_dind, h->path_index, h->center_n, h->win_rows, h->win_v};
    hipLaunchKernelGGL(ref_window_kernel, dim3(1), dim3(REFWIN_BLOCK), 0, (hipStream_t)stream, w,
                       state_dev ? state_dev : h->x0_cur, h->ref);
    HIP_TRY(h, hipGetLastError());
    h->ctx.ref = h->ref;
    h->ctx.ref_rows = h->win_rows;
    return MPPI_OK;
}

int mppi_set_path_index(mppi_handle_t h, int32_t cind, void* stream) {
    if (!h || cind < 0) return fail(h, MPPI_E_INVALID, "bad path index");
    if (!h->path_index) return fail(h, MPPI_E_STATE, "mppi_set_path_index before mppi_set_center_path");
    HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(h, hipMemcpy(h->path_index, &cind, sizeof(int32_t), hipMemcpyHostToDevice));
    return MPPI_OK;
}

int mppi_get_path_index(mppi_handle_t h, int32_t* cind_out_host, void* stream) {
    if (!h || !cind_out_host) return fail(h, MPPI_E_INVALID, "null");
    if (!h->path_index) return fail(h, MPPI_E_STATE, "mppi_get_path_index before mppi_set_center_path");
    HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(h, hipMemcpy(cind_out_host, h->path_index, sizeof(int32_t), hipMemcpyDeviceToHost));
    return MPPI_OK;
}

int mppi_get_reference(mppi_handle_t h, float* ref_out, int rows, int on_device, void* stream) {
    if (!h || !ref_out || rows < 1) return fail(h, MPPI_E_INVALID, "bad get_reference arguments");
    if (!h->ctx.ref || rows > h->ctx.ref_rows) return fail(h, MPPI_E_STATE, "no reference window of that many rows");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(h, hipMemcpy2DAsync(ref_out, 4 * sizeof(float), h->ref, 8 * sizeof(float), 4 * sizeof(float), (size_t)rows,
                                on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    if (!on_device) HIP_TRY(h, hipStreamSynchronize(s));
    return MPPI_OK;
}

int mppi_model_step(int model, const float* params_host, int n_params, const float* u_min_host, const float* u_max_host,
                    const float* state_dev, const float* action_dev, float* next_state_dev, const float* goal_xy_host,
                    float goal_threshold, uint8_t* reached_out_dev, void* stream) {
    ModelDims md{};
    if (!model_dims(model, md) || n_params < 0 || n_params > MPPI_MAX_PARAMS || (n_params && !params_host) ||
        !state_dev || !action_dev || !next_state_dev || (reached_out_dev && !goal_xy_host) || md.dc > MPPI_MAX_DIM_CONTROL)
        return MPPI_E_INVALID;
    const int need = model == MPPI_MODEL_RACING ? MPPI_RP_COUNT : model == MPPI_MODEL_NAV2D ? MPPI_NP_COUNT
                     : model == MPPI_MODEL_GOALZONE ? MPPI_GP_COUNT : 0;
    if (n_params < need) return MPPI_E_INVALID;  // (the cost weights at the tail are not read by the dynamics)
    ModelCtx ctx;
    std::memset(&ctx, 0, sizeof(ctx));
    for (int i = 0; i < n_params; ++i) ctx.P[i] = params_host[i];
    StepBounds ub;
    for (int k = 0; k < MPPI_MAX_DIM_CONTROL; ++k) {
        ub.lo[k] = (u_min_host && k < md.dc) ? u_min_host[k] : -INFINITY;
        ub.hi[k] = (u_max_host && k < md.dc) ? u_max_host[k] : INFINITY;
    }
    const float gx = goal_xy_host ? goal_xy_host[0] : 0.0f, gy = goal_xy_host ? goal_xy_host[1] : 0.0f;
    hipStream_t s = (hipStream_t)stream;
#define CALL_STEP(MODEL)                                                                              \
    hipLaunchKernelGGL((model_step_kernel<MODEL>), dim3(1), dim3(WAVE), 0, s, ctx, state_dev, action_dev, ub,  \
                       next_state_dev, gx, gy, goal_threshold, reached_out_dev)
    switch (model) {
    case MPPI_MODEL_PENDULUM: CALL_STEP(MPPI_MODEL_PENDULUM); break;
    case MPPI_MODEL_CARTPOLE: CALL_STEP(MPPI_MODEL_CARTPOLE); break;
    case MPPI_MODEL_MOUNTAINCAR: CALL_STEP(MPPI_MODEL_MOUNTAINCAR); break;
    case MPPI_MODEL_NAV2D: CALL_STEP(MPPI_MODEL_NAV2D); break;
    case MPPI_MODEL_RACING: CALL_STEP(MPPI_MODEL_RACING); break;
    case MPPI_MODEL_MJCARTPOLE: CALL_STEP(MPPI_MODEL_MJCARTPOLE); break;
    case MPPI_MODEL_GOALZONE: CALL_STEP(MPPI_MODEL_GOALZONE); break;
    }
#undef CALL_STEP
    return hipGetLastError() == hipSuccess ? MPPI_OK : MPPI_E_HIP;
}

int mppi_grid_lookup(const float* map_dev, int nx, int ny, float cell_size, float origin_x, float origin_y, const float* xy_dev,
                     int64_t n, int64_t stride, float* out_dev, void* stream) {
    if (!map_dev || !xy_dev || !out_dev || nx < 1 || ny < 1 || !(cell_size > 0.0f) || n < 0 || stride < 2) return MPPI_E_INVALID;
    if (n == 0) return MPPI_OK;
    hipLaunchKernelGGL(grid_lookup_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, map_dev, nx,
                       ny, cell_size, origin_x, origin_y, xy_dev, n, stride, out_dev);
    return hipGetLastError() == hipSuccess ? MPPI_OK : MPPI_E_HIP;
}

// device <-> device / device -> host copies of small vectors
static int copy_small(mppi_handle_t h, void* dst, const void* src, size_t bytes, bool dst_dev, bool src_dev, hipStream_t s) {
    if (dst_dev && !src_dev) return upload_small(h, (float*)dst, (const float*)src, bytes / sizeof(float), s);
    const hipMemcpyKind kind = dst_dev ? hipMemcpyDeviceToDevice : (src_dev ? hipMemcpyDeviceToHost : hipMemcpyHostToHost);
    HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, kind, s));
    if (!dst_dev) HIP_TRY(h, hipStreamSynchronize(s));
    return MPPI_OK;
}

int mppi_set_mean(mppi_handle_t h, const float* mean, int on_device, void* stream) {
    if (!h || !mean) return fail(h, MPPI_E_INVALID, "null");
    return copy_small(h, h->mean, mean, sizeof(float) * (size_t)h->d.row, true, on_device != 0, (hipStream_t)stream);
}
int mppi_get_mean(mppi_handle_t h, float* out, int on_device, void* stream) {
    if (!h || !out) return fail(h, MPPI_E_INVALID, "null");
    return copy_small(h, out, h->mean, sizeof(float) * (size_t)h->d.row, on_device != 0, true, (hipStream_t)stream);
}
int mppi_set_state(mppi_handle_t h, const float* x0, int on_device, void* stream) {
    if (!h || !x0) return fail(h, MPPI_E_INVALID, "null");
    h->x0_cur = h->x0;
    return copy_small(h, h->x0, x0, sizeof(float) * (size_t)h->ds, true, on_device != 0, (hipStream_t)stream);
}
int mppi_bind_state(mppi_handle_t h, const float* x0_dev) {
    if (!h || !x0_dev) return fail(h, MPPI_E_INVALID, "null");
    h->x0_cur = x0_dev;
    return MPPI_OK;
}

static int materialize_tiles(mppi_handle_t h, hipStream_t s) {
    const unsigned grid = (unsigned)((h->d.tiles + 3) / 4);
    if (!h->limits_set) return fail(h, MPPI_E_STATE, "dim_control > 4: call mppi_set_control_limits first");
    if (h->wide) hipLaunchKernelGGL(sample_kernel<true>, dim3(grid), dim3(BLOCK), 0, s, h->noise, h->d, h->gen, h->coltab);
    else hipLaunchKernelGGL(sample_kernel<false>, dim3(grid), dim3(BLOCK), 0, s, h->noise, h->d, h->gen, (const float*)nullptr);
    HIP_TRY(h, hipGetLastError());
    h->tiles_valid = true;
    return MPPI_OK;
}

int mppi_sample(mppi_handle_t h, uint32_t solve_idx, void* stream) {
    if (!h) return MPPI_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    h->gen.solve_idx = solve_idx;
    h->injected = false;
    h->tiles_valid = false;
    if (h->noise_regen && !h->wide) return MPPI_OK;  // consumers regenerate eps(seed, solve, i, t, k) in registers
    StageTimer tm(h, 0, s);
    return materialize_tiles(h, s);
}

// the tiles must hold the current noise for the layout/gather entry points
static int need_tiles(mppi_handle_t h, hipStream_t s) {
    if (h->tiles_valid) return MPPI_OK;
    return materialize_tiles(h, s);
}

int mppi_inject_noise(mppi_handle_t h, const float* eps_dev, void* stream) {
    if (!h || !eps_dev) return fail(h, MPPI_E_INVALID, "null");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)h->d.tiles, (unsigned)((h->d.row + CONV_COLS - 1) / CONV_COLS));
    hipLaunchKernelGGL(inject_kernel, grid, dim3(BLOCK), 0, s, eps_dev, h->noise, h->d);
    HIP_TRY(h, hipGetLastError());
    h->injected = true;
    h->tiles_valid = true;
    return MPPI_OK;
}

int mppi_export_noise(mppi_handle_t h, float* eps_out, float* act_out, void* stream) {
    if (!h) return MPPI_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = need_tiles(h, s)) return rc;
    const dim3 grid((unsigned)h->d.tiles, (unsigned)((h->d.row + CONV_COLS - 1) / CONV_COLS));
    hipLaunchKernelGGL(export_kernel, grid, dim3(BLOCK), 0, s, h->noise, h->mean, eps_out, act_out, h->d,
                       h->wide ? h->coltab : (const float*)nullptr);
    HIP_TRY(h, hipGetLastError());
    return MPPI_OK;
}

int mppi_rollout_cost(mppi_handle_t h, void* stream) {
    if (!h) return MPPI_E_INVALID;
    if (int rc = check_ready(h)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (h->mapping == 1) {  // comparison variant: one wavefront per trajectory, reference-layout noise
        if (int rc = flush_state_seq(h, s)) return rc;
        if (!h->noise_std) HIP_TRY(h, hipMalloc(&h->noise_std, sizeof(float) * (size_t)h->d.N * h->d.row));
        if (int rc = need_tiles(h, s)) return rc;
        const dim3 cgrid((unsigned)h->d.tiles, (unsigned)((h->d.row + CONV_COLS - 1) / CONV_COLS));
        hipLaunchKernelGGL(export_kernel, cgrid, dim3(BLOCK), 0, s, h->noise, h->mean, h->noise_std, (float*)nullptr, h->d,
                           (const float*)nullptr);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(h->mean_used, h->mean, sizeof(float) * (size_t)h->d.row, hipMemcpyDeviceToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(h->x0_used, h->x0_cur, sizeof(float) * (size_t)h->ds, hipMemcpyDeviceToDevice, s));
        StageTimer tmw(h, 1, s);
        h->min_slot ^= 1;
        unsigned* mkw = h->min_key + h->min_slot;
        unsigned* mkw_next = h->min_key + (h->min_slot ^ 1);
        const unsigned wgrid = (unsigned)std::min<int64_t>((h->d.N + 3) / 4, 256 * 8 * 4);
#define CALL_WAVE(MODEL, FASTV)                                                                       \
        do {                                                                                          \
            const size_t shw = sizeof(float) * 4 * ((size_t)4 * h->d.R + (size_t)(h->d.T + 1) * ModelT<MODEL, FASTV>::DS); \
            hipLaunchKernelGGL((rollout_cost_wave_kernel<MODEL, FASTV>), dim3(wgrid), dim3(BLOCK), shw, s, h->noise_std, \
                               h->mean, h->x0_cur, h->costs, mkw, mkw_next, h->d, h->ctx);           \
        } while (0)
        MPPI_DISPATCH(h, CALL_WAVE);
#undef CALL_WAVE
        HIP_TRY(h, hipGetLastError());
        return MPPI_OK;
    }
    StageTimer tm(h, 1, s);
    const bool gen = h->noise_regen && !h->injected;
    if (!gen && !h->tiles_valid) return fail(h, MPPI_E_STATE, "no noise: call mppi_sample or mppi_inject_noise first");
    h->min_slot ^= 1;
    unsigned* mk = h->min_key + h->min_slot;
    unsigned* mk_next = h->min_key + (h->min_slot ^ 1);
    // a state sequence still pending from the previous solve (option "lazy_state_seq") rides in one extra block of this
    // launch: its T dependent steps hide behind the N-sample rollout instead of extending the previous solve's tail
    float* ride = h->pending_state_out;
    if (ride) { if (int rc = order_behind_pending(h, s)) return rc; }
    const unsigned grid = (unsigned)((h->d.tiles + 3) / 4) + (ride ? 1u : 0u);
#define CALL_ROLLOUT(MODEL, FASTV)                                                                    \
    do {                                                                                              \
        const size_t shmem = sizeof(float) * std::max((size_t)8 * h->d.R + (size_t)h->d.T * ModelT<MODEL, FASTV>::KROW, \
                                                      (size_t)h->d.row + MPPI_MAX_DIM_STATE);         \
        constexpr bool UCV = FASTV != 0;  /* the FAST kernels exist in the u_in_bounds form only (see use_fast) */ \
        if (gen)                                                                                      \
            hipLaunchKernelGGL((rollout_cost_kernel<MODEL, FASTV, true, UCV>), dim3(grid), dim3(BLOCK), shmem, s, \
                               h->noise, h->mean, h->x0_cur, h->costs, mk, mk_next, h->mean_used, h->x0_used, h->d, h->gen, h->ctx, \
                               (const float*)h->b1, ride);                                           \
        else                                                                                          \
            hipLaunchKernelGGL((rollout_cost_kernel<MODEL, FASTV, false, UCV>), dim3(grid), dim3(BLOCK), shmem, s, \
                               h->noise, h->mean, h->x0_cur, h->costs, mk, mk_next, h->mean_used, h->x0_used, h->d, h->gen, h->ctx, \
                               (const float*)h->b1, ride);                                           \
    } while (0)
    MPPI_DISPATCH(h, CALL_ROLLOUT);
#undef CALL_ROLLOUT
    HIP_TRY(h, hipGetLastError());
    if (ride) h->pending_state_out = nullptr;  // (cleared only once the launch that carries it went through)
    return MPPI_OK;
}

int mppi_get_costs(mppi_handle_t h, float* dst, int on_device, void* stream) {
    if (!h || !dst) return fail(h, MPPI_E_INVALID, "null");
    return copy_small(h, dst, h->costs, sizeof(float) * (size_t)h->d.N, on_device != 0, true, (hipStream_t)stream);
}

__global__ void min_cost_kernel(const float* __restrict__ costs, int64_t N, unsigned* __restrict__ min_key) {
    float m = INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        m = fminf(m, costs[i]);
    m = wave_min(m);
    if ((threadIdx.x & 63) == 0 && m < INFINITY) atomicMin(min_key, float_to_key(m));
}

int mppi_set_costs(mppi_handle_t h, const float* src, int on_device, void* stream) {
    if (!h || !src) return fail(h, MPPI_E_INVALID, "null");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = copy_small(h, h->costs, src, sizeof(float) * (size_t)h->d.N, true, on_device != 0, s)) return rc;
    HIP_TRY(h, hipMemsetAsync(h->min_key + h->min_slot, 0xFF, sizeof(unsigned), s));
    const unsigned grid = (unsigned)std::min<int64_t>((h->d.N + BLOCK - 1) / BLOCK, 1024);
    hipLaunchKernelGGL(min_cost_kernel, dim3(grid), dim3(BLOCK), 0, s, h->costs, h->d.N, h->min_key + h->min_slot);
    HIP_TRY(h, hipGetLastError());
    return MPPI_OK;
}

// lambda argument of the reduce / finalize entry points -> (launch constant, device pointer or null)
static int resolve_lambda(mppi_handle_t h, float lambda, const float** lam_dev) {
    *lam_dev = nullptr;
    if (lambda == MPPI_LAMBDA_DEVICE) {
        if (!h->lambda_dev_valid) return fail(h, MPPI_E_STATE, "MPPI_LAMBDA_DEVICE: no temperature on the device (run a device-resident rule or mppi_mpo_reset first)");
        *lam_dev = h->lambda_dev;
        return MPPI_OK;
    }
    if (!(lambda > 0.0f)) return fail(h, MPPI_E_INVALID, "lambda must be > 0");
    return MPPI_OK;
}

int mppi_weights_reduce(mppi_handle_t h, float lambda, float* summary_out_dev, void* stream) {
    if (!h) return MPPI_E_INVALID;
    const float* lam_dev = nullptr;
    if (int rc = resolve_lambda(h, lambda, &lam_dev)) return rc;
    hipStream_t s = (hipStream_t)stream;
    StageTimer tm(h, 2, s);
    // one wave per tile up to reduce_blocks blocks (dense weights need the parallelism; with sparse
    // weights most waves only run the phase-A check)
    int64_t blocks = std::min<int64_t>(h->reduce_blocks, (h->d.tiles + 3) / 4);
    blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, REDUCE_MAX_BLOCKS));
    h->last_reduce_blocks = (int)blocks;
    const dim3 grid((unsigned)blocks, (unsigned)h->nchunks);
    const bool gen = h->noise_regen && !h->injected && !h->wide;
    if (!gen && !h->tiles_valid) return fail(h, MPPI_E_STATE, "no noise: call mppi_sample or mppi_inject_noise first");
    const unsigned* mk = h->min_key + h->min_slot;
#define CALL_REDUCE(GPWV, GENV, WIDEV, CHAINSV, REMV)                                                 \
    hipLaunchKernelGGL((weights_reduce_kernel<GPWV, GENV, WIDEV, CHAINSV, REMV>), grid, dim3(BLOCK), 0, s, h->noise, h->mean, h->costs, mk, \
                       h->partials, h->heads, h->d, h->gen, lambda, lam_dev, (const float*)h->coltab)
    // regenerated noise: four chains per basic block while a SIMD holds one or two reduction waves, two beyond (see the kernel)
    const bool chains4 = h->reduce_chains == 4 || (h->reduce_chains == 0 && blocks * (int64_t)h->nchunks <= 2 * (int64_t)h->cu_count);
    const bool rem = (h->d.R % 4) != 0;  // some chunk of the row leaves groups over (chunks hold 32 groups: R % 32 % 4)
    if (h->wide) CALL_REDUCE(8, false, true, 2, true);
    else if (gen && chains4 && rem) CALL_REDUCE(8, true, false, 4, true);
    else if (gen && chains4) CALL_REDUCE(8, true, false, 4, false);
    else if (gen && rem) CALL_REDUCE(8, true, false, 2, true);
    else if (gen) CALL_REDUCE(8, true, false, 2, false);
    else if (rem) CALL_REDUCE(8, false, false, 2, true);
    else CALL_REDUCE(8, false, false, 2, false);
#undef CALL_REDUCE
    HIP_TRY(h, hipGetLastError());
    // Fold the published partial rows into the shard summary.  Sharded use needs the summary before the
    // collective; otherwise mppi_finalize folds the rows itself when the previous solves published few of them
    // (*live_hint, written by finalize_kernel to mapped host memory and read here without synchronising: it only
    // steers this choice: both folds use the same summation tree, so the summary is bit-identical either way).
    h->summary_valid = false;
    P2pCtx p2p{};
    if (h->p2p_enabled) {  // summarize_kernel also hands the summary to every peer (and to this rank's own slot)
        ++h->p2p_seq;
        if (h->p2p_seq == 0) h->p2p_seq = 1;
        p2p = p2p_ctx(h);
    }
    const bool many_rows = h->fold_mode == 0 ? *(volatile int*)h->live_hint > FOLD_IN_FINALIZE_MAX_ROWS : h->fold_mode == 2;
    const bool comm = h->comm_enabled && !h->p2p_enabled;
    if (comm && summary_out_dev) return fail(h, MPPI_E_INVALID, "exchange_comm: the library gathers the summaries itself (pass NULL)");
    if (summary_out_dev || h->p2p_enabled || comm || !fold_fits(h) || many_rows) {
        const unsigned sgrid = (unsigned)((h->colsp + SUM_COLS - 1) / SUM_COLS + 1);
        hipLaunchKernelGGL(summarize_kernel, dim3(sgrid), dim3(SUM_BLOCK), 0, s, h->partials, h->heads, mk, (int)blocks,
                           h->colsp, h->d.row, h->summary, comm ? h->comm_send : summary_out_dev, h->live_hint_dev, p2p);
        HIP_TRY(h, hipGetLastError());
        h->summary_valid = true;
    }
    if (comm)  // the solve's only exchange: 4 + T*dc floats per rank, on the solve's own stream
        RCCL_TRY(h, rccl().all_gather(h->comm_send, h->comm_recv, (size_t)(MPPI_SUMMARY_HEAD + h->d.row), ncclFloat, h->comm, s));
    return MPPI_OK;
}

int mppi_finalize(mppi_handle_t h, const float* summaries_dev, int num_shards, float lambda, int store_mean,
                  float* action_out, float* state_out, float* stats_out, void* stream) {
    if (!h || num_shards < 1) return fail(h, MPPI_E_INVALID, "bad finalize arguments");
    const float* lam_dev = nullptr;
    if (int rc = resolve_lambda(h, lambda, &lam_dev)) return rc;
    const bool generic = h->cfg.model == MPPI_MODEL_GENERIC;
    if (generic && state_out) return fail(h, MPPI_E_INVALID, "generic model: roll the action out with the host dynamics");
    if (!generic) { if (int rc = check_ready(h)) return rc; }
    hipStream_t s = (hipStream_t)stream;
    P2pCtx p2p{};  // seq == 0: off
    if (!summaries_dev) {  // this handle's own reduction (mppi_weights_reduce)
        if (h->last_reduce_blocks < 1) return fail(h, MPPI_E_STATE, "mppi_finalize before mppi_weights_reduce");
        if (h->p2p_enabled) {
            p2p = p2p_ctx(h);  // all shards' summaries of this solve, through the exchange buffer
            num_shards = h->p2p_world;
        } else if (h->comm_enabled) {
            summaries_dev = h->comm_recv;  // gathered by mppi_weights_reduce
            num_shards = h->comm_world;
        } else {
            if (h->summary_valid) summaries_dev = h->summary;  // else the kernel folds the partial rows itself
            num_shards = 1;
        }
    }
    // the filter replaces the stored warm start, so it only runs when this call stores it (mppi.py:441-452)
    const SgFilter sg{h->sg_coeffs, h->sg_history, (store_mean && h->sg_window > 0) ? h->sg_window : 0};
    const bool fold_here = !summaries_dev && !p2p.seq;  // the kernel folds the partial rows itself
    const size_t shmem = finalize_lds_floats(h, p2p.seq ? p2p.world : 1, sg.window, fold_here) * sizeof(float);
    if (shmem > 64 * 1024) return fail(h, MPPI_E_INVALID, "finalize: horizon too long for the exchange / filter staging");
    const unsigned* mk = h->min_key + h->min_slot;
    // Option "lazy_state_seq": the batch-1 rollout leaves this kernel (and the solve's critical path).  The kernel writes the
    // rollout's inputs to h->b1; the rollout itself rides in the next mppi_rollout_cost launch on this stream, or is
    // launched by mppi_join_state_seq when somebody reads the state sequence first.
    const bool defer = h->lazy_state && state_out && !generic && h->mapping == 0;
    if (h->pending_state_out) { if (int rc = flush_state_seq(h, s)) return rc; }  // (an older one nobody picked up)
    {
        StageTimer tm(h, 3, s);
#define CALL_FINALIZE(MODEL, FASTV)                                                                   \
    hipLaunchKernelGGL((finalize_kernel<MODEL, FASTV>), dim3(1), dim3(FIN_BLOCK), shmem, s, summaries_dev, num_shards, \
                       h->partials, h->heads, mk, h->last_reduce_blocks, h->colsp, h->summary, h->live_hint_dev,  \
                       lambda, lam_dev, h->d.row, h->d.T, h->x0_cur, store_mean ? h->mean : (float*)nullptr, action_out,  \
                       defer ? (float*)nullptr : state_out, stats_out, h->solve_stats, sg, p2p, h->ctx,          \
                       defer ? h->b1 : (float*)nullptr, defer ? state_out : (float*)nullptr)
        MPPI_DISPATCH(h, CALL_FINALIZE);
#undef CALL_FINALIZE
    }
    HIP_TRY(h, hipGetLastError());
    ++h->finalize_serial;
    if (defer) { h->pending_state_out = state_out; h->pending_serial = h->finalize_serial; h->pending_stream = s; }
    return MPPI_OK;
}

// A pending state sequence is about to be completed on `s`: if that is not the stream its finalize_kernel ran on, order `s`
// behind everything enqueued there so far (an event recorded NOW on the producing stream sits after finalize's write of b1).
static int order_behind_pending(mppi_handle_t h, hipStream_t s) {
    if (s == h->pending_stream) return MPPI_OK;
    if (!h->lazy_ev) HIP_TRY(h, hipEventCreateWithFlags(&h->lazy_ev, hipEventDisableTiming));
    HIP_TRY(h, hipEventRecord(h->lazy_ev, h->pending_stream));
    HIP_TRY(h, hipStreamWaitEvent(s, h->lazy_ev, 0));
    return MPPI_OK;
}

// The pending batch-1 rollout as its own one-wave kernel on `s` (same code and bits as the in-kernel rollout).  The pending
// mark is cleared only once the launch went through.
static int flush_state_seq(mppi_handle_t h, hipStream_t s) {
    if (!h->pending_state_out) return MPPI_OK;
    float* out = h->pending_state_out;
    if (int rc = order_behind_pending(h, s)) return rc;
    const size_t sh1 = sizeof(float) * ((size_t)h->d.row + MPPI_MAX_DIM_STATE);
    StageTimer tm(h, 4, s);
#define CALL_STATE_SEQ(MODEL, FASTV)                                                                  \
    hipLaunchKernelGGL((state_seq_kernel<MODEL, FASTV>), dim3(1), dim3(WAVE), sh1, s, (const float*)h->b1, h->d.row, h->d.T, \
                       out, h->ctx)
    MPPI_DISPATCH(h, CALL_STATE_SEQ);
#undef CALL_STATE_SEQ
    HIP_TRY(h, hipGetLastError());
    h->pending_state_out = nullptr;
    return MPPI_OK;
}
// Before anything that changes which kernel variant MPPI_DISPATCH picks or what the model context holds (math level,
// mapping, maps, model parameters): complete a pending state sequence with the settings of the solve it belongs to, on the
// stream that solve ran on.
static int settle_state_seq(mppi_handle_t h) { return h->pending_state_out ? flush_state_seq(h, h->pending_stream) : MPPI_OK; }

// Complete the state sequence of the last mppi_finalize / mppi_solve on `stream` if its rollout is still pending (option
// "lazy_state_seq"); a no-op otherwise.  `serial` = 0, or the value mppi_state_seq_serial returned right after that solve:
// a reader of an OLDER solve's state sequence (already completed by a later rollout launch) then launches nothing.
int mppi_join_state_seq(mppi_handle_t h, uint32_t serial, void* stream) {
    if (!h) return MPPI_E_INVALID;
    if (!h->pending_state_out || (serial && serial != h->pending_serial)) return MPPI_OK;
    return flush_state_seq(h, (hipStream_t)stream);
}
// Serial number of the last mppi_finalize (for mppi_join_state_seq), and whether its state sequence is still pending.
int mppi_state_seq_serial(mppi_handle_t h, uint32_t* serial_out, int* pending_out) {
    if (!h) return MPPI_E_INVALID;
    if (serial_out) *serial_out = h->finalize_serial;
    if (pending_out) *pending_out = h->pending_state_out != nullptr && h->pending_serial == h->finalize_serial;
    return MPPI_OK;
}

// ---- the single-launch solve (solve_fused_kernel)
static constexpr int64_t FUSED_AUTO_MAX_SAMPLES = 4096;          // fixed temperature / MPO
static constexpr int64_t FUSED_AUTO_MAX_SAMPLES_SEARCH = 16384;  // ESSPS / LBPS on the device
static bool fused_applies(mppi_handle_t h, float lambda) {
    if (!h->fused_mode || h->cfg.model == MPPI_MODEL_GENERIC || h->mapping != 0) return false;
    // measured (profiles/r03_experiments.md, r03_visitD_fused_crossover.txt): a cell round trip costs about as much as a
    // kernel boundary, so the single launch wins where it replaces more kernel boundaries than it needs round trips — with
    // a fixed temperature up to a few thousand samples (27 vs 32 us for racing at N = 1024, 29 vs 32 at 4096, 33 vs 32 at
    // 8192), under a temperature search further (nav2d ESSPS 32 vs 47 us at N = 1024, 48 vs 52 at 16 384, 51 vs 52 at 32 768)
    const bool search = lambda == MPPI_LAMBDA_DEVICE && (h->auto_rule == MPPI_AUTO_ESSPS || h->auto_rule == MPPI_AUTO_LBPS);
    // (the single launch searches LBPS on 32-temperature grids; the reference's Brent search is a kernel of its own)
    if (lambda == MPPI_LAMBDA_DEVICE && h->auto_rule == MPPI_AUTO_LBPS && !h->lbps_grid) return false;
    if (h->fused_mode == 1 && h->d.N > (search ? FUSED_AUTO_MAX_SAMPLES_SEARCH : FUSED_AUTO_MAX_SAMPLES)) return false;
    if (!(h->noise_regen && !h->injected && !h->wide)) return false;         // the noise is regenerated in registers
    if (h->p2p_enabled || h->comm_enabled) return false;                      // sharded solves exchange between devices
    if (h->d.row > FUSED_MAX_ROW) return false;
    if (h->d.N > (int64_t)FUSED_BLOCK * std::min(FUSED_MAX_BLOCKS, h->cu_count)) return false;  // every block must be resident at once
    if (h->fused_error && *(volatile int*)h->fused_error) return false;       // a poll timed out once: stay on the multi-kernel path
    if (lambda == MPPI_LAMBDA_DEVICE && h->auto_rule == MPPI_AUTO_MPO && !h->lambda_dev_valid) return false;
    if (h->timing == 1) return false;                                         // per-stage timing brackets the separate kernels
    return check_ready(h) == MPPI_OK;
}

// The state of the device-resident ESSPS search for [lam_min, lam_max]: a cold (geometric) first grid; after that every
// finished search leaves the first grid of the next one behind (host_search.hpp: essps_first_grid).  Set-up path, blocking.
static int essps_prepare(mppi_handle_t h, double lam_min, double lam_max) {
    if (h->essps_lo == lam_min && h->essps_hi == lam_max) return MPPI_OK;
    EsspsDev st{};
    float lamf[STATS_L];
    h->essps_range = mppi::host::essps_range(lam_min, lam_max);
    mppi::host::essps_first_grid<STATS_L>(false, 0.0, h->essps_range, st.grid0, st.lgrid0);
    for (int j = 0; j < STATS_L; ++j) { lamf[j] = (float)st.grid0[j]; st.grid1[j] = st.grid0[j]; st.lgrid1[j] = st.lgrid0[j]; }
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(h->lams_dev + STATS_L, lamf, sizeof(lamf), hipMemcpyHostToDevice));
    // (round 1's grid is rewritten by every search; a valid one for the searches that end after round 0)
    HIP_TRY(h, hipMemcpy(h->lams_dev + 2 * STATS_L, lamf, sizeof(lamf), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->essps_dev, &st, sizeof(st), hipMemcpyHostToDevice));
    h->essps_lo = lam_min; h->essps_hi = lam_max;
    return MPPI_OK;
}

static int solve_fused(mppi_handle_t h, float lambda, float* action_out, float* state_out, float* stats_out, hipStream_t s,
                       bool* declined) {
    *declined = false;
    if (!h->fused_cells) {  // first use: set-up path (blocking)
        const size_t bytes = sizeof(unsigned long long) * FX_PHASES * FUSED_MAX_BLOCKS * FX_CELLS;
        HIP_TRY(h, hipMalloc(&h->fused_cells, bytes));
        HIP_TRY(h, hipMemset(h->fused_cells, 0, bytes));
        HIP_TRY(h, hipMalloc(&h->grid0_dev, sizeof(double) * STATS_L));
        HIP_TRY(h, hipHostMalloc((void**)&h->fused_error, sizeof(int) * 64, hipHostMallocMapped));
        *h->fused_error = 0;
        HIP_TRY(h, hipHostGetDevicePointer((void**)&h->fused_error_dev, h->fused_error, 0));
        HIP_TRY(h, hipDeviceSynchronize());
    }
    const bool dev = lambda == MPPI_LAMBDA_DEVICE;
    int rule = FUSED_RULE_NONE;
    if (dev && h->auto_rule == MPPI_AUTO_ESSPS) rule = FUSED_RULE_ESSPS;
    if (dev && h->auto_rule == MPPI_AUTO_LBPS) rule = FUSED_RULE_LBPS;
    if (rule == FUSED_RULE_ESSPS)
        if (int rc = essps_prepare(h, h->auto_lo, h->auto_hi)) return rc;
    if (rule == FUSED_RULE_LBPS && (h->grid0_lo != h->auto_lo || h->grid0_hi != h->auto_hi)) {  // set-up path, blocking
        double g0[STATS_L];
        mppi::host::essps_make_grid<STATS_L>(h->auto_lo, h->auto_hi, g0);
        HIP_TRY(h, hipDeviceSynchronize());
        HIP_TRY(h, hipMemcpy(h->grid0_dev, g0, sizeof(g0), hipMemcpyHostToDevice));
        h->grid0_lo = h->auto_lo; h->grid0_hi = h->auto_hi;
    }
    if (!dev && !(lambda > 0.0f)) return fail(h, MPPI_E_INVALID, "lambda must be > 0");
    h->min_slot ^= 1;
    ++h->fused_seq;
    if (h->fused_seq == 0) h->fused_seq = 1;
    double* host_lam = nullptr;
    host_lam = h->stats_host_dev;
    host_lam += 8 + STATS_L * 3;
    FusedArgs A{};
    A.mean = h->mean; A.x0 = h->x0_cur; A.costs = h->costs;
    A.min_key = h->min_key + h->min_slot; A.next_min_key = h->min_key + (h->min_slot ^ 1);
    A.mean_used = h->mean_used; A.x0_used = h->x0_used;
    A.rule = rule; A.rule_param = h->auto_param; A.lam_min = h->auto_lo; A.lam_max = h->auto_hi;
    A.lambda_arg = dev ? -1.0f : lambda;
    A.lambda_dev = h->lambda_dev; A.lambda_host = host_lam;
    A.grid0 = rule == FUSED_RULE_ESSPS ? h->essps_dev->grid0 : h->grid0_dev;
    A.lams0 = h->lams_dev + STATS_L;
    A.essps = h->essps_dev; A.range = h->essps_range;
    A.mean_store = h->mean; A.action_out = action_out; A.state_out = state_out; A.stats_out = stats_out;
    A.stats_keep = h->solve_stats; A.summary_out = h->summary;
    const SgFilter sg{h->sg_coeffs, h->sg_history, h->sg_window};
    const FusedCtx fx{h->fused_cells, h->fused_error_dev, h->fused_seq, h->fused_timeout_ticks};
    // G = min(#CUs, ceil(N / 64)) blocks, each owning spb (a multiple of 64, <= 1024) consecutive trajectories: ONE wave
    // of rollouts per block as long as there are CUs left (the rest of its 1024 threads share the block's reductions and
    // the regeneration of its weighted noise rows, which a block of 256 trajectories spends ~5 us on)
    const int64_t gmax = std::min(FUSED_MAX_BLOCKS, h->cu_count);
    unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(gmax, (h->d.N + 63) / 64));
    // up to FUSED_AUTO_MAX_SAMPLES trajectories: at most FUSED_SMALL_BLOCKS blocks, which then need neither the hop for the
    // global minimum nor the broadcast of the temperature (solve_fused_kernel: `small`)
    if (h->d.N <= FUSED_AUTO_MAX_SAMPLES) grid = std::min<unsigned>(grid, (unsigned)FUSED_SMALL_BLOCKS);
    A.spb = (int)(((h->d.N + grid - 1) / grid + 63) / 64 * 64);
    grid = (unsigned)((h->d.N + A.spb - 1) / A.spb);  // (no block without trajectories)
#define CALL_FUSED(MODEL, FASTV)                                                                      \
    do {                                                                                              \
        const size_t shmem = sizeof(float) * ((size_t)8 * h->d.R + (size_t)h->d.T * ModelT<MODEL, FASTV>::KROW + 2 * (size_t)h->d.row + \
                                              MPPI_SUMMARY_HEAD + (sg.window ? (size_t)(2 * h->d.T - 1 + 2 * (sg.window / 2)) * h->dc : 0)); \
        /* the blocks synchronise through cells in HBM: every one of them must be resident at once.  Checked against the   \
           kernel's own occupancy (cached per (math level, LDS size)); what OTHER work holds of the GPU at run time is what  \
           the polls' time-out is for */                                                                                   \
        const uint64_t okey = ((uint64_t)(FASTV + 1) << 48) | (uint64_t)shmem;                                             \
        if (h->fused_occ_key != okey) {                                                                                    \
            int nb = 0;                                                                                                    \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, solve_fused_kernel<MODEL, FASTV>, FUSED_BLOCK, shmem) != hipSuccess) nb = 0; \
            h->fused_occ_key = okey; h->fused_occ_blocks = nb;                                                             \
        }                                                                                                                  \
        if ((int64_t)grid > (int64_t)h->fused_occ_blocks * h->cu_count) { *declined = true; break; }                       \
        StageTimer tm(h, 1, s);  /* (after the occupancy check: a declined launch leaves no empty event pair behind) */    \
        hipLaunchKernelGGL((solve_fused_kernel<MODEL, FASTV>), dim3(grid), dim3(FUSED_BLOCK), shmem, s, A, h->d, h->gen, h->ctx, sg, fx); \
    } while (0)
    MPPI_DISPATCH(h, CALL_FUSED);
#undef CALL_FUSED
    if (*declined) {  // (undo the bookkeeping of a solve that did not start: the multi-kernel path takes it from here)
        h->min_slot ^= 1;
        --h->fused_seq;
        return MPPI_OK;
    }
    HIP_TRY(h, hipGetLastError());
    if (rule != FUSED_RULE_NONE) h->lambda_dev_valid = true;
    // (the single launch rolls the solution out itself; a state sequence still pending from an earlier multi-kernel solve
    // was completed by mppi_solve before it got here)
    h->last_reduce_blocks = 0;   // no partial rows of a separate reduction exist for this solve
    h->summary_valid = true;     // ... but its summary does (h->summary)
    return MPPI_OK;
}

// The temperature rule mppi_solve applies when it is called with lambda = MPPI_LAMBDA_DEVICE (mppi.py:183-210).
int mppi_set_auto_lambda(mppi_handle_t h, int rule, double param, double lam_min, double lam_max) {
    if (!h || rule < MPPI_AUTO_NONE || rule > MPPI_AUTO_MPO) return fail(h, MPPI_E_INVALID, "bad temperature rule");
    if ((rule == MPPI_AUTO_ESSPS || rule == MPPI_AUTO_LBPS) && (!(lam_min > 0.0) || !(lam_max > lam_min) || !(param > 0.0)))
        return fail(h, MPPI_E_INVALID, "bad temperature rule arguments");
    h->auto_rule = rule; h->auto_param = param; h->auto_lo = lam_min; h->auto_hi = lam_max;
    return MPPI_OK;
}

// MPPI.forward() for a native model in ONE call (mppi.py:223-460): bind the state, fix the noise identity, rollout +
// costs, the temperature (fixed, or the configured rule resident on the device), weights + reduction, finalize with the
// warm start stored.  Exactly the sequence of the individual entry points (same kernels, same results): one
// host -> library transition per solve for callers that need nothing in between.
int mppi_solve(mppi_handle_t h, const float* x0_dev, uint32_t solve_idx, float lambda, float* action_out_dev,
               float* state_seq_out_dev, float* stats_out_dev, void* stream) {
    if (!h) return MPPI_E_INVALID;
    const bool dev = lambda == MPPI_LAMBDA_DEVICE;
    if (dev && h->auto_rule == MPPI_AUTO_NONE)
        return fail(h, MPPI_E_STATE, "MPPI_LAMBDA_DEVICE: no temperature rule configured (mppi_set_auto_lambda)");
    if (x0_dev) { if (int rc = mppi_bind_state(h, x0_dev)) return rc; }
    if (int rc = mppi_sample(h, solve_idx, stream)) return rc;
    if (fused_applies(h, lambda)) {
        if (int rc = flush_state_seq(h, (hipStream_t)stream)) return rc;  // (pending from an earlier multi-kernel solve)
        bool declined = false;
        if (int rc = solve_fused(h, lambda, action_out_dev, state_seq_out_dev, stats_out_dev, (hipStream_t)stream, &declined)) return rc;
        if (!declined) {
            if (h->auto_rule == MPPI_AUTO_MPO) return mppi_mpo_step_device(h, stream);
            return MPPI_OK;
        }
    }
    if (int rc = mppi_rollout_cost(h, stream)) return rc;
    if (dev && h->auto_rule == MPPI_AUTO_ESSPS) {
        if (int rc = mppi_essps_lambda_device(h, h->auto_param, h->auto_lo, h->auto_hi, stream)) return rc;
    } else if (dev && h->auto_rule == MPPI_AUTO_LBPS) {
        if (int rc = h->lbps_grid ? mppi_lbps_lambda_device(h, h->auto_param, h->auto_lo, h->auto_hi, stream)
                                  : mppi_lbps_brent_device(h, h->auto_param, h->auto_lo, h->auto_hi, stream)) return rc;
    }
    if (int rc = mppi_weights_reduce(h, lambda, nullptr, stream)) return rc;
    if (int rc = mppi_finalize(h, nullptr, 1, lambda, 1, action_out_dev, state_seq_out_dev, stats_out_dev, stream)) return rc;
    // MPO: the dual steps after every solve, whatever temperature this solve's weights were given (mppi.py:387-398)
    if (h->auto_rule == MPPI_AUTO_MPO) return mppi_mpo_step_device(h, stream);
    return MPPI_OK;
}

// Savitzky-Golay smoothing of the solution inside mppi_finalize (step 7, mppi.py:423-443): taps = first row of
// pinv(vander) computed by the caller (mppi.py:568-596), history = `_actions_history_for_sg`.
int mppi_set_sg_filter(mppi_handle_t h, const float* coeffs_host, int window, const float* history_host) {
    if (!h || window < 0) return fail(h, MPPI_E_INVALID, "bad sg filter arguments");
    if (window == 0) { h->sg_window = 0; return MPPI_OK; }
    if (!coeffs_host || window % 2 == 0 || window > 255) return fail(h, MPPI_E_INVALID, "sg window must be odd and <= 255");
    if (h->d.row > FIN_BLOCK) return fail(h, MPPI_E_INVALID, "sg filter on the device supports T*dim_control <= 1024");
    if (window / 2 > 2 * h->d.T - 1) return fail(h, MPPI_E_INVALID, "sg window too wide for the horizon");
    const size_t hist_floats = (size_t)std::max(h->d.T - 1, 1) * h->dc;
    if (!h->sg_coeffs) HIP_TRY(h, hipMalloc(&h->sg_coeffs, sizeof(float) * 256));
    if (!h->sg_history) {
        HIP_TRY(h, hipMalloc(&h->sg_history, sizeof(float) * hist_floats));
        HIP_TRY(h, hipMemset(h->sg_history, 0, sizeof(float) * hist_floats));
    }
    HIP_TRY(h, hipMemcpy(h->sg_coeffs, coeffs_host, sizeof(float) * (size_t)window, hipMemcpyHostToDevice));
    if (history_host)
        HIP_TRY(h, hipMemcpy(h->sg_history, history_host, sizeof(float) * (size_t)(h->d.T - 1) * h->dc, hipMemcpyHostToDevice));
    h->sg_window = window;
    return MPPI_OK;
}

int mppi_get_sg_history(mppi_handle_t h, float* history_host) {
    if (!h || !history_host) return fail(h, MPPI_E_INVALID, "null");
    if (!h->sg_history) return fail(h, MPPI_E_STATE, "sg filter not set");
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(history_host, h->sg_history, sizeof(float) * (size_t)(h->d.T - 1) * h->dc, hipMemcpyDeviceToHost));
    return MPPI_OK;
}

int mppi_softmax_stats(mppi_handle_t h, float lambda, double* out5_host, void* stream) {
    if (!h || !out5_host || !(lambda > 0.0f)) return fail(h, MPPI_E_INVALID, "bad softmax_stats arguments");
    hipStream_t s = (hipStream_t)stream;
    const unsigned* mk = h->min_key + h->min_slot;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(STATS_BLOCKS, (h->d.N + BLOCK - 1) / BLOCK));
    hipLaunchKernelGGL(stats_partial_kernel, dim3(blocks), dim3(BLOCK), 0, s, h->costs, h->d.N, mk, lambda,
                       (const float*)nullptr, h->stats_part);
    HIP_TRY(h, hipGetLastError());
    double* dev_out = nullptr;
    dev_out = h->stats_host_dev;
    hipLaunchKernelGGL(stats_combine_kernel, dim3(1), dim3(WAVE), 0, s, h->stats_part, blocks, mk, dev_out);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(s));
    for (int i = 0; i < 5; ++i) out5_host[i] = h->stats_host[i];
    return MPPI_OK;
}

static int stats_blocks(mppi_handle_t h) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(STATS_BLOCKS, (h->d.N + STATS_THREADS - 1) / STATS_THREADS));
}

int mppi_softmax_stats_multi(mppi_handle_t h, const float* lambdas_host, int count, double* out_host, void* stream) {
    if (!h || !lambdas_host || !out_host || count < 1 || count > STATS_L)
        return fail(h, MPPI_E_INVALID, "bad softmax_stats_multi arguments (1..32 lambdas)");
    float lam[STATS_L];
    for (int l = 0; l < STATS_L; ++l) {
        lam[l] = l < count ? lambdas_host[l] : 1.0f;
        if (!(lam[l] > 0.0f)) return fail(h, MPPI_E_INVALID, "lambda must be > 0");
    }
    hipStream_t s = (hipStream_t)stream;
    if (int rc = upload_small(h, h->lams_dev, lam, STATS_L, s)) return rc;
    const unsigned* mk = h->min_key + h->min_slot;
    const int blocks = stats_blocks(h);
    hipLaunchKernelGGL(stats_multi_kernel, dim3(blocks), dim3(STATS_THREADS), 0, s, h->costs, h->d.N, mk,
                       (const float*)h->lams_dev, h->stats_part, (float*)nullptr);
    HIP_TRY(h, hipGetLastError());
    double* dev_out = nullptr;
    dev_out = h->stats_host_dev;
    hipLaunchKernelGGL(stats_multi_combine_kernel, dim3(1), dim3(1024), 0, s, h->stats_part, blocks, dev_out + 8);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(s));
    for (int j = 0; j < count * 3; ++j) out_host[j] = h->stats_host[8 + j];
    return MPPI_OK;
}

// ESSPS with no host synchronisation (mppi.py:351-370): statistics pass over the preset round-0 grid -> select
// (end-point rules / refined grid, on the device) -> statistics pass over that grid -> select (root) -> the
// temperature stays in HBM, where mppi_weights_reduce / mppi_finalize read it when called with MPPI_LAMBDA_DEVICE;
// mppi_get_lambda fetches it (synchronises).  Same arithmetic as mppi_essps_lambda: both run host_search.hpp.
int mppi_essps_lambda_device(mppi_handle_t h, double target_ess, double lam_min, double lam_max, void* stream) {
    if (!h || !(lam_min > 0.0) || !(lam_max > lam_min) || !(target_ess > 0.0))
        return fail(h, MPPI_E_INVALID, "bad essps arguments");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = essps_prepare(h, lam_min, lam_max)) return rc;
    const unsigned* mk = h->min_key + h->min_slot;
    const int blocks = stats_blocks(h);
    double* host_lam = nullptr;
    host_lam = h->stats_host_dev;
    host_lam += 8 + STATS_L * 3;
    float* lams0 = h->lams_dev + STATS_L;
    float* lams1 = h->lams_dev + 2 * STATS_L;
    for (int r = 0; r < 2; ++r) {
        if (++h->round1_seq == 0u) h->round1_seq = 1u;  // (the cells start out zeroed: 0 tags nothing)
        if (r == 0 && h->essps_merge0)
            hipLaunchKernelGGL(essps_round_kernel<0>, dim3(blocks), dim3(STATS_THREADS), 0, s, h->costs, h->d.N, mk, target_ess,
                               h->essps_range, h->essps_dev, lams1, lams0, h->lambda_dev, host_lam, h->round1_cells,
                               h->round1_seq);
        else if (r == 0) {
            hipLaunchKernelGGL(stats_multi_kernel, dim3(blocks), dim3(STATS_THREADS), 0, s, h->costs, h->d.N, mk,
                               (const float*)lams0, h->stats_part, (float*)nullptr);
            hipLaunchKernelGGL(essps_select_kernel, dim3(1), dim3(1024), 0, s, (const float*)h->stats_part, blocks, target_ess,
                               h->essps_range, h->essps_dev, lams1, lams0, h->lambda_dev, host_lam);
        } else
            hipLaunchKernelGGL(essps_round_kernel<1>, dim3(blocks), dim3(STATS_THREADS), 0, s, h->costs, h->d.N, mk, target_ess,
                               h->essps_range, h->essps_dev, lams1, lams0, h->lambda_dev, host_lam, h->round1_cells,
                               h->round1_seq);
    }
    HIP_TRY(h, hipGetLastError());
    h->lambda_dev_valid = true;
    return MPPI_OK;
}

// The temperature a device-resident rule left behind (and, optionally, the one the last solve's weights used: the same
// for ESSPS / LBPS, the previous one for MPO).  Synchronises the stream.
int mppi_get_lambda(mppi_handle_t h, double* lambda_out_host, double* lambda_used_out_host, void* stream) {
    if (!h || !lambda_out_host) return fail(h, MPPI_E_INVALID, "null");
    if (!h->lambda_dev_valid) return fail(h, MPPI_E_STATE, "no temperature on the device");
    HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    *lambda_out_host = h->stats_host[8 + STATS_L * 3];
    if (lambda_used_out_host) *lambda_used_out_host = h->stats_host[8 + STATS_L * 3 + 1];
    return MPPI_OK;
}

// Passes over the costs (32-temperature grids) the last device-resident ESSPS / LBPS search took: ESSPS 1 when an
// end-point rule decided or the warm-started first grid was enough, else 2; LBPS always LBPS_ROUNDS.  Synchronises.
int mppi_search_passes(mppi_handle_t h, void* stream) {
    if (!h || !h->lambda_dev_valid) return 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 0;
    return (int)h->stats_host[8 + STATS_L * 3 + 2];
}

// LBPS with no host synchronisation (mppi.py:341-349): LBPS_ROUNDS x (32-temperature statistics pass -> one-block
// grid step), the temperature stays in HBM (MPPI_LAMBDA_DEVICE).  See lbps_select_kernel.
int mppi_lbps_lambda_device(mppi_handle_t h, double delta, double lam_min, double lam_max, void* stream) {
    if (!h || !(lam_min > 0.0) || !(lam_max > lam_min) || !(delta > 0.0) || !(delta < 1.0))
        return fail(h, MPPI_E_INVALID, "bad lbps arguments");
    hipStream_t s = (hipStream_t)stream;
    float* lams0 = h->lams_dev;              // (the caller's-grid slot doubles as LBPS's preset round-0 grid)
    float* lams1 = h->lams_dev + 2 * STATS_L;
    if (h->lbps_lo != lam_min || h->lbps_hi != lam_max) {  // (re)build the round-0 grid: set-up path, blocking
        LbpsDev st{};
        float lamf[STATS_L];
        mppi::host::essps_make_grid<STATS_L>(lam_min, lam_max, st.grid0);
        for (int j = 0; j < STATS_L; ++j) { lamf[j] = (float)st.grid0[j]; st.grid[j] = st.grid0[j]; }
        HIP_TRY(h, hipDeviceSynchronize());
        HIP_TRY(h, hipMemcpy(h->lbps_dev, &st, sizeof(st), hipMemcpyHostToDevice));
        h->lbps_lo = lam_min; h->lbps_hi = lam_max;
    }
    {   // the caller's-grid slot may have been overwritten by mppi_softmax_stats_multi: refresh it from the search state
        float lamf[STATS_L];
        double g0[STATS_L];
        mppi::host::essps_make_grid<STATS_L>(lam_min, lam_max, g0);
        for (int j = 0; j < STATS_L; ++j) lamf[j] = (float)g0[j];
        if (int rc = upload_small(h, lams0, lamf, STATS_L, s)) return rc;
    }
    const unsigned* mk = h->min_key + h->min_slot;
    const int blocks = stats_blocks(h);
    double* host_lam = nullptr;
    host_lam = h->stats_host_dev;
    host_lam += 8 + STATS_L * 3;
    for (int r = 0; r < LBPS_ROUNDS; ++r) {
        hipLaunchKernelGGL(stats_multi_kernel, dim3(blocks), dim3(STATS_THREADS), 0, s, h->costs, h->d.N, mk,
                           (const float*)(r == 0 ? lams0 : lams1), h->stats_part, h->stats_max);
        if (r == 0)
            hipLaunchKernelGGL((lbps_select_kernel<false, true>), dim3(1), dim3(1024), 0, s, (const float*)h->stats_part,
                               (const float*)h->stats_max, blocks, mk, delta, h->lbps_dev, lams1, h->lambda_dev, host_lam);
        else if (r < LBPS_ROUNDS - 1)
            hipLaunchKernelGGL((lbps_select_kernel<false, false>), dim3(1), dim3(1024), 0, s, (const float*)h->stats_part,
                               (const float*)h->stats_max, blocks, mk, delta, h->lbps_dev, lams1, h->lambda_dev, host_lam);
        else
            hipLaunchKernelGGL((lbps_select_kernel<true, false>), dim3(1), dim3(1024), 0, s, (const float*)h->stats_part,
                               (const float*)h->stats_max, blocks, mk, delta, h->lbps_dev, lams1, h->lambda_dev, host_lam);
    }
    HIP_TRY(h, hipGetLastError());
    h->lambda_dev_valid = true;
    return MPPI_OK;
}

// ESSPS temperature (mppi.py:351-370,559-566): the root of ESS(lambda) = target on [lam_min, lam_max] with the
// reference's end-point rules, found on the host from device statistics — two 32-point geometric grids
// (mppi_softmax_stats_multi: one pass over the costs each) and an inverse polynomial interpolation in
// (ESS, log lambda): host::essps_lambda in host_search.hpp.  Same algorithm as pi_mpc/_host.py::essps_lambda_grid
// (which sharded solvers use, with an all_gather per grid); kept in the library so that the single-GPU solve has no
// interpreter work per probe.
int mppi_essps_lambda(mppi_handle_t h, double target_ess, double lam_min, double lam_max, double* lambda_out,
                      void* stream) {
    if (!h || !lambda_out || !(lam_min > 0.0) || !(lam_max > lam_min) || !(target_ess > 0.0))
        return fail(h, MPPI_E_INVALID, "bad essps arguments");
    constexpr int P = STATS_L;
    int rc = MPPI_OK;
    if (h->essps_prev_lo != lam_min || h->essps_prev_hi != lam_max) h->essps_prev_host.warm = false;  // another range: a cold search
    const bool ok = mppi::host::essps_lambda<P>(
        [&](const double* grid, double* ess) {
            float lamf[P];
            double raw[P * 3];
            for (int j = 0; j < P; ++j) lamf[j] = (float)grid[j];
            rc = mppi_softmax_stats_multi(h, lamf, P, raw, stream);
            if (rc) return false;
            for (int j = 0; j < P; ++j) ess[j] = raw[3 * j] * raw[3 * j] / raw[3 * j + 1];
            return true;
        },
        target_ess, lam_min, lam_max, *lambda_out, h->essps_prev_host);
    h->essps_prev_lo = lam_min; h->essps_prev_hi = lam_max;
    if (!ok) h->essps_prev_host.warm = false;
    return ok ? MPPI_OK : rc;
}

// LBPS as the reference searches it (mppi.py:341-349: scipy's bounded Brent, host::fminbound step for step) with NO host
// synchronisation: ONE launch of lbps_brent_kernel (mppi_search.hpp) runs every probe — the statistics of
// mppi_softmax_stats bit for bit, gathered by every block through tagged cells — and leaves the temperature in HBM
// (MPPI_LAMBDA_DEVICE) and in mapped host memory.  The same temperature as mppi_lbps_lambda, to the bit.
int mppi_lbps_brent_device(mppi_handle_t h, double delta, double lam_min, double lam_max, void* stream) {
    if (!h || !(lam_min > 0.0) || !(lam_max > lam_min) || !(delta > 0.0) || !(delta < 1.0))
        return fail(h, MPPI_E_INVALID, "bad lbps arguments");
    hipStream_t s = (hipStream_t)stream;
    // the geometry of mppi_softmax_stats (stats_partial_kernel): nvb blocks of 256 threads, grid-stride over the costs;
    // block l of this launch runs the virtual blocks l, l + 64, ... (lane l's rows of stats_combine_kernel)
    const int nvb = (int)std::max<int64_t>(1, std::min<int64_t>(STATS_BLOCKS, (h->d.N + BLOCK - 1) / BLOCK));
    const int64_t per_thread64 = (h->d.N + (int64_t)nvb * BLOCK - 1) / ((int64_t)nvb * BLOCK);
    const int grid = std::min(nvb, BRENT_LANES);
    const int threads = BLOCK * ((nvb + BRENT_LANES - 1) / BRENT_LANES);
    if (grid > h->cu_count) return fail(h, MPPI_E_STATE, "lbps_brent: more blocks than CUs (they must be resident at once)");
    int per_thread = (int)std::min<int64_t>(per_thread64, BRENT_STAGE_MAX + 1);  // (beyond the staging limit only the flag matters)
    size_t shmem = per_thread <= BRENT_STAGE_MAX ? sizeof(float) * (size_t)per_thread * threads : 0;
    if (shmem + sizeof(BrentLds) + 256 > (size_t)h->lds_max) { per_thread = BRENT_STAGE_MAX + 1; shmem = 0; }
    if (shmem > 48 * 1024) {
        static size_t granted = 0;  // (per process: the attribute belongs to the kernel, not to a handle)
        if (shmem > granted) {
            (void)hipFuncSetAttribute((const void*)lbps_brent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            (void)hipGetLastError();
            granted = shmem;
        }
    }
    if (h->brent_seq > 0xFFFFFFFFu - 2 * BRENT_SEQ_STRIDE) {  // tag space used up (after 8 million searches): start over on clean cells
        HIP_TRY(h, hipMemsetAsync(h->brent_cells, 0, sizeof(unsigned long long) * 2 * BRENT_LANES * BRENT_CELLS, s));
        h->brent_seq = 0;
    }
    const BrentCtx bx{h->brent_cells, h->search_error_dev, h->brent_seq, h->fused_timeout_ticks};
    h->brent_seq += BRENT_SEQ_STRIDE;
    double* host_lam = h->stats_host_dev + 8 + STATS_L * 3;
    // (test hook: with the last block missing, its lane's sums never arrive — what a block that is not resident looks like to
    // the others: every poll runs into the budget, the flag is raised and the temperature is NaN)
    hipLaunchKernelGGL(lbps_brent_kernel, dim3(grid - (h->brent_drop_block && grid > 1 ? 1 : 0)), dim3(threads), shmem, s, (const float*)h->costs, h->d.N,
                       (const unsigned*)(h->min_key + h->min_slot), nvb, per_thread, delta, lam_min, lam_max, bx, h->lambda_dev,
                       host_lam);
    HIP_TRY(h, hipGetLastError());
    h->lambda_dev_valid = true;
    return MPPI_OK;
}
// 1 once a poll of a device-resident temperature search timed out on this handle (a block of lbps_brent_kernel never became
// resident: the GPU is shared with other work): that solve's temperature — and with it its outputs — are NaN.
int mppi_search_error(mppi_handle_t h) { return (h && h->search_error) ? *(volatile int*)h->search_error : 0; }
#ifdef MPPI_BRENT_TRACE
extern "C" int mppi_debug_brent_trace(mppi_handle_t h, int* out8) {  // 10 ns ticks of the last search per phase (block 0)
    if (!h || !h->search_error) return MPPI_E_STATE;
    (void)hipDeviceSynchronize();
    for (int k = 0; k < 8; ++k) out8[k] = h->search_error[1 + k];
    return MPPI_OK;
}
#endif

// LBPS temperature (mppi.py:341-349,534-557): scipy's bounded Brent minimiser (host::fminbound, xatol 1e-5) of the
// lower-bound objective over [lam_min, lam_max]; every probe is one mppi_softmax_stats round trip (two tiny launches
// + a 40-byte read-back through mapped host memory), with no interpreter in the loop.  Unsharded handles; synchronises.
int mppi_lbps_lambda(mppi_handle_t h, double delta, double lam_min, double lam_max, double* lambda_out, void* stream) {
    if (!h || !lambda_out || !(lam_min > 0.0) || !(lam_max > lam_min) || !(delta > 0.0) || !(delta < 1.0))
        return fail(h, MPPI_E_INVALID, "bad lbps arguments");
    int rc = MPPI_OK;
    const bool ok = mppi::host::lbps_lambda(
        [&](double lam, mppi::host::SoftmaxStats& st) {
            double o[5];
            rc = mppi_softmax_stats(h, (float)lam, o, stream);
            if (rc) return false;
            st = mppi::host::SoftmaxStats{o[0], o[1], o[2], o[3], o[4]};
            return true;
        },
        delta, lam_min, lam_max, *lambda_out);
    return ok ? MPPI_OK : rc;
}

// MPO temperature (mppi.py:191-200,387-398): the dual variable log T and its Adam moments live in DEVICE memory.
int mppi_mpo_reset(mppi_handle_t h, double lambda0, double epsilon, double lr) {
    if (!h || !(lambda0 > 0.0) || !(lr > 0.0)) return fail(h, MPPI_E_INVALID, "bad mpo arguments");
    return mpo_upload(h, lambda0, epsilon, lr, true);
}
static int mpo_upload(mppi_handle_t h, double lambda0, double epsilon, double lr, bool lambda_too) {
    mppi::host::MpoState st;
    mppi::host::mpo_reset(st, lambda0, epsilon, lr);
    const float lam0 = (float)lambda0, temp0 = st.temperature();
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(h->mpo_dev, &st, sizeof(st), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->mpo_temp_dev, &temp0, sizeof(float), hipMemcpyHostToDevice));
    if (!lambda_too) return MPPI_OK;  // (mppi_create: the dual exists, but no temperature has been asked for yet)
    HIP_TRY(h, hipMemcpy(h->lambda_dev, &lam0, sizeof(float), hipMemcpyHostToDevice));  // the first solve's temperature
    h->stats_host[8 + STATS_L * 3] = h->stats_host[8 + STATS_L * 3 + 1] = lambda0;
    h->lambda_dev_valid = true;
    return MPPI_OK;
}
// One Adam step of the dual on the last solve's costs with NO host synchronisation: statistics at softplus(logT) (read
// from device memory) + a one-thread step; lambda = exp(logT) — the temperature of the NEXT solve — replaces the one in
// HBM that this solve's weights used (MPPI_LAMBDA_DEVICE).  Call it after mppi_finalize.
int mppi_mpo_step_device(mppi_handle_t h, void* stream) {
    if (!h) return MPPI_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const unsigned* mk = h->min_key + h->min_slot;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(STATS_BLOCKS, (h->d.N + BLOCK - 1) / BLOCK));
    double* host_lam = nullptr;
    host_lam = h->stats_host_dev;
    host_lam += 8 + STATS_L * 3;
    hipLaunchKernelGGL(stats_partial_kernel, dim3(blocks), dim3(BLOCK), 0, s, h->costs, h->d.N, mk, 1.0f,
                       (const float*)h->mpo_temp_dev, h->stats_part);
    hipLaunchKernelGGL(mpo_step_kernel, dim3(1), dim3(WAVE), 0, s, (const float*)h->stats_part, blocks, mk, h->mpo_dev,
                       h->lambda_dev, h->mpo_temp_dev, host_lam);
    HIP_TRY(h, hipGetLastError());
    h->lambda_dev_valid = true;
    return MPPI_OK;
}
// The same step, returning lambda_out = exp(logT) = the temperature of the NEXT solve.  Unsharded handles; synchronises.
int mppi_mpo_step(mppi_handle_t h, double* lambda_out, void* stream) {
    if (!h || !lambda_out) return fail(h, MPPI_E_INVALID, "null");
    if (int rc = mppi_mpo_step_device(h, stream)) return rc;
    return mppi_get_lambda(h, lambda_out, nullptr, stream);
}
// {log T, first moment, second moment, step count} of the dual (inspection / tests).  Synchronises the device.
int mppi_mpo_state(mppi_handle_t h, double* out4_host) {
    if (!h || !out4_host) return fail(h, MPPI_E_INVALID, "null");
    mppi::host::MpoState st;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(&st, h->mpo_dev, sizeof(st), hipMemcpyDeviceToHost));
    out4_host[0] = st.log_temperature; out4_host[1] = st.m; out4_host[2] = st.v; out4_host[3] = st.t;
    return MPPI_OK;
}

// The inverse of mppi_mpo_state (restoring a saved solver): {log T, first moment, second moment, step count} -> the dual; the
// temperature of the next solve becomes exp(log T) (mppi.py:398).  epsilon / lr keep their values.  Synchronises the device.
int mppi_mpo_set_state(mppi_handle_t h, const double* in4_host) {
    if (!h || !in4_host) return fail(h, MPPI_E_INVALID, "null");
    mppi::host::MpoState st;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(&st, h->mpo_dev, sizeof(st), hipMemcpyDeviceToHost));
    st.log_temperature = (float)in4_host[0]; st.m = (float)in4_host[1]; st.v = (float)in4_host[2]; st.t = (int32_t)in4_host[3];
    const float temp = st.temperature(), lam = (float)exp(st.log_temperature);
    HIP_TRY(h, hipMemcpy(h->mpo_dev, &st, sizeof(st), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->mpo_temp_dev, &temp, sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->lambda_dev, &lam, sizeof(float), hipMemcpyHostToDevice));
    h->stats_host[8 + STATS_L * 3] = h->stats_host[8 + STATS_L * 3 + 1] = (double)lam;
    h->lambda_dev_valid = true;
    return MPPI_OK;
}

// Device address of the dual's log T (one fp32, first field of the state mppi_mpo_step_device updates): lets a caller
// expose it without a copy (this build's MPPI makes it the storage of its `log_temperature` nn.Parameter, mppi.py:194-199).
// Read-only for the caller: the library derives the temperatures it uses when the dual steps.
int mppi_mpo_log_temperature_ptr(mppi_handle_t h, float** out_dev) {
    if (!h || !out_dev || !h->mpo_dev) return fail(h, MPPI_E_INVALID, "null");
    *out_dev = &h->mpo_dev->log_temperature;
    return MPPI_OK;
}

// get_samples_from_posterior, the sampling half (mppi.py:489-503): k unclamped action sequences
// loc + eps, eps ~ N(0, diag(sigma^2)) from the Philox stream of the RESERVED solve index `solve_idx` (callers advance
// their solve counter: the draw consumes the stream like the reference's global generator).  Roll them out with
// mppi_rollout_actions(x0_dev = the posterior's state).
int mppi_sample_posterior(mppi_handle_t h, uint32_t solve_idx, const float* loc_dev, int k, float* samples_out_dev,
                          void* stream) {
    if (!h || !loc_dev || !samples_out_dev || k < 1) return fail(h, MPPI_E_INVALID, "bad sample_posterior arguments");
    if (!h->limits_set) return fail(h, MPPI_E_STATE, "dim_control > 4: call mppi_set_control_limits first");
    const GenCtx g{h->gen.seed_lo, h->gen.seed_hi, solve_idx};
    const unsigned grid = (unsigned)(((int64_t)k * h->d.R + BLOCK - 1) / BLOCK);
    if (h->wide)
        hipLaunchKernelGGL(posterior_sample_kernel<true>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, loc_dev, k,
                           samples_out_dev, h->d, g, (const float*)h->coltab);
    else
        hipLaunchKernelGGL(posterior_sample_kernel<false>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, loc_dev, k,
                           samples_out_dev, h->d, g, (const float*)nullptr);
    HIP_TRY(h, hipGetLastError());
    return MPPI_OK;
}

int mppi_weights(mppi_handle_t h, float lambda, float cmin, float sum_e, float* w_out, void* stream) {
    if (!h || !w_out || !(lambda > 0.0f)) return fail(h, MPPI_E_INVALID, "bad weights arguments");
    const unsigned grid = (unsigned)((h->d.N + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(weights_kernel, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, h->costs, h->d.N, lambda, cmin,
                       sum_e, w_out);
    HIP_TRY(h, hipGetLastError());
    return MPPI_OK;
}

int mppi_rollout_actions(mppi_handle_t h, const float* actions_dev, int k, const float* x0_dev, float* states_out,
                         void* stream) {
    if (!h || !actions_dev || !states_out || k < 1) return fail(h, MPPI_E_INVALID, "bad rollout_actions arguments");
    if (int rc = check_ready(h)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const float* x0 = x0_dev ? x0_dev : h->x0_cur;  // an explicit start state leaves the solver's own untouched
    const unsigned grid = (unsigned)((k + WAVE - 1) / WAVE);
#define CALL_RA(MODEL, FASTV)                                                                         \
    hipLaunchKernelGGL((rollout_actions_kernel<MODEL, FASTV>), dim3(grid), dim3(WAVE), 0, s, actions_dev, k, h->d.T, \
                       x0, states_out, h->ctx)
    MPPI_DISPATCH(h, CALL_RA);
#undef CALL_RA
    HIP_TRY(h, hipGetLastError());
    return MPPI_OK;
}

int mppi_rollout_samples(mppi_handle_t h, const int64_t* idx_dev, int k, float* states_out, void* stream) {
    if (!h || !idx_dev || !states_out || k < 1) return fail(h, MPPI_E_INVALID, "bad rollout_samples arguments");
    if (int rc = check_ready(h)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = need_tiles(h, s)) return rc;
    const unsigned grid = (unsigned)((k + WAVE - 1) / WAVE);
#define CALL_RS(MODEL, FASTV)                                                                         \
    hipLaunchKernelGGL((rollout_samples_kernel<MODEL, FASTV>), dim3(grid), dim3(WAVE), 0, s, h->noise, h->mean_used, \
                       idx_dev, k, h->x0_used, states_out, h->d, h->ctx)
    MPPI_DISPATCH(h, CALL_RS);
#undef CALL_RS
    HIP_TRY(h, hipGetLastError());
    return MPPI_OK;
}

// candidate buffer for k words, padded to a power of two (the large-k sort works on 2^m words)
static int topk_reserve(mppi_handle_t h, int k) {
    size_t need = TOPK_MAX;
    while (need < (size_t)k) need <<= 1;
    if (need <= h->topk_cap) return MPPI_OK;
    HIP_TRY(h, hipDeviceSynchronize());
    (void)hipFree(h->topk_cand);
    h->topk_cand = nullptr; h->topk_cap = 0;
    HIP_TRY(h, hipMalloc(&h->topk_cand, sizeof(unsigned long long) * need));
    h->topk_cap = need;
    return MPPI_OK;
}

// radix select of this handle's k smallest costs -> h->topk_cand (unordered, global indices)
static int topk_select(mppi_handle_t h, int k, hipStream_t s) {
    if (k < 1 || k > h->d.N) return fail(h, MPPI_E_INVALID, "top samples: need 1 <= k <= num_samples");
    if (h->d.sample_offset + h->d.N >= ((int64_t)1 << 32)) return fail(h, MPPI_E_INVALID, "top samples: global sample indices must be < 2^32");
    if (int rc = topk_reserve(h, k)) return rc;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((h->d.N + BLOCK * 8 - 1) / (BLOCK * 8), 1024));
    unsigned* hist = h->topk_hist;
    unsigned* counters = h->topk_hist + 3 * TOPK_BINS;
    hipLaunchKernelGGL(topk_hist_kernel<0>, dim3(grid), dim3(BLOCK), 0, s, h->costs, h->d.N, (unsigned)k, hist, h->topk_sel);
    hipLaunchKernelGGL(topk_hist_kernel<1>, dim3(grid), dim3(BLOCK), 0, s, h->costs, h->d.N, (unsigned)k, hist, h->topk_sel);
    hipLaunchKernelGGL(topk_hist_kernel<2>, dim3(grid), dim3(BLOCK), 0, s, h->costs, h->d.N, (unsigned)k, hist, h->topk_sel);
    hipLaunchKernelGGL(topk_collect_kernel, dim3(grid), dim3(BLOCK), 0, s, h->costs, h->d.N, (unsigned)k, h->d.sample_offset,
                       hist, h->topk_sel, h->topk_cand, counters);
    HIP_TRY(h, hipGetLastError());
    return MPPI_OK;
}

// ascending sort of the k > TOPK_MAX candidate words at h->topk_cand (see topk_sort_local_kernel)
static int topk_sort_large(mppi_handle_t h, int k, hipStream_t s) {
    int P = TOPK_MAX;
    while (P < k) P <<= 1;
    if (P > k) hipLaunchKernelGGL(topk_pad_kernel, dim3((unsigned)((P - k + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, h->topk_cand, k, P);
    hipLaunchKernelGGL(topk_sort_local_kernel<true>, dim3((unsigned)(P / TOPK_MAX)), dim3(TOPK_MAX), 0, s, h->topk_cand, 0);
    for (int size = 2 * TOPK_MAX; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride >= TOPK_MAX; stride >>= 1)
            hipLaunchKernelGGL(topk_sort_global_kernel, dim3((unsigned)((P / 2 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, h->topk_cand,
                               P, size, stride);
        hipLaunchKernelGGL(topk_sort_local_kernel<false>, dim3((unsigned)(P / TOPK_MAX)), dim3(TOPK_MAX), 0, s, h->topk_cand, size);
    }
    HIP_TRY(h, hipGetLastError());
    return MPPI_OK;
}

// A launch whose dynamic LDS grows with the length of a control row (T * dim_control): beyond 32 KiB it is held against the
// kernel's own static LDS and the 64 KiB a block gets without opting in, so that a horizon too long for the staging fails
// with a message instead of a bare launch error.  Queried only in that rare case.
static int check_row_lds(mppi_handle_t h, const void* kernel, size_t dyn, const char* what) {
    if (dyn <= 32 * 1024) return MPPI_OK;
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) { (void)hipGetLastError(); return MPPI_OK; }
    if (fa.sharedSizeBytes + dyn > 64 * 1024)
        return fail(h, MPPI_E_INVALID, std::string(what) + ": control rows of T * dim_control = " + std::to_string(h->d.row) +
                    " floats need " + std::to_string((fa.sharedSizeBytes + dyn + 1023) / 1024) + " KiB of LDS per block (limit 64 KiB): "
                    "horizon x dim_control too long for this query");
    return MPPI_OK;
}

// sort k candidates, weigh and re-roll them; `clean` also resets the select state (after topk_select).  `cand` is
// h->topk_cand when k > TOPK_MAX (sorted in place).
static int topk_rollout(mppi_handle_t h, const unsigned long long* cand, int k, float lambda, float* states_out,
                        float* weights_out, bool clean, bool need_local, hipStream_t s, bool direct = false) {
    const bool gen = h->noise_regen && !h->injected;
    if (!gen && !h->tiles_valid) return fail(h, MPPI_E_STATE, "no noise: solve first");
    if (!gen && !need_local)
        return fail(h, MPPI_E_STATE, "candidates of other shards can only be re-rolled from regenerated noise (noise_regen = 1, no injection)");
    unsigned* hist = clean ? h->topk_hist : nullptr;
    unsigned* counters = clean ? h->topk_hist + 3 * TOPK_BINS : nullptr;
    if (k <= TOPK_MAX) {
#define CALL_TOPK(MODEL, FASTV)                                                                       \
    if (int rc = check_row_lds(h, (const void*)topk_rollout_kernel<MODEL, FASTV, false>, topk_rollout_lds(h->d.R, false, gen), "get_top_samples")) return rc; \
    hipLaunchKernelGGL((topk_rollout_kernel<MODEL, FASTV, false>), dim3((unsigned)((k + WAVE - 1) / WAVE)), dim3(TOPK_MAX), topk_rollout_lds(h->d.R, false, gen), s, cand, k, \
                       direct ? (const float*)h->costs : (const float*)nullptr, direct ? (int)h->d.N : 0, h->noise, gen,  \
                       h->mean_used, h->x0_used, h->solve_stats, lambda, states_out, weights_out, hist, counters,  \
                       h->d, h->gen, h->ctx)
        MPPI_DISPATCH(h, CALL_TOPK);
#undef CALL_TOPK
    } else {
        if (int rc = topk_sort_large(h, k, s)) return rc;
        const unsigned grid = (unsigned)((k + WAVE - 1) / WAVE);
#define CALL_TOPK_SORTED(MODEL, FASTV)                                                                \
    if (int rc = check_row_lds(h, (const void*)topk_rollout_kernel<MODEL, FASTV, true>, topk_rollout_lds(h->d.R, true, gen), "get_top_samples")) return rc; \
    hipLaunchKernelGGL((topk_rollout_kernel<MODEL, FASTV, true>), dim3(grid), dim3(WAVE), topk_rollout_lds(h->d.R, true, gen), s, (const unsigned long long*)h->topk_cand, \
                       k, (const float*)nullptr, 0, h->noise, gen, h->mean_used, h->x0_used, h->solve_stats, lambda, states_out,  \
                       weights_out, hist, counters, h->d, h->gen, h->ctx)
        MPPI_DISPATCH(h, CALL_TOPK_SORTED);
#undef CALL_TOPK_SORTED
    }
    HIP_TRY(h, hipGetLastError());
    return MPPI_OK;
}

int mppi_top_samples(mppi_handle_t h, int k, float lambda, float* states_out, float* weights_out, void* stream) {
    if (!h || !states_out || !weights_out || !(lambda > 0.0f || lambda == MPPI_LAMBDA_DEVICE))
        return fail(h, MPPI_E_INVALID, "bad top_samples arguments");
    if (int rc = check_ready(h)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (h->d.N <= TOPK_DIRECT_MAX && k <= TOPK_MAX) {  // small problems (the reference examples' sizes): ONE launch
        if (k < 1 || k > h->d.N) return fail(h, MPPI_E_INVALID, "top samples: need 1 <= k <= num_samples");
        if (h->d.sample_offset + h->d.N >= ((int64_t)1 << 32)) return fail(h, MPPI_E_INVALID, "top samples: global sample indices must be < 2^32");
        return topk_rollout(h, nullptr, k, lambda, states_out, weights_out, false, true, s, true);
    }
    if (int rc = topk_select(h, k, s)) return rc;
    return topk_rollout(h, h->topk_cand, k, lambda, states_out, weights_out, true, true, s);
}

int mppi_top_candidates(mppi_handle_t h, int k, uint64_t* cand_out_dev, void* stream) {
    if (!h || !cand_out_dev) return fail(h, MPPI_E_INVALID, "bad top_candidates arguments");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = topk_select(h, k, s)) return rc;
    HIP_TRY(h, hipMemcpyAsync(cand_out_dev, h->topk_cand, sizeof(uint64_t) * (size_t)k, hipMemcpyDeviceToDevice, s));
    // leave the select state clean for the next call
    HIP_TRY(h, hipMemsetAsync(h->topk_hist, 0, sizeof(unsigned) * (3 * TOPK_BINS + 2), s));
    return MPPI_OK;
}

int mppi_rollout_candidates(mppi_handle_t h, const uint64_t* cand_dev, int k, float lambda, float* states_out,
                            float* weights_out, void* stream) {
    if (!h || !cand_dev || !states_out || !weights_out || !(lambda > 0.0f || lambda == MPPI_LAMBDA_DEVICE) || k < 1)
        return fail(h, MPPI_E_INVALID, "bad rollout_candidates arguments");
    if (int rc = check_ready(h)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (k > TOPK_MAX) {  // the large-k sort works in place on the handle's padded buffer
        if (int rc = topk_reserve(h, k)) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->topk_cand, cand_dev, sizeof(uint64_t) * (size_t)k, hipMemcpyDeviceToDevice, s));
        return topk_rollout(h, h->topk_cand, k, lambda, states_out, weights_out, false, false, s);
    }
    return topk_rollout(h, reinterpret_cast<const unsigned long long*>(cand_dev), k, lambda, states_out, weights_out, false,
                        false, s);
}

// ---- in-library collective: RCCL all_gather of the shard summaries on the solve's stream (SURVEY 8e variant A)
int mppi_comm_unique_id(void* id_out128) {
    if (!id_out128) return MPPI_E_INVALID;
    if (!rccl().ok) return MPPI_E_STATE;
    static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
    return rccl().get_unique_id(reinterpret_cast<ncclUniqueId*>(id_out128)) == ncclSuccess ? MPPI_OK : MPPI_E_HIP;
}

int mppi_comm_init(mppi_handle_t h, int world, int rank, const void* id128) {
    if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return fail(h, MPPI_E_INVALID, "bad comm arguments");
    if (h->comm) return fail(h, MPPI_E_STATE, "communicator already initialised");
    if (!rccl().ok) return fail(h, MPPI_E_STATE, "librccl.so.1 not found (or incomplete): no in-library collective");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const size_t len = (size_t)(MPPI_SUMMARY_HEAD + h->d.row);
    HIP_TRY(h, hipMalloc(&h->comm_send, sizeof(float) * len));
    HIP_TRY(h, hipMalloc(&h->comm_recv, sizeof(float) * len * (size_t)world));
    RCCL_TRY(h, rccl().comm_init_rank(&h->comm, world, id, rank));  // collective: every rank of the job calls it
    h->comm_world = world; h->comm_rank = rank;
    return MPPI_OK;
}

int mppi_comm_destroy(mppi_handle_t h) {
    if (!h) return MPPI_E_INVALID;
    h->comm_enabled = false;
    if (h->comm) { HIP_TRY(h, hipDeviceSynchronize()); (void)rccl().comm_destroy(h->comm); h->comm = nullptr; }
    return MPPI_OK;
}

// One stand-alone all_gather of data_dev [4 + T*dc] (self-test; every rank calls it the same number of times):
// gathered_out_dev [world][4 + T*dc].  Synchronises.
// What RCCL itself says about the communicator of this handle: ncclCommCount / ncclCommUserRank (diagnostics: a
// multi-GPU bench line records them next to the world size the launcher claims).
int mppi_comm_info(mppi_handle_t h, int* count_out, int* rank_out) {
    if (!h || !h->comm) return fail(h, MPPI_E_STATE, "no communicator (mppi_comm_init)");
    if (!rccl().comm_count || !rccl().comm_user_rank) return fail(h, MPPI_E_STATE, "librccl lacks ncclCommCount / ncclCommUserRank");
    int c = -1, r = -1;
    RCCL_TRY(h, rccl().comm_count(h->comm, &c));
    RCCL_TRY(h, rccl().comm_user_rank(h->comm, &r));
    if (count_out) *count_out = c;
    if (rank_out) *rank_out = r;
    return MPPI_OK;
}

int mppi_comm_exchange(mppi_handle_t h, const float* data_dev, float* gathered_out_dev, void* stream) {
    if (!h || !data_dev || !gathered_out_dev) return fail(h, MPPI_E_INVALID, "bad comm arguments");
    if (!h->comm) return fail(h, MPPI_E_STATE, "comm: not initialised");
    hipStream_t s = (hipStream_t)stream;
    RCCL_TRY(h, rccl().all_gather(data_dev, gathered_out_dev, (size_t)(MPPI_SUMMARY_HEAD + h->d.row), ncclFloat, h->comm, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return MPPI_OK;
}

// ---- peer-to-peer exchange of the shard summaries (sharded solves; see P2pCtx in mppi_kernels.hpp)
int mppi_p2p_alloc(mppi_handle_t h, int world, int rank, void* ipc_handle_out64) {
    if (!h || !ipc_handle_out64 || world < 2 || world > 64 || rank < 0 || rank >= world)
        return fail(h, MPPI_E_INVALID, "bad p2p arguments");
    if (h->p2p_local) return fail(h, MPPI_E_STATE, "p2p buffer already allocated");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "ipc handle size");
    h->p2p_world = world; h->p2p_rank = rank;
    h->p2p_lenp = ((MPPI_SUMMARY_HEAD + h->d.row + 15) / 16) * 16;
    const size_t bytes = sizeof(unsigned long long) * 2 * (size_t)world * h->p2p_lenp;
    HIP_TRY(h, hipExtMallocWithFlags((void**)&h->p2p_local, bytes, hipDeviceMallocFinegrained));
    HIP_TRY(h, hipMemset(h->p2p_local, 0, bytes));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipHostMalloc((void**)&h->p2p_error, sizeof(int), hipHostMallocMapped));
    *h->p2p_error = 0;
    HIP_TRY(h, hipHostGetDevicePointer((void**)&h->p2p_error_dev, h->p2p_error, 0));
    HIP_TRY(h, hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(ipc_handle_out64), h->p2p_local));
    return MPPI_OK;
}

int mppi_p2p_connect(mppi_handle_t h, const void* ipc_handles_world_x64, const int32_t* peer_devices) {
    if (!h || !ipc_handles_world_x64 || !peer_devices) return fail(h, MPPI_E_INVALID, "bad p2p arguments");
    if (!h->p2p_local || h->p2p_connected) return fail(h, MPPI_E_STATE, "p2p: allocate first, connect once");
    // every peer GPU must be directly addressable from this one (xGMI / PCIe peer access) before any store goes out
    int ndev = 0;
    HIP_TRY(h, hipGetDeviceCount(&ndev));
    for (int w = 0; w < h->p2p_world; ++w) {
        const int pd = peer_devices[w];
        if (w == h->p2p_rank || pd == h->cfg.device) continue;
        if (pd < 0 || pd >= ndev) return fail(h, MPPI_E_STATE, "p2p: a peer's device is not visible to this process");
        int can = 0;
        HIP_TRY(h, hipDeviceCanAccessPeer(&can, h->cfg.device, pd));
        if (!can) return fail(h, MPPI_E_STATE, "p2p: no peer access to a rank's device");
    }
    std::vector<unsigned long long*> peers((size_t)h->p2p_world, nullptr);
    const hipIpcMemHandle_t* hs = reinterpret_cast<const hipIpcMemHandle_t*>(ipc_handles_world_x64);
    for (int w = 0; w < h->p2p_world; ++w) {
        if (w == h->p2p_rank) { peers[w] = h->p2p_local; continue; }
        void* pm = nullptr;
        HIP_TRY(h, hipIpcOpenMemHandle(&pm, hs[w], hipIpcMemLazyEnablePeerAccess));
        h->p2p_opened.push_back(pm);
        peers[w] = static_cast<unsigned long long*>(pm);
    }
    HIP_TRY(h, hipMalloc(&h->p2p_peers_dev, sizeof(unsigned long long*) * (size_t)h->p2p_world));
    HIP_TRY(h, hipMemcpy(h->p2p_peers_dev, peers.data(), sizeof(unsigned long long*) * (size_t)h->p2p_world, hipMemcpyHostToDevice));
    h->p2p_connected = true;
    return MPPI_OK;
}

// One exchange of `data_dev` [4 + T*dc] outside a solve (self-test; every rank must call it the same number of
// times): gathered_out_dev [world][4 + T*dc].  Returns MPPI_E_STATE when a poll timed out.  Synchronises.
int mppi_p2p_exchange(mppi_handle_t h, const float* data_dev, float* gathered_out_dev, void* stream) {
    if (!h || !data_dev || !gathered_out_dev) return fail(h, MPPI_E_INVALID, "bad p2p arguments");
    if (!h->p2p_connected) return fail(h, MPPI_E_STATE, "p2p: not connected");
    hipStream_t s = (hipStream_t)stream;
    ++h->p2p_seq;
    if (h->p2p_seq == 0) h->p2p_seq = 1;
    const int len = MPPI_SUMMARY_HEAD + h->d.row;
    hipLaunchKernelGGL(p2p_publish_kernel, dim3(1), dim3(BLOCK), 0, s, data_dev, len, p2p_ctx(h));
    hipLaunchKernelGGL(p2p_collect_kernel, dim3(1), dim3(BLOCK), 0, s, p2p_ctx(h), len, gathered_out_dev);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(s));
    if (*(volatile int*)h->p2p_error) return fail(h, MPPI_E_STATE, "p2p exchange timed out");
    return MPPI_OK;
}

// 1 once a poll of the single-launch solve timed out on this handle (read without synchronising): that solve's outputs
// are void (NaN) and the handle has returned to the multi-kernel path
#ifdef MPPI_FUSED_TRACE
extern "C" int mppi_debug_fused_trace(mppi_handle_t h, int* out10) {  // (out: 56 ints)  // 10 ns ticks since block 0 started, per phase boundary
    if (!h || !h->fused_error) return MPPI_E_STATE;
    (void)hipDeviceSynchronize();
    for (int k = 0; k < 24; ++k) { out10[k] = h->fused_error[1 + k]; h->fused_error[1 + k] = 0; }
    for (int k = 0; k < 32; ++k) out10[24 + k] = h->fused_error[32 + k];
    return MPPI_OK;
}
#endif
int mppi_fused_error(mppi_handle_t h) { return (h && h->fused_error) ? *(volatile int*)h->fused_error : 0; }

// 1 if a poll of the exchange buffer ever timed out on this handle (read without synchronising)
int mppi_p2p_error(mppi_handle_t h) { return (h && h->p2p_error) ? *(volatile int*)h->p2p_error : 0; }

int mppi_set_option(mppi_handle_t h, const char* key, int64_t value) {
    if (!h || !key) return MPPI_E_INVALID;
    const std::string k(key);
    if (k == "math" || k == "mapping") { if (int rc = settle_state_seq(h)) return rc; }  // (a pending state sequence keeps ITS solve's variant)
    if (k == "math") { h->math_fast = value < 0 ? 0 : value > 2 ? 2 : (int)value; return MPPI_OK; }
    if (k == "reduce_blocks") { h->reduce_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(value, 2048)); return MPPI_OK; }
    if (k == "timing") { h->timing = (int)value; return MPPI_OK; }
    if (k == "mapping") { h->mapping = value ? 1 : 0; return MPPI_OK; }
    if (k == "essps_cold") {  // the next ESSPS search (device chain and host loop) starts from the geometric grid
        h->essps_lo = h->essps_hi = 0.0;
        h->essps_prev_host.warm = false;
        return MPPI_OK;
    }
    if (k == "reduce_chains") { h->reduce_chains = value == 2 ? 2 : value == 4 ? 4 : 0; return MPPI_OK; }
    if (k == "fused_solve") { h->fused_mode = value < 0 ? 0 : value > 2 ? 2 : (int)value; return MPPI_OK; }
    if (k == "fused_timeout_us") {  // poll budget of the single-launch solve (default 20 000 us; 100 MHz ticks inside)
        if (value < 100 || value > 60000000) return fail(h, MPPI_E_INVALID, "fused_timeout_us: 100 us .. 60 s");
        h->fused_timeout_ticks = (long long)value * 100;
        return MPPI_OK;
    }
    if (k == "fused_rearm") {  // after a timed-out poll demoted the handle: allow the single launch again
        if (h->fused_error) *(volatile int*)h->fused_error = 0;
        return MPPI_OK;
    }
    if (k == "lazy_state_seq") {  // see mppi_join_state_seq
        if (value && !h->b1) HIP_TRY(h, hipMalloc(&h->b1, sizeof(float) * ((size_t)h->d.row + MPPI_MAX_DIM_STATE)));
        h->lazy_state = value ? 1 : 0;
        return MPPI_OK;
    }
    if (k == "lbps_search") { h->lbps_grid = value ? 1 : 0; return MPPI_OK; }  // what mppi_solve's LBPS rule runs: 0 Brent (default), 1 grids
    if (k == "search_test_drop_block") { h->brent_drop_block = value ? 1 : 0; return MPPI_OK; }
    if (k == "search_rearm") { if (h->search_error) *(volatile int*)h->search_error = 0; return MPPI_OK; }
    if (k == "essps_merge0") { h->essps_merge0 = value != 0; return MPPI_OK; }  // A/B: round 0 of the ESSPS chain as one launch
    if (k == "fold_path") { h->fold_mode = (value >= 0 && value <= 2) ? (int)value : 0; return MPPI_OK; }
    if (k == "exchange_p2p") {  // sharded solves: summaries travel through the peer-to-peer buffer, no collective
        if (value && !h->p2p_connected) return fail(h, MPPI_E_STATE, "exchange_p2p: call mppi_p2p_alloc / mppi_p2p_connect first");
        h->p2p_enabled = value != 0;
        return MPPI_OK;
    }
    if (k == "exchange_comm") {  // sharded solves: mppi_weights_reduce all_gathers the summaries itself (RCCL, same stream)
        if (value && !h->comm) return fail(h, MPPI_E_STATE, "exchange_comm: call mppi_comm_init first");
        h->comm_enabled = value != 0;
        return MPPI_OK;
    }
    if (k == "noise_regen") { h->noise_regen = value ? 1 : 0; h->tiles_valid = h->tiles_valid && h->injected; return MPPI_OK; }
    return fail(h, MPPI_E_INVALID, "unknown option " + k);
}

// mean device time [ms] and launch count of the stand-alone state-sequence kernel (mppi_join_state_seq; the rollouts that
// rode in a rollout launch are not separate kernels) since the last call (option "timing" = 1)
int mppi_get_state_seq_timing(mppi_handle_t h, float* out2) {
    if (!h || !out2) return MPPI_E_INVALID;
    out2[0] = -1.0f; out2[1] = 0.0f;
    const size_t pairs = h->ev_used[4] / 2;
    double sum = 0.0;
    for (size_t p = 0; p < pairs; ++p) {
        float ms = 0.0f;
        HIP_TRY(h, hipEventSynchronize(h->ev_pool[4][2 * p + 1]));
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev_pool[4][2 * p], h->ev_pool[4][2 * p + 1]));
        sum += ms;
    }
    if (pairs) { out2[0] = (float)(sum / (double)pairs); out2[1] = (float)pairs; }
    h->ev_used[4] = 0;
    return MPPI_OK;
}

int mppi_get_timing(mppi_handle_t h, float* out) {
    if (!h || !out) return MPPI_E_INVALID;
    for (int i = 0; i < 4; ++i) {
        out[i] = -1.0f;
        out[4 + i] = 0.0f;
        const size_t pairs = h->ev_used[i] / 2;
        double sum = 0.0;
        for (size_t p = 0; p < pairs; ++p) {
            float ms = 0.0f;
            HIP_TRY(h, hipEventSynchronize(h->ev_pool[i][2 * p + 1]));
            HIP_TRY(h, hipEventElapsedTime(&ms, h->ev_pool[i][2 * p], h->ev_pool[i][2 * p + 1]));
            sum += ms;
        }
        if (pairs) { out[i] = (float)(sum / (double)pairs); out[4 + i] = (float)pairs; }
        h->ev_used[i] = 0;
    }
    return MPPI_OK;
}

}  // extern "C"
