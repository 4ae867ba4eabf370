// mppi_models.hpp — the five shipped dynamics/cost plugins as inlineable functors.
//
// Compiled for gfx950 by hipcc (device) and, for pre-GPU debugging only, by g++ into the test-only
// harness tests/host_emul (never loaded by the product).  All arithmetic is fp32 in the
// reference's operation order; the translation unit is built with -ffp-contract=off and fusion is
// spelled out with fmaf() where it is wanted.
//
// FAST=false uses the library sinf/cosf/tanf/fmodf and IEEE division.
// FAST=true  replaces them by branch-free fast paths that are either bit-identical (angle wrap,
//            division by the map cell size) or <=1 ulp (sin/cos/tan polynomials).  Each fast path
//            has a validity range; instead of branching to the library call inside the horizon
//            loop it raises the per-lane `bad` flag, and the caller re-runs that trajectory with
//            FAST=false after the loop (never taken for the shipped models: angles are wrapped
//            every step and the steering angle is clamped).  Preconditions that are uniform over
//            the launch (tan polynomial range, Markstein reciprocal, fused map) are checked on the
//            host, which otherwise selects the FAST=false kernel.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/mppi_hip.h"

#if defined(__HIPCC__)
#define MPPI_HD __host__ __device__ __forceinline__
#else
#define MPPI_HD inline
#endif

namespace mppi {

constexpr float PI_F = 3.14159274f;      // (float)torch.pi
constexpr float TWO_PI_F = 6.28318548f;  // (float)(2*torch.pi)

struct MapView {
    const uint8_t* cells;  // [nx][ny] occupancy 0/1 (or 0..2 when fused)
    int32_t nx, ny;
    float cell, inv_cell;  // inv_cell = RN(1/cell) for the Markstein division
    float ox, oy;
};

struct ModelCtx {
    float P[MPPI_MAX_PARAMS];
    MapView maps[2];
    const uint8_t* fused;  // racing: obstacle+lane summed (0..2) when both maps share geometry
    const float* ref;      // racing: [rows][8] = x, y, yaw, v, sin(yaw), cos(yaw), 0, 0
    int32_t ref_rows;
    int32_t tan_small;     // steer bounds within [-0.25, 0.25]: polynomial tan is valid
    float inv_L;           // racing: RN(1/L) for the Markstein division by the wheel base (0 = unusable)
    int32_t u_in_bounds;   // the solver's [u_min, u_max] lies inside the model's own action clamp
    int32_t wrap_safe;     // per-step heading increments are < pi: wrapped angles stay in the narrow range
};

// torch.clamp(x, lo, hi) = min(max(x, lo), hi).  On the device this is one v_med3_f32 (identical for
// lo <= hi and non-NaN x; NaN inputs are outside the contract).
MPPI_HD float clampf(float x, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(x, lo, hi);
#else
    return fminf(fmaxf(x, lo), hi);
#endif
}

// ---------------------------------------------------------------------------------- math
// angle_normalize (src/envs/racing_env.py:20-22): ((x + pi) % (2 pi)) - pi with torch.remainder
// semantics (fmod + sign fix).  fmod is exact, so any exact evaluation of a - k*2pi with the right
// integer k reproduces it bit for bit:
//   narrow fast path (|a| < 4 pi, models that wrap their angle every step): at most one exact
//     subtraction (Sterbenz);
//   WIDE fast path (|a| < 1e5): k = trunc(a/2pi) from a reciprocal multiply, repaired when off by
//     one, and r = fma(-k, 2pi, a) (the product is exact inside the FMA and the true remainder is
//     representable, so the single rounding is exact).
template <bool FAST, bool WIDE = false, bool CHECK = true>
MPPI_HD float angle_normalize(float x, bool& bad) {
    const float a = x + PI_F;
    float r;
    if (FAST && !WIDE) {
        if (CHECK) bad = bad || !(fabsf(a) < 2.0f * TWO_PI_F);
        r = a;
        if (a >= TWO_PI_F) r = a - TWO_PI_F;
        if (a <= -TWO_PI_F) r = a + TWO_PI_F;
    } else if (FAST && WIDE) {
        bad = bad || !(fabsf(a) < 1.0e5f);
        const float sg = copysignf(1.0f, a);
        float k = truncf(a * 0.159154937f);
        const float r0 = fmaf(-k, TWO_PI_F, a);
        float adj = 0.0f;
        if (r0 * sg < 0.0f) adj = -sg;
        if (fabsf(r0) >= TWO_PI_F) adj = sg;
        k += adj;
        r = fmaf(-k, TWO_PI_F, a);
    } else {
        r = fmodf(a, TWO_PI_F);
    }
    if (r != 0.0f && r < 0.0f) r += TWO_PI_F;
    return r - PI_F;
}

// sin and cos of one argument.  Fast path: k = rint(x*2/pi), two-term Cody-Waite reduction with
// FMA (exact product), degree-7 / degree-6 minimax polynomials on [-pi/4, pi/4]
// (measured <= 0.76 / 0.80 ulp against double, tests/test_fast_math.py).  Valid for |x| <= 200.
template <bool FAST, bool CHECK = true>
MPPI_HD void sincos_f(float x, float& s, float& c, bool& bad) {
    if (FAST) {
        if (CHECK) bad = bad || !(fabsf(x) <= 200.0f);
        const float kf = rintf(x * 0.636619747f);
        float r = fmaf(-kf, 1.57079637f, x);
        r = fmaf(-kf, -4.37113883e-8f, r);
        const float z = r * r;
        float ps = fmaf(-0.00019582892f, z, 0.008332725f);
        ps = fmaf(ps, z, -0.16666664f);
        const float sr = fmaf(ps, z * r, r);
        float pc = fmaf(2.4542922e-05f, z, -0.0013888279f);
        pc = fmaf(pc, z, 0.041666664f);
        const float t = fmaf(z * z, -pc, 0.5f * z);
        const float cr = 1.0f - t;
        const int q = (int)kf;
        const float a = (q & 1) ? cr : sr;
        const float b = (q & 1) ? sr : cr;
        s = (q & 2) ? -a : a;
        c = ((q + 1) & 2) ? -b : b;
    } else {
        s = sinf(x);
        c = cosf(x);
    }
}

// tan of the (clamped) steering angle.  Fast path (host-checked precondition |x| <= 0.25): odd
// Taylor polynomial to x^13, truncation < 1e-10 relative.
template <bool FAST>
MPPI_HD float tan_f(float x) {
    if (FAST) {
        const float z = x * x;
        float p = fmaf(0.00359212803f, z, 0.00886323552f);  // 21844/6081075, 1382/155925
        p = fmaf(p, z, 0.0218694885f);                      // 62/2835
        p = fmaf(p, z, 0.0539682540f);                      // 17/315
        p = fmaf(p, z, 0.133333333f);                       // 2/15
        p = fmaf(p, z, 0.333333333f);                       // 1/3
        return fmaf(p, z * x, x);
    }
    return tanf(x);
}

// x / cell, correctly rounded.  Fast path (host-checked precondition inv_cell = RN(1/cell) and the
// significand of cell is not all ones): Markstein's sequence q0 = RN(x*y), r = x - cell*q0 (exact
// in FMA), q = RN(q0 + r*y), which returns RN(x/cell).
template <bool FAST>
MPPI_HD float div_cell(float x, const MapView& m) {
    if (FAST) {
        const float q0 = x * m.inv_cell;
        const float r = fmaf(-m.cell, q0, x);
        return fmaf(r, m.inv_cell, q0);
    }
    return x / m.cell;
}

// ObstacleMap.compute_cost / LaneMap.compute_cost (src/envs/obstacle_map_2d.py:168-200,
// src/envs/lane_map_2d.py:90-122): idx = round_half_even(x / cell + origin); out of bound -> oob,
// else cells[ix][iy].  Branch-free: the index is forced in range and the load is unconditional.
template <bool FAST>
MPPI_HD float occ_lookup(const MapView& m, const uint8_t* cells, float px, float py, float oob) {
    const float qx = rintf(div_cell<FAST>(px, m) + m.ox);
    const float qy = rintf(div_cell<FAST>(py, m) + m.oy);
    // |q| < 2^24 for any position the models can reach (they clamp to the map limits), so the
    // conversions are exact; negative indices wrap to huge unsigned values and fail the test
    const int ix = (int)qx, iy = (int)qy;
    const bool inb = ((unsigned)ix < (unsigned)m.nx) & ((unsigned)iy < (unsigned)m.ny);
    const unsigned idx = inb ? (unsigned)ix * (unsigned)m.ny + (unsigned)iy : 0u;
    const float v = (float)cells[idx];
    return inb ? v : oob;
}

// ---------------------------------------------------------------------------------- models
// step(ctx, s, u, sn, ss, bad): sn = dynamics(s, u); ss = what the reference leaves in S[:, t]
// after the call (== s except for mountaincar, whose dynamics mutates its input views).
// load_k(tab, t): the wave-uniform per-step constants of cost_func's info["t"] (racing: the
//   reference row, KROW floats per step); the rollout kernel copies the table into LDS once per
//   block and fetches row t+1 while step t computes.
// cost(ctx, k, s, u, pu, bad): cost_func(state, action, info{prev_action, t}).
template <int MODEL, bool FAST>
struct Model;

struct NoStepConst {};

template <bool FAST>
struct Model<MPPI_MODEL_PENDULUM, FAST> {  // example/pendulum.py:17-47
    using K = NoStepConst;
    static constexpr int KROW = 0;
    static MPPI_HD K load_k(const float*, int) { return K{}; }
    static MPPI_HD void check_state(const float*, bool&) {}  // every fast path checks its own range
    static constexpr int DS = 2, DC = 1;
    static MPPI_HD void step(const ModelCtx&, const float* s, const float* u, float* sn, float* ss, bool& bad, bool uc = false) {
        const float th = s[0], thdot = s[1];
        const float uu = clampf(u[0], -2.0f, 2.0f);
        float sth, cth;
        sincos_f<FAST>(th + PI_F, sth, cth, bad);
        float newthdot = thdot + (-15.0f * sth + 3.0f * uu) * 0.05f;
        const float newth = th + newthdot * 0.05f;
        newthdot = clampf(newthdot, -8.0f, 8.0f);
        ss[0] = s[0]; ss[1] = s[1];
        sn[0] = newth; sn[1] = newthdot;
    }
    static MPPI_HD float cost(const ModelCtx&, const K&, const float* s, const float*, const float*, bool& bad) {
        const float a = angle_normalize<FAST, true>(s[0], bad);  // theta is never wrapped by the dynamics
        return a * a + 0.1f * (s[1] * s[1]);
    }
};

template <bool FAST>
struct Model<MPPI_MODEL_CARTPOLE, FAST> {  // example/cartpole.py:17-81
    using K = NoStepConst;
    static constexpr int KROW = 0;
    static MPPI_HD K load_k(const float*, int) { return K{}; }
    static MPPI_HD void check_state(const float*, bool&) {}  // every fast path checks its own range
    static constexpr int DS = 4, DC = 1;
    static MPPI_HD void step(const ModelCtx&, const float* s, const float* u, float* sn, float* ss, bool& bad, bool uc = false) {
        const float x = s[0], x_dt = s[1], theta = s[2], theta_dt = s[3];
        float force = 0.0f;
        if (u[0] >= 0.0f) force = 10.0f;
        if (u[0] < 0.0f) force = -10.0f;
        float sintheta, costheta;
        sincos_f<FAST>(theta, sintheta, costheta, bad);
        const float temp = (force + 0.05f * (theta_dt * theta_dt) * sintheta) / 1.1f;
        const float thetaacc = (9.8f * sintheta - costheta * temp) /
                               (0.5f * (1.33333337f - 0.1f * (costheta * costheta) / 1.1f));
        const float xacc = temp - 0.05f * thetaacc * costheta / 1.1f;
        float newx = x + 0.02f * x_dt;
        const float newx_dt = x_dt + 0.02f * xacc;
        float newtheta = theta + 0.02f * theta_dt;
        const float newtheta_dt = theta_dt + 0.02f * thetaacc;
        newx = clampf(newx, -2.4f, 2.4f);
        newtheta = clampf(newtheta, -0.20943951f, 0.20943951f);
        ss[0] = s[0]; ss[1] = s[1]; ss[2] = s[2]; ss[3] = s[3];
        sn[0] = newx; sn[1] = newx_dt; sn[2] = newtheta; sn[3] = newtheta_dt;
    }
    static MPPI_HD float cost(const ModelCtx&, const K&, const float* s, const float*, const float*, bool& bad) {
        const float a = angle_normalize<FAST>(s[2], bad);
        return a * a + 0.1f * (s[3] * s[3]) + 0.1f * (s[0] * s[0]);
    }
};

template <bool FAST>
struct Model<MPPI_MODEL_MOUNTAINCAR, FAST> {  // example/mountaincar.py:17-55
    using K = NoStepConst;
    static constexpr int KROW = 0;
    static MPPI_HD K load_k(const float*, int) { return K{}; }
    static MPPI_HD void check_state(const float*, bool&) {}  // every fast path checks its own range
    static constexpr int DS = 2, DC = 1;
    static MPPI_HD void step(const ModelCtx&, const float* s, const float* u, float* sn, float* ss, bool& bad, bool uc = false) {
        const float position = s[0], velocity = s[1];
        const float force = clampf(u[0], -1.0f, 1.0f);
        float s3, c3;
        sincos_f<FAST>(3.0f * position, s3, c3, bad);
        const float v1 = velocity + (force * 0.0015f - 0.0025f * c3);
        const float v2 = clampf(v1, -0.07f, 0.07f);
        const float p1 = position + v2;
        const float p2 = clampf(p1, -1.2f, 0.6f);
        ss[0] = p1; ss[1] = v1;  // `velocity +=` / `position +=` act on views of S[:, t]
        sn[0] = p2; sn[1] = v2;
    }
    static MPPI_HD float cost(const ModelCtx&, const K&, const float* s, const float*, const float*, bool&) {
        const float d = 0.45f - s[0];
        return d * d;
    }
};

template <bool FAST>
struct Model<MPPI_MODEL_NAV2D, FAST> {  // src/envs/navigation_2d.py:218-279
    using K = NoStepConst;
    static constexpr int KROW = 0;
    static MPPI_HD K load_k(const float*, int) { return K{}; }
    // FAST range check of the trajectory's initial heading; inside the loop the heading is always a
    // wrapped angle plus a host-bounded increment (ctx.wrap_safe), so no per-step checks are needed
    static MPPI_HD void check_state(const float* s, bool& bad) { bad = bad || !(fabsf(s[2] + PI_F) < 2.0f * TWO_PI_F); }
    static constexpr int DS = 3, DC = 2;
    static MPPI_HD void step(const ModelCtx& c, const float* s, const float* u, float* sn, float* ss, bool& bad, bool uc = false) {
        const float* P = c.P;
        const float x = s[0], y = s[1];
        (void)uc;
        const float v = clampf(u[0], P[MPPI_NP_VMIN], P[MPPI_NP_VMAX]);
        const float omega = clampf(u[1], P[MPPI_NP_WMIN], P[MPPI_NP_WMAX]);
        // range checks: only the incoming heading can be out of the narrow wrap range (host-checked
        // ctx.wrap_safe bounds the per-step increment); the wrapped heading feeds sin/cos
        const float theta = angle_normalize<FAST, false, false>(s[2], bad);
        float sn_, cs_;
        sincos_f<FAST, false>(theta, sn_, cs_, bad);
        const float new_x = x + v * cs_ * P[MPPI_NP_DT];
        const float new_y = y + v * sn_ * P[MPPI_NP_DT];
        const float new_theta = angle_normalize<FAST, false, false>(theta + omega * P[MPPI_NP_DT], bad);
        ss[0] = s[0]; ss[1] = s[1]; ss[2] = s[2];
        sn[0] = clampf(new_x, P[MPPI_NP_XLO], P[MPPI_NP_XHI]);
        sn[1] = clampf(new_y, P[MPPI_NP_YLO], P[MPPI_NP_YHI]);
        sn[2] = new_theta;
    }
    static MPPI_HD float cost(const ModelCtx& c, const K&, const float* s, const float*, const float*, bool&) {
        const float* P = c.P;
        const float dx = s[0] - P[MPPI_NP_GX], dy = s[1] - P[MPPI_NP_GY];
        const float goal_cost = sqrtf(dx * dx + dy * dy);
        const float occ = occ_lookup<FAST>(c.maps[0], c.maps[0].cells, s[0], s[1], 1.0f);
        return goal_cost + P[MPPI_NP_QO] * occ;
    }
};

template <bool FAST>
struct Model<MPPI_MODEL_RACING, FAST> {  // src/envs/racing_env.py:327-372, example/racing.py:110-159
    static constexpr int DS = 4, DC = 2;
    struct K { float xr, yr, vr, sinp, cosp; };  // reference_path[t] with sin/cos of its yaw
    static constexpr int KROW = 8;                // floats per row of the step-constant table
    static MPPI_HD K load_k(const float* tab, int t) {  // tab = ctx.ref or its LDS copy
        const float* r = tab + 8 * t;
        return K{r[0], r[1], r[3], r[4], r[5]};
    }
    static MPPI_HD void check_state(const float* s, bool& bad) { bad = bad || !(fabsf(s[2] + PI_F) < 2.0f * TWO_PI_F); }
    static MPPI_HD void step(const ModelCtx& c, const float* s, const float* u, float* sn, float* ss, bool& bad, bool uc = false) {
        const float* P = c.P;
        const float x = s[0], y = s[1], v = s[3];
        (void)uc;
        const float accel = clampf(u[0], P[MPPI_RP_AMIN], P[MPPI_RP_AMAX]);
        const float steer = clampf(u[1], P[MPPI_RP_SMIN], P[MPPI_RP_SMAX]);
        const float theta = angle_normalize<FAST, false, false>(s[2], bad);
        float sn_, cs_;
        sincos_f<FAST, false>(theta, sn_, cs_, bad);
        const float dx = v * cs_;
        const float dy = v * sn_;
        const float vt = v * tan_f<FAST>(steer);
        float dtheta;
        if (FAST) {  // vt / L by Markstein's sequence (host-checked precondition on inv_L); exact for L = 1
            const float q0 = vt * c.inv_L;
            dtheta = fmaf(fmaf(-P[MPPI_RP_L], q0, vt), c.inv_L, q0);
        } else {
            dtheta = vt / P[MPPI_RP_L];
        }
        const float new_x = x + dx * P[MPPI_RP_DT];
        const float new_y = y + dy * P[MPPI_RP_DT];
        const float new_theta = angle_normalize<FAST, false, false>(theta + dtheta * P[MPPI_RP_DT], bad);
        const float new_v = v + accel * P[MPPI_RP_DT];
        ss[0] = s[0]; ss[1] = s[1]; ss[2] = s[2]; ss[3] = s[3];
        sn[0] = clampf(new_x, P[MPPI_RP_XLO], P[MPPI_RP_XHI]);
        sn[1] = clampf(new_y, P[MPPI_RP_YLO], P[MPPI_RP_YHI]);
        sn[2] = new_theta;
        sn[3] = clampf(new_v, -P[MPPI_RP_VMAX], P[MPPI_RP_VMAX]);
    }
    static MPPI_HD float cost(const ModelCtx& c, const K& k, const float* s, const float* u, const float* pu, bool&) {
        const float* P = c.P;
        const float sinp = k.sinp, cosp = k.cosp;
        const float ex = s[0] - k.xr, ey = s[1] - k.yr;
        const float ec = sinp * ex - cosp * ey;
        const float el = -cosp * ex - sinp * ey;
        const float path_cost = P[MPPI_RP_QC] * (ec * ec) + P[MPPI_RP_QL] * (el * el);
        const float dv = s[3] - k.vr;
        const float velocity_cost = P[MPPI_RP_QV] * (dv * dv);
        float occ;
        if (FAST) {  // host-checked precondition: both maps share geometry and are fused (0..2)
            occ = occ_lookup<FAST>(c.maps[0], c.fused, s[0], s[1], 2.0f);
        } else {
            occ = occ_lookup<FAST>(c.maps[0], c.maps[0].cells, s[0], s[1], 1.0f);
            occ += occ_lookup<FAST>(c.maps[1], c.maps[1].cells, s[0], s[1], 1.0f);
        }
        const float obstacle_cost = P[MPPI_RP_QO] * occ;
        float input_cost = P[MPPI_RP_QIN] * (u[0] * u[0] + u[1] * u[1]);
        const float d0 = u[0] - pu[0], d1 = u[1] - pu[1];
        input_cost += P[MPPI_RP_QDIN] * (d0 * d0 + d1 * d1);
        return path_cost + velocity_cost + obstacle_cost + input_cost;
    }
};

}  // namespace mppi
