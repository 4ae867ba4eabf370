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
    const uint8_t* cells;  // [nx][ny] occupancy 0/1
    int32_t nx, ny;
    float cell, inv_cell;  // inv_cell = RN(1/cell) for the Markstein division
    float ox, oy;
};

struct ModelCtx {
    float P[MPPI_MAX_PARAMS];
    MapView maps[2];
    // FAST racing / nav2d: the occupancy grid(s) copied into one (nx+1) x (ny+1) grid whose extra row and
    // column hold the out-of-bounds value (racing: obstacle + lane summed, 0..2), addressed without any
    // bounds test (see occ_lookup_pad); `pad` is the grid's base minus the constant the index trick adds.
    const uint8_t* pad;
    int32_t pad_stride;    // ny + 1
    const float* ref;      // racing: [rows][8] = x, y, yaw, v, sin(yaw), cos(yaw), 0, 0
    int32_t ref_rows;
    int32_t tan_small;     // steer bounds within [-0.25, 0.25]: polynomial tan is valid
    float inv_L;           // racing: RN(1/L) for the Markstein division by the wheel base (0 = unusable)
    int32_t unit_L;        // racing: the wheel base is exactly 1 (the reference's value): the division disappears
    int32_t u_in_bounds;   // the solver's [u_min, u_max] lies inside the model's own action clamp
    int32_t wrap_safe;     // per-step heading increments are < pi: wrapped angles stay in the narrow range
};

// Host-side plan of the padded grid.  The models clamp positions to [xlo, xhi] x [ylo, yhi]; if every index
// round(p/cell + origin) reachable under that clamp lies in [0, nx] x [0, ny], the lookup needs no bounds test.
// koff = the constant (mod 2^32) that occ_lookup_pad's index carries; returns false when the plan does not apply.
inline bool pad_map_plan(const MapView& m, float xlo, float xhi, float ylo, float yhi, uint32_t& koff) {
    if (!(m.cell > 0.0f) || m.nx < 1 || m.ny < 1 || m.nx >= (1 << 20) || m.ny >= (1 << 20)) return false;
    const float ix0 = rintf(xlo / m.cell + m.ox), ix1 = rintf(xhi / m.cell + m.ox);  // IEEE division = the
    const float iy0 = rintf(ylo / m.cell + m.oy), iy1 = rintf(yhi / m.cell + m.oy);  // device's Markstein quotient
    if (!(ix0 >= 0.0f && ix1 <= (float)m.nx && iy0 >= 0.0f && iy1 <= (float)m.ny)) return false;
    const uint64_t k = 0x400000ull * (uint64_t)(m.ny + 1) + 0x4B400000ull;
    koff = (uint32_t)k;
    return (uint64_t)koff + (uint64_t)(m.nx + 1) * (uint64_t)(m.ny + 1) < (1ull << 32);
}

// torch.clamp(x, lo, hi) = min(max(x, lo), hi).  On the device this is one v_med3_f32 (identical for
// lo <= hi and non-NaN x; NaN inputs are outside the contract).
MPPI_HD float clampf(float x, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(x, lo, hi);
#else
    return fminf(fmaxf(x, lo), hi);
#endif
}

// Total cost of a trajectory = sum_t stage cost + terminal cost (mppi.py:333-334: torch.sum(costs, dim=1) + terminal).
// The order of torch.sum is not part of the reference's contract (a vectorised cascade on the CPU that depends on the
// ISA, a tree on a GPU), and softmax(-c / lambda) amplifies the last bits of c by |c| / lambda: the reference's own
// action sequence moves by 2e-5 .. 7e-5 (nav2d, goal zone at lambda = 1) when its costs are re-summed in another valid
// fp32 order (tests/golden/make_golden.py: the recorded bands).  EXACT = true adds the fp32 stage costs in DOUBLE and
// rounds once: the value every fp32 order approximates, so the distance to ANY build of the reference is that build's own
// rounding error only (the sequential fp32 sum of rounds 1-3 sat at the edge of the reference's spread: 2.7e-5 against a
// band of 2.7e-5 on nav2d).  EXACT = false is the plain sequential fp32 sum, kept for racing: its costs / lambda are
// far from that regime (an arg-min at lambda = 1, 1e4-scale obstacle costs over lambda >= 100 in the dense cases: the
// reference's own spread is <= 6e-6 and the kernel's distance to it <= 2e-6), and in its VALU-issue-bound kernel the
// conversion + half-rate add + register-pair moves are 6 of 209 instructions per two steps: 125.1 against 121.7 us
// (profiles/r04_experiments.md).
template <bool EXACT>
struct CostSum {
    float a = 0.0f;
    MPPI_HD void add(float c) { a += c; }
    MPPI_HD float total(float terminal) const { return a + terminal; }
};
template <>
struct CostSum<true> {
    double a = 0.0;
    MPPI_HD void add(float c) { a += (double)c; }
    MPPI_HD float total(float terminal) const { return (float)(a + (double)terminal); }
};
#ifndef MPPI_EXACT_SUM_RACING
#define MPPI_EXACT_SUM_RACING 0  // (A/B knob of scripts/build_variant.sh: the double accumulator in the racing kernel too)
#endif
constexpr bool exact_cost_sum(int model) { return MPPI_EXACT_SUM_RACING || model != MPPI_MODEL_RACING; }

}  // namespace mppi

// The model code is compiled twice:
//   mppi::strict — FP contraction off: every multiply and add rounds separately, in the reference's
//                  (unfused torch) operation order.  Used with FAST=false (library math).
//   mppi::fused  — FP contraction on: a*b+c may become one FMA (one rounding instead of two).  Used with
//                  FAST=true, whose sin/cos/tan already differ from the library math by <= 1.5 ulp; the
//                  fused form saves ~13 VALU instructions per racing step.  Parity tests bound both
//                  against the oracle at 1e-5.
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
#define MPPI_MODEL_NS strict
#define MPPI_MODEL_HWTRIG false
#include "mppi_models.inc"
#undef MPPI_MODEL_NS
#undef MPPI_MODEL_HWTRIG
//   mppi::fused_hw — as mppi::fused, with sin/cos of arguments the MODEL bounds — the wrapped headings of the kinematic
//                  models (racing, nav2d, goal zone: the `CHECK = false` call sites of sincos_f, argument in [-pi, pi))
//                  and the clamped pole angle / position of the cart-pole and the mountain car (sincos_b, |x| <= 4
//                  or the lane is redone) — evaluated by the hardware v_sin_f32 / v_cos_f32 (argument in revolutions):
//                  measured max abs error 2.7e-7 against 7.8e-8 of the polynomials (scripts/ubench/hw_sincos_acc.hip)
//                  — below half an ulp of any position beyond 4 m — for 3 instead of 24 instructions per step.
//                  The pendulum's free angle and the MuJoCo-style cart-pole keep the exactly reduced polynomials.
//                  Device only.
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
#define MPPI_MODEL_NS fused
#define MPPI_MODEL_HWTRIG false
#include "mppi_models.inc"
#undef MPPI_MODEL_NS
#undef MPPI_MODEL_HWTRIG
#if defined(__HIPCC__)
#define MPPI_MODEL_NS fused_hw
#define MPPI_MODEL_HWTRIG true
#include "mppi_models.inc"
#undef MPPI_MODEL_NS
#undef MPPI_MODEL_HWTRIG
#endif
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace mppi {
// The math level selects the functor: 0 = library math, unfused (the reference's operation order);
// 1 = range-checked fast paths, fused; 2 = level 1 + hardware sin/cos for the wrapped headings.
template <int MODEL, int MATH>
struct ModelSel { using type = strict::Model<MODEL, false>; };
template <int MODEL>
struct ModelSel<MODEL, 1> { using type = fused::Model<MODEL, true>; };
#if defined(__HIPCC__)
template <int MODEL>
struct ModelSel<MODEL, 2> { using type = fused_hw::Model<MODEL, true>; };
#endif
template <int MODEL, int MATH>
using ModelT = typename ModelSel<MODEL, MATH>::type;

// Models whose fast-math cost loop takes ANY finite initial state (Model::enter_any / start_in_box / cost(..., general)):
// no lane of theirs can leave a fast path's validity range, so their cost kernels carry no library-math redo.
template <class M, class = void>
struct EntryGeneral { static constexpr bool value = false; };
template <class M>
struct EntryGeneral<M, decltype((void)M::ENTRY_GENERAL)> { static constexpr bool value = M::ENTRY_GENERAL; };
}  // namespace mppi
