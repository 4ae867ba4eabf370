// mppi_kernels.hpp — gfx950 kernels of the MPPI.forward() hot path.
//
// Mapping (CDNA4, wave64): one LANE per trajectory, one WAVEFRONT per tile of 64 trajectories.
// The horizon recurrence is serial in t, so the 64 lanes of a wave advance 64 independent
// trajectories in lock-step; the noise is stored lane-major (see include/mppi_hip.h) so each
// wave-level load/store is one contiguous 1 KiB segment, no LDS transpose is needed on the hot
// path, and state stays in VGPRs for the whole horizon.  LDS is used only for the block-level
// reductions and for the [N][T][dc] <-> tile layout conversions (inject/export).
#pragma once
//
// One header per stage (round 5; this file only includes them):
//   mppi_common.hpp    Dims, GenCtx, wave reductions
//   mppi_sample.hpp    step 1: the noise stream (gen_noise4), sample_kernel, posterior draws
//   mppi_rollout.hpp   steps 1b-3: rollout_cost_kernel (THE hot loop: trajectory_cost), the wavefront-per-trajectory variant
//   mppi_reduce.hpp    steps 5-6: weights_reduce_kernel
//   mppi_exchange.hpp  sharded solves: peer-to-peer buffers
//   mppi_finalize.hpp  summarize_kernel, finalize_kernel (combine, normalise, SG filter, warm start, batch-1 rollout)
//   mppi_search.hpp    step 4 on the device: statistics, ESSPS / LBPS / MPO
//   mppi_fused.hpp     the whole solve as one launch (small problems)
//   mppi_topk.hpp      queries after a solve: weights, re-rolls, get_top_samples
//   mppi_env.hpp       map lookups, calc_ref_trajectory, env.step on the device
//   mppi_layout.hpp    reference layout <-> lane-major tiles
//   mppi_maps.hpp      map construction
#include "mppi_common.hpp"
#include "mppi_sample.hpp"
#include "mppi_rollout.hpp"
#include "mppi_reduce.hpp"
#include "mppi_exchange.hpp"
#include "mppi_finalize.hpp"
#include "mppi_search.hpp"
#include "mppi_fused.hpp"
#include "mppi_topk.hpp"
#include "mppi_env.hpp"
#include "mppi_layout.hpp"
#include "mppi_maps.hpp"
