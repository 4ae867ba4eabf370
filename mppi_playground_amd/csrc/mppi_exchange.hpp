// mppi_exchange.hpp — The sharded solve's only exchange without a collective launch: peer-to-peer buffers over xGMI (tagged 8-byte cells).
// Part of the MPPI.forward() hot path for gfx950; see mppi_kernels.hpp for the map of the files.
#pragma once
#include "mppi_common.hpp"

namespace mppi {

// ------------------------------------------------------------------------------------------
// One-shot peer-to-peer exchange of the shard summaries (the sharded solve's only exchange) without a collective
// launch: every rank stores its summary straight into all peers' exchange buffers over xGMI and the consumer
// polls its own buffer.  Cells are 8 bytes {fp32 value, 32-bit sequence number} written with ONE store, so data and
// "ready" flag cannot be seen apart (the idea of RCCL's low-latency protocol): no fence ordering is relied on.
// Buffer of rank r (fine-grained device memory, IPC-mapped into every peer): cells[2][W][lenp]; solve `seq` uses
// parity seq & 1 — a rank can be at most one solve ahead of the slowest one, because its next finalize needs
// everybody's summary of that solve.
struct P2pCtx {
    unsigned long long* const* peers;  // [W] base of every rank's buffer as mapped here (device array)
    unsigned long long* local;         // this rank's buffer
    int* error;                        // mapped host flag: set when a poll timed out
    int world, rank, lenp;
    unsigned seq;                      // 0 = exchange off
};

__device__ __forceinline__ void p2p_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long p2p_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(BLOCK) void p2p_publish_kernel(const float* __restrict__ summary, int len, P2pCtx x) {
    const size_t slot = ((size_t)(x.seq & 1u) * x.world + x.rank) * x.lenp;
    for (int j = threadIdx.x; j < len; j += BLOCK) {
        const unsigned long long cell = ((unsigned long long)x.seq << 32) | (unsigned long long)__float_as_uint(summary[j]);
        for (int w = 0; w < x.world; ++w) p2p_store(x.peers[w] + slot + j, cell);
    }
}

// Block-wide: wait for the `len` cells of every rank of solve x.seq and unpack them to out[w * stride + j].
// Polls give up after ~20 s of wall clock (100 MHz counter) and raise *x.error; the caller's results are then void.
template <int NT>
__device__ __forceinline__ void p2p_collect(const P2pCtx& x, int len, float* __restrict__ out, int stride,
                                            int* __restrict__ s_timed_out = nullptr) {
    const long long t0 = wall_clock64();
    for (int idx = threadIdx.x; idx < x.world * len; idx += NT) {
        const int w = idx / len, j = idx - w * len;
        const unsigned long long* cellp = x.local + ((size_t)(x.seq & 1u) * x.world + w) * x.lenp + j;
        unsigned long long cell = p2p_load(cellp);
        unsigned spins = 0;
        while ((unsigned)(cell >> 32) != x.seq) {
            if ((++spins & 255u) == 0u && wall_clock64() - t0 > 2000000000ll) {
                *x.error = 1;
                if (s_timed_out) *s_timed_out = 1;  // (LDS) the block voids this solve's outputs
                break;
            }
            // back-off: a peer's summary normally lands within a few microseconds (a handful of polls); a poll that is still
            // waiting after that is waiting for a straggler or — ranks time-sharing one device in the dry runs — for the
            // peer's kernel to get the device at all, and should leave the memory system and the issue slots alone
            if (spins < 64u) __builtin_amdgcn_s_sleep(2);
            else if (spins < 1024u) __builtin_amdgcn_s_sleep(32);
            else __builtin_amdgcn_s_sleep(127);
            cell = p2p_load(cellp);
        }
        out[w * stride + j] = __uint_as_float((unsigned)cell);
    }
    __syncthreads();
}

// self-test / generic use: collect into a plain device array [W][len]
__global__ __launch_bounds__(BLOCK) void p2p_collect_kernel(P2pCtx x, int len, float* __restrict__ out) {
    p2p_collect<BLOCK>(x, len, out, len);
}

}  // namespace mppi
