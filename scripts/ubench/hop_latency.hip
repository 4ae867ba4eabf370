// hop_latency — how long one all-to-all hop through memory takes between G resident blocks on MI355X, per publish / poll
// protocol: every round each block publishes 6 tagged 8-byte cells (one 64-byte line) and wave 0 of every block polls the
// line of every block (lane l -> block l) until all carry the round's tag — the exchange step of lbps_brent_kernel
// (csrc/mppi_search.hpp) with nothing else in the round.  Prints microseconds per round.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/hop_latency.hip -o scripts/ubench/hop_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum { P_SC1 = 0, P_SYS = 1, P_ATOMIC_RMW = 2, P_FENCE = 3, P_SAME_WAVE = 4, P_NT = 5 };

template <int PROTO>
__device__ __forceinline__ void publish(unsigned long long* p, unsigned long long v) {
    if (PROTO == P_SC1 || PROTO == P_SAME_WAVE) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (PROTO == P_SYS) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (PROTO == P_ATOMIC_RMW) (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (PROTO == P_NT) __builtin_nontemporal_store(v, p);
    else { *(volatile unsigned long long*)p = v; }
}
template <int PROTO>
__device__ __forceinline__ unsigned long long peek(const unsigned long long* p) {
    if (PROTO == P_SYS) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (PROTO == P_ATOMIC_RMW) return __hip_atomic_fetch_add((unsigned long long*)p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int PROTO>
__global__ __launch_bounds__(1024) void hop(unsigned long long* cells, int rounds, unsigned base, int sleep, int* gave_up) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int G = gridDim.x;
    const long long t0 = wall_clock64();
    __shared__ int s_stop;
    if (tid == 0) s_stop = 0;
    __syncthreads();
    for (int r = 1; r <= rounds; ++r) {
        if (s_stop) return;
        const unsigned tag = base + (unsigned)r;
        unsigned long long* buf = cells + (size_t)(r & 1) * 256 * 8;
        __syncthreads();  // (the barrier the probe has before its publish)
        const int pub_wave = PROTO == P_SAME_WAVE ? 0 : 1;
        if (wid == pub_wave && lane < 6) {
            publish<PROTO>(buf + blockIdx.x * 8 + lane, ((unsigned long long)tag << 32) | (unsigned)(r * 7 + lane));
            if (PROTO == P_FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        if (wid == 0) {
            unsigned long long c[6];
            bool ok = lane >= G;
            const int lines = (G + 63) / 64;  // lane l polls blocks l, l + 64, ... (G = 256: four lines per lane)
            int line = 0;
            const unsigned long long* theirs = buf + lane * 8;
            unsigned spins = 0;
            while (!__all(ok)) {
                // (a protocol whose stores never become visible to another XCD must not hang the box: 0.2 s budget per launch)
                if ((++spins & 1023u) == 0u && wall_clock64() - t0 > 20000000ll) { *gave_up = 1; s_stop = 1; break; }
                if (!ok) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) c[j] = peek<PROTO>(theirs + j);
                    ok = true;
#pragma unroll
                    for (int j = 0; j < 6; ++j) ok = ok && (unsigned)(c[j] >> 32) == tag;
                    if (ok && ++line < lines && lane + 64 * line < G) { ok = false; theirs = buf + (lane + 64 * line) * 8; }
                }
                if (sleep) __builtin_amdgcn_s_sleep(1);
            }
        }
    }
}

template <int PROTO>
static void run(const char* name, int G, int threads, int sleep) {
    unsigned long long* cells;
    int* gave_up;
    (void)hipHostMalloc((void**)&gave_up, sizeof(int), hipHostMallocMapped);
    *gave_up = 0;
    (void)hipMalloc(&cells, 2 * 256 * 8 * 8);
    (void)hipMemset(cells, 0, 2 * 256 * 8 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int rounds = 2000;
    unsigned base = 0;
    hipLaunchKernelGGL(hop<PROTO>, dim3(G), dim3(threads), 0, 0, cells, 200, base, sleep, gave_up); base += 4096;
    (void)hipDeviceSynchronize();
    if (*(volatile int*)gave_up) { std::printf("%-44s G=%2d threads=%4d sleep=%d: GAVE UP (the stores never became visible to every block)\n", name, G, threads, sleep); std::fflush(stdout); (void)hipFree(cells); return; }
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(hop<PROTO>, dim3(G), dim3(threads), 0, 0, cells, rounds, base, sleep, gave_up); base += 4096;
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::printf("%-44s G=%2d threads=%4d sleep=%d: %.2f us per round%s\n", name, G, threads, sleep, best * 1e3f / rounds, *gave_up ? " (GAVE UP in a timed run)" : "");
    std::fflush(stdout);
    (void)hipFree(cells);
}

int main() {
    run<P_SC1>("agent-scope, 256 blocks x 256 threads, 4 lines per lane", 256, 256, 1);
    run<P_SC1>("agent-scope, 128 blocks x 512 threads, 2 lines per lane", 128, 512, 1);
    for (int G : {2, 16, 64}) {
        run<P_SC1>("agent-scope store / load (shipped)", G, 1024, 1);
        run<P_SC1>("agent-scope store / load, no sleep", G, 1024, 0);
        run<P_SC1>("agent-scope, 256-thread blocks", G, 256, 1);
        run<P_SAME_WAVE>("agent-scope, the polling wave publishes", G, 1024, 1);
        run<P_SYS>("system-scope store / load", G, 1024, 1);
        run<P_ATOMIC_RMW>("atomic exchange / fetch_add(0)", G, 1024, 1);
        run<P_FENCE>("plain store + release fence, agent load", G, 1024, 1);
        run<P_NT>("nontemporal store, agent load", G, 1024, 1);
    }
    return 0;
}
