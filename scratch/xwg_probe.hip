// Cross-workgroup exchange latency through L2 (round 6, VERDICT r05 item 2: would a sampler that splits one cloud over several
// workgroups pay?). W workgroups on ONE XCD (block ids = multiples of 8) exchange a 16-byte candidate per round: each writes
// its slot {round, payload} and spins until all W slots carry the round. Prints ns per round for W = 2, 4, 8, and the same
// with the workgroups on DIFFERENT XCDs (consecutive block ids).
//   hipcc --offload-arch=gfx950 -O3 -o scratch/xwg_probe scratch/xwg_probe.hip && scratch/xwg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct alignas(16) Slot { uint32_t round, a, b, c; };

// the floor: one relaxed 64-bit atomic per candidate {round : 32 | payload : 32}, no release / acquire fences (no cache
// write-back or invalidate): device-scope atomics are served at the coherence point
__global__ void exchange_relaxed(unsigned long long *slots, int w, int stride, int rounds, unsigned long long *cycles_out, uint32_t *sink) {
    if (blockIdx.x % stride != 0) return;
    const int me = blockIdx.x / stride;
    if (me >= w || threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    uint32_t acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds; ++r) {
        if (lane == 0)
            __hip_atomic_store(&slots[me * 8], ((unsigned long long)r << 32) | (uint32_t)(r * 3 + me + acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool done = lane >= w;
        uint32_t got = 0;
        while (!__all(done)) {
            if (!done) {
                const unsigned long long v = __hip_atomic_load(&slots[lane * 8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(v >> 32) >= (uint32_t)r) { done = true; got = (uint32_t)v; }
            }
        }
        acc += got;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { cycles_out[me] = t1 - t0; sink[me] = acc; }
}

__global__ void exchange(Slot *slots, int w, int stride, int rounds, unsigned long long *cycles_out, uint32_t *sink) {
    if (blockIdx.x % stride != 0) return;
    const int me = blockIdx.x / stride;
    if (me >= w || threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    uint32_t acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds; ++r) {
        if (lane == 0) {
            // payload first, then the round word (one 16-byte store would also do; two stores + release order is the safe form)
            Slot s{(uint32_t)r, (uint32_t)(r * 3 + me), acc, 7u};
            __hip_atomic_store(&slots[me].a, s.a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&slots[me].b, s.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&slots[me].round, s.round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        // lane i < w polls slot i
        bool done = lane >= w;
        uint32_t got = 0;
        while (!__all(done)) {
            if (!done) {
                const uint32_t rr = __hip_atomic_load(&slots[lane].round, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (rr >= (uint32_t)r) { done = true; got = __hip_atomic_load(&slots[lane].a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            }
        }
        acc += got;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { cycles_out[me] = t1 - t0; sink[me] = acc; }
}

int main() {
    Slot *slots; unsigned long long *cyc; uint32_t *sink;
    hipMalloc(&slots, 256 * sizeof(Slot)); hipMalloc(&cyc, 64 * 8); hipMalloc(&sink, 64 * 4);
    const int rounds = 2000;
    for (int stride : {8, 1}) {
        for (int w : {2, 4, 8}) {
            hipMemset(slots, 0, 256 * sizeof(Slot));
            hipLaunchKernelGGL(exchange, dim3(w * stride), dim3(64), 0, 0, slots, w, stride, rounds, cyc, sink);
            hipDeviceSynchronize();
            hipMemset(slots, 0, 256 * sizeof(Slot));
            hipLaunchKernelGGL(exchange, dim3(w * stride), dim3(64), 0, 0, slots, w, stride, rounds, cyc, sink);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            std::vector<unsigned long long> h(w);
            hipMemcpy(h.data(), cyc, w * 8, hipMemcpyDeviceToHost);
            // s_memrealtime ticks at 100 MHz
            printf("%s XCD, %d workgroups: %.0f ns per exchange round (all-to-all through L2%s)\n", stride == 8 ? "same" : "different", w,
                   h[0] * 10.0 / rounds, stride == 8 ? "" : " / fabric");
            unsigned long long *s64 = reinterpret_cast<unsigned long long *>(slots);
            hipMemset(slots, 0, 256 * sizeof(Slot));
            hipLaunchKernelGGL(exchange_relaxed, dim3(w * stride), dim3(64), 0, 0, s64, w, stride, rounds, cyc, sink);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            hipMemcpy(h.data(), cyc, w * 8, hipMemcpyDeviceToHost);
            printf("                                      %.0f ns with one relaxed 64-bit atomic per candidate (no fences)\n", h[0] * 10.0 / rounds);
        }
    }
    return 0;
}
