// Round 6, VERDICT item 5: what would ONE cooperative launch save over the two launches of a one-pass call at small N?
// A = a partition-shaped kernel (256 x 1024 threads) busy for ta microseconds, B = a tile-shaped kernel (512 x 512) busy for tb.
//   two launches, one stream        : A, kernel boundary, B
//   one cooperative launch, 256x1024: A's work, a grid barrier (agent-scope ticket + spin, what a hand-over on the partition's
//                                     own ticket would be), then every workgroup does B's work for TWO tiles (as two halves)
// Prints the average duration of a call (HIP events around 200 back-to-back calls) for both forms and several (ta, tb).
//   hipcc --offload-arch=gfx950 -O3 tools/coop_probe.hip -o tools/coop_probe.bin && tools/coop_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void busy_us(float us) {
    const uint64_t t0 = wall_clock64();           // 100 MHz
    const uint64_t dt = (uint64_t)(us * 100.0f);
    while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(2);
}
__global__ void __launch_bounds__(1024) k_a(float us, uint32_t *sink) {
    busy_us(us);
    if (threadIdx.x == 0 && us < 0) atomicAdd(sink, 1u);
}
__global__ void __launch_bounds__(512) k_b(float us, uint32_t *sink) {
    busy_us(us);
    if (threadIdx.x == 0 && us < 0) atomicAdd(sink, 1u);
}
__global__ void __launch_bounds__(1024) k_fused(float ta, float tb, uint32_t *ticket, uint32_t epoch, uint32_t *sink) {
    busy_us(ta);
    // grid barrier: every workgroup drains, takes a ticket; all spin until the count of this epoch is complete
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * gridDim.x) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    busy_us(tb);   // (the two 512-thread halves of the workgroup each run one tile: same duration as one k_b workgroup)
    if (threadIdx.x == 0 && ta < 0) atomicAdd(sink, 1u);
}

int main() {
    uint32_t *ticket, *sink;
    hipMalloc(&ticket, 4); hipMalloc(&sink, 4);
    hipMemset(ticket, 0, 4); hipMemset(sink, 0, 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    uint32_t epoch = 0;
    int coop = 0; hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0);
    printf("cooperative launch supported: %d\n", coop);
    for (int warm = 0; warm < 2000; ++warm) k_a<<<256, 1024, 0, s>>>(5.0f, sink);   // clocks up
    hipStreamSynchronize(s);
    const float cases[][2] = {{0.f, 0.f}, {4.f, 3.f}, {8.f, 5.f}, {12.f, 8.f}};
    for (auto &c : cases) {
        float ta = c[0], tb = c[1], ms2 = 0, ms1 = 0, msc = 0;
        hipEventRecord(e0, s);
        for (int i = 0; i < reps; ++i) { k_a<<<256, 1024, 0, s>>>(ta, sink); k_b<<<512, 512, 0, s>>>(tb, sink); }
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms2, e0, e1);
        hipEventRecord(e0, s);
        for (int i = 0; i < reps; ++i) { ++epoch; k_fused<<<256, 1024, 0, s>>>(ta, tb, ticket, epoch, sink); }
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
        if (coop) {
            hipEventRecord(e0, s);
            for (int i = 0; i < reps; ++i) {
                ++epoch;
                void *args[] = {&ta, &tb, &ticket, &epoch, &sink};
                hipLaunchCooperativeKernel((const void *)k_fused, dim3(256), dim3(1024), args, 0, s);
            }
            hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&msc, e0, e1);
        }
        printf("A %4.1f us + B %4.1f us: two launches %6.2f us per call | one launch (plain <<<>>>, all 256 workgroups resident) %6.2f us | "
               "hipLaunchCooperativeKernel %6.2f us\n", ta, tb, ms2 / reps * 1e3, ms1 / reps * 1e3, msc / reps * 1e3);
    }
    return 0;
}
