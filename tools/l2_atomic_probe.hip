// Microbenchmark: global float atomics at different scopes.  Agent-scope atomics execute memory-side (~21 G/s measured);
// workgroup-scope atomics are executed by the XCD's own L2 -- if fast, per-XCD private partial grids would beat bucketing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int SCOPE>
__global__ void __launch_bounds__(256) probe(float* base, uint32_t cells_per_copy, int iters, int copies) {
    uint32_t xcc = 0;
    if (copies > 1) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7; }
    float* img = base + (size_t)(copies > 1 ? xcc : 0) * cells_per_copy;
    uint32_t a = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a = a * 1664525u + 1013904223u;
            const uint32_t c = (a >> 8) % cells_per_copy;
            if (SCOPE == 0) __hip_atomic_fetch_add(img + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (SCOPE == 1) __hip_atomic_fetch_add(img + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (SCOPE == 2) __hip_atomic_fetch_add(img + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (SCOPE == 3) __hip_atomic_fetch_add((int*)img + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

__global__ void xcc_census(int* out) {
    uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) atomicAdd(out + (xcc & 7), 1);
}

int main() {
    const uint32_t cells = 614400;       // 2 bins x 640 x 480 (the working set of one time slice of the voxel grid)
    float* d; (void)hipMalloc(&d, (size_t)cells * 8 * 4); (void)hipMemset(d, 0, (size_t)cells * 8 * 4);
    int* cen; (void)hipMalloc(&cen, 32); (void)hipMemset(cen, 0, 32);
    xcc_census<<<2048, 64>>>(cen); int h[8]; (void)hipMemcpy(h, cen, 32, hipMemcpyDeviceToHost);
    printf("blocks per XCC id: %d %d %d %d %d %d %d %d\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 2048, iters = 250;
    const char* names[] = {"agent f32", "workgroup f32", "wavefront f32", "workgroup i32"};
    for (int copies : {1, 8}) {
        for (int scope = 0; scope < 4; ++scope) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                switch (scope) {
                    case 0: probe<0><<<blocks, 256>>>(d, cells, iters, copies); break;
                    case 1: probe<1><<<blocks, 256>>>(d, cells, iters, copies); break;
                    case 2: probe<2><<<blocks, 256>>>(d, cells, iters, copies); break;
                    case 3: probe<3><<<blocks, 256>>>(d, cells, iters, copies); break;
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            }
            double ops = (double)blocks * 256 * iters * 4;
            printf("copies=%d %-14s %8.3f ms  %8.1f G atomics/s\n", copies, names[scope], ms, ops / ms / 1e6);
        }
    }
    // correctness of the per-XCD private-copy idea: total over the 8 copies must equal the number of atomics issued
    (void)hipMemset(d, 0, (size_t)cells * 8 * 4);
    probe<1><<<blocks, 256>>>(d, cells, 10, 8); (void)hipDeviceSynchronize();
    float* hbuf = (float*)malloc((size_t)cells * 8 * 4); (void)hipMemcpy(hbuf, d, (size_t)cells * 8 * 4, hipMemcpyDeviceToHost);
    double tot = 0; for (size_t i = 0; i < (size_t)cells * 8; ++i) tot += hbuf[i];
    printf("workgroup-scope, 8 private copies: sum %.0f expected %.0f\n", tot, (double)blocks * 256 * 10 * 4);
    return 0;
}
