// Microbenchmark 3: RETURNING LDS integer atomics (the scatter kernel's slot allocation) vs non-returning.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(1024) probe(const uint32_t* __restrict__ idx, int iters, int cells, uint32_t* out) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < cells; i += 1024) lds[i] = 0;
    __syncthreads();
    uint32_t a = idx[blockIdx.x * 1024 + threadIdx.x], acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a = a * 1664525u + 1013904223u;
            const uint32_t c = (a >> 10) % cells;
            if (MODE == 0) __hip_atomic_fetch_add(lds + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) acc += __hip_atomic_fetch_add(lds + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) { const uint32_t r = __hip_atomic_fetch_add(lds + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); lds[cells + (r & 1023)] = r; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + acc;
}
int main() {
    const int blocks = 256, iters = 2000;
    std::vector<uint32_t> h(blocks * 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
    uint32_t *d, *o; (void)hipMalloc(&d, h.size() * 4); (void)hipMalloc(&o, blocks * 4);
    (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"ds_add_u32", "ds_add_rtn_u32 (sum)", "ds_add_rtn_u32 + dependent ds_write"};
    for (int cells : {600, 1200}) for (int mode = 0; mode < 3; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            size_t lds = (cells + 1024) * 4;
            if (mode == 0) probe<0><<<blocks, 1024, lds>>>(d, iters, cells, o);
            if (mode == 1) probe<1><<<blocks, 1024, lds>>>(d, iters, cells, o);
            if (mode == 2) probe<2><<<blocks, 1024, lds>>>(d, iters, cells, o);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        double ops = (double)blocks * 1024 * iters * 4;
        printf("cells=%5d %-38s %8.3f ms  %8.1f Gop/s  (%.2f lane-ops/clk/CU)\n", cells, names[mode], ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
    }
    return 0;
}
