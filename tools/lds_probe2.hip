// Microbenchmark 2: LDS 64-bit atomic throughput with the loop overhead amortised (8 atomics per iteration,
// addresses from a cheap LCG), random cells in a tile -- f64 add vs u64 add (fixed-point accumulation candidate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ idx, int iters, int cells, double* out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    for (int i = threadIdx.x; i < cells; i += 256) lds[i] = 0.0;
    __syncthreads();
    uint32_t a = idx[blockIdx.x * 256 + threadIdx.x];
    const double v = 1.0 + threadIdx.x * 1e-3;
    const long long vi = (long long)(v * 4294967296.0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a = a * 1664525u + 1013904223u;
            const uint32_t c = (a >> 10) % cells;
            if (MODE == 0) __hip_atomic_fetch_add(lds + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) __hip_atomic_fetch_add((unsigned long long*)lds + c, (unsigned long long)vi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) __hip_atomic_fetch_add((unsigned int*)lds + c, (unsigned int)vi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) { /* address generation only */ lds[0] = (MODE == 3 && c == 0xffffffffu) ? 1.0 : lds[0]; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + (double)a;
}

int main() {
    const int blocks = 2048, iters = 500;
    std::vector<uint32_t> h(blocks * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
    uint32_t* d; double* o;
    (void)hipMalloc(&d, h.size() * 4); (void)hipMalloc(&o, blocks * 8);
    (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"ds_add_f64", "ds_add_u64", "ds_add_u32", "addr-gen only"};
    for (int cells : {576, 960, 2304}) {
        for (int mode = 0; mode < 4; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                size_t lds = cells * 8;
                switch (mode) {
                    case 0: probe<0><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                    case 1: probe<1><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                    case 2: probe<2><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                    case 3: probe<3><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            double ops = (double)blocks * 256 * iters * 8;
            printf("cells=%5d %-14s %8.3f ms  %8.1f Gop/s  (%.2f lane-ops/clk/CU @2.4GHz)\n", cells, names[mode], ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
        }
    }
    return 0;
}
