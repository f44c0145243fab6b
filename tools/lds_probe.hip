// Microbenchmark: LDS atomic throughput on gfx950 (random addresses within a tile), to choose the accumulate primitive.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ idx, int iters, int cells, float* out) {
    extern __shared__ __attribute__((aligned(16))) double lds_d[]; float* lds = (float*)lds_d;
    for (int i = threadIdx.x; i < cells * 2; i += 256) lds[i] = 0.f;
    __syncthreads();
    uint32_t a = idx[blockIdx.x * 256 + threadIdx.x];
    float v = 1.0f + threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        a = a * 1664525u + 1013904223u;
        const uint32_t c = (a >> 8) % cells;
        if (MODE == 0) __hip_atomic_fetch_add(lds + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 1) __hip_atomic_fetch_add((uint32_t*)lds + c, (uint32_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) __hip_atomic_fetch_add((unsigned long long*)lds_d + c, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) lds[c] = v;                       // plain store (racy) - LDS write rate reference
        if (MODE == 4) v += lds[c];                      // plain load
        if (MODE == 5) { float o = __hip_atomic_fetch_add(lds + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); v += o * 1e-9f; }
        if (MODE == 6) __hip_atomic_fetch_add(lds_d + c, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 7) { /* no LDS op: loop overhead */ v += (float)c; }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + v;
}

int main() {
    const int blocks = 2048, iters = 2000;
    std::vector<uint32_t> h(blocks * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
    uint32_t* d; float* o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, blocks * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_write_b32", "ds_read_b32", "ds_add_rtn_f32", "ds_add_f64", "no-op loop"};
    for (int cells : {512, 2560, 4096}) {
        for (int mode = 0; mode < 8; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                size_t lds = cells * 8;
                switch (mode) {
                    case 0: probe<0><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                    case 1: probe<1><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                    case 2: probe<2><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                    case 3: probe<3><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                    case 4: probe<4><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                    case 5: probe<5><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                    case 6: probe<6><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                    case 7: probe<7><<<blocks, 256, lds>>>(d, iters, cells, o); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            double ops = (double)blocks * 256 * iters;
            printf("cells=%5d %-15s %8.3f ms  %8.1f Gop/s  (%.2f lane-ops/clk/CU @2.4GHz)\n", cells, names[mode], ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
        }
    }
    return 0;
}
