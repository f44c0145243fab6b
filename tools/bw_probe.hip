// Streaming-read bandwidth probe (gfx950): how fast can 160 MB / 800 MB be read, as a function of workgroup count,
// loads in flight per lane and the load flavour (plain / nontemporal)?   hipcc --offload-arch=gfx950 -O3 bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(1024) k_read(const uint4 *__restrict__ p, size_t n16, unsigned *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if constexpr (NT) {
                const unsigned *q = reinterpret_cast<const unsigned *>(p + i + u * stride);
                v[u].x = __builtin_nontemporal_load(q), v[u].y = __builtin_nontemporal_load(q + 1);
                v[u].z = __builtin_nontemporal_load(q + 2), v[u].w = __builtin_nontemporal_load(q + 3);
            } else {
                v[u] = p[i + u * stride];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) acc += p[i].x;
    if (acc == 0x12345678u) *sink = acc;
}

// contiguous chunk per block (what the partition kernels do) instead of a grid-stride interleave
template <int UNROLL>
__global__ void __launch_bounds__(1024) k_read_chunk(const uint4 *__restrict__ p, size_t n16, unsigned *sink) {
    const size_t chunk = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n16 ? lo + chunk : n16;
    unsigned acc = 0;
    size_t i = lo + threadIdx.x;
    for (; i + (UNROLL - 1) * (size_t)blockDim.x < hi; i += UNROLL * (size_t)blockDim.x) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * (size_t)blockDim.x];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < hi; i += blockDim.x) acc += p[i].x;
    if (acc == 0x12345678u) *sink = acc;
}

template <typename F>
static double time_us(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / reps;
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atol(argv[1]) : 160;
    const size_t bytes = mb * 1000 * 1000, n16 = bytes / 16;
    uint4 *p;
    unsigned *sink;
    hipMalloc(&p, bytes), hipMalloc(&sink, 4);
    hipMemset(p, 1, bytes);
    printf("%zu MB streaming read\n", mb);
    for (int threads : {256, 512, 1024})
        for (int bpc : {1, 2, 4, 8}) {
            const int blocks = 256 * bpc;
            if (threads * bpc > 2048 * 1) {}
            double t1 = time_us([&] { k_read<1, false><<<blocks, threads>>>(p, n16, sink); }, 20);
            double t2 = time_us([&] { k_read<2, false><<<blocks, threads>>>(p, n16, sink); }, 20);
            double t4 = time_us([&] { k_read<4, false><<<blocks, threads>>>(p, n16, sink); }, 20);
            double t8 = time_us([&] { k_read<8, false><<<blocks, threads>>>(p, n16, sink); }, 20);
            double n4 = time_us([&] { k_read<4, true><<<blocks, threads>>>(p, n16, sink); }, 20);
            double c4 = time_us([&] { k_read_chunk<4><<<blocks, threads>>>(p, n16, sink); }, 20);
            printf("threads %4d blocks %4d : unroll1 %.2f  unroll2 %.2f  unroll4 %.2f  unroll8 %.2f  nt4 %.2f  chunk4 %.2f TB/s\n",
                   threads, blocks, bytes / t1 / 1e6, bytes / t2 / 1e6, bytes / t4 / 1e6, bytes / t8 / 1e6, bytes / n4 / 1e6,
                   bytes / c4 / 1e6);
        }
    return 0;
}
