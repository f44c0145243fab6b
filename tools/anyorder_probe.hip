// Does a kernel launched with hipExtAnyOrderLaunch start while its predecessor IN THE SAME STREAM is still running
// (AQL barrier bit clear), and do a 1024-thread / 72 KB workgroup and a 512-thread / 82 KB workgroup pair up on every CU?
// (round 5: the question behind the "live" voxel tile kernel that consumes the partition's runs while it is still sorting)
//
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/anyorder_probe.bin && tools/anyorder_probe.bin
//
// A: 256 workgroups x 1024 threads, 72 KB of LDS; spins until `flag` is set or `limit_us` have passed (never hangs).
// B: 256 workgroups x 512 threads, 82 KB of LDS; records its start time, sets the flag.
// Reported: A's duration with B launched (a) plainly, (b) any-order; B's start relative to A's; how many CUs hold one of each.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

__device__ __forceinline__ uint32_t hw_where() {
    // XCC_ID (gfx94x/95x: hwreg 20) and HW_ID (hwreg 4): cu_id [11:8], sh_id [12], se_id [15:13]
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    return ((xcc & 15u) << 16) | (hw & 0xFF00u);
}

__global__ void __launch_bounds__(1024) kA(unsigned long long *t, uint32_t *where, uint32_t *flag, int limit_us) {
    extern __shared__ unsigned char lds[];
    lds[threadIdx.x] = 1;
    const unsigned long long s = wall_clock64();   // 100 MHz
    if (threadIdx.x == 0) {
        where[blockIdx.x] = hw_where();
        while (wall_clock64() - s < (unsigned long long)limit_us * 100ull) {
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            __builtin_amdgcn_s_sleep(20);
        }
        t[2 * blockIdx.x] = s, t[2 * blockIdx.x + 1] = wall_clock64();
    }
    __syncthreads();
}

__global__ void __launch_bounds__(512) kB(unsigned long long *t, uint32_t *where, uint32_t *flag, int hold_us) {
    extern __shared__ unsigned char lds[];
    lds[threadIdx.x] = 1;
    const unsigned long long s = wall_clock64();
    if (threadIdx.x == 0) {
        where[blockIdx.x] = hw_where();
        t[2 * blockIdx.x] = s;
        // stay resident for a while, so that the census sees who shares a CU with whom
        while (wall_clock64() - s < (unsigned long long)hold_us * 100ull) __builtin_amdgcn_s_sleep(20);
        if (blockIdx.x == gridDim.x - 1) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t[2 * blockIdx.x + 1] = wall_clock64();
    }
    __syncthreads();
}

int main() {
    const int NB = 256, LIMIT = 2000, HOLD = 50;
    unsigned long long *ta, *tb;
    uint32_t *wa, *wb, *flag;
    CHECK(hipMalloc(&ta, NB * 16));
        CHECK(hipMalloc(&tb, NB * 16));
    CHECK(hipMalloc(&wa, NB * 4));
        CHECK(hipMalloc(&wb, NB * 4));
        CHECK(hipMalloc(&flag, 4));
    CHECK(hipFuncSetAttribute((const void *)kA, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    CHECK(hipFuncSetAttribute((const void *)kB, hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024));
    hipStream_t s, s2;
    CHECK(hipStreamCreate(&s));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int mode = 0; mode < 6; ++mode) {   // 0: plain launch of B, 1, 2: any-order (same stream), 3-5: B on a SECOND stream
        CHECK(hipMemsetAsync(flag, 0, 4, s));
        CHECK(hipMemsetAsync(ta, 0, NB * 16, s));
        CHECK(hipMemsetAsync(tb, 0, NB * 16, s));
        int limit = LIMIT, hold = HOLD;
        void *argsA[] = {&ta, &wa, &flag, &limit};
        void *argsB[] = {&tb, &wb, &flag, &hold};
        if (mode >= 3) CHECK(hipStreamSynchronize(s));   // (the memsets are done before the second stream starts)
        if (mode >= 3) {
            // the partition's own stream first, the consumer's right behind it, no event between them
            CHECK(hipExtLaunchKernel((const void *)kA, dim3(NB), dim3(1024), argsA, 72 * 1024, s, nullptr, nullptr, 0));
            CHECK(hipExtLaunchKernel((const void *)kB, dim3(NB), dim3(512), argsB, 82 * 1024, s2, nullptr, nullptr, 0));
            CHECK(hipStreamSynchronize(s2));
            CHECK(hipStreamSynchronize(s));
        } else {
        CHECK(hipExtLaunchKernel((const void *)kA, dim3(NB), dim3(1024), argsA, 72 * 1024, s, nullptr, nullptr, 0));
        CHECK(hipExtLaunchKernel((const void *)kB, dim3(NB), dim3(512), argsB, 82 * 1024, s, nullptr, nullptr,
                                 (mode == 1 || mode == 2) ? hipExtAnyOrderLaunch : 0));
        CHECK(hipStreamSynchronize(s));
        }
        std::vector<unsigned long long> ha(2 * NB), hb(2 * NB);
        std::vector<uint32_t> pa(NB), pb(NB);
        CHECK(hipMemcpy(ha.data(), ta, NB * 16, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hb.data(), tb, NB * 16, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(pa.data(), wa, NB * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(pb.data(), wb, NB * 4, hipMemcpyDeviceToHost));
        unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
        for (int i = 0; i < NB; ++i) {
            if (ha[2 * i] < a0) a0 = ha[2 * i];
            if (ha[2 * i + 1] > a1) a1 = ha[2 * i + 1];
            if (hb[2 * i] < b0) b0 = hb[2 * i];
            if (hb[2 * i] > b1) b1 = hb[2 * i];
        }
        std::map<uint32_t, int> ca, cb;
        for (int i = 0; i < NB; ++i) ++ca[pa[i]], ++cb[pb[i]];
        int paired = 0, maxa = 0, maxb = 0;
        for (auto &kv : ca) {
            if (cb.count(kv.first)) ++paired;
            if (kv.second > maxa) maxa = kv.second;
        }
        for (auto &kv : cb)
            if (kv.second > maxb) maxb = kv.second;
        printf("mode %d (%s): A ran %.1f us; B's first / last workgroup started %.1f / %.1f us after A's first; "
               "distinct CUs A %zu (max %d per CU) B %zu (max %d); CUs holding both %d\n",
               mode, mode >= 3 ? "second stream" : (mode ? "any-order" : "plain"), (a1 - a0) / 100.0, ((double)b0 - (double)a0) / 100.0,
               ((double)b1 - (double)a0) / 100.0, ca.size(), maxa, cb.size(), maxb, paired);
    }
    printf("verdict: concurrent iff A ran far less than %d us in the any-order modes\n", LIMIT);
    return 0;
}
