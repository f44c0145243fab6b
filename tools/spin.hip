// Occupies `blocks` CUs-worth of workgroups for roughly `usec` microseconds (s_memrealtime runs at 100 MHz), each holding
// `lds_bytes` of LDS: stands in for a collective's kernel (RCCL channels) running next to the event kernels.
// Used by tools/contention_probe.py.   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/spin.hip -o tools/libspin.so
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(512) k_spin(long long ticks, int *sink, int words) {
    extern __shared__ int lds[];
    for (int i = threadIdx.x; i < words; i += 512) lds[i] = i;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) lds[(threadIdx.x * 7) % words] += 1;
    if (lds[threadIdx.x % words] == -1) *sink = 1;
}
extern "C" int spin_launch(int blocks, double usec, void *stream, int *sink, int lds_bytes) {
    if (lds_bytes < 2048) lds_bytes = 2048;
    static bool attr = false;
    if (!attr) (void)hipFuncSetAttribute((const void *)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), attr = true;
    k_spin<<<blocks, 512, lds_bytes, (hipStream_t)stream>>>((long long)(usec * 100.0), sink, lds_bytes / 4);
    return (int)hipGetLastError();
}
