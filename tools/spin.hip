// Occupies `blocks` CUs-worth of workgroups for roughly `usec` microseconds (s_memrealtime runs at 100 MHz): stands in
// for a collective's kernel (RCCL channels) running next to the event kernels.  Used by tools/contention_probe.py.
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(512) k_spin(long long ticks, int *sink) {
    __shared__ int lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) lds[(threadIdx.x * 7) & 1023] += 1;
    if (lds[threadIdx.x] == -1) *sink = 1;
}
extern "C" int spin_launch(int blocks, double usec, void *stream, int *sink) {
    k_spin<<<blocks, 512, 0, (hipStream_t)stream>>>((long long)(usec * 100.0), sink);
    return (int)hipGetLastError();
}
