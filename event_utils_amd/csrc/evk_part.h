// Pieces shared by the one-pass partition kernels (evk_voxel3.hip; the round-2 kernels of evk_voxel2.hip keep their own
// copies behind EVK_EXPERIMENTS): an LDS-only workgroup barrier, the workgroup exclusive scan, the nearest-pixel tile key
// with the pixel inside the tile, and the layout of the self-resetting index buffer.
#pragma once
#include "evk_tiles.h"

namespace evk {

// index buffer (uint32 words, zeroed ONCE by the caller; every call leaves the counters it used zero again)
#define VP_HDR 8             // [0] t_first bits, [1] t_last bits, [2] ticket, [3] escaped records (info), [4] fixed-point range errors
#define VP_MAX_TILES 2048    // totals live at a FIXED offset so that they are zero again after every call
#define VP_TOTALS VP_HDR
#define VP_PART (VP_HDR + VP_MAX_TILES)            // part_start[T + 1]
#define VP_COUNTER(T) (VP_PART + (T) + 1)          // counters[T]   (split-tile combine)
#define VP_ITEM(T) (VP_PART + 2 * (T) + 1)         // item_tile[max_items]

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release, for which the
// compiler drains this wave's outstanding GLOBAL stores (s_waitcnt vmcnt(0)): a full store round trip at every barrier,
// and no load can be in flight across it.  Inside the partition kernels only LDS is shared between the waves.
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// vmcnt(0), expcnt / lgkmcnt untouched -- as a BUILTIN, so that the compiler's wait-count pass knows the loads are in
#define EVK_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)

template <int THREADS>
__device__ __forceinline__ uint32_t wg_excl_scan(uint32_t mine, uint32_t *tmp, uint32_t &total) {
    constexpr int NW = THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) tmp[wave] = incl;
    lds_only_barrier();
    if (wave == 0) {
        const uint32_t w = lane < NW ? tmp[lane] : 0u;
        uint32_t wi = w;
#pragma unroll
        for (int off = 1; off < NW; off <<= 1) {
            const uint32_t v = __shfl_up(wi, off, 64);
            if (lane >= off) wi += v;
        }
        if (lane < NW) tmp[32 + lane] = wi - w;
        if (lane == NW - 1) tmp[64] = wi;
    }
    lds_only_barrier();
    total = tmp[64];
    return tmp[32 + wave] + incl - mine;   // the caller puts a barrier before tmp is used again
}

// Nearest-pixel key (EVK_KEY_NEAREST of tile_key) that also returns the pixel inside the tile.  Branch-free: sixteen of
// these per thread with early returns became thirty-two divergent branches and the register allocator spilled across them.
__device__ __forceinline__ int nearest_key_local(float x, float y, const TileGrid &g, uint32_t &local) {
    int xi = (int)x, yi = (int)y;   // .long() truncation (saturating v_cvt_i32_f32; NaN -> 0, rejected below)
    xi += xi < 0 ? g.dom_w : 0;     // negative indices wrap once, as index_put_ does
    yi += yi < 0 ? g.dom_h : 0;
    const bool ok = (x == x) & (y == y) & ((uint32_t)xi < (uint32_t)g.dom_w) & ((uint32_t)yi < (uint32_t)g.dom_h);
    const int tw1 = (1 << g.tw_log2) - 1, th1 = (1 << g.th_log2) - 1;
    local = (uint32_t)(((yi & th1) << g.tw_log2) | (xi & tw1));
    const int key = (yi >> g.th_log2) * g.tiles_x + (xi >> g.tw_log2);
    return ok ? key : -1;
}

}  // namespace evk
