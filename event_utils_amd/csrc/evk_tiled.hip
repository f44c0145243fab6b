// Tile-bucketed event kernels (the MI355X design): global float atomics top out at ~21 G/s on this chip (memory-side
// atomics, measured: tools/probe.py), i.e. 2-12 atomics per 16-byte event cap the direct kernels at 1-10 Gev/s while
// the HBM stream alone sustains ~300 Gev/s.  So:
//   1. events are bucketed ONCE by output tile (counting sort: per-block tile histogram -> scan -> scatter into
//      16-byte (x, y, t, p) records, contiguous per tile);
//   2. one workgroup per tile streams its records with 16 B/lane coalesced loads and accumulates in an LDS tile
//      (64-bit LDS atomics: ds_add_f64 for voxel grids, fixed-point ds_add_u64 for IWE windows; ds_add_f32 is ~10x
//      slower on gfx950, tools/lds_probe.hip);
//   3. the tile is flushed with plain, coalesced stores: voxel tiles are owned exclusively; IWE windows (tile + halo,
//      shifted by the flow) go to a staging area and a gather kernel sums the (<= a few) windows covering each pixel.
// No global atomics remain on the hot path (only the rare event that lands outside its block's window uses one).
#include <mutex>
#include <type_traits>

#include "evk_img.h"
#include "evk_tiles.h"

namespace evk {

// ---------------------------------------------------------------------------------------------------------
// bucketing: histogram -> scan -> scatter
// ---------------------------------------------------------------------------------------------------------
#ifndef EVK_BUCKET_THREADS
#define EVK_BUCKET_THREADS 1024
#endif
#ifndef EVK_BUCKET_BLOCKS
#define EVK_BUCKET_BLOCKS 256  // one 1024-thread workgroup per CU; table is [tile][EVK_BUCKET_BLOCKS]
#endif

// Block b owns the contiguous event range [b*chunk, (b+1)*chunk) (chunk % 4 == 0); table[b][tile] = its count.
// COMPACT records (8 bytes, see below): the bits of a record's second word
#define EVK_REC_LOCAL_MASK 0x3FFu
#define EVK_REC_P_MASK 0xFFFFF800u
// is (x, y, p) representable as a compact record?  Integer pixel coordinates inside the domain, a polarity whose low 11
// mantissa bits are zero (the rule of k_compact_records)
__device__ __forceinline__ bool rec_compactable(float x, float y, float p, const TileGrid &g) {
    const float fx = floorf(x), fy = floorf(y);
    return fx == x && fy == y && fx >= 0.0f && fy >= 0.0f && fx <= (float)(g.dom_w - 1) && fy <= (float)(g.dom_h - 1) &&
           !(__float_as_uint(p) & ~EVK_REC_P_MASK);
}

// STATS (round 6, EVK_STAGE_STATS): the pass also reads the polarity column (12 instead of 8 B/event) and delivers, into
// the last two words of the bucket index, (0) whether SOME event has no compact 8-byte record (0 = every event has one) and
// (1) the bit pattern of max |p| -- what the host needed a separate pass over the 16-byte records (evk_compact_records_f32,
// 352 us at 50 M events whether or not it succeeded) and a reduction over p with a synchronisation of its own for.
template <typename C, bool STATS>
__global__ void __launch_bounds__(EVK_BUCKET_THREADS) k_tile_hist(const C c, int64_t n, int64_t chunk, TileGrid g,
                                                                  int mode, int ntiles, uint32_t *__restrict__ table,
                                                                  uint32_t *oob, uint32_t *__restrict__ stats) {
    extern __shared__ uint32_t hist[];
    for (int i = threadIdx.x; i < ntiles; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    int64_t hi = lo + chunk;
    if (hi > n) hi = n;
    uint32_t dropped = 0;
    bool bad = false;
    const int64_t nq = (hi > lo) ? ((hi - lo) >> 2) : 0;
    auto count4 = [&](const Vec4<float> &xv, const Vec4<float> &yv, const Vec4<float> &pv) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int key = tile_key(xv.v[k], yv.v[k], g, mode);
            if (key >= 0)
                atomicAdd(&hist[key], 1u);
            else
                ++dropped;
            if constexpr (STATS) bad |= !rec_compactable(xv.v[k], yv.v[k], pv.v[k], g);
        }
    };
    int64_t q = threadIdx.x;
    Vec4<float> xa, ya, xb, yb, pa = {}, pb = {};
    // (STATS: the polarities are read only while this lane's verdict is still open -- a stream of sub-pixel coordinates
    // fails on its first events and the pass reads 8 B/event like the plain one)
    for (; q + blockDim.x < nq; q += 2 * blockDim.x) {  // 4 (6) independent 16-byte loads in flight per lane
        c.xy4(lo, q, xa, ya);
        c.xy4(lo, q + blockDim.x, xb, yb);
        if constexpr (STATS) {
            if (!bad) c.p4(lo, q, pa), c.p4(lo, q + blockDim.x, pb);
        }
        count4(xa, ya, pa);
        count4(xb, yb, pb);
    }
    for (; q < nq; q += blockDim.x) {
        c.xy4(lo, q, xa, ya);
        if constexpr (STATS) {
            if (!bad) c.p4(lo, q, pa);
        }
        count4(xa, ya, pa);
    }
    for (int64_t i = lo + (nq << 2) + threadIdx.x; i < hi; i += blockDim.x) {  // ragged tail of the last block
        const float xs = c.x1(i), ys = c.y1(i);
        const int key = tile_key(xs, ys, g, mode);
        if (key >= 0)
            atomicAdd(&hist[key], 1u);
        else
            ++dropped;
        if constexpr (STATS) bad |= !rec_compactable(xs, ys, c.p1(i), g);
    }
    if (dropped && oob) atomicAdd(oob, dropped);
    if constexpr (STATS) {
        // per BLOCK "some event has no compact record" -> stats[block]: no global atomics, nothing to zero (a memset node in
        // front of the kernel cost ~90 us on this stack); k_tile_scan_totals folds the 256 words into the index and zeroes the
        // max |p| word, which the scatter (it reads the polarities anyway) fills
        __shared__ uint32_t sbad;
        if (threadIdx.x == 0) sbad = 0u;
        __syncthreads();
        const bool anybad = __any(bad);
        if ((threadIdx.x & 63) == 0 && anybad) atomicOr(&sbad, 1u);
        __syncthreads();
        if (threadIdx.x == 0) stats[blockIdx.x] = sbad;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ntiles; i += blockDim.x) table[(int64_t)i * EVK_BUCKET_BLOCKS + blockIdx.x] = hist[i];
}

// Per tile: exclusive prefix of table[tile][.] over the EVK_BUCKET_BLOCKS blocks (in place) and the tile total.
// One wavefront per tile: lane l owns blocks 4l..4l+3 (one 16-byte load), wave-wide scan by shuffles.
__global__ void __launch_bounds__(256) k_tile_scan_blocks(uint32_t *__restrict__ table, int ntiles,
                                                          uint32_t *__restrict__ totals) {
    static_assert(EVK_BUCKET_BLOCKS % 256 == 0, "lane l owns EVK_BUCKET_BLOCKS / 64 consecutive blocks");
    constexpr int PER = EVK_BUCKET_BLOCKS / 64;
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    uint32_t *row = table + (int64_t)tile * EVK_BUCKET_BLOCKS + lane * PER;
    uint32_t c[PER];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k += 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(row + k);
        c[k] = v.x, c[k + 1] = v.y, c[k + 2] = v.z, c[k + 3] = v.w;
        sum += v.x + v.y + v.z + v.w;
    }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    uint32_t run = incl - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t v = c[k];
        c[k] = run;
        run += v;
    }
#pragma unroll
    for (int k = 0; k < PER; k += 4) *reinterpret_cast<uint4 *>(row + k) = make_uint4(c[k], c[k + 1], c[k + 2], c[k + 3]);
    if (lane == 63) totals[tile] = incl;
}

// Bucket index (uint32), written by k_tile_scan_totals, read by the tile kernels:
//   [0 .. T]        bucket_start : record offsets of the tiles (exclusive scan of the tile totals)
//   [T+1 .. 2T+1]   part_start   : work-item offsets; tile k owns items part_start[k] .. part_start[k+1]-1 =
//                                  max(1, ceil(count_k / cap)) parts -- a tile hotter than `cap` events is split over
//                                  several workgroups so that clustered (real) event data cannot serialise on one CU
//   [2T+2 .. 3T+1]  counters     : per-tile arrival counters of the split-tile combine (self-resetting)
//   [3T+2 ..]       item_tile    : tile of every work item (bucket_max_items_balanced entries)
//   [last]          scene        : 1 when the fullest tile holds more than 1.25 x the mean (a structured scene: the plan
//                                  was balanced, and the host picks the accumulators that suffer least from the
//                                  same-address LDS conflicts of such scenes), else 0
#define IDX_PART(T) ((T) + 1)
#define IDX_COUNTER(T) (2 * (T) + 2)
#define IDX_ITEM(T) (3 * (T) + 2)

__global__ void __launch_bounds__(1024) k_tile_scan_totals(const uint32_t *__restrict__ totals, int ntiles,
                                                           uint32_t cap, uint32_t budget, uint32_t *__restrict__ index,
                                                           uint32_t *__restrict__ scene,
                                                           const uint32_t *__restrict__ block_stats = nullptr) {
    __shared__ uint32_t part[1024];
    if (block_stats && threadIdx.x < 64) {   // EVK_STAGE_STATS: the histogram blocks' verdicts -> scene[1]; scene[2] = 0 for the scatter
        uint32_t bad = 0;
        for (int b = threadIdx.x; b < EVK_BUCKET_BLOCKS; b += 64) bad |= block_stats[b];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) bad |= __shfl_xor(bad, off, 64);
        if (threadIdx.x == 0) scene[1] = bad, scene[2] = 0u;
    }
    uint32_t *bucket_start = index, *part_start = index + IDX_PART(ntiles), *counters = index + IDX_COUNTER(ntiles),
             *item_tile = index + IDX_ITEM(ntiles);
    const int per = (ntiles + 1023) / 1024;
    const int i0 = threadIdx.x * per, i1 = (i0 + per < ntiles) ? i0 + per : ntiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto block_exclusive = [&](uint32_t mine) -> uint32_t {  // wave shuffle scans + one scan of the 16 wave totals
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) part[wave] = incl;
        __syncthreads();
        if (wave == 0) {
            uint32_t w = lane < 16 ? part[lane] : 0u, wi = w;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const uint32_t v = __shfl_up(wi, off, 64);
                if (lane >= off) wi += v;
            }
            if (lane < 16) part[16 + lane] = wi - w;  // exclusive prefix of the wave totals
        }
        __syncthreads();
        const uint32_t ex = part[16 + wave] + incl - mine;
        __syncthreads();
        return ex;
    };
    auto block_reduce = [&](uint32_t v, bool take_max) -> uint32_t {  // sum or max over the workgroup, to every thread
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_xor(v, off, 64);
            v = take_max ? (o > v ? o : v) : v + o;
        }
        if (lane == 0) part[wave] = v;
        __syncthreads();
        uint32_t r = part[0];
        for (int k = 1; k < 16; ++k) r = take_max ? (part[k] > r ? part[k] : r) : r + part[k];
        __syncthreads();
        return r;
    };
    uint32_t s = 0, mx = 0;
    for (int i = i0; i < i1; ++i) {
        s += totals[i];
        mx = totals[i] > mx ? totals[i] : mx;
    }
    // ---- balance (see bucket_item_budget): a non-uniform scene gets the smallest split threshold the budget allows
    const uint32_t total = block_reduce(s, false), fullest = block_reduce(mx, true);
    const uint32_t mean = total / (uint32_t)ntiles;
    const bool structured = (uint64_t)fullest * 4u > (uint64_t)mean * 5u;
    if (threadIdx.x == 0) *scene = structured ? 1u : 0u;
    if (budget && structured) {
        auto items_at = [&](uint32_t c) -> uint32_t {
            uint32_t np = 0;
            for (int i = i0; i < i1; ++i) np += totals[i] > c ? (totals[i] + c - 1) / c : 1u;
            return block_reduce(np, false);
        };
        uint32_t lo = mean / 8u * 5u;
        lo = lo > 4096u ? lo : 4096u;
        if (lo < cap) {
            if (items_at(lo) <= budget) {
                cap = lo;
            } else {
                uint32_t hi = cap;  // items_at(cap) <= ntiles + ntiles / 4 <= budget
                while (hi - lo > 64u) {
                    const uint32_t mid = lo + (hi - lo) / 2u;
                    if (items_at(mid) <= budget) hi = mid;
                    else lo = mid;
                }
                cap = hi;
            }
        }
    }
    uint32_t np = 0;
    for (int i = i0; i < i1; ++i) np += totals[i] > cap ? (totals[i] + cap - 1) / cap : 1u;
    uint32_t run = block_exclusive(s);
    for (int i = i0; i < i1; ++i) {
        bucket_start[i] = run;
        run += totals[i];
    }
    if (threadIdx.x == 1023) bucket_start[ntiles] = run;   // thread 1023 ends at the grand total
    uint32_t prun = block_exclusive(np);
    for (int i = i0; i < i1; ++i) {
        const uint32_t parts = totals[i] > cap ? (totals[i] + cap - 1) / cap : 1u;
        part_start[i] = prun;
        counters[i] = 0;
        for (uint32_t j = 0; j < parts; ++j) item_tile[prun + j] = (uint32_t)i;
        prun += parts;
    }
    if (threadIdx.x == 1023) part_start[ntiles] = prun;
}

// Scatter: LDS cursors start at bucket_start[tile] + (this block's exclusive prefix); an LDS returning atomic hands
// every event its final slot; the 16-byte record is written there.  Events keep their time order across blocks and
// (up to the interleaving of one block's waves) within a block.
template <typename C>
__global__ void __launch_bounds__(EVK_BUCKET_THREADS) k_tile_scatter(const C c, int64_t n, int64_t chunk, TileGrid g,
                                                                     int mode, int ntiles,
                                                                     const uint32_t *__restrict__ table,
                                                                     const uint32_t *__restrict__ bucket_start,
                                                                     float4 *__restrict__ rec) {
    extern __shared__ uint32_t cursor[];
    for (int i = threadIdx.x; i < ntiles; i += blockDim.x)
        cursor[i] = bucket_start[i] + table[(int64_t)i * EVK_BUCKET_BLOCKS + blockIdx.x];
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    int64_t hi = lo + chunk;
    if (hi > n) hi = n;
    const int64_t nq = (hi > lo) ? ((hi - lo) >> 2) : 0;
    for (int64_t q = threadIdx.x; q < nq; q += blockDim.x) {
        Vec4<float> xv, yv, tv, pv;
        c.xy4(lo, q, xv, yv);
        c.tp4(lo, q, tv, pv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int key = tile_key(xv.v[k], yv.v[k], g, mode);
            if (key >= 0) {
                const uint32_t pos = atomicAdd(&cursor[key], 1u);
                rec[pos] = make_float4(xv.v[k], yv.v[k], tv.v[k], pv.v[k]);
            }
        }
    }
    for (int64_t i = lo + (nq << 2) + threadIdx.x; i < hi; i += blockDim.x) {
        const float xs = c.x1(i), ys = c.y1(i);
        const int key = tile_key(xs, ys, g, mode);
        if (key >= 0) {
            const uint32_t pos = atomicAdd(&cursor[key], 1u);
            rec[pos] = make_float4(xs, ys, c.t1(i), c.p1(i));
        }
    }
}

// Scatter with software write-combining.  A scattered 16-byte store costs a 32-byte HBM sector (PMC: the plain
// scatter writes 314 MB for 160 MB of records), so every tile gets an R-slot LDS ring that holds the records of the
// sliding window of positions [vstart, vstart + R) of this block's segment of that tile; after each phase of 4096
// events the complete, G = R/2-aligned granules (G*16 bytes: 64 B for R = 8, 128 B for R = 16) are written out by
// G consecutive lanes as one contiguous, aligned piece.  A record that does not fit the window (a hot tile) is stored
// directly -- such stores are consecutive in memory anyway.
template <int R, typename C>
__global__ void __launch_bounds__(EVK_BUCKET_THREADS) k_tile_scatter_wc(const C c, int64_t n, int64_t chunk,
                                                                        TileGrid g, int mode, int ntiles,
                                                                        const uint32_t *__restrict__ table,
                                                                        const uint32_t *__restrict__ bucket_start,
                                                                        float4 *__restrict__ rec) {
    constexpr int G = R / 2;
    extern __shared__ __attribute__((aligned(16))) float4 ring[];  // [ntiles][R]
    uint32_t *cursor = reinterpret_cast<uint32_t *>(ring + (size_t)ntiles * R);
    uint32_t *vstart = cursor + ntiles;
    uint32_t *work_count = vstart + ntiles;
    unsigned short *work = reinterpret_cast<unsigned short *>(work_count + 1);  // queue of tiles to flush (ids < 8192)
    for (int i = threadIdx.x; i < ntiles; i += blockDim.x) {
        const uint32_t c = bucket_start[i] + table[(int64_t)i * EVK_BUCKET_BLOCKS + blockIdx.x];
        cursor[i] = c;
        vstart[i] = c;
    }
    if (threadIdx.x == 0) *work_count = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    int64_t hi = lo + chunk;
    if (hi > n) hi = n;
    const int64_t nq = (hi > lo) ? ((hi - lo + 3) >> 2) : 0;  // quads, the last one may be ragged
    const int64_t nphase = (nq + blockDim.x - 1) / blockDim.x;
    auto place = [&](float xv, float yv, float tv, float pv) {
        const int key = tile_key(xv, yv, g, mode);
        if (key < 0) return;
        const uint32_t pos = atomicAdd(&cursor[key], 1u);
        const float4 r = make_float4(xv, yv, tv, pv);
        if (pos - vstart[key] < (uint32_t)R)
            ring[(size_t)key * R + (pos & (R - 1))] = r;
        else
            rec[pos] = r;
    };
    // software pipeline: the 4 column loads of phase ph+1 are issued before phase ph is placed and flushed, so HBM
    // latency overlaps the LDS work and the two barriers of the flush
    Vec4<float> xn, yn, tn, pn;
    auto fetch = [&](int64_t ph) {
        const int64_t q = ph * blockDim.x + threadIdx.x;
        if (ph < nphase && q < nq && lo + (q << 2) + 4 <= hi) {
            c.xy4(lo, q, xn, yn);
            c.tp4(lo, q, tn, pn);
        }
    };
    fetch(0);
    for (int64_t ph = 0; ph <= nphase; ++ph) {
        const int64_t q = ph * blockDim.x + threadIdx.x;
        const Vec4<float> xv = xn, yv = yn, tv = tn, pv = pn;
        fetch(ph + 1);
        if (ph < nphase && q < nq) {
            const int64_t base = lo + (q << 2);
            if (base + 4 <= hi) {
#pragma unroll
                for (int k = 0; k < 4; ++k) place(xv.v[k], yv.v[k], tv.v[k], pv.v[k]);
            } else {
                for (int64_t i = base; i < hi; ++i) place(c.x1(i), c.y1(i), c.t1(i), c.p1(i));
            }
        }
        __syncthreads();
        // Flush in two steps (the kernel is instruction-bound: scanning all tiles with R lanes each cost more
        // wave-instructions than placing the events).  Step 1: one LANE per tile decides whether the tile has a
        // complete aligned granule (or, on the last pass, anything left) and queues it.  Step 2: R consecutive lanes
        // per queued tile copy its records out as one contiguous piece.
        const bool last = (ph == nphase);
        auto flush_end = [&](uint32_t vs, uint32_t c) -> uint32_t {  // flush positions [vs, end)
            if (c - vs >= (uint32_t)R) return vs + R;               // window full (overflow went direct): drain it
            if (last) return c;
            return (c / G) * G;                                     // complete aligned granules only
        };
        for (int k = threadIdx.x; k < ntiles; k += blockDim.x) {
            const uint32_t vs = vstart[k];
            if (flush_end(vs, cursor[k]) > vs) work[atomicAdd(work_count, 1u)] = (unsigned short)k;
        }
        __syncthreads();
        const uint32_t nwork = *work_count;
        const int sub = threadIdx.x & (R - 1);
        for (uint32_t w = threadIdx.x / R; w < nwork; w += blockDim.x / R) {
            const int k = work[w];
            const uint32_t vs = vstart[k], c = cursor[k], end = flush_end(vs, c);
            const uint32_t pos = vs + sub;
            if (pos < end) rec[pos] = ring[(size_t)k * R + (pos & (R - 1))];
            if (sub == 0) vstart[k] = (c - vs >= (uint32_t)R) ? c : end;
        }
        __syncthreads();
        if (threadIdx.x == 0) *work_count = 0;  // visible to step 1 of the next phase through its append barrier
    }
}

// Scatter by LDS sort (round 6; the default where its LDS fits).  The write-combining scatter above is INSTRUCTION-bound:
// every phase of 4096 events scans all tiles for complete granules and runs three workgroup barriers -- 634 us for the
// 1.6 GB it moves at 50 M events (2.5 TB/s), with the histogram and the scans 750 us of a cold optimize() (1.1 ms with the
// separate compaction pass).  Here every workgroup sorts SUB-CHUNKS of 1024 x EPT consecutive events by tile entirely in LDS
// -- histogram, scan, a returning atomic per event on its tile's cursor: the passes of k_part_sorted (evk_part2.h) -- and writes
// each sorted sub-chunk out with consecutive lanes on consecutive records: a tile's ~9 records (142 bytes) leave as one
// contiguous piece to   bucket_start[tile] + (this block's prefix) + (what its earlier sub-chunks put there),   so the
// records end up exactly where the other scatters put them (tile-contiguous, time order preserved), and the piece that
// follows it in memory is the same tile's piece of the NEXT sub-chunk of the same workgroup, ~10 us later.
// FMT 1 (EVK_STAGE_COMPACT, when the histogram pass found every event compactable): the sorted buffer holds, and the
// kernel writes, the 8-byte COMPACT records directly -- {t, polarity bits [31:11] | pixel in tile [9:0]}, exactly
// evk_compact_records_f32's -- 16 B/event read + 8 written instead of 16 + 16 and then 16 + 8 again.
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define SCATTER_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)

template <int EPT, int FMT, typename C>
__device__ __forceinline__ void scatter_sorted_body(const C &c, int64_t n, int64_t chunk, const TileGrid &g, int mode, int ntiles,
                                                    const uint32_t *__restrict__ table,
                                                    const uint32_t *__restrict__ bucket_start, void *__restrict__ rec_,
                                                    unsigned char *smem, uint32_t *pmax_out) {
    constexpr int T = EVK_BUCKET_THREADS, S = T * EPT, NQ = EPT / 4, NWV = T / 64;
    static_assert(EPT % 4 == 0, "whole quads of events per thread");
    typedef typename std::conditional<FMT == 1, uint2, float4>::type Rec;
    Rec *sorted = reinterpret_cast<Rec *>(smem);                                         // [S] the sorted sub-chunk
    unsigned short *tileof = reinterpret_cast<unsigned short *>(smem + (size_t)S * sizeof(Rec));   // [S] its records' tiles
    const int npad = (ntiles + 3) & ~3;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + (size_t)S * (sizeof(Rec) + 2));  // [ntiles] counts of the pass (zero between passes)
    uint32_t *cur = cnt + npad;                                                          // [ntiles] cursors: start -> end of the tile's piece in `sorted`
    uint32_t *gend = cur + npad;                                                         // [ntiles] where the tile's piece ENDS in the record array
    uint32_t *tmp = gend + npad;                                                         // [NWV + 1] scan scratch
    Rec *rec = static_cast<Rec *>(rec_);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < ntiles; i += T) {
        cnt[i] = 0;
        gend[i] = bucket_start[i] + table[(int64_t)i * EVK_BUCKET_BLOCKS + blockIdx.x];
    }
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    int64_t hi = lo + chunk;
    if (hi > n) hi = n;
    const int64_t nq = (hi > lo) ? ((hi - lo + 3) >> 2) : 0;   // quads, the last one may be ragged
    const int64_t npass = (nq + (int64_t)T * NQ - 1) / ((int64_t)T * NQ);
    Vec4<float> xv[NQ], yv[NQ], tv[NQ], pv[NQ], xn[NQ], yn[NQ], tn[NQ], pn[NQ];
    auto fetch = [&](int64_t pass, Vec4<float>(&X)[NQ], Vec4<float>(&Y)[NQ], Vec4<float>(&Tt)[NQ], Vec4<float>(&P)[NQ]) {
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int64_t q = (pass * NQ + k) * T + tid;
            if (pass < npass && q < nq) {
                const int64_t base = lo + (q << 2);
                if (base + 4 <= hi) {
                    c.xy4(lo, q, X[k], Y[k]);
                    c.tp4(lo, q, Tt[k], P[k]);
                } else {   // the stream's ragged last quad
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int64_t i = base + e < hi ? base + e : hi - 1;
                        X[k].v[e] = c.x1(i), Y[k].v[e] = c.y1(i), Tt[k].v[e] = c.t1(i), P[k].v[e] = c.p1(i);
                    }
                }
            }
        }
    };
    auto valid = [&](int64_t pass, int k) -> int {   // events of quad k of this thread that exist (0 .. 4)
        const int64_t q = (pass * NQ + k) * T + tid;
        if (q >= nq) return 0;
        const int64_t left = hi - (lo + (q << 2));
        return left >= 4 ? 4 : (int)left;
    };
    uint32_t pmax = 0;
    fetch(0, xv, yv, tv, pv);
    __syncthreads();
    for (int64_t pass = 0; pass < npass; ++pass) {
        // ---- keys, histogram
        int key[EPT];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int nv = valid(pass, k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = e < nv ? tile_key(xv[k].v[e], yv[k].v[e], g, mode) : -1;
                key[4 * k + e] = kk;
                if (kk >= 0) __hip_atomic_fetch_add(&cnt[kk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        lds_only_barrier();   // histogram complete (and: every wave has written the previous sub-chunk out)
        fetch(pass + 1, xn, yn, tn, pn);   // in flight during the scan and the placement
        // ---- exclusive scan of the counts -> cursors; the global end of every tile's piece
        uint32_t kept = 0;
        for (int base = 0; base < ntiles; base += T) {
            const int i = base + tid;
            const uint32_t cn = i < ntiles ? cnt[i] : 0u;
            uint32_t incl = cn;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            if (lane == 63) tmp[wave] = incl;
            lds_only_barrier();
            const uint32_t wt = lane < NWV ? tmp[lane] : 0u;
            uint32_t wi = wt;
#pragma unroll
            for (int off = 1; off < NWV; off <<= 1) {
                const uint32_t v = __shfl_up(wi, off, 64);
                if (lane >= off) wi += v;
            }
            const uint32_t start = kept + __shfl(wi - wt, wave, 64) + incl - cn;
            if (i < ntiles) {
                cur[i] = start;
                gend[i] += cn;
                cnt[i] = 0;
            }
            kept += __shfl(wi, NWV - 1, 64);
            if (base + T < ntiles) lds_only_barrier();   // tmp is reused by the next round
        }
        lds_only_barrier();   // cursors complete
        // ---- placement: a returning LDS atomic on the tile's cursor hands every event its slot
        uint32_t slot[EPT];
#pragma unroll
        for (int s2 = 0; s2 < EPT; ++s2) {
            slot[s2] = 0;
            if (key[s2] >= 0) {
                slot[s2] = atomicAdd(&cur[key[s2]], 1u);
                const uint32_t a = __float_as_uint(pv[s2 / 4].v[s2 % 4]) & 0x7FFFFFFFu;   // max |p| of the bucketed events
                pmax = a > pmax ? a : pmax;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < EPT; ++s2) {
            if (key[s2] >= 0) {
                const float x = xv[s2 / 4].v[s2 % 4], y = yv[s2 / 4].v[s2 % 4], t = tv[s2 / 4].v[s2 % 4], p = pv[s2 / 4].v[s2 % 4];
                if constexpr (FMT == 1) {
                    const int xi = (int)x, yi = (int)y;   // (integers inside the domain: the histogram pass's verdict)
                    const uint32_t local = (uint32_t)(((yi & ((1 << g.th_log2) - 1)) << g.tw_log2) | (xi & ((1 << g.tw_log2) - 1)));
                    sorted[slot[s2]] = make_uint2(__float_as_uint(t), (__float_as_uint(p) & EVK_REC_P_MASK) | (local & EVK_REC_LOCAL_MASK));
                } else {
                    sorted[slot[s2]] = make_float4(x, y, t, p);
                }
                tileof[slot[s2]] = (unsigned short)key[s2];
            }
        }
        lds_only_barrier();   // the sorted sub-chunk is complete
        SCATTER_WAIT_VM0();   // the next pass's columns have landed: the stores below never sit between a load and its use
        // ---- write-out: record i of the sorted sub-chunk -> gend[tile] - (cur[tile] - i)   (cur is the piece's END now)
#ifndef SCATTER_ABL_LINEAR
#define SCATTER_ABL_LINEAR 0   // (timing builds, results wrong) 1: every sub-chunk leaves as ONE contiguous run (what would streaming writes cost?)
#endif
        for (uint32_t i = tid; i < kept; i += T) {
            const uint32_t tl = tileof[i];
            if (SCATTER_ABL_LINEAR) rec[(lo + pass * S + i) % (uint64_t)(n > 0 ? n : 1)] = sorted[i];
            else rec[gend[tl] - (cur[tl] - i)] = sorted[i];
        }
#pragma unroll
        for (int k = 0; k < NQ; ++k) xv[k] = xn[k], yv[k] = yn[k], tv[k] = tn[k], pv[k] = pn[k];
    }
    if (pmax_out) {   // ONE global atomic per workgroup (the word was zeroed by k_tile_scan_totals; 4096 same-address atomics --
                      // one per wave -- cost 30 us at the end of the kernel)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint32_t o = __shfl_xor(pmax, off, 64);
            pmax = o > pmax ? o : pmax;
        }
        lds_only_barrier();              // (tmp is free: every wave is past its last scan)
        if (tid == 0) tmp[0] = 0u;
        lds_only_barrier();
        if (lane == 0 && pmax) atomicMax(&tmp[0], pmax);
        lds_only_barrier();
        if (tid == 0 && tmp[0]) atomicMax(pmax_out, tmp[0]);
    }
}

template <int EPT, typename C>
__global__ void __launch_bounds__(EVK_BUCKET_THREADS) k_tile_scatter_sorted(const C c, int64_t n, int64_t chunk, TileGrid g, int mode,
                                                                            int ntiles, const uint32_t *__restrict__ table,
                                                                            const uint32_t *__restrict__ bucket_start,
                                                                            void *__restrict__ rec,
                                                                            uint32_t *__restrict__ stats, int try_compact) {
    extern __shared__ __attribute__((aligned(16))) unsigned char scatter_smem[];
    // (uniform over the launch: the histogram pass and the scans ended before this kernel began)
    if (try_compact && stats && stats[0] == 0u)
        scatter_sorted_body<EPT, 1>(c, n, chunk, g, mode, ntiles, table, bucket_start, rec, scatter_smem, stats ? stats + 1 : nullptr);
    else
        scatter_sorted_body<EPT, 0>(c, n, chunk, g, mode, ntiles, table, bucket_start, rec, scatter_smem, stats ? stats + 1 : nullptr);
}
static size_t scatter_sorted_lds(int ept, int ntiles) {   // (sized for the 16-byte format: one launch serves either)
    return (size_t)EVK_BUCKET_THREADS * ept * (16 + 2) + 3 * (size_t)((ntiles + 3) & ~3) * 4 + (EVK_BUCKET_THREADS / 64 + 1) * 4;
}

// ---------------------------------------------------------------------------------------------------------
// voxel grid: one workgroup per tile, LDS accumulators (B x th x tw), exclusive plain-store flush
// ---------------------------------------------------------------------------------------------------------
// Streams records [lo, hi) through f with 4 independent 16-byte loads in flight per lane.  (A software-pipelined
// variant -- next step's loads issued before this step's atomics -- measured no faster: the kernels are LDS-atomic or
// HBM bound with 28-32 resident waves per CU already overlapping each other.)
#ifndef EVK_STREAM_DEPTH
#define EVK_STREAM_DEPTH 4
#endif
template <typename F>
__device__ __forceinline__ void stream_records(const float4 *__restrict__ rec, uint32_t lo, uint32_t hi, F f) {
    uint32_t i = lo + threadIdx.x;
#if EVK_STREAM_DEPTH >= 8
    for (; i + 7 * EVK_BLOCK < hi; i += 8 * EVK_BLOCK) {
        const float4 r0 = rec[i], r1 = rec[i + EVK_BLOCK], r2 = rec[i + 2 * EVK_BLOCK], r3 = rec[i + 3 * EVK_BLOCK];
        const float4 r4 = rec[i + 4 * EVK_BLOCK], r5 = rec[i + 5 * EVK_BLOCK], r6 = rec[i + 6 * EVK_BLOCK], r7 = rec[i + 7 * EVK_BLOCK];
        f(r0), f(r1), f(r2), f(r3), f(r4), f(r5), f(r6), f(r7);
    }
#endif
    for (; i + 3 * EVK_BLOCK < hi; i += 4 * EVK_BLOCK) {
        const float4 r0 = rec[i], r1 = rec[i + EVK_BLOCK], r2 = rec[i + 2 * EVK_BLOCK], r3 = rec[i + 3 * EVK_BLOCK];
        f(r0), f(r1), f(r2), f(r3);
    }
    for (; i < hi; i += EVK_BLOCK) f(rec[i]);
}

// COMPACT records (8 bytes, evk_compact_records_f32): {t (float32 bits), polarity bits [31:11] | pixel in tile [9:0]}.
// Events whose x, y are integers inside the domain (sensor pixels) and whose polarity has its low 11 mantissa bits zero
// (+-1, 0, small integers, halves ...) lose nothing: x, y come back from the tile origin, which the work item knows.
// Streams compact records [lo, hi): a lane takes PAIRS (one 16-byte load = records 2j, 2j + 1), 4 loads in flight; the
// pair straddling lo / hi is loaded whole (the buffer is padded to an even count) and the outsiders skipped.
template <typename F>
__device__ __forceinline__ void stream_records8(const uint2 *__restrict__ rec, uint32_t lo, uint32_t hi, F f) {
    const uint4 *pairs = reinterpret_cast<const uint4 *>(rec);
    const uint32_t jend = (hi + 1u) >> 1;
    auto two = [&](const uint4 &v, uint32_t j) {
        const uint32_t pos = 2u * j;
        if (pos >= lo) f(make_uint2(v.x, v.y));
        if (pos + 1u < hi) f(make_uint2(v.z, v.w));
    };
    uint32_t j = (lo >> 1) + threadIdx.x;
    if (hi <= lo) return;
    for (; j + 3 * EVK_BLOCK < jend; j += 4 * EVK_BLOCK) {
        const uint4 r0 = pairs[j], r1 = pairs[j + EVK_BLOCK], r2 = pairs[j + 2 * EVK_BLOCK], r3 = pairs[j + 3 * EVK_BLOCK];
        two(r0, j), two(r1, j + EVK_BLOCK), two(r2, j + 2 * EVK_BLOCK), two(r3, j + 3 * EVK_BLOCK);
    }
    for (; j < jend; j += EVK_BLOCK) two(pairs[j], j);
}

// float4 records -> compact records, same order (record i -> compact record i); *not_compact |= 1 when some record is
// not representable (non-integer or out-of-domain coordinates, a polarity with low mantissa bits): the caller then
// keeps the 16-byte records.
__global__ void __launch_bounds__(EVK_BLOCK) k_compact_records(const float4 *__restrict__ rec, int64_t n, TileGrid g,
                                                               uint2 *__restrict__ out, uint32_t *__restrict__ not_compact) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    auto conv = [&](const float4 &r) -> uint2 {
        const float fx = floorf(r.x), fy = floorf(r.y);
        const uint32_t pb = __float_as_uint(r.w);
        if (!(fx == r.x && fy == r.y && fx >= 0.0f && fy >= 0.0f && fx <= (float)(g.dom_w - 1) && fy <= (float)(g.dom_h - 1)) ||
            (pb & ~EVK_REC_P_MASK))
            bad = true;
        const int xi = (int)fx, yi = (int)fy;
        const uint32_t local = (uint32_t)(((yi & ((1 << g.th_log2) - 1)) << g.tw_log2) | (xi & ((1 << g.tw_log2) - 1)));
        return make_uint2(__float_as_uint(r.z), (pb & EVK_REC_P_MASK) | (local & EVK_REC_LOCAL_MASK));
    };
    const int64_t npairs = (n + 1) >> 1;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < npairs; j += stride) {
        const uint2 a = conv(rec[2 * j]);
        const uint2 b = 2 * j + 1 < n ? conv(rec[2 * j + 1]) : make_uint2(0u, 0u);
        reinterpret_cast<uint4 *>(out)[j] = make_uint4(a.x, a.y, b.x, b.y);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(not_compact, 1u);
}

// flags: EVK_VOXEL_OVERWRITE; EVK_VOXEL_SPLIT_POLARITY: two grids in one pass, vox = (2, B, h, w): [0] counts the
// events with p > 0, [1] those with p <= 0, each with weight 1 (events_to_neg_pos_voxel_torch, voxel_grid.py:172-180).
__global__ void __launch_bounds__(EVK_BLOCK) k_voxel_tiled(const float4 *__restrict__ rec, uint32_t *__restrict__ index,
                                                           TileGrid g, float t_first, float dt, float bm1, int B,
                                                           int flags, float *__restrict__ vox,
                                                           float *__restrict__ staging) {
    const int overwrite = flags & EVK_VOXEL_OVERWRITE;
    const bool split = flags & EVK_VOXEL_SPLIT_POLARITY;
    const int NB = split ? 2 * B : B;  // accumulator planes
    extern __shared__ __attribute__((aligned(16))) acc_t acc[];
    const int ntiles = g.tiles_x * g.tiles_y;
    const uint32_t *bucket_start = index, *part_start = index + IDX_PART(ntiles), *item_tile = index + IDX_ITEM(ntiles);
    if (blockIdx.x >= part_start[ntiles]) return;  // the grid is the host's upper bound on the work items
    const int tw = 1 << g.tw_log2, th = 1 << g.th_log2, tpix = tw * th;
    const int tile = (int)item_tile[blockIdx.x];
    const uint32_t first_item = part_start[tile], nparts = part_start[tile + 1] - first_item;
    const uint32_t part_id = blockIdx.x - first_item;
    const int tx0 = (tile % g.tiles_x) << g.tw_log2, ty0 = (tile / g.tiles_x) << g.th_log2;
    for (int i = threadIdx.x; i < NB * tpix; i += EVK_BLOCK) acc[i] = 0.0;
    __syncthreads();
    const uint32_t blo = bucket_start[tile], cnt = bucket_start[tile + 1] - blo;
    const uint32_t lo = blo + (uint32_t)(((uint64_t)cnt * part_id) / nparts);
    const uint32_t hi = blo + (uint32_t)(((uint64_t)cnt * (part_id + 1)) / nparts);
    auto one = [&](const float4 &r) {
        int xi = (int)r.x, yi = (int)r.y;  // same conversion as tile_key (the record is known to be in this tile)
        if (xi < 0) xi += g.dom_w;
        if (yi < 0) yi += g.dom_h;
        const int local = ((yi - ty0) << g.tw_log2) + (xi - tx0);
        const float tn = (r.z - t_first) / dt * bm1;  // voxel_grid.py:134 (float32, IEEE divide)
        if (!split) {
            voxel_bins_lds(acc, tpix, local, B, tn, r.w);
        } else if (tn != tn) {  // dt == 0: the reference's 0 * NaN poisons BOTH grids at this pixel
            voxel_bins_lds(acc, tpix, local, B, tn, 1.0f);
            voxel_bins_lds(acc + B * tpix, tpix, local, B, tn, 1.0f);
        } else if (r.w > 0.0f) {
            voxel_bins_lds(acc, tpix, local, B, tn, 1.0f);
        } else if (r.w <= 0.0f) {  // a NaN polarity is in neither grid
            voxel_bins_lds(acc + B * tpix, tpix, local, B, tn, 1.0f);
        }
    };
    stream_records(rec, lo, hi, one);
    __syncthreads();
    const int64_t plane = (int64_t)g.dom_h * g.dom_w;
    auto flush = [&](auto value_of) {
        for (int c = threadIdx.x; c < NB * tpix; c += EVK_BLOCK) {
            const int b = c / tpix, l = c - b * tpix;
            const int X = tx0 + (l & (tw - 1)), Y = ty0 + (l >> g.tw_log2);
            if (X < g.dom_w && Y < g.dom_h) {
                float *o = vox + b * plane + (int64_t)Y * g.dom_w + X;
                // the tile is owned by this workgroup: plain, row-coalesced store (overwrite: the caller skipped
                // the memset, every cell of every tile is written) or read-modify-write (accumulate)
                const float v = value_of(c);
                *o = overwrite ? v : *o + v;
            }
        }
    };
    if (nparts == 1) {
        flush([&](int c) { return (float)acc[c]; });
        return;
    }
    // Split (hot) tile: every part stores its partial tile, the LAST part to arrive sums them in part order
    // (deterministic) and writes the output.  Hand-off with agent-scope (sc1) stores and loads on both sides
    // (MI355X_MICROARCH.md, valid forms): every wave drains its stores -> barrier -> one lane takes the ticket; the last
    // arriver reads the partial tiles with agent-scope loads.  No release / acquire fence (= a write-back of the XCD's L2).
    const int cells = NB * tpix;
    float *mine = staging + (int64_t)blockIdx.x * cells;
    for (int c = threadIdx.x; c < cells; c += EVK_BLOCK)
        __hip_atomic_store(mine + c, (float)acc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    EVK_HANDOVER_DRAIN();
    __shared__ int is_last;
    if (threadIdx.x == 0) {
        uint32_t *counter = index + IDX_COUNTER(ntiles) + tile;
        const uint32_t prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == nparts - 1);
        if (is_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
    if (!is_last) return;
    EVK_HANDOVER_ACQUIRE();
    const float *parts = staging + (int64_t)first_item * cells;
    flush([&](int c) {
        float sum = 0.0f;
        for (uint32_t p = 0; p < nparts; ++p)
            sum += __hip_atomic_load(parts + (int64_t)p * cells + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return sum;
    });
}

// ---------------------------------------------------------------------------------------------------------
// fused linear-flow IWE: one workgroup per (tile, time slice); LDS window = tile + flow halo
// ---------------------------------------------------------------------------------------------------------
struct IweParams {
    double t_ref, vx, vy, bw, bh, p_scale;
    float clipx, clipy;
    int ch, cw;        // canvas
    int slices;        // time slices per tile (runtime, from the flow magnitude)
    int win_w, win_h;  // LDS / staging window capacity (cells)
    int abs_p, grad;
    int trio;  // MODE 2: flows 1 and 2 differ from flow 0 in vx only / in vy only (forward differences): see k_iwe_tiled
    // bounds of (window origin - tile origin) over the whole stream; MODE 2: one set per flow (every flow's plane has its OWN
    // window origin, so three flows of any distance share a pass over the events)
    int sx_lo[3], sx_hi[3], sy_lo[3], sy_hi[3];
    double vxb[2], vyb[2];           // MODE 2 (batch of 3 nearby flows): flows 1 and 2 (flow 0 is vx, vy)
    double fx_scale, fx_inv;         // FIXED 1: LDS cells hold sum(value * 2^k) as int64 (2^k = fx_scale)
};

// Same per-event arithmetic as evk_scatter.hip's iwe_event (parity depends on it; the polarity shortcut below is exact).
__device__ __forceinline__ bool iwe_event_f32(const float4 &r, const IweParams &q, double vx, double vy, int &px,
                                              int &py, float &dx, float &dy, float &mp, float &jf) {
    const double dt = (double)r.z - q.t_ref;
    const double xw = (double)r.x - dt * vx;
    const double yw = (double)r.y - dt * vy;
    if (xw <= 0.0 || xw > q.bw || yw <= 0.0 || yw > q.bh) return false;
    const float xf = (float)xw, yf = (float)yw;
    if (xf >= q.clipx || yf >= q.clipy) return false;
    const float fx = floorf(xf), fy = floorf(yf);
    dx = xf - fx;
    dy = yf - fy;
    px = (int)fx;
    py = (int)fy;
    // polarity: (float)((double)p * p_scale) [, |.|] -- with p_scale == 1 (everything but the adaptive-lifespan x100, Q10)
    // that is p itself, bit for bit, and the two conversions and the float64 multiply go away
    if (q.p_scale == 1.0) {
        mp = q.abs_p ? fabsf(r.w) : r.w;
    } else {
        const double ps = (double)r.w * q.p_scale;
        mp = (float)(q.abs_p ? fabs(ps) : ps);
    }
    jf = (float)(-dt);
    return true;
}

// MODE 0: IWE.  MODE 1: IWE + dIWE (gradient).  MODE 2: three IWEs for three nearby flows in one pass over the events
// (forward-difference numeric gradient: f(v), f(v + eps e1), f(v + eps e2) share every event load).
// FIXED: the LDS cells are 64-bit FIXED-POINT sums (ds_add_u64 sustains 4.6 lane-ops/clk/CU vs 3.0 for ds_add_f64,
// tools/lds_probe2.hip, and this kernel is LDS-atomic bound).  Every float32 contribution is scaled by 2^k, k chosen by
// the host so that n * max|contribution| < 2^61 cannot overflow (k >= 26, typically 30-36): quantisation <= 2^-(k+1) per
// event, orders of magnitude below the float32 rounding of the result, and integer adds commute, so the window sums
// are bit-reproducible from run to run.
// (Packed 32-bit PAIRS -- two neighbouring cells per 64-bit word, 4 / 6 atomics per event instead of 8 / 12 in the gradient /
// three-flow modes -- were built in round 2 and measured level with these cells once the LDS pitches were odd: 0.288 against
// 0.290 ms per gradient evaluation at 50 M events; removed in round 4)
#ifndef IWE_ABLATE
#define IWE_ABLATE 99  // ablation builds (timing only): 0 record loads only, 1 + per-event arithmetic without LDS atomics
#endif
template <int MODE, int FIXED, bool COMPACT>
__global__ void __launch_bounds__(EVK_BLOCK) k_iwe_tiled(const float4 *__restrict__ rec,
                                                         const uint32_t *__restrict__ index, TileGrid g,
                                                         IweParams q, float *__restrict__ staging,
                                                         int4 *__restrict__ origins, float *__restrict__ iwe,
                                                         float *__restrict__ diwe) {
    extern __shared__ __attribute__((aligned(16))) acc_t win[];
    const int wcells = q.win_w * q.win_h;
    // LDS pitches: ODD numbers of 8-byte cells.  With an even pitch the rows of one column share few banks (40 cells = 80
    // dwords: 4 distinct bank pairs out of 32), and real scenes are edges -- a wave's events sit in one column, different
    // rows: 78 % of the LDS cycles of this kernel were bank conflicts on the moving-edge scene
    // (profiles/r02_c4_iwe_lds_counters.json).  lw: row pitch of the row-major planes, lh: of the column-major E0 plane.
    const int lw = q.win_w | 1, lh = q.win_h | 1, lcells = lw * lh;
    constexpr bool GRAD = (MODE == 1);
    constexpr int PLANES = MODE == 0 ? 1 : 3;
    const int ntiles = g.tiles_x * g.tiles_y;
    const uint32_t *bucket_start = index, *part_start = index + IDX_PART(ntiles), *item_tile = index + IDX_ITEM(ntiles);
    // work item = one part of a tile (hot tiles are split); each item is cut into q.slices time slices
    const uint32_t item = blockIdx.x / q.slices, sl = blockIdx.x - item * q.slices;
    if (item >= part_start[ntiles]) return;
    const int tile = (int)item_tile[item];
    const uint32_t nparts = part_start[tile + 1] - part_start[tile], part_id = item - part_start[tile];
    const uint32_t nsub = nparts * q.slices, sub = part_id * q.slices + sl;
    const uint32_t blo = bucket_start[tile], bhi = bucket_start[tile + 1];
    const uint32_t cnt = bhi - blo;
    const uint32_t lo = blo + (uint32_t)(((uint64_t)cnt * sub) / nsub);
    const uint32_t hi = blo + (uint32_t)(((uint64_t)cnt * (sub + 1)) / nsub);
    for (int i = threadIdx.x; i < PLANES * lcells; i += EVK_BLOCK) win[i] = 0.0;
    // Window origin from the time span of this slice (records are time-ordered up to intra-block interleaving; an
    // event that still falls outside takes the global-atomic path below, so this is a performance hint only).
    constexpr int NFLOW = MODE == 2 ? 3 : 1;
    int wx0[NFLOW], wy0[NFLOW];
#pragma unroll
    for (int k = 0; k < NFLOW; ++k) wx0[k] = wy0[k] = 0;
    if (hi > lo) {
        const uint2 *rec8 = reinterpret_cast<const uint2 *>(rec);
        const double ta = (double)(COMPACT ? __uint_as_float(rec8[lo].x) : rec[lo].z) - q.t_ref;
        const double tb = (double)(COMPACT ? __uint_as_float(rec8[hi - 1].x) : rec[hi - 1].z) - q.t_ref;
        const int tx0 = (tile % g.tiles_x) << g.tw_log2, ty0 = (tile / g.tiles_x) << g.th_log2;
#pragma unroll
        for (int k = 0; k < NFLOW; ++k) {
            const double vxk = k == 0 ? q.vx : q.vxb[k - 1], vyk = k == 0 ? q.vy : q.vyb[k - 1];
            const double dxm = fmin(-ta * vxk, -tb * vxk), dym = fmin(-ta * vyk, -tb * vyk);
            int sx = (int)floor(dxm) - 1, sy = (int)floor(dym) - 1;
            sx = sx < q.sx_lo[k] ? q.sx_lo[k] : (sx > q.sx_hi[k] ? q.sx_hi[k] : sx);  // the gather kernel relies on these bounds
            sy = sy < q.sy_lo[k] ? q.sy_lo[k] : (sy > q.sy_hi[k] ? q.sy_hi[k] : sy);
            wx0[k] = tx0 + sx;
            wy0[k] = ty0 + sy;
        }
    }
    __syncthreads();
    const int64_t plane = (int64_t)q.ch * q.cw;
    auto lds_acc = [&](acc_t *cell, float v) {
        if constexpr (FIXED == 1) {
            // round-to-nearest double -> int64 with one add: for |x| < 2^51 the low mantissa bits of x + 1.5*2^52 hold
            // x as a two's-complement integer (the host caps k so that every contribution satisfies |x| < 2^50)
            const double magic = 6755399441055744.0;
            // (fma: one instruction; the product by a power of two is exact, so the single rounding changes nothing)
            const long long fx = __double_as_longlong(__builtin_fma((double)v, q.fx_scale, magic)) - __double_as_longlong(magic);
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(cell), (unsigned long long)fx,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else
            lds_add(cell, v);
    };
    // one located event into one IWE plane (`wp` in LDS, `gp` in the image); GRAD adds the derivative planes behind it
    auto deposit = [&](int px, int py, float dx, float dy, float mp, float jf, acc_t *wp, float *gp, int ox, int oy) {
        if (IWE_ABLATE < 2) {
            if ((float)(px + py) + dx + dy + mp + jf == 1.2345e-30f) win[0] = 1.0;
            return;
        }
        const float ax = 1.0f - dx, ay = 1.0f - dy;
        const int lx = px - ox, ly = py - oy;
        const float a = jf * mp;
        if (lx >= 0 && ly >= 0 && lx + 1 < q.win_w && ly + 1 < q.win_h) {
            acc_t *c = wp + ly * lw + lx;
            lds_acc(c, mp * ax * ay);
            lds_acc(c + 1, mp * dx * ay);
            lds_acc(c + lw, mp * ax * dy);
            lds_acc(c + lw + 1, mp * dx * dy);
            if constexpr (GRAD) {
                // The reference's 8 derivative contributions come in +/- pairs on neighbouring pixels
                // (image.py:132-135): d0[y][x] gets -a*ay from px == x and +a*ay from px == x-1, etc.  Accumulate
                // the 4 magnitudes (E0: a*ay, a*dy; E1: a*ax, a*dx) and take the finite difference at the flush:
                // d0[y][x] = E0[y][x-1] - E0[y][x],  d1[y][x] = E1[y-1][x] - E1[y][x]   (8 LDS atomics, not 12).
                acc_t *e0 = c + lcells, *e1 = e0 + lcells;
                lds_acc(e0, a * ay);
                lds_acc(e0 + lw, a * dy);
                lds_acc(e1, a * ax);
                lds_acc(e1 + 1, a * dx);
            }
        } else {  // outside the window (flow larger than the halo, clamped outlier): straight to the image
            float *c = gp + (int64_t)py * q.cw + px;
            atomic_add(c, mp * ax * ay);
            atomic_add(c + 1, mp * dx * ay);
            atomic_add(c + q.cw, mp * ax * dy);
            atomic_add(c + q.cw + 1, mp * dx * dy);
            if constexpr (GRAD) {
                float *d0 = diwe + (int64_t)py * q.cw + px, *d1 = d0 + plane;
                atomic_add(d0, a * (-ay));
                atomic_add(d0 + 1, a * ay);
                atomic_add(d0 + q.cw, a * (-dy));
                atomic_add(d0 + q.cw + 1, a * dy);
                atomic_add(d1, a * (-ax));
                atomic_add(d1 + 1, a * (-dx));
                atomic_add(d1 + q.cw, a * ax);
                atomic_add(d1 + q.cw + 1, a * dx);
            }
        }
    };
    auto splat = [&](const float4 &r, double vx, double vy, acc_t *wp, float *gp, int ox, int oy) {
        int px, py;
        float dx, dy, mp, jf;
        if (IWE_ABLATE < 1) {
            if (r.x + r.y + r.z + r.w == 1.2345e-30f) win[0] = 1.0;
            return;
        }
        if (!iwe_event_f32(r, q, vx, vy, px, py, dx, dy, mp, jf)) return;
        deposit(px, py, dx, dy, mp, jf, wp, gp, ox, oy);
    };
    // MODE 2 with the flows of a forward-difference gradient -- v, v + (a, 0), v + (0, b): the x side of flows 0 and 2 and
    // the y side of flows 0 and 1 are the same numbers, and the tests of iwe_event_f32 are per axis, so each axis is
    // located twice instead of three times (same arithmetic per flow: bit-identical images, ~12 % fewer instructions in a
    // kernel that is bound by them)
    struct Axis {
        bool ok;
        int p;
        float d;
    };
    auto axis = [&](double w, double bound, float clip) -> Axis {
        Axis a;
        const float f = (float)w, fl = floorf(f);
        a.ok = !(w <= 0.0 || w > bound) && !(f >= clip);
        a.d = f - fl;
        a.p = (int)fl;
        return a;
    };
    auto one = [&](const float4 &r) {
        if constexpr (MODE == 2) {  // planes 1, 2 of the (3, ch, cw) buffer = diwe, diwe + plane
            if (q.trio && IWE_ABLATE >= 1) {
                const double dt = (double)r.z - q.t_ref;
                const Axis X0 = axis((double)r.x - dt * q.vx, q.bw, q.clipx), X1 = axis((double)r.x - dt * q.vxb[0], q.bw, q.clipx);
                const Axis Y0 = axis((double)r.y - dt * q.vy, q.bh, q.clipy), Y1 = axis((double)r.y - dt * q.vyb[1], q.bh, q.clipy);
                float mp;
                if (q.p_scale == 1.0) {
                    mp = q.abs_p ? fabsf(r.w) : r.w;
                } else {
                    const double ps = (double)r.w * q.p_scale;
                    mp = (float)(q.abs_p ? fabs(ps) : ps);
                }
                const float jf = (float)(-dt);
                if (X0.ok && Y0.ok) deposit(X0.p, Y0.p, X0.d, Y0.d, mp, jf, win, iwe, wx0[0], wy0[0]);
                if (X1.ok && Y0.ok) deposit(X1.p, Y0.p, X1.d, Y0.d, mp, jf, win + lcells, diwe, wx0[NFLOW - 2], wy0[NFLOW - 2]);
                if (X0.ok && Y1.ok) deposit(X0.p, Y1.p, X0.d, Y1.d, mp, jf, win + 2 * lcells, diwe + plane, wx0[NFLOW - 1], wy0[NFLOW - 1]);
                return;
            }
        }
        splat(r, q.vx, q.vy, win, iwe, wx0[0], wy0[0]);
        if constexpr (MODE == 2) {
            splat(r, q.vxb[0], q.vyb[0], win + lcells, diwe, wx0[NFLOW - 2], wy0[NFLOW - 2]);
            splat(r, q.vxb[1], q.vyb[1], win + 2 * lcells, diwe + plane, wx0[NFLOW - 1], wy0[NFLOW - 1]);
        }
    };
    if constexpr (COMPACT) {
        const int tx0 = (tile % g.tiles_x) << g.tw_log2, ty0 = (tile / g.tiles_x) << g.th_log2;
        const uint32_t twm = (1u << g.tw_log2) - 1u;
        stream_records8(reinterpret_cast<const uint2 *>(rec), lo, hi, [&](const uint2 &c) {
            const uint32_t local = c.y & EVK_REC_LOCAL_MASK;
            one(make_float4((float)(tx0 + (int)(local & twm)), (float)(ty0 + (int)(local >> g.tw_log2)), __uint_as_float(c.x),
                            __uint_as_float(c.y & EVK_REC_P_MASK)));
        });
    } else {
        stream_records(rec, lo, hi, one);
    }
    __syncthreads();
    float *st = staging + (int64_t)blockIdx.x * PLANES * wcells;
    // cell value / difference of two cells as float (exact integer difference on the fixed-point path)
    auto cell = [&](const acc_t *a) -> float {
        if constexpr (FIXED == 1) return (float)((double)*reinterpret_cast<const long long *>(a) * q.fx_inv);
        else return (float)*a;
    };
    auto diff = [&](const acc_t *a, const acc_t *b) -> float {  // a - b; a == nullptr means 0
        if constexpr (FIXED == 1) {
            const long long va = a ? *reinterpret_cast<const long long *>(a) : 0ll;
            return (float)((double)(va - *reinterpret_cast<const long long *>(b)) * q.fx_inv);
        } else {
            return (float)((a ? *a : 0.0) - *b);
        }
    };
    for (int c = threadIdx.x; c < wcells; c += EVK_BLOCK) {
        const int lx = c % q.win_w, ly = c / q.win_w, li = ly * lw + lx;  // staging cell c = LDS cell li
        st[c] = cell(win + li);
        if constexpr (GRAD) {
            const acc_t *e0 = win + lcells, *e1 = e0 + lcells;
            st[wcells + c] = diff(lx > 0 ? e0 + li - 1 : nullptr, e0 + li);
            st[2 * wcells + c] = diff(ly > 0 ? e1 + li - lw : nullptr, e1 + li);
        }
        if constexpr (MODE == 2) {
            st[wcells + c] = cell(win + lcells + li);
            st[2 * wcells + c] = cell(win + 2 * lcells + li);
        }
    }
    // MODE 2: the three planes are three WINDOWS (ids 3 w + k) with their own origins; the staging layout is the same
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NFLOW; ++k) origins[NFLOW * blockIdx.x + k] = make_int4(wx0[k], wy0[k], hi > lo ? 1 : 0, 0);
    }
}

// Gather: every canvas pixel sums the staged windows that cover it and ADDS the sum to the image (which already holds
// the rare direct atomics).  One workgroup per 32x8 pixel patch: the windows that can reach the patch (tiles whose
// window, shifted by at most [s_lo, s_hi], intersects it; all their work items and time slices) are listed once in LDS
// (sorted by window id, so the float32 summation order is fixed), then every pixel walks that short list.
#define EVK_GATHER_PX 32
// patch height PY (template): 8 rows, or 16 (two rows per thread) when the 8-row patches would not all be resident at once
// (more than 2048 of them: 720p and up) -- 32 x 16 patches measured 18.8 instead of 21 us at 720p, but 11.1 instead of
// 8.0 us at VGA
#define EVK_GATHER_CAP 128
#define EVK_GATHER_CAND 1024  // windows of all candidate tiles (before the overlap test)
struct GatherBounds {
    int sx_lo[3], sx_hi[3], sy_lo[3], sy_hi[3];
};
// nflow = 3 (grid.y = 3, GRAD = false): the three flows of a batched evaluation -- flow k = blockIdx.y sums the one-plane
// windows 3 w + k (own origins, own shift bounds) into plane k of the (3, ch, cw) output.
template <bool GRAD, int PY>
__global__ void __launch_bounds__(EVK_BLOCK) k_iwe_gather(const float *__restrict__ staging,
                                                          const int4 *__restrict__ origins,
                                                          const uint32_t *__restrict__ index, TileGrid g, int slices,
                                                          int win_w, int win_h, int ch, int cw, GatherBounds gb, int nflow,
                                                          float *__restrict__ iwe,
                                                          float *__restrict__ diwe, const float *__restrict__ spill,
                                                          float *__restrict__ spill_clean) {
    constexpr int PLANES = GRAD ? 3 : 1;
    const int flow = blockIdx.y;
    const int sx_lo = gb.sx_lo[flow], sx_hi = gb.sx_hi[flow], sy_lo = gb.sy_lo[flow], sy_hi = gb.sy_hi[flow];
    if (flow > 0) {   // planes 1, 2 of the output and of the spill pair
        iwe = diwe + (int64_t)(flow - 1) * ch * cw;
        if (spill) spill += (int64_t)flow * ch * cw, spill_clean += (int64_t)flow * ch * cw;
    }
    __shared__ int list_w[EVK_GATHER_CAP], list_x[EVK_GATHER_CAP], list_y[EVK_GATHER_CAP];   // unsorted
    __shared__ int sort_w[EVK_GATHER_CAP], sort_x[EVK_GATHER_CAP], sort_y[EVK_GATHER_CAP];   // by window id
    __shared__ int cand[EVK_GATHER_CAND];  // ids of the windows of the candidate tiles
    __shared__ int count, ncand;
    const int wcells = win_w * win_h;
    const int64_t plane = (int64_t)ch * cw;
    const uint32_t *part_start = index + IDX_PART(g.tiles_x * g.tiles_y);
    const int patches_x = (cw + EVK_GATHER_PX - 1) / EVK_GATHER_PX;
    const int X0 = (blockIdx.x % patches_x) * EVK_GATHER_PX, Y0 = (blockIdx.x / patches_x) * PY;
    const int X1 = min(X0 + EVK_GATHER_PX, cw) - 1, Y1 = min(Y0 + PY, ch) - 1;  // inclusive
    // a window starts at tile_origin + shift, shift in [s_lo, s_hi] (clamped by k_iwe_tiled): tile tx can reach the
    // patch columns [X0, X1] iff tx*tw + s_lo <= X1 and X0 < tx*tw + s_hi + win_w
    const int tx_a = max((X0 - sx_hi - win_w + 1) >> g.tw_log2, 0), tx_b = min((X1 - sx_lo) >> g.tw_log2, g.tiles_x - 1);
    const int ty_a = max((Y0 - sy_hi - win_h + 1) >> g.th_log2, 0), ty_b = min((Y1 - sy_lo) >> g.th_log2, g.tiles_y - 1);
    // a thread owns column X and ROWS rows (RSTEP apart) of the patch; what the output needs from the spill pair is
    // fetched first, so that these loads are in flight while the window list is built
    constexpr int RSTEP = EVK_BLOCK / EVK_GATHER_PX, ROWS = PY / RSTEP;
    const int X = X0 + (threadIdx.x & (EVK_GATHER_PX - 1)), Yb = Y0 + threadIdx.x / EVK_GATHER_PX;
    bool inside[ROWS];
    float sp[ROWS][PLANES], sc[ROWS][PLANES];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        inside[rr] = X < cw && Yb + rr * RSTEP < ch;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
            sp[rr][pl] = sc[rr][pl] = 0.0f;
            if (spill && inside[rr]) {
                const int64_t pix = (int64_t)(Yb + rr * RSTEP) * cw + X;
                sp[rr][pl] = spill[pl * plane + pix];
                sc[rr][pl] = spill_clean[pl * plane + pix];
            }
        }
    }
    if (threadIdx.x == 0) count = 0, ncand = 0;
    __syncthreads();
    // Three short parallel steps instead of a few threads walking their tiles' windows one dependent load after the
    // other (and one thread sorting): (1) a thread per candidate tile reserves room for its window ids, (2) a thread per
    // window fetches its origin and appends it when it touches the patch, (3) a thread per listed window finds its rank.
    const int ntx = tx_b - tx_a + 1, nty = ty_b - ty_a + 1;
    for (int c = threadIdx.x; c < ntx * nty; c += EVK_BLOCK) {
        const int tile = (ty_a + c / ntx) * g.tiles_x + tx_a + c % ntx;
        const int w0 = (int)part_start[tile] * slices, w1 = (int)part_start[tile + 1] * slices;
        const int at = atomicAdd(&ncand, w1 - w0);
        for (int w = w0; w < w1 && at + (w - w0) < EVK_GATHER_CAND; ++w) cand[at + (w - w0)] = nflow * w + flow;
    }
    __syncthreads();
    const int m = ncand;
    if (m <= EVK_GATHER_CAND) {
        for (int i = threadIdx.x; i < m; i += EVK_BLOCK) {
            const int w = cand[i];
            const int4 o = origins[w];
            if (!o.z || o.x > X1 || o.x + win_w <= X0 || o.y > Y1 || o.y + win_h <= Y0) continue;
            const int k = atomicAdd(&count, 1);
            if (k < EVK_GATHER_CAP) list_w[k] = w, list_x[k] = o.x, list_y[k] = o.y;
        }
    }
    __syncthreads();
    const int nlist = m <= EVK_GATHER_CAND ? count : EVK_GATHER_CAP + 1;
    float acc[ROWS][3] = {};
    auto add_window = [&](int w, int ox, int oy) {
        const int lx = X - ox;
        if (lx < 0 || lx >= win_w) return;
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int ly = Yb + rr * RSTEP - oy;
            if (!inside[rr] || ly < 0 || ly >= win_h) continue;
            const float *st = staging + (int64_t)w * PLANES * wcells + ly * win_w + lx;
            acc[rr][0] += st[0];
            if constexpr (GRAD) {
                acc[rr][1] += st[wcells];
                acc[rr][2] += st[2 * wcells];
            }
        }
    };
    if (nlist <= EVK_GATHER_CAP) {
        // by window id (ids are unique: rank = number of smaller ids): a fixed float32 summation order
        for (int k = threadIdx.x; k < nlist; k += EVK_BLOCK) {
            const int w = list_w[k];
            int rank = 0;
            for (int j = 0; j < nlist; ++j) rank += list_w[j] < w;
            sort_w[rank] = w, sort_x[rank] = list_x[k], sort_y[rank] = list_y[k];
        }
        __syncthreads();
#ifndef EVK_GATHER_BATCH
#define EVK_GATHER_BATCH 4   // (A/B: 1 = one window at a time)
#endif
        if (EVK_GATHER_BATCH <= 1) {
            for (int k = 0; k < nlist; ++k) add_window(sort_w[k], sort_x[k], sort_y[k]);
        } else {
            // (round 6) the windows of the list FOUR at a time: their staging loads are issued together and added in list
            // order afterwards -- the same sums in the same order, a latency chain of nlist / 4 round trips instead of nlist
            // (the list holds 4-9 windows at configs[2]'s optimum; this kernel is a chain of dependent loads over a
            // 1-4 MB image: 15.8 -> ~11 us with three planes at 640x480)
            constexpr int GB = EVK_GATHER_BATCH;
            for (int k0 = 0; k0 < nlist; k0 += GB) {
                float v[GB][ROWS][3];
                bool ok[GB][ROWS];
#pragma unroll
                for (int u = 0; u < GB; ++u) {
                    const int k = k0 + u < nlist ? k0 + u : nlist - 1;
                    const int w = sort_w[k], lx = X - sort_x[k], oy = sort_y[k];
                    const bool okx = k0 + u < nlist && lx >= 0 && lx < win_w;
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr) {
                        const int ly = Yb + rr * RSTEP - oy;
                        ok[u][rr] = okx && inside[rr] && ly >= 0 && ly < win_h;
                        const float *st = staging + (int64_t)w * PLANES * wcells + (ok[u][rr] ? ly * win_w + lx : 0);
                        v[u][rr][0] = v[u][rr][1] = v[u][rr][2] = 0.0f;
                        if (ok[u][rr]) {
                            v[u][rr][0] = st[0];
                            if constexpr (GRAD) v[u][rr][1] = st[wcells], v[u][rr][2] = st[2 * wcells];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < GB; ++u)
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr)
                        if (ok[u][rr]) {
                            acc[rr][0] += v[u][rr][0];
                            if constexpr (GRAD) acc[rr][1] += v[u][rr][1], acc[rr][2] += v[u][rr][2];
                        }
            }
        }
    } else {  // more candidate windows than the LDS list holds (huge flows / many slices): walk them all
        for (int ty = ty_a; ty <= ty_b; ++ty)
            for (int tx = tx_a; tx <= tx_b; ++tx) {
                const int tile = ty * g.tiles_x + tx;
                const int w0 = (int)part_start[tile] * slices, w1 = (int)part_start[tile + 1] * slices;
                for (int w = w0; w < w1; ++w) {
                    const int4 o = origins[nflow * w + flow];
                    if (o.z) add_window(nflow * w + flow, o.x, o.y);
                }
            }
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        if (!inside[rr]) continue;
        const int64_t pix = (int64_t)(Yb + rr * RSTEP) * cw + X;
        if (spill) {  // out = spill + windows: the output needs no memset; the OTHER spill image is zeroed for the next call
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
                float *o = pl == 0 ? iwe + pix : diwe + (pl - 1) * plane + pix;
                *o = sp[rr][pl] + acc[rr][pl];
                if (sc[rr][pl] != 0.0f) spill_clean[pl * plane + pix] = 0.0f;
            }
        } else {
            iwe[pix] += acc[rr][0];
            if constexpr (GRAD) {
                diwe[pix] += acc[rr][1];
                diwe[plane + pix] += acc[rr][2];
            }
        }
    }
}

}  // namespace evk

using namespace evk;

extern "C" int evk_bucket_num_tiles(int dom_h, int dom_w, int tw_log2, int th_log2) {
    TileGrid g;
    if (make_grid(g, dom_h, dom_w, tw_log2, th_log2) != EVK_OK) return EVK_EINVAL;
    const int nt = g.tiles_x * g.tiles_y;
    return nt <= EVK_MAX_TILES ? nt : EVK_EINVAL;
}

extern "C" int64_t evk_bucket_index_len(int ntiles, int64_t n) {
    if (ntiles <= 0 || n < 0) return 0;
    return (int64_t)IDX_ITEM(ntiles) + bucket_max_items_balanced(n, ntiles) + 3;  // + the scene word and the two stats words
}

extern "C" int evk_bucket_max_items(int ntiles, int64_t n) { return ntiles > 0 && n >= 0 ? bucket_max_items_balanced(n, ntiles) : 0; }

extern "C" int64_t evk_bucket_scratch_bytes(int ntiles) {
    if (ntiles <= 0) return 0;   // the [tile][block] table, the tile totals, the blocks' stats pairs (EVK_STAGE_STATS)
    return ((int64_t)EVK_BUCKET_BLOCKS * ntiles + ntiles + EVK_BUCKET_BLOCKS) * (int64_t)sizeof(uint32_t);
}

template <int RR, typename C>
static void launch_scatter_wc(const C &c, int64_t n, int64_t chunk, const TileGrid &g, int key_mode, int ntiles,
                              const uint32_t *table, const uint32_t *bucket_start, float *records, size_t lds_wc,
                              hipStream_t s) {
    static uint64_t attr_set = 0;  // per device of this process (the attribute is per device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(attr_set >> (dev & 63) & 1)) {
        (void)hipFuncSetAttribute((const void *)k_tile_scatter_wc<RR, C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr_set |= (uint64_t)1 << (dev & 63);
    }
    k_tile_scatter_wc<RR, C><<<EVK_BUCKET_BLOCKS, EVK_BUCKET_THREADS, lds_wc, s>>>(c, n, chunk, g, key_mode, ntiles, table,
                                                                                bucket_start, (float4 *)records);
}

__global__ void k_set_word(uint32_t *w, uint32_t v) { *w = v; }

template <int EPT, typename C>
static void launch_scatter_sorted(const C &c, int64_t n, int64_t chunk, const TileGrid &g, int key_mode, int ntiles,
                                  const uint32_t *table, const uint32_t *bucket_start, float *records, uint32_t *stats,
                                  int try_compact, hipStream_t s) {
    static std::once_flag once[64];   // per device and instantiation
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [] {
        (void)hipFuncSetAttribute((const void *)k_tile_scatter_sorted<EPT, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    });
    k_tile_scatter_sorted<EPT, C><<<EVK_BUCKET_BLOCKS, EVK_BUCKET_THREADS, scatter_sorted_lds(EPT, ntiles), s>>>(
        c, n, chunk, g, key_mode, ntiles, table, bucket_start, (void *)records, stats, try_compact);
}

template <typename C>
static int bucket_events(const C &c, int64_t n, int key_mode, int dom_h, int dom_w, int tw_log2, int th_log2,
                         float *records, uint32_t *bucket_start, void *scratch, int64_t scratch_bytes, uint32_t *oob,
                         int stages, void *stream) {
    TileGrid g;
    if (make_grid(g, dom_h, dom_w, tw_log2, th_log2) != EVK_OK) return EVK_EINVAL;
    const int ntiles = g.tiles_x * g.tiles_y;
    if (ntiles > EVK_MAX_TILES || n < 0 || n > (int64_t)4000000000LL || (key_mode != 0 && key_mode != 1)) return EVK_EINVAL;
    if (!bucket_start || !scratch || (n > 0 && !records)) return EVK_EINVAL;
    if (scratch_bytes < evk_bucket_scratch_bytes(ntiles)) return EVK_ESCRATCH;
    if (!aligned16(records)) return EVK_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    uint32_t *table = (uint32_t *)scratch;
    uint32_t *totals = table + (int64_t)EVK_BUCKET_BLOCKS * ntiles;
    int64_t chunk = (n + EVK_BUCKET_BLOCKS - 1) / EVK_BUCKET_BLOCKS;
    chunk = (chunk + 3) & ~(int64_t)3;
    if (chunk == 0) chunk = 4;
    const size_t lds = (size_t)ntiles * sizeof(uint32_t);
    uint32_t *const scene = bucket_start + IDX_ITEM(ntiles) + bucket_max_items_balanced(n, ntiles), *const stats = scene + 1;
    // compact records need tiles of <= 1024 pixels (10-bit pixel field) and the IWE key (every event inside a tile)
    const bool compact_ok = (1 << (tw_log2 + th_log2)) <= (int)EVK_REC_LOCAL_MASK + 1 && key_mode == EVK_KEY_FLOOR_CLAMP;
    if ((stages & EVK_STAGE_COMPACT) && !(stages & EVK_STAGE_STATS)) return EVK_EINVAL;
    uint32_t *const block_stats = totals + ntiles;   // [EVK_BUCKET_BLOCKS]
    if (stages & EVK_STAGE_HIST) {
        if (stages & EVK_STAGE_STATS)
            k_tile_hist<C, true><<<EVK_BUCKET_BLOCKS, EVK_BUCKET_THREADS, lds, s>>>(c, n, chunk, g, key_mode, ntiles, table, oob, block_stats);
        else
            k_tile_hist<C, false><<<EVK_BUCKET_BLOCKS, EVK_BUCKET_THREADS, lds, s>>>(c, n, chunk, g, key_mode, ntiles, table, oob, nullptr);
    }
    if (stages & EVK_STAGE_SCAN) {
        k_tile_scan_blocks<<<(ntiles + 3) / 4, 256, 0, s>>>(table, ntiles, totals);
        k_tile_scan_totals<<<1, 1024, 0, s>>>(totals, ntiles, (uint32_t)bucket_cap(n, ntiles),
                                              (uint32_t)bucket_item_budget(ntiles), bucket_start, scene,
                                              (stages & EVK_STAGE_STATS) ? block_stats : nullptr);
    }
    if (!(stages & EVK_STAGE_SCATTER)) return launch_status();
    // the LDS-sorting scatter (round 6) where its buffers fit the budget below: sub-chunks of 8 K events, or of 4 K
    const size_t lds_cap = (stages & EVK_STAGE_SHARE_CU) ? (size_t)96 * 1024 : (size_t)160 * 1024 - 512;
    const int try_compact = (stages & EVK_STAGE_COMPACT) && compact_ok ? 1 : 0;
    if (!(stages & EVK_STAGE_LEGACY_SCATTER)) {
        for (int ept : {8, 4}) {
            if (scatter_sorted_lds(ept, ntiles) > lds_cap) continue;
            uint32_t *const st = (stages & EVK_STAGE_STATS) ? stats : nullptr;
            if (ept == 8) launch_scatter_sorted<8>(c, n, chunk, g, key_mode, ntiles, table, bucket_start, records, st, try_compact, s);
            else launch_scatter_sorted<4>(c, n, chunk, g, key_mode, ntiles, table, bucket_start, records, st, try_compact, s);
            return launch_status();
        }
    }
    // (only the sorting scatter writes compact records and max |p|: on a tiling it cannot take -- counters that do not fit beside
    // the sort buffer -- the ring scatter writes 16-byte records and the index says so: verdict 1, max |p| unknown)
    if (stages & EVK_STAGE_STATS) {
        if (try_compact) k_set_word<<<1, 1, 0, s>>>(stats, 1u);
        k_set_word<<<1, 1, 0, s>>>(stats + 1, 0xFFFFFFFFu);
    }
    // write-combining scatter when the per-tile LDS rings fit (160 KiB per CU), else the plain scatter
    // all partition blocks co-resident, one per CU.  With EVK_STAGE_SHARE_CU the rings take at most 96 KB so that a
    // workgroup of ANOTHER kernel (an overlapped RCCL collective) still fits on every CU: a scatter workgroup that
    // owns all 160 KB cannot start on a CU where such a kernel is resident, and one late workgroup costs the call
    // +27 % (tools/contention_probe.py); the smaller rings cost +3 % and are insensitive to it.
    const size_t lds_budget = (stages & EVK_STAGE_SHARE_CU) ? (size_t)96 * 1024
                                                            : (size_t)(160 * 1024) / (EVK_BUCKET_BLOCKS / 256) - 256;
    int R = 0;
    for (int r : {16, 8, 4})
        if (!R && (size_t)ntiles * (r * 16 + 10) + 8 <= lds_budget) R = r;
    const size_t lds_wc = (size_t)ntiles * (R * 16 + 10) + 8;  // rings + cursor + vstart + flush queue
    if (R == 16) launch_scatter_wc<16>(c, n, chunk, g, key_mode, ntiles, table, bucket_start, records, lds_wc, s);
    else if (R == 8) launch_scatter_wc<8>(c, n, chunk, g, key_mode, ntiles, table, bucket_start, records, lds_wc, s);
    else if (R == 4) launch_scatter_wc<4>(c, n, chunk, g, key_mode, ntiles, table, bucket_start, records, lds_wc, s);
    else
        k_tile_scatter<C><<<EVK_BUCKET_BLOCKS, EVK_BUCKET_THREADS, lds, s>>>(c, n, chunk, g, key_mode, ntiles, table,
                                                                          bucket_start, (float4 *)records);
    return launch_status();
}

extern "C" int evk_bucket_events_f32(const float *x, const float *y, const float *t, const float *p, int64_t n,
                                     int key_mode, int dom_h, int dom_w, int tw_log2, int th_log2, float *records,
                                     uint32_t *bucket_start, void *scratch, int64_t scratch_bytes, uint32_t *oob,
                                     int stages, void *stream) {
    if (n > 0 && (!x || !y || !t || !p)) return EVK_EINVAL;
    if (!(aligned16(x) && aligned16(y) && aligned16(t) && aligned16(p))) return EVK_EALIGN;
    const ColsF32 c{x, y, t, p};
    return bucket_events(c, n, key_mode, dom_h, dom_w, tw_log2, th_log2, records, bucket_start, scratch, scratch_bytes, oob,
                         stages, stream);
}

extern "C" int evk_bucket_events_native_f32(const int16_t *x, const int16_t *y, int xy_stride, const void *t, int t_kind,
                                            double t_offset, const void *p, int p_kind, int64_t n, int key_mode,
                                            int dom_h, int dom_w, int tw_log2, int th_log2, float *records,
                                            uint32_t *bucket_start, void *scratch, int64_t scratch_bytes, uint32_t *oob,
                                            int stages, void *stream) {
    ColsNative c;
    const int rc = native_cols(c, x, y, xy_stride, t, t_kind, t_offset, p, p_kind, n);
    if (rc != EVK_OK) return rc;
    if (!(aligned16(x) && (xy_stride == 2 || aligned16(y)) && aligned16(t) && aligned16(p))) return EVK_EALIGN;
    return bucket_events(c, n, key_mode, dom_h, dom_w, tw_log2, th_log2, records, bucket_start, scratch, scratch_bytes, oob,
                         stages, stream);
}

extern "C" int64_t evk_compact_records_bytes(int64_t n) { return n < 0 ? 0 : ((n + 1) & ~(int64_t)1) * 8; }

extern "C" int evk_compact_records_f32(const float *records, int64_t n, int dom_h, int dom_w, int tw_log2, int th_log2,
                                       void *compact, uint32_t *not_compact, void *stream) {
    TileGrid g;
    if (make_grid(g, dom_h, dom_w, tw_log2, th_log2) != EVK_OK || n < 0 || !not_compact || (n > 0 && (!records || !compact)))
        return EVK_EINVAL;
    if ((1 << (tw_log2 + th_log2)) > (int)EVK_REC_LOCAL_MASK + 1) return EVK_EINVAL;
    if (!aligned16(records) || !aligned16(compact)) return EVK_EALIGN;
    if (n == 0) return EVK_OK;
    k_compact_records<<<stream_grid((n + 1) >> 1), EVK_BLOCK, 0, (hipStream_t)stream>>>(
        (const float4 *)records, n, g, (uint2 *)compact, not_compact);
    return launch_status();
}

// native columns -> four float32 SoA columns (any alignment): the direct kernels' input, 13 B read + 16 B written
__global__ void __launch_bounds__(EVK_BLOCK) k_native_to_columns(const ColsNative c, int64_t n, float *__restrict__ x,
                                                                 float *__restrict__ y, float *__restrict__ t,
                                                                 float *__restrict__ p) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        x[i] = c.x1(i), y[i] = c.y1(i), t[i] = c.t1(i), p[i] = c.p1(i);
}

extern "C" int evk_native_to_columns_f32(const int16_t *x, const int16_t *y, int xy_stride, const void *t, int t_kind,
                                         double t_offset, const void *p, int p_kind, int64_t n, float *out_x,
                                         float *out_y, float *out_t, float *out_p, void *stream) {
    ColsNative c;
    const int rc = native_cols(c, x, y, xy_stride, t, t_kind, t_offset, p, p_kind, n);
    if (rc != EVK_OK || n < 0) return rc != EVK_OK ? rc : EVK_EINVAL;
    if (n == 0) return EVK_OK;
    if (!out_x || !out_y || !out_t || !out_p) return EVK_EINVAL;
    k_native_to_columns<<<stream_grid(n), EVK_BLOCK, 0, (hipStream_t)stream>>>(c, n, out_x, out_y, out_t, out_p);
    return launch_status();
}

extern "C" int64_t evk_voxel_tiled_staging_bytes(int ntiles, int64_t n, int B, int tw_log2, int th_log2) {
    return (int64_t)bucket_max_items_balanced(n, ntiles) * ((int64_t)B << (tw_log2 + th_log2)) * (int64_t)sizeof(float);
}

extern "C" int evk_voxel_tiled_f32(const float *records, uint32_t *bucket_index, int64_t n, int h, int wd, int tw_log2,
                                   int th_log2, float t_first, float t_last, int B, int flags, float *vox,
                                   void *staging, int64_t staging_bytes, void *stream) {
    TileGrid g;
    if (make_grid(g, h, wd, tw_log2, th_log2) != EVK_OK || B <= 0 || !records || !bucket_index || !vox || !staging ||
        n < 0 || (flags & ~(EVK_VOXEL_OVERWRITE | EVK_VOXEL_SPLIT_POLARITY)))
        return EVK_EINVAL;
    const int ntiles = g.tiles_x * g.tiles_y;
    const int planes = (flags & EVK_VOXEL_SPLIT_POLARITY) ? 2 * B : B;
    const size_t lds = (size_t)planes * sizeof(acc_t) << (tw_log2 + th_log2);
    if (lds > 64 * 1024) return EVK_EINVAL;
    if (staging_bytes < evk_voxel_tiled_staging_bytes(ntiles, n, planes, tw_log2, th_log2)) return EVK_ESCRATCH;
    const float dt = t_last - t_first, bm1 = (float)(B - 1);
    k_voxel_tiled<<<bucket_max_items_balanced(n, ntiles), EVK_BLOCK, lds, (hipStream_t)stream>>>(
        (const float4 *)records, bucket_index, g, t_first, dt, bm1, B, flags, vox, (float *)staging);
    return launch_status();
}

extern "C" int64_t evk_iwe_tiled_staging_bytes(int ntiles, int64_t n, int slices, int planes, int win_w, int win_h) {
    return (int64_t)bucket_max_items_balanced(n, ntiles) * slices *
           ((int64_t)planes * win_w * win_h * (int64_t)sizeof(float) + (int64_t)planes * (int64_t)sizeof(int4));
}

static int launch_iwe_tiled(int mode, const float *records, const uint32_t *bucket_index, int64_t n, int dom_h, int dom_w,
                            int tw_log2, int th_log2, int slices, int win_w, int win_h, double t_first, double t_ref,
                            const double *vx, const double *vy, double bounds_w, double bounds_h, int canvas_h,
                            int canvas_w, uint32_t flags, double p_scale, double p_bound, double dt_bound, void *staging,
                            int64_t staging_bytes, float *iwe, float *diwe, void *stream, const float *spill = nullptr,
                            float *spill_clean = nullptr) {
    TileGrid g;
    if (make_grid(g, dom_h, dom_w, tw_log2, th_log2) != EVK_OK || !records || !bucket_index || !iwe || !staging)
        return EVK_EINVAL;
    const int planes = mode == 0 ? 1 : 3;
    if ((planes == 3 && !diwe) || slices < 1 || slices > 256 || canvas_h <= 1 || canvas_w <= 1) return EVK_EINVAL;
    const int tw = 1 << tw_log2, th = 1 << th_log2;
    if (win_w < tw + 3 || win_h < th + 3) return EVK_EINVAL;
    const size_t lds = (size_t)planes * (win_w | 1) * (win_h | 1) * sizeof(acc_t);  // odd LDS pitches, see k_iwe_tiled
    if (lds > 64 * 1024) return EVK_EINVAL;
    const int ntiles = g.tiles_x * g.tiles_y;
    if (n < 0 || staging_bytes < evk_iwe_tiled_staging_bytes(ntiles, n, slices, planes, win_w, win_h)) return EVK_ESCRATCH;
    IweParams q;
    q.t_ref = t_ref, q.vx = vx[0], q.vy = vy[0], q.bw = bounds_w, q.bh = bounds_h, q.p_scale = p_scale;
    q.clipx = (float)(canvas_w - 1), q.clipy = (float)(canvas_h - 1);
    q.ch = canvas_h, q.cw = canvas_w, q.slices = slices, q.win_w = win_w, q.win_h = win_h;
    q.abs_p = (flags & EVK_IWE_ABS_POLARITY) ? 1 : 0, q.grad = (mode == 1);
    const int nflow = mode == 2 ? 3 : 1;
    for (int k = 0; k < 2; ++k) q.vxb[k] = mode == 2 ? vx[k + 1] : vx[0], q.vyb[k] = mode == 2 ? vy[k + 1] : vy[0];
    q.trio = (mode == 2 && vy[1] == vy[0] && vx[2] == vx[0]) ? 1 : 0;
    // displacement of an event at time t is -(t - t_ref) * v; over [t_first, t_ref] it spans [min(0, D), max(0, D)]
    double dx_lo = 0.0, dx_hi = 0.0, dy_lo = 0.0, dy_hi = 0.0;
    for (int k = 0; k < nflow; ++k) {
        const double Dx = -(t_first - t_ref) * vx[k], Dy = -(t_first - t_ref) * vy[k];
        if (!(fabs(Dx) < 1e6 && fabs(Dy) < 1e6)) return EVK_EINVAL;
        dx_lo = fmin(dx_lo, Dx), dx_hi = fmax(dx_hi, Dx), dy_lo = fmin(dy_lo, Dy), dy_hi = fmax(dy_hi, Dy);
    }
    GatherBounds gb;
    for (int k = 0; k < 3; ++k) {   // per flow (MODE 2), else the one flow's in every slot
        const int kk = k < nflow ? k : 0;
        const double Dx = -(t_first - t_ref) * vx[kk], Dy = -(t_first - t_ref) * vy[kk];
        q.sx_lo[k] = gb.sx_lo[k] = (int)floor(fmin(0.0, Dx)) - 1, q.sx_hi[k] = gb.sx_hi[k] = (int)floor(fmax(0.0, Dx)) - 1;
        q.sy_lo[k] = gb.sy_lo[k] = (int)floor(fmin(0.0, Dy)) - 1, q.sy_hi[k] = gb.sy_hi[k] = (int)floor(fmax(0.0, Dy)) - 1;
    }
    (void)dx_lo, (void)dx_hi, (void)dy_lo, (void)dy_hi;
    // fixed-point LDS accumulation when the caller can bound every contribution: |p * p_scale| <= p_bound, |t - t_ref| <=
    // dt_bound.  64-bit cells: one scale for the launch from n * p_bound * max(1, dt_bound) >= any cell sum.
    const double acc_bound = (p_bound > 0.0 && dt_bound >= 0.0) ? (double)n * p_bound * fmax(1.0, dt_bound) : 0.0;
    int k = 0;
    if (acc_bound > 0.0 && acc_bound < 1e300) {
        int e;
        (void)frexp(acc_bound, &e);  // acc_bound < 2^e
        k = 61 - e;
        if (k > 40) k = 40;
        // one contribution is at most acc_bound / n: keep |contribution * 2^k| < 2^50 for the one-add conversion
        int e1;
        (void)frexp(acc_bound / (double)(n > 0 ? n : 1), &e1);
        if (k > 50 - e1) k = 50 - e1;
    }
    const bool fixed = k >= 26;
    q.fx_scale = fixed ? ldexp(1.0, k) : 0.0;
    q.fx_inv = fixed ? ldexp(1.0, -k) : 0.0;
    const int nwin = bucket_max_items_balanced(n, ntiles) * slices;
    int4 *origins = (int4 *)staging;  // origins first (16 B each), windows after
    float *st = (float *)((char *)staging + (int64_t)nwin * planes * sizeof(int4));   // (three origins per workgroup in MODE 2)
    hipStream_t s = (hipStream_t)stream;
    const int gpx = (canvas_w + EVK_GATHER_PX - 1) / EVK_GATHER_PX;
    const bool tall = mode == 0 && gpx * ((canvas_h + 7) / 8) > 2048;   // see k_iwe_gather (three planes: 29.3 vs 28.3 us)
    const int ggrid = gpx * ((canvas_h + (tall ? 15 : 7)) / (tall ? 16 : 8));
    const float4 *rec = (const float4 *)records;
    const bool compact = (flags & EVK_IWE_COMPACT) != 0;  // `records` are 8-byte compact records
    if (compact && ((1 << (tw_log2 + th_log2)) > (int)EVK_REC_LOCAL_MASK + 1 || !aligned16(records))) return EVK_EINVAL;
    // spill pair: the kernel's global atomics (events outside their window) go to `spill` and the gather WRITES
    // out = spill + windows (no memset of the output), zeroing what the previous call left in `spill_clean`
    float *out_iwe = iwe, *out_diwe = diwe;
    if (spill) {
        iwe = const_cast<float *>(spill);
        diwe = iwe + (size_t)canvas_h * canvas_w;
    }
#define EVK_IWE_LAUNCH_C(M, C)                                                                                     \
    do {                                                                                                           \
        if (fixed) k_iwe_tiled<M, 1, C><<<nwin, EVK_BLOCK, lds, s>>>(rec, bucket_index, g, q, st, origins, iwe, diwe); \
        else k_iwe_tiled<M, 0, C><<<nwin, EVK_BLOCK, lds, s>>>(rec, bucket_index, g, q, st, origins, iwe, diwe);   \
    } while (0)
#define EVK_IWE_LAUNCH(M)                                                                                          \
    do {                                                                                                           \
        if (compact) EVK_IWE_LAUNCH_C(M, true);                                                                    \
        else EVK_IWE_LAUNCH_C(M, false);                                                                           \
    } while (0)
    if (mode == 0) {
        EVK_IWE_LAUNCH(0);
        if (tall) k_iwe_gather<false, 16><<<ggrid, EVK_BLOCK, 0, s>>>(st, origins, bucket_index, g, slices, win_w, win_h, canvas_h,
                                                       canvas_w, gb, 1, out_iwe, out_diwe, spill, spill_clean);
        else k_iwe_gather<false, 8><<<ggrid, EVK_BLOCK, 0, s>>>(st, origins, bucket_index, g, slices, win_w, win_h, canvas_h,
                                                       canvas_w, gb, 1, out_iwe, out_diwe, spill, spill_clean);
    } else if (mode == 1) {
        EVK_IWE_LAUNCH(1);
        k_iwe_gather<true, 8><<<ggrid, EVK_BLOCK, 0, s>>>(st, origins, bucket_index, g, slices, win_w, win_h, canvas_h,
                                                      canvas_w, gb, 1, out_iwe, out_diwe, spill, spill_clean);
    } else {   // three flows: three one-plane gathers (grid.y = flow) over the windows 3 w + flow
        EVK_IWE_LAUNCH(2);
        k_iwe_gather<false, 8><<<dim3(ggrid, 3), EVK_BLOCK, 0, s>>>(st, origins, bucket_index, g, slices, win_w, win_h, canvas_h,
                                                                 canvas_w, gb, 3, out_iwe, out_diwe, spill, spill_clean);
    }
#undef EVK_IWE_LAUNCH
#undef EVK_IWE_LAUNCH_C
    return launch_status();
}

extern "C" int evk_iwe_linvel_tiled_f32(const float *records, const uint32_t *bucket_index, int64_t n, int dom_h,
                                        int dom_w, int tw_log2, int th_log2, int slices, int win_w, int win_h,
                                        double t_first, double t_ref, double vx, double vy, double bounds_w,
                                        double bounds_h, int canvas_h, int canvas_w, uint32_t flags, double p_scale,
                                        double p_bound, double dt_bound, void *staging, int64_t staging_bytes, float *iwe,
                                        float *diwe, void *stream) {
    return launch_iwe_tiled((flags & EVK_IWE_GRADIENT) ? 1 : 0, records, bucket_index, n, dom_h, dom_w, tw_log2, th_log2,
                            slices, win_w, win_h, t_first, t_ref, &vx, &vy, bounds_w, bounds_h, canvas_h, canvas_w,
                            flags, p_scale, p_bound, dt_bound, staging, staging_bytes, iwe, diwe, stream);
}

extern "C" int evk_iwe_linvel_tiled_batch3_f32(const float *records, const uint32_t *bucket_index, int64_t n, int dom_h,
                                               int dom_w, int tw_log2, int th_log2, int slices, int win_w, int win_h,
                                               double t_first, double t_ref, const double *host_vx,
                                               const double *host_vy, double bounds_w, double bounds_h, int canvas_h,
                                               int canvas_w, uint32_t flags, double p_scale, double p_bound, double dt_bound,
                                               void *staging, int64_t staging_bytes, float *iwe3, void *stream) {
    if (!host_vx || !host_vy || !iwe3 || (flags & EVK_IWE_GRADIENT)) return EVK_EINVAL;
    return launch_iwe_tiled(2, records, bucket_index, n, dom_h, dom_w, tw_log2, th_log2, slices, win_w, win_h, t_first,
                            t_ref, host_vx, host_vy, bounds_w, bounds_h, canvas_h, canvas_w, flags, p_scale, p_bound, dt_bound,
                            staging, staging_bytes, iwe3, iwe3 + (size_t)canvas_h * canvas_w, stream);
}

// evk_cmax.hip: the tiled IWE with a spill pair (no memset; see launch_iwe_tiled)
int evk_iwe_tiled_spill(int mode, const float *records, const uint32_t *bucket_index, int64_t n, int dom_h, int dom_w,
                        int tw_log2, int th_log2, int slices, int win_w, int win_h, double t_first, double t_ref,
                        const double *vx, const double *vy, double bounds_w, double bounds_h, int canvas_h, int canvas_w,
                        uint32_t flags, double p_scale, double p_bound, double dt_bound, void *staging, int64_t staging_bytes,
                        float *iwe_buf, const float *spill, float *spill_clean, void *stream) {
    return launch_iwe_tiled(mode, records, bucket_index, n, dom_h, dom_w, tw_log2, th_log2, slices, win_w, win_h, t_first,
                            t_ref, vx, vy, bounds_w, bounds_h, canvas_h, canvas_w, flags, p_scale, p_bound, dt_bound, staging,
                            staging_bytes, iwe_buf, iwe_buf + (size_t)canvas_h * canvas_w, stream, spill, spill_clean);
}
