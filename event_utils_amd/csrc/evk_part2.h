// One-pass partition shared by the voxel grid (evk_voxel2.hip) and the event images (evk_image2.hip): the sub-chunk sort
// kernel k_part_sorted, the layout of its index / table / runs, and the host-side geometry.  See evk_voxel2.hip for the
// design notes of the voxel path; the image record formats are described at k_part_sorted below.
#pragma once
#include <cstdio>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "evk_part.h"

namespace evk {

#define V2_HDR 16            // [0] t_first bits, [1] t_last bits, [2] ticket, [3] events with a wide polarity (info),
                             // [4] contributions the deterministic mode refused, [5] events whose polarity is not +1, -1 or
                             // +0 (cumulative), [6] its value after the previous call, [7] 1 if THIS call had any,
                             // [8] ticket of the EARLY report of dropped events (round 5), [9..15] spare
#define V2_MAX_TILES 2048    // totals live at a FIXED offset so that they are zero again after every call
#define V2_TOTALS V2_HDR
// words of the LIVE hand-over (evk_voxel_live.h), at a FIXED place whatever the tiling and the event count -- a buffer that
// serves calls of several shapes must never show these kernels a word that something else wrote there:
// progress[256] (one per partition workgroup), status[V2_MAX_TILES] (one per tile)
#define V2_LIVE_PROGRESS (V2_HDR + V2_MAX_TILES)
#define V2_LIVE_STATUS (V2_LIVE_PROGRESS + 256)
#define V2_PART (V2_LIVE_STATUS + V2_MAX_TILES)    // part_start[T + 1]
#define V2_COUNTER(T) (V2_PART + (T) + 1)          // counters[T]   (split-tile combine)
#define V2_ITEM(T) (V2_PART + 2 * (T) + 1)         // item_tile[max_items]
#ifndef V2_LB
#define V2_LB 10  // bits of the pixel-in-tile field (tiles of <= 2^V2_LB pixels); the polarity keeps 32 - V2_LB - 1 bits
#endif
#define EVK_VOXEL2_COUNT (1 << 20)   // kernel-internal flag: the launch has the LDS of the counting mode (k_voxel_tiles2)
#define EVK_VOXEL2_COUNT2 (1 << 21)  // kernel-internal flag: unit polarities are counted in the B planes of the float64 mode
#define V2_LOCAL_MASK ((1u << V2_LB) - 1u)
#define V2_WIDE (1u << V2_LB)
#define V2_P_MASK (~((2u << V2_LB) - 1u))
// ablation builds (tools/v2_ablate.sh): stop the partition kernel's per-sub-chunk work after stage A (0 loads, 1 ranks,
// 2 scan + table, 3 placement, 4 = everything) / the tile kernel's after stage B (0 table entries, 1 record loads,
// 2 decode, 3 = everything).  Results are wrong below the last stage; timing only.
// waves per SIMD the tile kernel must fit: 6 (<= 80 registers; 3 workgroups of 8 waves per CU) with 8-byte records, 4 (128
// registers) with 4-byte records, and the table entries a lane takes per batch (see the kernel)
#ifndef V2_TILES_WAVES
#define V2_TILES_WAVES(REC) ((REC) == 4 ? 4 : 6)
#endif
#ifndef V2_STORE_SC1
#define V2_STORE_SC1 1   // (A/B) write-through stores for the runs of 8-byte records
#endif
#ifndef V2_ENT
#define V2_ENT(REC) ((REC) == 4 ? 3 : 1)
#endif
#ifndef V2_XY_PREFETCH
#define V2_XY_PREFETCH 1   // load x, y of sub-chunk j + 1 before the placement of j (else at the top of j + 1)
#endif
#ifndef V2_NT_COLUMNS
#define V2_NT_COLUMNS true   // (A/B) nontemporal loads of the event columns (evk_part.h, load_col16)
#endif
#ifndef V2_PLACE_BATCH
#define V2_PLACE_BATCH 1   // (A/B) the placement's returning LDS atomics all issued before the first record is built
#endif
#ifndef V2_ABLATE_A
#define V2_ABLATE_A 99
#endif
#ifndef V2_ABLATE_B
#define V2_ABLATE_B 99
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release, for which the
// compiler drains this wave's outstanding GLOBAL stores (s_waitcnt vmcnt(0)): a full store round trip at every barrier,
// and no load can be in flight across it.  Inside these kernels only LDS is shared between the waves of a workgroup.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t mine, uint32_t *tmp, uint32_t &total) {
    constexpr int NW = THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) tmp[wave] = incl;
    lds_barrier();
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
    lds_barrier();
    total = tmp[64];
    return tmp[32 + wave] + incl - mine;   // the caller puts a barrier before tmp is used again
}

// -DV2_PHASE_TIMING (experiments builds): every wave adds the shader cycles (s_memtime) it spends in each phase of a
// sub-chunk pass to v2_phase_cycles[]; evk_debug_phase_cycles() reads and clears them (tools/phase_timing.py)
#ifdef V2_PHASE_TIMING
__device__ unsigned long long v2_phase_cycles[16];
#define V2_T0()                                            \
    unsigned long long pt_ = __builtin_readcyclecounter(); \
    unsigned long long pa_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define V2_T(i)                                                     \
    do {                                                            \
        const unsigned long long n_ = __builtin_readcyclecounter(); \
        pa_[i] += n_ - pt_;                                         \
        pt_ = n_;                                                   \
    } while (0)
#define V2_TEND()                                                                   \
    do {                                                                            \
        if ((threadIdx.x & 63) == 0)                                                \
            for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&v2_phase_cycles[i_], pa_[i_]); \
    } while (0)
// (the same for the tile kernel's phases, v2_tile_cycles[])
__device__ unsigned long long v2_tile_cycles[16];
#define V2_U(i) V2_T(i)
#define V2_UEND()                                                                  \
    do {                                                                           \
        if ((threadIdx.x & 63) == 0)                                               \
            for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&v2_tile_cycles[i_], pa_[i_]); \
    } while (0)
#else
#define V2_T0() do {} while (0)
#define V2_T(i) do {} while (0)
#define V2_TEND() do {} while (0)
#define V2_U(i) do {} while (0)
#define V2_UEND() do {} while (0)
#endif

// floats of staging per work item (the partial tile of a cut tile's piece): a multiple of 32 = whole 128-byte lines
__host__ __device__ static inline int64_t v2_staging_stride(int64_t cells) { return (cells + 31) & ~(int64_t)31; }

struct Part2 {
    int S;          // events per sub-chunk (% 4 == 0, <= THREADS * EPT)
    int per_block;  // consecutive sub-chunks per partition block
    int nsc;        // sub-chunks in the stream
    int nt_pad;     // table row stride: table[sub-chunk][tile]
    int nblk;
};

// C = column source (evk_part.h): SrcF32, or SrcNative<.> for the reference's on-disk dtypes.  G = C::G consecutive events
// per lane and load instruction.
//
// REC = bytes per record.  8: {t_norm, polarity | cell} as described at the top.  4 (round 3): ONE word,
//     [31:12] t_norm as a bit-pattern DELTA from the sub-chunk's first event   [11:10] polarity code   [9:0] cell
// * t_norm is carried exactly: the events of a sub-chunk are consecutive in a time-sorted stream, so their t_norm values lie
//   within a few thousand float32 steps of the first one's (`bases[sub-chunk]` holds its bit pattern); delta < 2^20.
// * polarity code 0 / 1 / 2 = +1.0 / -1.0 / +0.0 -- what the reference's loaders and the bool / uint8 files produce.
// * anything else (another polarity, a delta out of range: unsorted or sparse streams, the first events of a stream where
//   float32 steps are tiny, NaN from dt == 0) ESCAPES: code 3, the delta field holds the index of an exact 8-byte
//   {t_norm bits, polarity bits} entry in the sub-chunk's slice of the side array.  Exact for any float32 input.
// The partition is bound by the bytes it moves (5 TB/s of 240 MB): 20 B/event instead of 24 is worth 5 us of 47 at 10 M
// events and 36 of 248 us at 50 M (measured by writing half of every run), the tile kernel then reads 4 B/event instead of 8.
//
// EVENT IMAGES (round 4; evk_image2.hip) go through the same kernel with two more record formats and column sources
// without a time column (evk_image2.hip: SrcImgF32, SrcImgI32):
// REC = V2_FMT_IMGN (nearest pixel, image.py:28-38 / :87-95): ONE word per record = the hi word of the 8-byte voxel record,
//     [31:11] the top 21 bits of the weight (float32 bits, or a signed integer that fits)   [10] wide   [9:0] cell
//   a wide weight goes, exactly, to the side array at the record's index.  12 B/event read, 4 written.
// REC = V2_FMT_IMGB (bilinear splat, image.py:79-86,102-115): the record run holds {x - tile x0, y - tile y0} as float32
//   (8 bytes: both differences are EXACT -- multiples of ulp(x) below x -- and so are the fractions and the pixel inside the
//   tile taken from them: floor(x - x0) = floor(x) - x0).  Both are non-negative, so their two SIGN bits carry the weight's
//   code -- 0 / 1 / 2 = +1.0 / -1.0 / +0.0, 3 = another value, found in the side run (the exact float32 weights of the
//   sub-chunk at the records' indices, written only when one of them is not +-1 / +0, as for IMGN).  12 B/event read, 8
//   written (12 with arbitrary weights); nothing is quantised.  Events the splat cannot take
//   from an LDS window (negative or out-of-range pixels, which wrap or raise in index_put_; non-finite coordinates) are
//   handed to the column source's `rare()` -- the direct kernel's global atomics -- by this kernel itself.
#define V2_FMT_IMGN 1
#define V2_FMT_IMGB 12
// REC = V2_FMT_IMGT (round 6: the average-timestamp images, image.py:219-353 -- four bilinear splats per event): the IMGB
//   record, its two sign bits carrying the event's CLASS instead of a weight code -- 0 = positive, 1 = non-positive polarity,
//   2 = neither (NaN: contributes nothing) -- and the side run, always written, the event's normalised time stamp (the
//   column source's w_bits).  16 B/event read, 12 written.  Rare events go to the column source's rare_ts().
#define V2_FMT_IMGT 13
// REC = V2_FMT_IMGX (round 6: interpolate_to_image on caller-computed pixels and fractions, image.py:102-115): the IMGB record
//   and side run; the only difference is the rare path -- an event the record cannot carry (its pixel + fraction is not a
//   float32 value, or a pixel that wraps / raises) is re-read by its INDEX in the stream (the column source's rare_at()).
#define V2_FMT_IMGX 14
// REC = V2_FMT_IMGD (round 6: the derivative splats -- interpolate_to_derivative_img, image.py:117-136, and events_to_image_drv,
//   image.py:162-217): the bilinear record {x - tile x0, y - tile y0}, and as side run the event's INDEX in the stream: the tile
//   kernel fetches the event's weights (four or five values: more than a sub-chunk's LDS could stage) from the caller's columns by
//   that index.  Rare events are re-read by their index (rare_at), as IMGX's.  n < 2^32.
#define V2_FMT_IMGD 15
// REC = V2_FMT_VOX8W: the 8-byte voxel records, with the EXACT polarities of a sub-chunk staged in LDS (4 more bytes per
// event) and written as a second dense run when one of them is wide -- instead of a scattered 4-byte store per wide
// polarity (arbitrary float32 weights: partition 78 -> ~50 us at 10 M events).  The geometry with 8 K-event sub-chunks
// only (100 KB of LDS; 12 K-event sub-chunks would need 147 KB), and not when the call shares its CUs with a collective.
#define V2_FMT_VOX8W 9
#define V2_DELTA_SHIFT 12
#define V2_DELTA_LIMIT (1u << 20)
#define V2_CODE_SHIFT 10
// bytes of LDS per event of the sorted buffer
__host__ __device__ constexpr int v2_fmt_lds_bytes(int rec) { return rec == V2_FMT_IMGN ? 8 : ((rec == V2_FMT_VOX8W || rec == V2_FMT_IMGT || rec == V2_FMT_IMGX || rec == V2_FMT_IMGD) ? 12 : rec); }
// LIVE (round 5; evk_voxel_live.hip): the runs are consumed WHILE the partition is still sorting, by a second kernel on a
// second stream (k_voxel_live: two tiles per workgroup, one workgroup per CU beside this kernel's).  What that needs here:
// the table row of a sub-chunk leaves with its run, as write-through 16-byte stores out of an LDS copy (plain 4-byte stores
// from the scan stay in this XCD's L2 until the kernel ends); and when every wave has drained the stores of run j -- the
// wait in front of pass j + 1's placement, then that pass's closing barrier -- one lane publishes
// {epoch, runs of this workgroup that are out} in `live_progress[blockIdx.x]` (an agent-scope store: the hand-over form
// "write-through payload, drained, flag" of MI355X_MICROARCH.md).  Nothing else changes; without LIVE the code is what it was.
template <int THREADS, int EPT, int REC, typename C, bool LIVE = false>
__global__ void __launch_bounds__(THREADS, 4) k_part_sorted(const C c, int64_t n, TileGridG g, int ntiles, Part2 q, float t_first,
                                                            float t_last, float bm1, int t_from_events,
                                                            void *__restrict__ rec_, void *__restrict__ side_,
                                                            uint32_t *__restrict__ bases,
                                                            uint32_t *__restrict__ table, uint32_t *__restrict__ index,
                                                            uint32_t cap, uint32_t part, uint32_t *oob, uint32_t *host_report,
                                                            uint32_t seq, uint32_t *live_progress = nullptr,
                                                            uint32_t live_epoch = 0) {
    static_assert(!LIVE || (REC == 8 && V2_STORE_SC1), "the live consumer reads 8-byte records written through");
    static_assert(REC == 8 || REC == 4 || REC == V2_FMT_IMGN || REC == V2_FMT_IMGB || REC == V2_FMT_IMGT || REC == V2_FMT_IMGX || REC == V2_FMT_IMGD || REC == V2_FMT_VOX8W, "record format");
    constexpr bool R8 = REC == 8 || REC == V2_FMT_VOX8W;      // 8-byte voxel records
    constexpr bool IMGBT = REC == V2_FMT_IMGB || REC == V2_FMT_IMGT || REC == V2_FMT_IMGX || REC == V2_FMT_IMGD;   // bilinear formats: {x, y} relative to the tile + a side run
    constexpr bool STAGE_W = REC == V2_FMT_VOX8W || REC == V2_FMT_IMGN || IMGBT;   // exact weights staged in LDS, dense side run on demand
    constexpr bool VOX = R8 || REC == 4;                      // voxel formats: a time column, t_norm in the record
    constexpr int LB = v2_fmt_lds_bytes(REC);
    uint2 *const rec = static_cast<uint2 *>(rec_);            // REC 8 / IMGB: 8-byte records | REC 4 / IMGN: viewed as uint32 below
    float *const pw = static_cast<float *>(side_);            // REC 8: exact polarity at the record's index
    uint2 *const wide2 = static_cast<uint2 *>(side_);         // REC 4: exact {t_norm, polarity} of an escaped record
    constexpr int G = C::G, NG = EPT / G;
    static_assert(EPT % G == 0, "events per thread");
    constexpr int PER_MAX = (V2_MAX_TILES + THREADS - 1) / THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2 *sorted = reinterpret_cast<uint2 *>(smem);                           // [THREADS * EPT] records + a trash slot
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + (size_t)THREADS * EPT * LB + 16);  // [ntiles] counts of the current pass
    uint32_t *cur = hist + ((ntiles + 4) & ~3);                                 // [ntiles] cursors of the current pass
    uint32_t *tot = cur + ((ntiles + 4) & ~3);                                  // [ntiles] this workgroup's totals
    uint32_t *tmp = tot + ((ntiles + 4) & ~3);                                  // [72] scan scratch; [67] escapes of the pass;
                                                                                // [65] IMGN: the pass has a wide weight; [66], [68]:
                                                                                // the workgroup's wide / non-unit weights
    uint32_t *trow_l = tmp + 72 + 4;                                            // LIVE: [nt_pad] the table row of the current pass (16-byte aligned)
    uint32_t *sorted4 = reinterpret_cast<uint32_t *>(smem);                     // REC 4 / IMGN: the same buffer, one word per record
    // IMGB: the weights, behind the 8-byte records; IMGN: the exact weights, behind the one-word records
    uint32_t *sortedp = reinterpret_cast<uint32_t *>(smem + (size_t)THREADS * EPT * (REC == V2_FMT_IMGN ? 4 : 8));
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t dropped = 0, nwide = 0;   // nwide: wide / escaped records in the low half, polarities other than +-1, +0 in the high
    float tb = 0.0f;   // REC 4: time stamp of the sub-chunk's first event
    if constexpr (VOX) {
        if (t_from_events) t_first = c.t1(0), t_last = c.t1(n - 1);   // ts[0], ts[-1] (voxel_grid.py:133)
    }
    const TimeNorm tnorm = make_time_norm(t_first, t_last, bm1);
    // IMGT: the two constants of the normalised time stamp -- the column source's, or (t_from_events: ts[0] / ts[-1] read HERE, no
    // device-to-host transfer before the launch) what image.py:326-329 forms from the stream's ends
    float ts_a = 0.0f, ts_d = 1.0f;
    if constexpr (REC == V2_FMT_IMGT) c.time_constants(n, t_from_events, ts_a, ts_d);
    // EARLY REPORT of dropped events (round 5).  A synchronous caller (EVK_ERRORS=strict, the default: the reference raises
    // before it returns, image.py:96-99) waits for {seq, dropped events} in its pinned slot, then prepares its next call while
    // the tile kernel runs.  The count is final long before this kernel ends: a workgroup's dropped events are known with the
    // keys of its last sub-chunk.  So ONE lane per workgroup adds the workgroup's count to *oob at that point -- a returning
    // atomic, so that its ticket is taken behind it -- and the last of these tickets publishes the report: most of a pass
    // and the plan's tail earlier than the end of the kernel, which is what lets the host's ~20 us between two calls
    // disappear behind the tile kernel.  (Not the bilinear image format: its rare path can still drop events in the placement.)
    constexpr bool EARLY = !IMGBT;
    uint32_t early_prev = 0;   // lane 0: what its early-report ticket returned
    // (three steps, none of which makes a wave wait where the others need it: the count leaves as a no-return atomic behind the
    // last pass's histogram barrier; the ticket is taken behind that pass's "everything outstanding has landed" wait, i.e. when
    // the count has been performed; its answer is looked at only after the placement)
    auto early_publish = [&]() {
        EVK_HANDOVER_ACQUIRE();
        __hip_atomic_store(index + 8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (host_report) {   // (an optional argument of its own: {seq, 0} for a caller that passes no counter)
            const uint32_t cnt = oob ? __hip_atomic_load(oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(host_report),
                               (unsigned long long)seq | ((unsigned long long)cnt << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };

    // Group k of sub-chunk sc = G consecutive events of thread tid.  A group that is only partly inside the stream is
    // loaded whole (the over-read stays inside an aligned block; the extra events are ignored); a group entirely outside
    // re-reads the sub-chunk's first group.  No branches: between the loads they cost waits and registers.
    // (tl_ = the thread index as the LOOP BODY sees it: re-materialised through an empty asm in every iteration, so that the
    // per-group offsets and predicates derived from it are recomputed -- a few VALU instructions -- instead of being hoisted
    // out of the loop into registers that stay live across it.)
    int tl_ = tid;
    auto valid_in = [&](int sc, int k) -> int {  // events of group k inside the stream (<= 0: none)
        const int64_t lo = (int64_t)sc * q.S;
        const int64_t hi = (lo + q.S < n) ? lo + q.S : n;
        return (int)(hi - lo) - G * (tl_ + k * THREADS);
    };
    // Addresses are (uniform base of the row of THREADS groups) + (one 32-bit lane offset): scalar registers and the
    // saddr form of the load, not a 64-bit VGPR pair per load.
    auto row_base = [&](int sc, int k) -> int64_t {   // first event of group row k; a row entirely outside: the first row
        const int64_t lo = (int64_t)sc * q.S, hi = (lo + q.S < n) ? lo + q.S : n, r = lo + (int64_t)G * k * THREADS;
        return r < hi ? r : lo;
    };
    // Software pipeline over the two halves of an event: x, y are needed first (tile key, histogram), t, p only at
    // placement.  t, p of sub-chunk j are loaded after its keys and land during its histogram + scan; x, y of j + 1 are
    // loaded before the placement of j and land during its placement.  (gfx9 counts loads and stores with ONE counter
    // that is in order only among loads, so a wave with stores in flight cannot wait for a particular load: the first use
    // of loaded data waits for everything outstanding -- hence the explicit wait points below, at moments when everything
    // outstanding is old.)
    const int sc0 = blockIdx.x * q.per_block;
    const int sc_end = (sc0 + q.per_block < q.nsc) ? sc0 + q.per_block : q.nsc;
    uint32_t xyr[NG * C::XYW], tpr[NG * C::TPW];   // RAW loaded words: decoded where they are used (evk_part.h)
    float tv[EPT];
    uint32_t kl[EPT];
    float xr[EPT], yr[EPT];   // IMGB: x, y relative to the tile's origin (of a rare event: x, y themselves)
    uint32_t rare = 0;        // IMGB: events of this pass that go to the column source's rare() (bit = event of the thread)
    auto load_xy = [&](int sc) {
#pragma unroll
        for (int k = 0; k < NG; ++k) c.template load_xy<V2_NT_COLUMNS>(row_base(sc, k), valid_in(sc, k) > 0 ? (uint32_t)tl_ : 0u, xyr + C::XYW * k);
    };
    auto load_tp = [&](int sc) {
#pragma unroll
        for (int k = 0; k < NG; ++k) c.template load_tp<V2_NT_COLUMNS>(row_base(sc, k), valid_in(sc, k) > 0 ? (uint32_t)tl_ : 0u, tpr + C::TPW * k);
        if constexpr (REC == 4) tb = c.t1((int64_t)sc * q.S);   // base of the t_norm deltas (same address in every lane)
    };
    auto fence = [&]() {   // for the compiler: loads hoisted above a compute phase keep their 2 * EPT registers live through it
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int i = tid; i < ntiles; i += THREADS) hist[i] = 0, tot[i] = 0;   // (every pass leaves hist zero again)
    if (tid < 8) tmp[64 + tid] = 0;
    if (sc0 < sc_end) load_xy(sc0);
    EVK_WAIT_VM0();
    lds_barrier();
    // One pass per sub-chunk.  Barriers: histogram | scan (wave totals) | cursors | placement -- four, and NONE at the end of
    // a pass: the sorted sub-chunk is written out at the START of the next pass (its LDS reads come before this wave's
    // keys; the buffer is rewritten only two barriers later), so the store burst -- 64 KB per CU, which the memory pipeline
    // takes at ~10 B/clk -- overlaps the key computation of the waves that got their stores in, instead of every wave
    // waiting for the last one at a closing barrier.  (Per-phase shader cycles of the round-2 order, tools/phase_timing.py:
    // write-out 11.3 us + closing barrier 14.5 us of a 52 us kernel; a scan by one wave 13 us.)
    uint32_t kept_prev = 0;   // records of the previous pass's run (uniform)
    int64_t lo_prev = 0;
    int sc_prev = 0;          // LIVE: its sub-chunk
    // one contiguous, coalesced run of `n16` 16-byte pieces from the sorted buffer
    auto store_run = [&](const uint4 *src, uint4 *dst, const int n16) {
        if constexpr (REC != 4 && V2_STORE_SC1) {
            // 8-byte records = cache-resident calls: WRITE-THROUGH (sc1) stores, through a buffer descriptor of this run.
            // Streaming ("nt") stores keep their lines in the XCD's L2, and what a kernel leaves dirty there is written
            // back at the kernel boundary behind it (MI355X_MICROARCH.md: + B / 6 TB/s for B bytes left dirty): the
            // boundary to the tile kernel was ~5 us instead of ~2 -- the whole call 0.0711 -> 0.068 ms at 10 M events.
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void *)dst, 0, (int)__builtin_amdgcn_readfirstlane(n16 * 16), 0x00020000);
            for (int i = tid; i < n16; i += THREADS) {
                const uint4 v = src[i];
                __builtin_amdgcn_raw_buffer_store_b128(u4v{v.x, v.y, v.z, v.w}, rs, i * 16, 0, /* sc1 */ 16);
            }
        } else {
            // 4-byte records = HBM-resident calls (write-through stores cost the partition 5 % there: 219 against 209 us at
            // 50 M events, and the boundary is 1 % of that call)
#ifndef V2_REC4_PLAIN_STORES
// Round 6: PLAIN 16-byte stores for the runs of 4-byte records.  Streaming ("nt") stores had been measured against plain ones
// on the partition kernel alone (-5 % at 50 M events) with the tile kernel timed BEHIND it on records that were still in the
// Infinity Cache either way; in the call itself (rocprofv3 over 300 back-to-back calls, tools/c5_loop.py) the streamed runs
// are gone from the cache when the tile kernel asks for them: k_voxel_tiles2 104.7 us behind streaming stores, 92.2 us behind
// plain ones, the partition 200 us either way (the columns keep their nontemporal loads: with plain loads both kernels lose).
#define V2_REC4_PLAIN_STORES 1
#endif
            for (int i = tid; i < n16; i += THREADS) {
                const uint4 v = src[i];
                if (V2_REC4_PLAIN_STORES) {
                    dst[i] = v;
                } else {
                    __builtin_nontemporal_store(v.x, &dst[i].x), __builtin_nontemporal_store(v.y, &dst[i].y);
                    __builtin_nontemporal_store(v.z, &dst[i].z), __builtin_nontemporal_store(v.w, &dst[i].w);
                }
            }
        }
    };
    auto write_out = [&]() {   // the previous pass's run(s) of records
        if (V2_ABLATE_A >= 4) {
            const uint4 *src = reinterpret_cast<const uint4 *>(sorted);
            if constexpr (R8 || IMGBT)
                store_run(src, reinterpret_cast<uint4 *>(rec + lo_prev), (int)((kept_prev + 1) >> 1));
            else
                store_run(src, reinterpret_cast<uint4 *>(reinterpret_cast<uint32_t *>(rec_) + lo_prev), (int)((kept_prev + 3) >> 2));
            if constexpr (LIVE)   // the run's table row (LDS copy, whole 16-byte pieces: nt_pad % 16 == 0), written through as well
                store_run(reinterpret_cast<const uint4 *>(trow_l), reinterpret_cast<uint4 *>(table + (int64_t)sc_prev * q.nt_pad), q.nt_pad >> 2);
            if constexpr (STAGE_W) {
                // the exact weights of the run, as a second run at the records' indices -- only when one of them does not fit
                // its record (tmp[65], set by the placement; cleared by the next pass's scan, i.e. after every wave has been
                // here): weights that need all 32 bits cost 4 more bytes per event, +-1 and small integers nothing
                if (tmp[65])
                    store_run(reinterpret_cast<const uint4 *>(sortedp),
                              reinterpret_cast<uint4 *>(static_cast<uint32_t *>(side_) + lo_prev), (int)((kept_prev + 3) >> 2));
            }
        }
    };
    V2_T0();
    for (int sc = sc0; sc < sc_end; ++sc) {
        asm volatile("" : "+v"(tl_));
        const int64_t lo = (int64_t)sc * q.S;
        if (sc > sc0) write_out();   // the previous pass's run (everything this wave loaded has landed: no load is in flight)
        V2_T(0);
        // ---- tile key + accumulator cell of every event
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int nv = valid_in(sc, k);
#pragma unroll
            for (int e = 0; e < G; ++e) {
                uint32_t cell = 0;
                int key;   // (of stale words beyond the stream)
                if constexpr (IMGBT) key = c.key_rel(xyr + C::XYW * k, e, g, xr[G * k + e], yr[G * k + e]);
                else key = c.key_of(xyr + C::XYW * k, e, g, cell);
                kl[G * k + e] = ((key >= 0) & (e < nv)) ? (((uint32_t)key << V2_LB) | cell) : 0xFFFFFFFFu;
                // image sources: -1 = outside the image (counted: the reference raises), -2 = contributes nothing, -3 = rare
                if constexpr (VOX) dropped += ((key < 0) & (e < nv)) ? 1u : 0u;
                else dropped += ((key == -1) & (e < nv)) ? 1u : 0u;
                if constexpr (IMGBT) rare |= ((key == -3) & (e < nv)) ? (1u << (G * k + e)) : 0u;
                asm volatile("" : "+v"(dropped));   // counted HERE: sunk to the end of the loop body it kept a copy of every key alive
                // one event at a time: GCN issues dependent VALU instructions back to back, while interleaving the EPT
                // independent chains (what the scheduler does for ILP) keeps ~4 temporaries per event live at once
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < EPT; ++s2) asm volatile("" : "+v"(kl[s2])::"memory");  // keys first, the t, p loads after
        load_tp(sc);    // land during the histogram and the scan
        V2_T(1);
        // ---- histogram (no-return LDS atomics; hist is zero: the scan of the previous pass left it so)
#pragma unroll
        for (int s2 = 0; s2 < EPT; ++s2)
            if (kl[s2] != 0xFFFFFFFFu)
                __hip_atomic_fetch_add(&hist[kl[s2] >> V2_LB], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (EARLY) {
            // this workgroup's dropped events are all counted once the keys of its LAST sub-chunk are (above)
            if (sc == sc_end - 1 && dropped) __hip_atomic_fetch_add(&tmp[69], dropped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        lds_barrier();  // histogram complete
        if constexpr (EARLY) {
            if (sc == sc_end - 1 && tid == 0 && oob && tmp[69])
                __hip_atomic_fetch_add(oob, tmp[69], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        V2_T(2);
        if constexpr (VOX) if (V2_ABLATE_A < 2) {
            uint32_t sink = 0;
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) sink += kl[s2] ^ __float_as_uint(c.t_of(tpr + C::TPW * (s2 / G), s2 % G)) ^ __float_as_uint(c.p_of(tpr + C::TPW * (s2 / G), s2 % G));
            if (sink == 0x12345u) tot[1] = 1;
            for (int i = tid; i < ntiles; i += THREADS) hist[i] = 0;
            if (sc + 1 < sc_end) load_xy(sc + 1);
            EVK_WAIT_VM0();
            lds_barrier();
            continue;
        }
        // ---- exclusive scan of the tile counts -> cursors, the table row, this workgroup's totals; every wave scans 64
        //      tiles, the wave totals meet in LDS (THREADS tiles per round: one round up to 1024 tiles)
        uint32_t kept = 0;
        {
            uint32_t *trow = table + (int64_t)sc * q.nt_pad;
            constexpr int NWV = THREADS / 64;
            const int wave = tid >> 6;
            for (int base = 0; base < ntiles; base += THREADS) {
                const int i = base + tid;
                const uint32_t cnt = i < ntiles ? hist[i] : 0u;
                uint32_t incl = cnt;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t v = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += v;
                }
                if (lane == 63) tmp[wave] = incl;
                lds_barrier();  // wave totals
                const uint32_t wt = lane < NWV ? tmp[lane] : 0u;   // every wave scans the (<= 16) wave totals itself
                uint32_t wi = wt;
#pragma unroll
                for (int off = 1; off < NWV; off <<= 1) {
                    const uint32_t v = __shfl_up(wi, off, 64);
                    if (lane >= off) wi += v;
                }
                const uint32_t carry = kept + __shfl(wi - wt, wave, 64);
                if (i < ntiles) {
                    const uint32_t start = carry + incl - cnt;
                    cur[i] = start;
                    hist[i] = 0;            // zero again for the next pass (this thread is the only one touching it now)
                    tot[i] += cnt;
                    if constexpr (LIVE) trow_l[i] = start | (cnt << 16);
                    else trow[i] = start | (cnt << 16);
                }
                kept += __shfl(wi, NWV - 1, 64);
                if (base + THREADS < ntiles) lds_barrier();   // tmp is reused by the next round
            }
            if (REC == 4 && tid == 0) tmp[67] = 0;   // escapes of this pass
            if (STAGE_W && tid == 0) tmp[65] = 0;   // wide weights of this pass
        }
        lds_barrier();  // cursors complete
        V2_T(3);
        // normalised time, in place (t has landed during the histogram and the scan), one division at a time
        if constexpr (VOX) {
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) {
                tv[s2] = time_norm(c.t_of(tpr + C::TPW * (s2 / G), s2 % G), tnorm);  // voxel_grid.py:134 (evk_part.h)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        V2_T(4);
        // Nothing outstanding from here (t, p are in; the previous run's stores are a histogram and a scan old) -- said with
        // the builtin so that the placement's uses of t, p get no wait of their own: with x, y of the next sub-chunk just
        // issued such a wait is a vmcnt(0), i.e. the full latency of those loads in every placement.
        EVK_WAIT_VM0();
        fence();
        if constexpr (EARLY) {   // (the workgroup's count has been performed: the wait above)
            if (sc == sc_end - 1 && tid == 0) {
                EVK_HANDOVER_RELEASE();
                early_prev = __hip_atomic_fetch_add(index + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (sc + 1 < sc_end) load_xy(sc + 1);  // in flight during the placement
        fence();
        if constexpr (VOX) if (V2_ABLATE_A < 3) {
            uint32_t sink = 0;
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) sink += kl[s2] ^ __float_as_uint(tv[s2]) ^ __float_as_uint(c.p_of(tpr + C::TPW * (s2 / G), s2 % G));
            if (sink == 0x12345u) tot[1] = 1;
            lds_barrier();
            EVK_WAIT_VM0();
            continue;
        }
        // ---- placement: a returning LDS atomic on the tile's cursor hands every event its slot of the sorted buffer, where
        //      its record is built
        // (V2_PLACE_BATCH, round 4: the EPT returning atomics are issued back to back and the records built behind them -- one
        // LDS round trip per pass instead of EPT dependent ones; the compiler cannot do it itself, cursors and sorted buffer
        // may alias for all it knows)
        uint32_t pos_[EPT];
        if constexpr (V2_PLACE_BATCH) {
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) {
                pos_[s2] = 0;
                if (kl[s2] != 0xFFFFFFFFu) pos_[s2] = atomicAdd(&cur[kl[s2] >> V2_LB], 1u);
            }
        }
        auto slot_of = [&](int s2) -> uint32_t {
            if constexpr (V2_PLACE_BATCH) return pos_[s2];
            else return atomicAdd(&cur[kl[s2] >> V2_LB], 1u);
        };
        if constexpr (REC == V2_FMT_VOX8W) {
            bool any_wide = false;
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) {
                if (kl[s2] != 0xFFFFFFFFu) {
                    const uint32_t pos = slot_of(s2);
                    const uint32_t pbits = __float_as_uint(c.p_of(tpr + C::TPW * (s2 / G), s2 % G));
                    const bool wide = ((pbits & ~V2_P_MASK) != 0u) | ((pbits & 0x7F800000u) == 0x7F800000u);
                    nwide += ((((pbits & 0x7FFFFFFFu) == 0x3F800000u) | (pbits == 0u)) ? 0u : 0x10000u) + (wide ? 1u : 0u);
                    sorted[pos] = make_uint2(__float_as_uint(tv[s2]), (wide ? V2_WIDE : (pbits & V2_P_MASK)) | (kl[s2] & V2_LOCAL_MASK));
                    sortedp[pos] = pbits;
                    any_wide |= wide;
                }
            }
            if (any_wide) tmp[65] = 1u;   // (every writer stores the same value)
        } else if constexpr (REC == 8) {
            uint32_t wide_mask = 0;
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) {
                if (kl[s2] != 0xFFFFFFFFu) {
                    const uint32_t pos = slot_of(s2);
                    const uint32_t pbits = __float_as_uint(c.p_of(tpr + C::TPW * (s2 / G), s2 % G));
                    // (a polarity that is not finite is always "wide": the tile kernel treats it in its rare branch)
                    const bool wide = ((pbits & ~V2_P_MASK) != 0u) | ((pbits & 0x7F800000u) == 0x7F800000u);
                    nwide += (((pbits & 0x7FFFFFFFu) == 0x3F800000u) | (pbits == 0u)) ? 0u : 0x10000u;
                    sorted[pos] = make_uint2(__float_as_uint(tv[s2]), (wide ? V2_WIDE : (pbits & V2_P_MASK)) | (kl[s2] & V2_LOCAL_MASK));
                    if (wide) wide_mask |= 1u << s2, kl[s2] = pos;   // kl is dead from here on: keep the slot instead
                }
            }
            if (__any(wide_mask != 0u)) {  // rare: exact float32 polarities go to the side array at the record's index
#pragma unroll
                for (int s2 = 0; s2 < EPT; ++s2)
                    if (wide_mask >> s2 & 1u) pw[lo + kl[s2]] = c.p_of(tpr + C::TPW * (s2 / G), s2 % G), ++nwide;
            }
        } else if constexpr (REC == V2_FMT_IMGN) {
            bool any_wide = false;
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) {
                if (kl[s2] != 0xFFFFFFFFu) {
                    const uint32_t pos = slot_of(s2);
                    bool wide, unit;
                    const uint32_t pay = c.payload(tpr + C::TPW * (s2 / G), s2 % G, wide, unit);
                    nwide += (unit ? 0u : 0x10000u) + (wide ? 1u : 0u);
                    sorted4[pos] = (wide ? V2_WIDE : (pay & V2_P_MASK)) | (kl[s2] & V2_LOCAL_MASK);
                    sortedp[pos] = c.w_bits(tpr + C::TPW * (s2 / G), s2 % G);
                    any_wide |= wide;
                }
            }
            if (any_wide) tmp[65] = 1u;   // (every writer stores the same value)
        } else if constexpr (REC == V2_FMT_IMGB || REC == V2_FMT_IMGX) {
            bool any_b = false;
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) {
                if (kl[s2] != 0xFFFFFFFFu) {
                    const uint32_t pos = slot_of(s2);
                    const uint32_t wb = c.w_bits(tpr + C::TPW * (s2 / G), s2 % G);
                    // +1.0 -> 0, -1.0 -> 1, +0.0 -> 2, anything else -> 3 (in the sign bits of the two coordinates, which are >= 0;
                    // a -0.0 coordinate is stored as +0.0: the same pixel, the same fractions)
                    const uint32_t code = (wb & 0x7FFFFFFFu) == 0x3F800000u ? wb >> 31 : 3u - (uint32_t)(wb == 0u);
                    nwide += code == 3u ? 0x10001u : 0u;
                    sorted[pos] = make_uint2((__float_as_uint(xr[s2]) & 0x7FFFFFFFu) | (code << 31),
                                             (__float_as_uint(yr[s2]) & 0x7FFFFFFFu) | ((code >> 1) << 31));
                    sortedp[pos] = wb;
                    any_b |= code == 3u;
                }
            }
            if (any_b) tmp[65] = 1u;   // (every writer stores the same value)
            if (__any(rare != 0u)) {   // rare: pixels that wrap or raise in index_put_ -- the direct kernel's global atomics
#pragma unroll
                for (int s2 = 0; s2 < EPT; ++s2)
                    if (rare >> s2 & 1u) {
                        if constexpr (REC == V2_FMT_IMGX)   // by its index in the stream (group s2 / G of this thread, event s2 % G)
                            dropped += c.rare_at(row_base(sc, s2 / G) + (int64_t)G * tl_ + (s2 % G)) ? 0u : 1u;
                        else
                            dropped += c.rare(xr[s2], yr[s2], __uint_as_float(c.w_bits(tpr + C::TPW * (s2 / G), s2 % G))) ? 0u : 1u;
                    }
            }
            rare = 0;
        } else if constexpr (REC == V2_FMT_IMGD) {
            bool any_d = false;
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) {
                if (kl[s2] != 0xFFFFFFFFu) {
                    const uint32_t pos = slot_of(s2);
                    sorted[pos] = make_uint2(__float_as_uint(xr[s2]), __float_as_uint(yr[s2]));
                    sortedp[pos] = (uint32_t)(row_base(sc, s2 / G) + (int64_t)G * tl_ + (s2 % G));   // the event's index in the stream
                    any_d = true;
                }
            }
            if (any_d) tmp[65] = 1u;   // the side run of this sub-chunk is always written
            if (__any(rare != 0u)) {
#pragma unroll
                for (int s2 = 0; s2 < EPT; ++s2)
                    if (rare >> s2 & 1u) dropped += c.rare_at(row_base(sc, s2 / G) + (int64_t)G * tl_ + (s2 % G)) ? 0u : 1u;
            }
            rare = 0;
        } else if constexpr (REC == V2_FMT_IMGT) {
            bool any_t = false;
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) {
                if (kl[s2] != 0xFFFFFFFFu) {
                    const uint32_t pos = slot_of(s2);
                    const uint32_t cls = c.cls(tpr + C::TPW * (s2 / G), s2 % G);   // 0 / 1 / 2: positive / non-positive / neither
                    sorted[pos] = make_uint2((__float_as_uint(xr[s2]) & 0x7FFFFFFFu) | ((cls & 1u) << 31),
                                             (__float_as_uint(yr[s2]) & 0x7FFFFFFFu) | ((cls >> 1) << 31));
                    sortedp[pos] = c.nts_bits(tpr + C::TPW * (s2 / G), s2 % G, ts_a, ts_d);   // the normalised time stamp
                    any_t = true;
                }
            }
            if (any_t) tmp[65] = 1u;   // the side run of this sub-chunk is always written
            if (__any(rare != 0u)) {   // rare: pixels that wrap or raise in index_put_ -- the direct kernel's global atomics
#pragma unroll
                for (int s2 = 0; s2 < EPT; ++s2)
                    if (rare >> s2 & 1u) dropped += c.rare_ts(xr[s2], yr[s2], tpr + C::TPW * (s2 / G), s2 % G, ts_a, ts_d) ? 0u : 1u;
            }
            rare = 0;
        } else {
            const uint32_t bbits = __float_as_uint(time_norm(tb, tnorm));
            if (tid == 0) bases[sc] = bbits;
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) {
                const uint32_t tbits = __float_as_uint(tv[s2]), pb = __float_as_uint(c.p_of(tpr + C::TPW * (s2 / G), s2 % G));
                const uint32_t d = tbits - bbits;
                // +1.0 -> 0, -1.0 -> 1, +0.0 -> 2, anything else -> 3 (two selects: a chain of equality tests on one value
                // becomes a switch with divergent branches)
                const uint32_t code = (pb & 0x7FFFFFFFu) == 0x3F800000u ? pb >> 31 : 3u - (uint32_t)(pb == 0u);
                const bool live = kl[s2] != 0xFFFFFFFFu;
                nwide += (live & (code == 3u)) ? 0x10000u : 0u;
                const bool esc = live & ((code == 3u) | (d >= V2_DELTA_LIMIT));
                uint32_t word = (d << V2_DELTA_SHIFT) | (code << V2_CODE_SHIFT) | (kl[s2] & V2_LOCAL_MASK);
                if (__any(esc)) {   // rare, wave-uniform test: the exact pair to the side array, its index into the record
                    if (esc) {
                        const uint32_t e = atomicAdd(&tmp[67], 1u);
                        wide2[lo + e] = make_uint2(tbits, pb);
                        word = (e << V2_DELTA_SHIFT) | (3u << V2_CODE_SHIFT) | (kl[s2] & V2_LOCAL_MASK);
                        ++nwide;
                    }
                }
                if (live) {
                    const uint32_t pos = slot_of(s2);
                    sorted4[pos] = word;
                }
            }
        }
        V2_T(5);
        lds_barrier();   // the sorted sub-chunk is complete
        if constexpr (EARLY) {
            if (sc == sc_end - 1 && tid == 0 && early_prev == gridDim.x - 1) early_publish();
        }
        if constexpr (LIVE) {
            // every wave waited for its stores of the PREVIOUS run (and its table row) in front of this placement: that run is
            // out.  runs published = sc - sc0.
            if (sc > sc0 && tid == 0)
                __hip_atomic_store(live_progress + blockIdx.x, (live_epoch << 8) | (uint32_t)(sc - sc0), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        V2_T(6);
        EVK_WAIT_VM0();   // x, y of the next sub-chunk have landed during the placement: the stores below (next pass, or the
                          // epilogue) then never sit between a load and its use
        V2_T(7);
        kept_prev = kept, lo_prev = lo, sc_prev = sc;
    }
    // ---- totals -> global (the last block to arrive builds the work-item plan), issued AHEAD of the last run's stores so
    //      that the two drain together
    uint32_t *gidx = index;
    if constexpr (EARLY) {
        if (sc0 >= sc_end && tid == 0) {   // a workgroup without a sub-chunk only takes its ticket
            early_prev = __hip_atomic_fetch_add(index + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (early_prev == gridDim.x - 1) early_publish();
        }
    } else {
        if (dropped && oob) atomicAdd(oob, dropped);
    }
    // (131 K atomics at 10 M events / 512 tiles: 1.5 us of the kernel, measured by leaving them out)
    for (int i = tid; i < ntiles; i += THREADS)
        if (tot[i]) __hip_atomic_fetch_add(gidx + V2_TOTALS + i, tot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // wide / non-unit counts: summed over the wave, then over the workgroup in LDS -- ONE global atomic per workgroup and
    // counter (round 4: every THREAD with a count added it itself, 262 K atomics on one address when the weights are
    // arbitrary floats: 45 us at the tail of the kernel)
    {
        uint32_t nw = nwide & 0xFFFFu, nu = nwide >> 16;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nw += __shfl_xor(nw, off, 64), nu += __shfl_xor(nu, off, 64);
        if (lane == 0) {
            if (nw) __hip_atomic_fetch_add(&tmp[66], nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (nu) __hip_atomic_fetch_add(&tmp[68], nu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (sc0 < sc_end) write_out();   // the last pass's run
    lds_barrier();
    if (tid == 0) {
        const uint32_t nw = tmp[66], nu = tmp[68];
        if (nw) __hip_atomic_fetch_add(gidx + 3, nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nu) __hip_atomic_fetch_add(gidx + 5, nu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    V2_T(8);
    V2_T(9);
    V2_T(10);
    if constexpr (VOX) {
        if (blockIdx.x == 0 && tid == 0 && n > 0) {
            __hip_atomic_store(gidx + 0, __float_as_uint(c.t1(0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gidx + 1, __float_as_uint(c.t1(n - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    EVK_HANDOVER_DRAIN();
    if constexpr (LIVE) {   // the last run is out (every wave drained its stores above)
        if (sc0 < sc_end && tid == 0)
            __hip_atomic_store(live_progress + blockIdx.x, (live_epoch << 8) | (uint32_t)(sc_end - sc0), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    // Everything the last block reads from the others -- tile totals, dropped-event count, wide-record count -- was written
    // with AGENT-SCOPE ATOMICS and is read with agent-scope atomic loads: performed at the level all XCDs share, and complete
    // (vmcnt) only when they are.  Every wave has drained its own (the wait above), so the ticket needs NO release / acquire
    // fence -- which on this chip is a write-back of the XCD's whole L2, with the 64 KB run every CU has just stored in it:
    // 4.5 us of a 47 us kernel (round 3, tools/ab.sh: 47.5 -> 43.0 us at 10 M events, 213 -> 208 us at 50 M).  The records,
    // the table and the plan are plain stores for the NEXT kernel: the kernel boundary publishes them.
    if (tid == 0) {
        const uint32_t prev = __hip_atomic_fetch_add(gidx + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == gridDim.x - 1);
    }
    __syncthreads();
    V2_T(11);  // totals, ticket
    V2_TEND();
    if (!is_last) return;
    EVK_HANDOVER_ACQUIRE();
    // ---- plan: part_start, per-tile combine counters, item -> tile; totals / ticket back to 0.  A tile with more than `cap`
    //      events is cut into pieces of at most `part` events (ranges of sub-chunks).
    const int per = (ntiles + THREADS - 1) / THREADS;
    const int i0 = tid * per, i1 = (i0 + per < ntiles) ? i0 + per : ntiles;
    uint32_t *part_start = index + V2_PART, *counters = index + V2_COUNTER(ntiles), *item_tile = index + V2_ITEM(ntiles);
    uint32_t tt[PER_MAX];
    uint32_t np = 0;
    // (this block is the tail of the whole kernel: the three loads its last lines need are issued here, ahead of the scan)
    uint32_t now5 = 0, prev6 = 0, cnt_oob = 0, now3 = 0, prev0 = 0;
    if (tid == 0) {
        now5 = __hip_atomic_load(gidx + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        prev6 = __hip_atomic_load(gidx + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (!VOX) {   // image formats: index[0] = index[3] after the previous call, index[1] = THIS call had wide weights
            now3 = __hip_atomic_load(gidx + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prev0 = __hip_atomic_load(gidx + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!EARLY && host_report && oob) cnt_oob = __hip_atomic_load(oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) {
        const int i = i0 + k;
        tt[k] = 0;
        if (k < per && i < i1) {
            tt[k] = __hip_atomic_load(gidx + V2_TOTALS + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gidx + V2_TOTALS + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            np += tt[k] > cap ? (tt[k] + part - 1) / part : 1u;
        }
    }
    uint32_t total_parts;
    uint32_t prun = block_excl_scan<THREADS>(np, tmp, total_parts);
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) {
        const int i = i0 + k;
        if (k < per && i < i1) {
            const uint32_t parts = tt[k] > cap ? (tt[k] + part - 1) / part : 1u;
            part_start[i] = prun;
            counters[i] = 0;
            for (uint32_t jj = 0; jj < parts; ++jj) item_tile[prun + jj] = (uint32_t)i;
            prun += parts;
        }
    }
    if (tid == 0) {
        part_start[ntiles] = total_parts;
        __hip_atomic_store(gidx + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // did THIS call see a polarity other than +1, -1, +0?  (the tile kernel counts unit polarities with integers)
        // (agent-scope accesses like everything else in this line of the index: its other words take the workgroups' atomics)
        __hip_atomic_store(gidx + 7, now5 != prev6 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gidx + 6, now5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (!VOX) {
            __hip_atomic_store(gidx + 1, now3 != prev0 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gidx + 0, now3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!EARLY && host_report) {  // every workgroup's dropped-event count is in *oob (added before its ticket): tell the host,
                            // in pinned memory, so that a deferred error check costs no copy and no event on the stream
            const uint32_t cnt = cnt_oob;
            // {seq, count} as ONE 8-byte system-scope store: the pair cannot be seen torn, and no release (a write-back of the
            // L2 at the very end of the kernel's critical path) is needed to order two stores
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(host_report),
                               (unsigned long long)seq | ((unsigned long long)cnt << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- host-side geometry -----------------------------------------------------------------------------------------
// Partition geometry = threads x events per thread, ONE workgroup per CU: 1024 x 8 (sub-chunks of 8 K events: 71 registers,
// 68 KB of LDS -- room for the workgroups of another kernel, e.g. an overlapped RCCL collective) and 1024 x 12 (12 K events:
// longer segments for the tile kernel, taken when there are more than 680 tiles and the call need not share its CUs).
// Measured and rejected: 512x32 / 1024x16 (16 K events: the whole register file, 67 / 74 us), two
// workgroups per CU (512x16 56.0 and 768x8 56.8 against 48 us: a second workgroup per CU only makes the sub-chunks shorter).
struct V2Config {
    int threads, ept;
};
#define V2_GEOMETRIES(X) X(1024, 8) X(1024, 12)
static const V2Config &v2_config(bool share = false, int ntiles = 0) {
    static const V2Config small{1024, 8}, large{1024, 12};
#ifndef V2_LARGE_ABOVE
#define V2_LARGE_ABOVE 680   // (A/B) tile count above which the 12 K-event geometry is taken
#endif
    return (share || ntiles <= V2_LARGE_ABOVE) ? small : large;
}
#define V2_MIN_SUBCHUNK 8192
#define V2_LDS_LIMIT (160 * 1024 - 512)   // (the partition kernel also has a few bytes of static LDS)
// LDS of the partition kernel: sorted records | counts | cursors | totals | scan scratch
static size_t v2_part_lds(int threads, int ept, int rec, int ntiles, bool live = false) {
    return (size_t)threads * ept * v2_fmt_lds_bytes(rec) + 16 + 3 * (size_t)((ntiles + 4) & ~3) * 4 + 72 * 4 + 16 +
           (live ? (size_t)((ntiles + 15) & ~15) * 4 + 16 : 0);   // LIVE: the table row's LDS copy (nt_pad words)
}

static Part2 v2_geometry(int64_t n, int ntiles, bool share = false) {
    const V2Config &c = v2_config(share, ntiles);
    const int64_t smax = (int64_t)c.threads * c.ept;
    int64_t nblk = (n + V2_MIN_SUBCHUNK - 1) / V2_MIN_SUBCHUNK;
    const int64_t maxblk = (int64_t)EVK_NUM_CU * (1024 / c.threads);
    if (nblk > maxblk) nblk = maxblk;
    if (nblk < 1) nblk = 1;
    int64_t per_block = (n + nblk * smax - 1) / (nblk * smax);
    if (per_block < 1) per_block = 1;
    int64_t S = (n + nblk * per_block - 1) / (nblk * per_block);
    S = (S + 3) & ~(int64_t)3;
    if (S < 4) S = 4;
    Part2 q;
    q.S = (int)S, q.per_block = (int)per_block, q.nblk = (int)nblk;
    q.nsc = (int)((n + S - 1) / S);
    if (q.nsc < 1) q.nsc = 1;
    q.nt_pad = (ntiles + 15) & ~15;
    return q;
}
static inline int64_t al256(int64_t b) { return (b + 255) & ~(int64_t)255; }

// Hot tiles.  The tile kernel runs one workgroup per work item and lasts as long as its busiest CU, so a tile holding more
// than V2_SPLIT_AT x the mean tile population is cut into pieces of about V2_PART x the mean (ranges of sub-chunks; the last
// piece to arrive sums the partial tiles).  Two numbers, because cutting costs (a staging store, a ticket, the combine --
// measured on the moving-edge scene, where most tiles hold 1.5-3 x the mean: cutting everything above 1.5 x made the kernel
// 30 % slower, 41 -> 53 us) but a blob that holds half of the events in twenty tiles must become many small pieces (cut at
// 4 x into pieces of < 4 x: 111 us; into pieces of ~1 x: see DESIGN.md).  Uniform events are never cut.  (3 / 1.5 until
// the tile kernel counted unit polarities and ran 768 threads; with both a piece is cheaper: 2.5 / 1.25, blob 48.5 -> 43.5 us;
// 2.5 / 1.6 once a wave lists all the chunks of a piece's long segments, in passes: 39.6 us -- and no cliff any more for
// bigger pieces: 2 x: 45, 2.5 x: 46, 3 x: 50, 4 x: 55 us; smaller ones pay their fixed costs: 1.25 x: 41.7 us.)
#define V2_SPLIT_AT 2.5
#define V2_SPLIT_PART 1.6
static int64_t v2_mean(int64_t n, int ntiles) { return n / (ntiles > 0 ? ntiles : 1); }
static int64_t v2_cap(int64_t n, int ntiles) {   // a tile with more events than this is cut ...
    const int64_t c = (int64_t)(V2_SPLIT_AT * (double)v2_mean(n, ntiles));
    return c > 16384 ? c : 16384;
}
static int64_t v2_part(int64_t n, int ntiles) {   // ... into pieces of at most this many
    const int64_t c = (int64_t)(V2_SPLIT_PART * (double)v2_mean(n, ntiles));
    return c > 8192 ? c : 8192;
}
static int v2_max_items(int64_t n, int ntiles) { return ntiles + (int)(n / v2_part(n, ntiles)) + 1; }

// Record size of a call.  4-byte compact records (k_part_sorted) once the call's streams no longer fit the 256 MB Infinity
// Cache (more than 16 M events): there the partition is bound by HBM bytes and 20 instead of 24 B/event make it 15 % faster
// (50 M events: 245 -> 208 us, whole call 0.397 -> 0.366 ms, same box).  Below, the streams are cache-resident and the delta /
// code / escape arithmetic costs what the bytes save, the tile kernel's extra decode 6 % (10 M events: 0.0838 against
// 0.0792 ms): 8-byte records.  EVK_VOXEL2_REC4 / _REC8 in `flags` force one (tests, measurements).
static int v2_rec_bytes(int64_t n, int flags) {
    if (flags & EVK_VOXEL2_REC4) return 4;
    if (flags & EVK_VOXEL2_REC8) return 8;
    return n * 16 > ((int64_t)256 << 20) ? 4 : 8;
}
struct V2Layout {
    int64_t table, bases, rec, pw, staging, total;
};
static V2Layout v2_layout(int ntiles, int64_t n, int planes, int tw, int th, bool share = false) {
    const Part2 q = v2_geometry(n, ntiles, share);
    const int64_t slots = (int64_t)q.nsc * q.S;
    V2Layout L;
    L.table = 0;
    L.bases = al256((int64_t)q.nsc * q.nt_pad * 4);
    L.rec = L.bases + al256((int64_t)q.nsc * 4);
    L.pw = L.rec + al256(slots * 8);        // (sized for either record format: 8 + 4 or 4 + 8 bytes per slot)
    L.staging = L.pw + al256(slots * 8);
    // (8 bytes per cell: the pieces of a cut tile hand over float32 partial tiles, or -- integer accumulators -- exact int64 ones)
    L.total = L.staging + al256((int64_t)v2_max_items(n, ntiles) * v2_staging_stride((int64_t)planes * tw * th) * 8);
    return L;
}

template <int THREADS, int EPT, int REC, typename C, bool LIVE = false>
static void launch_part(const C &c, int64_t n, const TileGridG &g, int ntiles, const Part2 &q, float t_first, float t_last,
                        float bm1, int t_from_events, void *rec, void *pw, uint32_t *bases, uint32_t *table, uint32_t *index,
                        uint32_t *oob, uint32_t *host_report, uint32_t seq, hipStream_t s, uint32_t *live_progress = nullptr,
                        uint32_t live_epoch = 0) {
    const size_t lds = v2_part_lds(THREADS, EPT, REC, ntiles, LIVE);
    static std::once_flag once[64];   // per device and instantiation: the attribute belongs to the loaded code object
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [] {   // (the kernel also has a few bytes of static LDS: ask for less than the full 160 KiB)
        (void)hipFuncSetAttribute((const void *)k_part_sorted<THREADS, EPT, REC, C, LIVE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024 - 256);
    });
    k_part_sorted<THREADS, EPT, REC, C, LIVE><<<q.nblk, THREADS, lds, s>>>(c, n, g, ntiles, q, t_first, t_last, bm1, t_from_events,
                                                                          rec, pw, bases, table, index, (uint32_t)v2_cap(n, ntiles), (uint32_t)v2_part(n, ntiles), oob,
                                                                          host_report, seq, live_progress, live_epoch);
}

}  // namespace evk
