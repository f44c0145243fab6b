// One-pass partition for the voxel grid (events_to_voxel_torch, voxel_grid.py:114-153).
//
// The three-pass counting sort of evk_tiled.hip (histogram -> scan -> scatter) reads every event twice and writes a
// 16-byte record: 581 MB of traffic for 166 MB of algorithmic bytes at 10 M events (profiles/r01_pmc_traffic.json).
// Here every partition workgroup sorts SUB-CHUNKS of <= 12 K consecutive events by tile entirely in LDS and writes each
// sorted sub-chunk back as ONE contiguous, fully coalesced run of 8-byte (or, above 16 M events, 4-byte) records, plus
// one 4-byte (start, count) entry per (sub-chunk, tile).  No global histogram, no scan kernels, no look-back: the events
// are read once (16 B) and written once (8 or 4 B).  The tile kernel (k_voxel_tiles2, below) then pulls its ~100-200 byte
// segments out of the runs, 64-byte chunks handed to groups of 4 lanes, and accumulates in LDS: float64 sums, or -- for
// calls whose polarities are all +1 / -1 / +0, where the LDS allows -- an integer count and one int64 fixed-point sum per
// event (the counting mode).  Tiles have any size (evk_part.h: TileGridG); tiled.voxel2_shape picks one whose tile COUNT is
// a multiple of the 256 CUs.
//
// 8-byte record: lo = float32 t_norm (raw bits); hi = [31:11] the top 21 bits of the float32 polarity, [10] "wide" flag,
// [9:0] accumulator cell inside the tile.  A polarity whose low 11 mantissa bits are zero (+-1, 0, small integers,
// halves, ...: every polarity the reference's loaders produce) is carried exactly; any other value sets the wide flag and
// is stored in a side array at the record's index (rare path, 4 extra bytes for that event only), so the result is exact
// for arbitrary float32 weights.  4-byte record: see k_part_sorted.
//
// Hot tiles (clustered data): every partition block adds its per-tile counts to global totals; the LAST block to
// finish (a relaxed ticket: everything it reads from the others is an agent-scope atomic, no fence) builds the work-item
// plan -- a tile with more than 2.5 x the mean is cut into pieces of 1.25 x the mean, by sub-chunk range.
#include <cstdio>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "evk_part.h"

namespace evk {

#define V2_HDR 8             // [0] t_first bits, [1] t_last bits, [2] ticket, [3] events with a wide polarity (info),
                             // [4] contributions the deterministic mode refused, [5] events whose polarity is not +1, -1 or
                             // +0 (cumulative), [6] its value after the previous call, [7] 1 if THIS call had any
#define V2_MAX_TILES 2048    // totals live at a FIXED offset so that they are zero again after every call
#define V2_TOTALS V2_HDR
#define V2_PART (V2_HDR + V2_MAX_TILES)            // part_start[T + 1]
#define V2_COUNTER(T) (V2_PART + (T) + 1)          // counters[T]   (split-tile combine)
#define V2_ITEM(T) (V2_PART + 2 * (T) + 1)         // item_tile[max_items]
#ifndef V2_LB
#define V2_LB 10  // bits of the pixel-in-tile field (tiles of <= 2^V2_LB pixels); the polarity keeps 32 - V2_LB - 1 bits
#endif
#define EVK_VOXEL2_COUNT (1 << 20)   // kernel-internal flag: the launch has the LDS of the counting mode (k_voxel_tiles2)
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
#define V2_DELTA_SHIFT 12
#define V2_DELTA_LIMIT (1u << 20)
#define V2_CODE_SHIFT 10
template <int THREADS, int EPT, int REC, typename C>
__global__ void __launch_bounds__(THREADS, 4) k_part_sorted(const C c, int64_t n, TileGridG g, int ntiles, Part2 q, float t_first,
                                                            float t_last, float bm1, int t_from_events,
                                                            void *__restrict__ rec_, void *__restrict__ side_,
                                                            uint32_t *__restrict__ bases,
                                                            uint32_t *__restrict__ table, uint32_t *__restrict__ index,
                                                            uint32_t cap, uint32_t part, uint32_t *oob, uint32_t *host_report,
                                                            uint32_t seq) {
    static_assert(REC == 8 || REC == 4, "record size");
    uint2 *const rec = static_cast<uint2 *>(rec_);            // REC 8: 8-byte records | REC 4: viewed as uint32 below
    float *const pw = static_cast<float *>(side_);            // REC 8: exact polarity at the record's index
    uint2 *const wide2 = static_cast<uint2 *>(side_);         // REC 4: exact {t_norm, polarity} of an escaped record
    constexpr int G = C::G, NG = EPT / G;
    static_assert(EPT % G == 0, "events per thread");
    constexpr int PER_MAX = (V2_MAX_TILES + THREADS - 1) / THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2 *sorted = reinterpret_cast<uint2 *>(smem);                           // [THREADS * EPT] records + a trash slot
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + (size_t)THREADS * EPT * REC + 16);  // [ntiles] counts of the current pass
    uint32_t *cur = hist + ((ntiles + 4) & ~3);                                 // [ntiles] cursors of the current pass
    uint32_t *tot = cur + ((ntiles + 4) & ~3);                                  // [ntiles] this workgroup's totals
    uint32_t *tmp = tot + ((ntiles + 4) & ~3);                                  // [68] scan scratch; [67] escapes of the pass
    uint32_t *sorted4 = reinterpret_cast<uint32_t *>(smem);                     // REC 4: the same buffer, one word per record
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t dropped = 0, nwide = 0;   // nwide: wide / escaped records in the low half, polarities other than +-1, +0 in the high
    float tb = 0.0f;   // REC 4: time stamp of the sub-chunk's first event
    if (t_from_events) t_first = c.t1(0), t_last = c.t1(n - 1);   // ts[0], ts[-1] (voxel_grid.py:133)
    const TimeNorm tnorm = make_time_norm(t_first, t_last, bm1);

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
    auto load_xy = [&](int sc) {
#pragma unroll
        for (int k = 0; k < NG; ++k) c.load_xy(row_base(sc, k), valid_in(sc, k) > 0 ? (uint32_t)tl_ : 0u, xyr + C::XYW * k);
    };
    auto load_tp = [&](int sc) {
#pragma unroll
        for (int k = 0; k < NG; ++k) c.load_tp(row_base(sc, k), valid_in(sc, k) > 0 ? (uint32_t)tl_ : 0u, tpr + C::TPW * k);
        if constexpr (REC == 4) tb = c.t1((int64_t)sc * q.S);   // base of the t_norm deltas (same address in every lane)
    };
    auto fence = [&]() {   // for the compiler: loads hoisted above a compute phase keep their 2 * EPT registers live through it
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int i = tid; i < ntiles; i += THREADS) hist[i] = 0, tot[i] = 0;   // (every pass leaves hist zero again)
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
    auto write_out = [&]() {   // one contiguous, coalesced run of records
        if (V2_ABLATE_A >= 4) {
            const uint4 *src = reinterpret_cast<const uint4 *>(sorted);
            uint4 *dst = REC == 8 ? reinterpret_cast<uint4 *>(rec + lo_prev)
                                  : reinterpret_cast<uint4 *>(reinterpret_cast<uint32_t *>(rec_) + lo_prev);
            const int n16 = REC == 8 ? (int)((kept_prev + 1) >> 1) : (int)((kept_prev + 3) >> 2);
            if constexpr (REC == 8 && V2_STORE_SC1) {
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
                // 4-byte records = HBM-resident calls: streaming stores (write-through ones cost the partition 5 % there:
                // 219 against 209 us at 50 M events, and the boundary is 1 % of that call)
                for (int i = tid; i < n16; i += THREADS) {
                    const uint4 v = src[i];
                    __builtin_nontemporal_store(v.x, &dst[i].x), __builtin_nontemporal_store(v.y, &dst[i].y);
                    __builtin_nontemporal_store(v.z, &dst[i].z), __builtin_nontemporal_store(v.w, &dst[i].w);
                }
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
                const int key = c.key_of(xyr + C::XYW * k, e, g, cell);   // (of stale words beyond the stream)
                kl[G * k + e] = ((key >= 0) & (e < nv)) ? (((uint32_t)key << V2_LB) | cell) : 0xFFFFFFFFu;
                dropped += ((key < 0) & (e < nv)) ? 1u : 0u;
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
        lds_barrier();  // histogram complete
        V2_T(2);
        if (V2_ABLATE_A < 2) {
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
                    trow[i] = start | (cnt << 16);
                }
                kept += __shfl(wi, NWV - 1, 64);
                if (base + THREADS < ntiles) lds_barrier();   // tmp is reused by the next round
            }
            if (REC == 4 && tid == 0) tmp[67] = 0;   // escapes of this pass
        }
        lds_barrier();  // cursors complete
        V2_T(3);
        // normalised time, in place (t has landed during the histogram and the scan), one division at a time
#pragma unroll
        for (int s2 = 0; s2 < EPT; ++s2) {
            tv[s2] = time_norm(c.t_of(tpr + C::TPW * (s2 / G), s2 % G), tnorm);  // voxel_grid.py:134 (evk_part.h)
            __builtin_amdgcn_sched_barrier(0);
        }
        V2_T(4);
        // Nothing outstanding from here (t, p are in; the previous run's stores are a histogram and a scan old) -- said with
        // the builtin so that the placement's uses of t, p get no wait of their own: with x, y of the next sub-chunk just
        // issued such a wait is a vmcnt(0), i.e. the full latency of those loads in every placement.
        EVK_WAIT_VM0();
        fence();
        if (sc + 1 < sc_end) load_xy(sc + 1);  // in flight during the placement
        fence();
        if (V2_ABLATE_A < 3) {
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
        if constexpr (REC == 8) {
            uint32_t wide_mask = 0;
#pragma unroll
            for (int s2 = 0; s2 < EPT; ++s2) {
                if (kl[s2] != 0xFFFFFFFFu) {
                    const uint32_t pos = atomicAdd(&cur[kl[s2] >> V2_LB], 1u);
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
                    const uint32_t pos = atomicAdd(&cur[kl[s2] >> V2_LB], 1u);
                    sorted4[pos] = word;
                }
            }
        }
        V2_T(5);
        lds_barrier();   // the sorted sub-chunk is complete
        V2_T(6);
        EVK_WAIT_VM0();   // x, y of the next sub-chunk have landed during the placement: the stores below (next pass, or the
                          // epilogue) then never sit between a load and its use
        V2_T(7);
        kept_prev = kept, lo_prev = lo;
    }
    // ---- totals -> global (the last block to arrive builds the work-item plan), issued AHEAD of the last run's stores so
    //      that the two drain together
    uint32_t *gidx = index;
    if (dropped && oob) atomicAdd(oob, dropped);
    // (131 K atomics at 10 M events / 512 tiles: 1.5 us of the kernel, measured by leaving them out)
    for (int i = tid; i < ntiles; i += THREADS)
        if (tot[i]) __hip_atomic_fetch_add(gidx + V2_TOTALS + i, tot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nwide & 0xFFFFu) __hip_atomic_fetch_add(gidx + 3, nwide & 0xFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nwide >> 16) __hip_atomic_fetch_add(gidx + 5, nwide >> 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sc0 < sc_end) write_out();   // the last pass's run
    V2_T(8);
    V2_T(9);
    V2_T(10);
    if (blockIdx.x == 0 && tid == 0 && n > 0) {
        __hip_atomic_store(gidx + 0, __float_as_uint(c.t1(0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gidx + 1, __float_as_uint(c.t1(n - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
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
    // ---- plan: part_start, per-tile combine counters, item -> tile; totals / ticket back to 0.  A tile with more than `cap`
    //      events is cut into pieces of at most `part` events (ranges of sub-chunks).
    const int per = (ntiles + THREADS - 1) / THREADS;
    const int i0 = tid * per, i1 = (i0 + per < ntiles) ? i0 + per : ntiles;
    uint32_t *part_start = index + V2_PART, *counters = index + V2_COUNTER(ntiles), *item_tile = index + V2_ITEM(ntiles);
    uint32_t tt[PER_MAX];
    uint32_t np = 0;
    // (this block is the tail of the whole kernel: the three loads its last lines need are issued here, ahead of the scan)
    uint32_t now5 = 0, prev6 = 0, cnt_oob = 0;
    if (tid == 0) {
        now5 = __hip_atomic_load(gidx + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        prev6 = __hip_atomic_load(gidx + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (host_report && oob) cnt_oob = __hip_atomic_load(oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        if (host_report) {  // every workgroup's dropped-event count is in *oob (added before its ticket): tell the host,
                            // in pinned memory, so that a deferred error check costs no copy and no event on the stream
            const uint32_t cnt = cnt_oob;
            // {seq, count} as ONE 8-byte system-scope store: the pair cannot be seen torn, and no release (a write-back of the
            // L2 at the very end of the kernel's critical path) is needed to order two stores
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(host_report),
                               (unsigned long long)seq | ((unsigned long long)cnt << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Voxel tiles from the sorted runs: one workgroup per work item (tile, or part of a hot tile = a range of sub-chunks).
// A tile's records sit in ~100-200 byte segments, one per sub-chunk.  Every thread fetches the table entry of one
// sub-chunk; each wave then cuts its 64 segments into 64-byte CHUNKS (8 records, from the segment's first one), lists the chunks
// in LDS (wave scan of the chunk counts) and hands them out to groups of 4 lanes, 16 bytes per lane: all lanes stay busy
// whatever the segment lengths are, and U chunk loads per lane are in flight at a time.  Segments longer than 56
// records (clustered scenes) are streamed by the whole wave instead.
#ifndef V2_MAX_CHUNKS
// chunks of a listed segment (longer ones are streamed by the whole wave): 7 = 56 records; 6 with 768-thread workgroups, whose
// twelve lists then leave room for the counting mode's accumulators of TWO workgroups per CU (VGA, 5 bins: 2 x 78.7 KB)
#define V2_MAX_CHUNKS(WG) ((WG) == 768 ? 6 : 7)
#endif
#define V2_CHUNK_CAP(WG) (64 * V2_MAX_CHUNKS(WG))  // per wave; 28 KB for 8 waves: with the padded accumulators (21 KB at VGA) three
                                           // workgroups still fit a CU's 160 KB
// FIXED (EVK_VOXEL_DETERMINISTIC): the cells are int64 multiples of 2^-32 instead of float64 -- integer adds commute, so the
// grid is bit-identical from run to run and for any order of the events.  |contribution| < 2^30 and finite, else it is
// counted in index[4] and left out (the wrapper raises).
#define V2_FIXED_ONE 4294967296.0
template <int WG, int U, bool SPLIT, bool FIXED, int REC>
__global__ void __launch_bounds__(WG, V2_TILES_WAVES(REC)) k_voxel_tiles2(const void *__restrict__ rec_, const void *__restrict__ side_,
                                                     const uint32_t *__restrict__ bases,
                                                     const uint32_t *__restrict__ table, uint32_t *__restrict__ index,
                                                     TileGridG g, Part2 q, int B, int flags, float *__restrict__ vox,
                                                     float *__restrict__ staging) {
    constexpr int NW = WG / 64, E = V2_ENT(REC);
    // REC 8: a lane takes 16 bytes = 2 records {t_norm, polarity | cell}; REC 4: 8 bytes = 2 one-word records (k_part_sorted),
    // decoded with the base of their sub-chunk, which travels with the chunk list
    typedef typename std::conditional<REC == 8, uint4, uint2>::type Pair;
    const Pair *const recp = static_cast<const Pair *>(rec_);           // indexed in PAIRS of records
    const float *const pw = static_cast<const float *>(side_);
    const uint2 *const wide2 = static_cast<const uint2 *>(side_);
    const int overwrite = flags & EVK_VOXEL_OVERWRITE;
    constexpr bool split = SPLIT;
    const int NB = split ? 2 * B : B;
    const float bm1 = (float)(B - 1);
    extern __shared__ __attribute__((aligned(16))) acc_t acc[];
    // chunk list of every wave: {first record of the chunk (even) | 1 if that record lies before the segment, end of the
    // segment}: ONE LDS read gives a lane group everything it needs for its load (a 2-byte (segment, chunk) entry followed
    // by a ds_bpermute of the table entry put two dependent LDS round trips, queued behind other waves' atomics, in front
    // of every load)
    __shared__ uint2 cseg[NW][V2_CHUNK_CAP(WG)];
    __shared__ uint32_t cbase[REC == 4 ? NW : 1][REC == 4 ? V2_CHUNK_CAP(WG) : 1];   // REC 4: t_norm base of the chunk's sub-chunk
    const int ntiles = g.tiles_x * g.tiles_y;
    const uint32_t *part_start = index + V2_PART, *item_tile = index + V2_ITEM(ntiles);
    V2_T0();
    const uint32_t nitems = part_start[ntiles];
    if (blockIdx.x >= nitems) return;
    // XCD-aware work-item order: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md), and the segments of NEIGHBOURING
    // tiles are neighbours in every run (and their table entries share a cache line), so XCD k takes a contiguous
    // range of the tile-ordered work items: the 128-byte lines two adjacent tiles share are then fetched into ONE L2
    // once instead of into two L2s
    uint32_t item = blockIdx.x;
    // (not when tiles were cut: the pieces of a hot tile are neighbours in the item order, a contiguous range would hand most
    // of a blob to two or three XCDs -- blob scene 68 -> 63 us in plain order, uniform events 29 -> 34 us)
    if (!(flags & EVK_VOXEL2_NO_XCD_ORDER) && nitems == (uint32_t)ntiles) {
        const uint32_t k = blockIdx.x & 7u, j = blockIdx.x >> 3, q8 = nitems >> 3, r8 = nitems & 7u;
        item = k * q8 + (k < r8 ? k : r8) + j;
    }
    const int tw = g.tw, th = g.th, tpix = tw * th;
    // LDS layout of the accumulators: rows of tw | 1 cells (8 bytes), i.e. an ODD pitch.  With a pitch of 32 cells = 64
    // dwords every row of one column lands on the same pair of banks, and the events of a real scene sit on edges: a
    // wave's events share a column and differ in the row (moving-edge scene: 66 us against 36 us on uniform events).
    // The records carry row * pitch + column (evk_part.h): the cell index itself.
    const int tpitch = g.pitch, ppix = tpitch * th;   // cells per accumulator plane
    // (no tile cut <=> as many items as tiles: the plan is the identity, and two dependent loads -- ~1.3 us at the head of
    // every workgroup of a launch whose workgroups all start together -- are not made)
    int tile = (int)item;
    uint32_t first_item = item, nparts = 1u;
    if (nitems != (uint32_t)ntiles) {
        tile = (int)item_tile[item];
        first_item = part_start[tile], nparts = part_start[tile + 1] - first_item;
    }
    const uint32_t part_id = item - first_item;
    const int tx0 = (tile % g.tiles_x) * tw, ty0 = (tile / g.tiles_x) * th;
    // (64-bit fixed-point cells as in k_iwe_tiled -- ds_add_u64 is the faster LDS atomic -- with the scale from a max |p|
    // the partition kernel collects: no faster here (33.9 vs 34.6 us at 10 M events, 152.6 vs 148 us at 50 M) and the
    // extra bookkeeping in the placement pushed the partition kernel from 48 to 91 us.  Float64 cells stay.)
    // UNIT-POLARITY COUNTING (round 3).  When every polarity of the call is +1, -1 or +0 (what the reference's loaders
    // produce; the partition kernel reports it in index[7]) an event's two contributions p (1 - f) and p f to bins b0 and
    // b0 + 1 (f = t_norm - b0) are not added as two float64 LDS atomics: the event adds p to an INTEGER count S0[b0] and
    // p f, as a multiple of 2^-31, to ONE int64 cell G[b0], and the flush forms  grid[b] = S0[b] - G[b] + G[b - 1].  An
    // int32 LDS atomic costs a fraction of a float64 one, the int64 one two thirds -- and much less when a wave's events
    // share cells, as on edges --, and the second weight is never computed: tile kernel 29.5 -> 24 us at 10 M uniform events,
    // 41 -> 30 us on the moving-edge scene.  Integer adds commute: such a call's grid is BIT-REPRODUCIBLE from run to run.
    // f is a float32 in [0, 1): exact in 2^-31 steps unless below 2^-7 (truncated by < 5e-10).  The sums differ from the
    // reference's in that p (1 - f) is not rounded to float32 per event (< 6e-8 per event, random sign; the bar is 1e-5 of
    // the grid's maximum).  Events outside [ts[0], ts[-1]] touch an edge bin only: bin 0 through G[-1] (the planes are
    // G[-1 .. B-1]), bin B - 1 through -G[B - 1]; a NaN t_norm (dt == 0) poisons its cell in every bin: one bit per cell.
    // Costs LDS: (B + 1) int64 + B int32 planes -- the host enables it (EVK_VOXEL2_COUNT in `flags`) where two workgroups
    // still fit a CU.
    constexpr double G_ONE = 2147483648.0;   // 2^31
    __shared__ uint32_t poison[(1 << V2_LB) / 32];
    constexpr bool COUNTING = !SPLIT;   // (also in the EVK_VOXEL_DETERMINISTIC instantiation: the counting mode IS deterministic)
    const bool unit = COUNTING && (flags & EVK_VOXEL2_COUNT) && index[7] == 0u;
    int *const s0 = reinterpret_cast<int *>(acc + (B + 1) * ppix);   // unit mode: acc = G[-1 .. B-1], then S0[0 .. B-1]
    V2_U(0);
    unsigned long long *const gq = reinterpret_cast<unsigned long long *>(acc);   // unit mode: G as int64
    if (unit) {
        for (int i = threadIdx.x; i < (B + 1) * ppix; i += WG) acc[i] = 0.0;
        for (int i = threadIdx.x; i < B * ppix; i += WG) s0[i] = 0;
        if (threadIdx.x < (1 << V2_LB) / 32) poison[threadIdx.x] = 0u;
    } else {
        for (int i = threadIdx.x; i < NB * ppix; i += WG) acc[i] = 0.0;
    }
    V2_U(1);
    const int sc_lo = (int)(((int64_t)q.nsc * part_id) / nparts), sc_hi = (int)(((int64_t)q.nsc * (part_id + 1)) / nparts);
    const uint32_t *col = table + tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 3, grp = lane >> 2;
    auto add = [&](acc_t *a, float v) {
        if constexpr (FIXED) {
            const double sc = (double)v * V2_FIXED_ONE;
            if (!(fabs(sc) < 4.0e18)) {   // NaN, infinity or beyond 2^30: not representable
                __hip_atomic_fetch_add(index + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a), (unsigned long long)__double2ll_rn(sc), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            lds_add(a, v);
        }
    };
    // any t_norm, any polarity (voxel_bins_lds with `add`): NaN t_norm (dt == 0, Q9) and a polarity that is not finite reach
    // EVERY bin, as in the reference -- its p * weight is added for all B bins, and NaN * 0 = inf * 0 = NaN
    auto bins_general = [&](acc_t *base, int local, float tn, float p) {
        if (tn != tn) {
            for (int b = 0; b < B; ++b) add(base + b * ppix + local, tn * p);
            return;
        }
        if (!(fabsf(p) <= 3.0e38f)) {
            for (int b = 0; b < B; ++b) add(base + b * ppix + local, p * fmaxf(0.0f, 1.0f - fabsf(tn - (float)b)));
            return;
        }
        const float fl = floorf(tn);
        const int b0 = (int)fmaxf(fminf(fl, (float)(B + 1)), -2.0f);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int b = b0 + k;
            if (b < 0 || b >= B) continue;
            const float val = p * fmaxf(0.0f, 1.0f - fabsf(tn - (float)b));
            if (val != 0.0f) add(base + b * ppix + local, val);
        }
    };
    // (The hot loop holds eight inlined copies of this per-record code, and its size matters: an extra inlined copy of the
    // general path in the rare branch cost 4 %, an out-of-line function for the rare cases -- the call's register saves
    // need scratch -- 15 %; a flag joining the range test 10 %.  A polarity that is not finite is therefore only looked for in
    // the rare branches, where such records always end up because the partition marks them wide / escaped, and sends the
    // record down the general path by POISONING the value the range test reads: the hot path is the two compares it was.)
    // unit mode, t_norm outside [0, B - 1] or NaN: see above
    auto unit_general = [&](int local, float tn, float p) {
        unsigned long long *gp = gq + local;   // plane k holds G[k - 1]
        if (tn != tn) {
            __hip_atomic_fetch_or(poison + (local >> 5), 1u << (local & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (tn < 0.0f) {
            const float val = p * fmaxf(0.0f, 1.0f - fabsf(tn - 0.0f));
            if (val != 0.0f)
                __hip_atomic_fetch_add(gp, (unsigned long long)__double2ll_rn((double)val * G_ONE), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            const float val = p * fmaxf(0.0f, 1.0f - fabsf(tn - bm1));
            if (val != 0.0f)
                __hip_atomic_fetch_add(gp + B * ppix, (unsigned long long)__double2ll_rn(-(double)val * G_ONE), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto one = [&](auto unit_tag, uint32_t lo_w, uint32_t hi_w, uint32_t ridx) {
        constexpr bool UNIT = decltype(unit_tag)::value;
        // REC 8: lo_w = t_norm bits, hi_w = polarity | cell.  REC 4: lo_w = the record word, hi_w = its sub-chunk's base.
        int local;
        float p, tn;
        float tr;   // t_norm as the range test below sees it: poisoned (-1) in the rare branches for a polarity that is not finite
        if constexpr (REC == 8) {
            local = (int)(hi_w & V2_LOCAL_MASK);   // row * pitch + column
            p = __uint_as_float(hi_w & V2_P_MASK);
            tn = tr = __uint_as_float(lo_w);  // normalised time, computed by the partition kernel
            if (hi_w & V2_WIDE) {   // rare: the exact float32 polarity from the side array.  The wait stays INSIDE the branch
                p = pw[ridx];       // (builtin: the compiler's scoreboard sees it) -- at the join it would be a vmcnt(0) on
                __builtin_amdgcn_s_waitcnt(0x0F70);   // every event, i.e. the next round's record loads could never stay in flight
                if (!SPLIT && !(fabsf(p) <= 3.0e38f)) tr = -1.0f;
            }
        } else {
            local = (int)(lo_w & V2_LOCAL_MASK);
            const uint32_t code = (lo_w >> V2_CODE_SHIFT) & 3u;
            if (code == 3u) {       // rare: escaped record, the exact pair from the side array (same wait discipline)
                const uint2 e = wide2[(uint64_t)(ridx / (uint32_t)q.S) * (uint32_t)q.S + (lo_w >> V2_DELTA_SHIFT)];
                __builtin_amdgcn_s_waitcnt(0x0F70);
                tn = tr = __uint_as_float(e.x), p = __uint_as_float(e.y);
                if (!SPLIT && !(fabsf(p) <= 3.0e38f)) tr = -1.0f;
            } else {
                tn = tr = __uint_as_float(hi_w + (lo_w >> V2_DELTA_SHIFT));
                p = __uint_as_float(code == 2u ? 0u : (0x3F800000u | (code << 31)));
            }
        }
        if (V2_ABLATE_B < 2) {
            if (tn * p == 1.2345e-30f) acc[local] = 1.0;
            return;
        }
        if constexpr (UNIT) {
            if (__builtin_expect(tr >= 0.0f && tr <= bm1, 1)) {
                const int b0 = (int)tn;
                const int off = __mul24(b0, ppix) + local;
                // G[b0] += p f (in 2^-31 steps: |p f| < 1 fits an int32), S0[b0] += p
                const int fx = (int)((p * (tn - (float)b0)) * 2147483648.0f);
                __hip_atomic_fetch_add(gq + ppix + off, (unsigned long long)(long long)fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(s0 + off, (int)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                unit_general(local, tn, p);
            }
            return;
        }
        if (__builtin_expect(tr >= 0.0f && tr <= bm1, 1)) {
            // the common case, straight-line: t inside [ts[0], ts[-1]].  Bins b0 = floor(t_norm) and b0 + 1 with the
            // weights of voxel_grid.py:138 -- 1 - |t_norm - b| evaluated exactly as there (for b0 the absolute value is
            // the identity; max(0, .) cannot bind for these two bins).  A zero weight is added like any other (x + 0 = x;
            // the reference's index_put_ adds it too): skipping it cost a compare and a branch per bin on every event.
            acc_t *a = acc + local;
            float w = p;
            if constexpr (split) {
                if (!(p > 0.0f) && !(p <= 0.0f)) return;  // a NaN polarity is in neither grid
                a += p > 0.0f ? 0 : B * ppix;
                w = 1.0f;
            }
            const int b0 = (int)tn;
            const float v0 = w * (1.0f - (tn - (float)b0)), v1 = w * (1.0f - fabsf(tn - (float)(b0 + 1)));
            // (b0 + 1 == B only for t_norm == B - 1, where v1 is a zero: added to bin B - 1, which it does not change, rather
            // than branched around)
            a += __mul24(b0, ppix);
            if (V2_ABLATE_B < 3) {   // (timing builds) weights and addresses, no atomics
                if (v0 + v1 == 1.2345e-30f) a[b0 + 1 < B ? ppix : 0] = 1.0;
                return;
            }
            add(a, v0);
            if (V2_ABLATE_B < 4) return;   // (timing builds) one atomic per event
            add(a + (b0 + 1 < B ? ppix : 0), v1);
        } else if (!split) {
            bins_general(acc, local, tn, p);
        } else if (tn != tn) {
            bins_general(acc, local, tn, 1.0f);
            bins_general(acc + B * ppix, local, tn, 1.0f);
        } else if (p > 0.0f) {
            bins_general(acc, local, tn, 1.0f);
        } else if (p <= 0.0f) {
            bins_general(acc + B * ppix, local, tn, 1.0f);
        }
    };
    // A lane's load: records pos, pos + 1 -- 16 (8) bytes at ANY record boundary: the chunks of a segment start at its first
    // record, not at the 16-byte boundary below it (global loads need no more than dword alignment; with aligned chunks the
    // first one of every other segment began with a record of the neighbouring tile: one lane idle and a test per pair)
    typedef typename std::conditional<REC == 8, uint2, uint32_t>::type Rec1;
    struct __attribute__((packed, aligned(REC))) PairU {
        Pair v;
    };
    auto load_pair = [&](uint32_t pos) -> Pair { return reinterpret_cast<const PairU *>(static_cast<const Rec1 *>(rec_) + pos)->v; };
    auto pair = [&](auto unit_tag, const Pair &v, uint32_t bbits, uint32_t pos, uint32_t end) {  // records pos, pos + 1 of a segment ending at `end`
        if constexpr (REC == 8) {
            one(unit_tag, v.x, v.y, pos);
            if (pos + 1 < end) one(unit_tag, v.z, v.w, pos + 1);
        } else {
            one(unit_tag, v.x, bbits, pos);
            if (pos + 1 < end) one(unit_tag, v.y, bbits, pos + 1);
        }
    };
    // Chunk rounds over a wave's list of `total` chunks, software-pipelined in three stages: list entries of round r + 2
    // (LDS) | record loads of round r + 1 (global) | accumulation of round r.  The loads are UNCONDITIONAL -- a lane group
    // without a chunk reads the head of the record buffer -- so that nothing but arithmetic sits between them and the
    // compiler can wait for the older round alone (`vmcnt(U)`).
    auto rounds = [&](auto unit_tag, const uint32_t total) {
        auto meta = [&](uint32_t j0, uint2(&cs)[U], uint32_t(&cb_)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t j = j0 + 16u * u + grp;
                cs[u] = j < total ? cseg[wave][j] : make_uint2(0u, 0u);
                if constexpr (REC == 4) cb_[u] = j < total ? cbase[wave][j] : 0u;
            }
        };
        auto fire = [&](const uint2(&cs)[U], Pair(&v)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pos = cs[u].x + 2u * sub;
                v[u] = load_pair(pos < cs[u].y ? pos : 2u * sub);
            }
        };
        auto eat = [&](const uint2(&cs)[U], const uint32_t(&cb_)[U], const Pair(&v)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pos = cs[u].x + 2u * sub;
                if (pos < cs[u].y) pair(unit_tag, v[u], cb_[u], pos, cs[u].y);
            }
        };
        constexpr uint32_t step = 16u * U;
        uint2 ca[U], cb[U], cn[U];
        uint32_t ba[U] = {}, bbv[U] = {}, bn[U] = {};
        Pair va[U], vb[U];
        meta(0u, ca, ba);
        fire(ca, va);
        meta(step, cb, bbv);
        for (uint32_t j0 = 0; j0 < total; j0 += 2u * step) {
            fire(cb, vb);               // round j0 + step
            meta(j0 + 2u * step, cn, bn);
            eat(ca, ba, va);            // round j0
            fire(cn, va);               // round j0 + 2 step
#pragma unroll
            for (int u = 0; u < U; ++u) ca[u] = cn[u], ba[u] = bn[u];
            meta(j0 + 3u * step, cn, bn);
            eat(cb, bbv, vb);           // round j0 + step
#pragma unroll
            for (int u = 0; u < U; ++u) cb[u] = cn[u], bbv[u] = bn[u];
        }
    };
    // A long segment [b2, e3) (> 7 chunks = 56 records; clustered scenes, the parts of a hot tile): the whole wave streams it,
    // 16 bytes per lane, four loads per lane in flight (with one dependent load at a time a part was a chain of ~2 us round
    // trips)
    auto stream_segment = [&](auto unit_tag, const uint32_t b2, const uint32_t e3, const uint32_t b2b) {
        for (uint32_t p2 = b2 + 2u * lane; p2 < e3; p2 += 512u) {
            Pair v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t pu = p2 + 128u * u;
                v[u] = load_pair(pu < e3 ? pu : p2);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t pu = p2 + 128u * u;
                if (pu < e3) pair(unit_tag, v[u], b2b, pu, e3);
            }
        }
    };
    auto wave_scan = [&](uint32_t v) {   // inclusive
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        return incl;
    };
    const int range = sc_hi - sc_lo;
    auto batches = [&](auto unit_tag) {
    if constexpr (E > 1) {
        // E table entries per lane and batch (HBM-resident calls, whose 4-byte records come with two workgroups per CU anyway
        // -- 80 KB of LDS at 720p -- so the registers of E entries are free): a tile's column of ~4000 sub-chunks is 3 batches
        // instead of 8, each with one list build and one fill and drain of the load pipeline (50 M events: 154 -> 143 us;
        // at 10 M events / VGA nothing, and the 128-register budget costs the structured scenes their third workgroup per
        // CU: E = 1 there).  Entry i of a batch goes to wave i % NW, there to lane (i / NW) % 64, register (i / NW) / 64.  When
        // a wave's chunks do not fit its list (clustered scenes: many segments of 5-7 chunks) it takes its registers one at a
        // time.  The lists are wave-private and a wave's LDS operations execute in order: no barrier inside the batch.
        const int nbatch = (range + WG * E - 1) / (WG * E);
        const int bsz = nbatch ? ((range + nbatch - 1) / nbatch + NW - 1) / NW * NW : NW;  // <= WG * E, a multiple of NW
        auto fetch = [&](int base, uint32_t(&en)[E], uint32_t(&bn)[E]) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int idx = (e * 64 + lane) * NW + wave, my = base + idx;
                const bool have = idx < bsz && my < sc_hi;
                en[e] = have ? col[(int64_t)my * q.nt_pad] : 0u;
                bn[e] = 0u;
                if constexpr (REC == 4) bn[e] = have ? bases[my] : 0u;
            }
        };
        __syncthreads();  // the accumulators are zero before the first adds
        for (int base = sc_lo; base < sc_hi; base += bsz) {
            // (no prefetch of the next batch's entries, and entries are fetched again where a rare path needs them after
            // the rounds: their registers are what the hot loop needs)
            uint32_t ent[E], bb[E];
            uint32_t sum = 0, longs = 0, packed = 0;   // packed: chunks of entry e in bits [4e, 4e + 4)
            fetch(base, ent, bb);
            auto mych_of = [&](int e) { return (packed >> (4 * e)) & 15u; };
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t start = ent[e] & 0xFFFFu, cnt = ent[e] >> 16;
                const uint32_t nch = (cnt + 7u) >> 3;
                const bool is_long = nch > (uint32_t)V2_MAX_CHUNKS(WG);
                packed |= (is_long ? 0u : nch) << (4 * e);
                sum += is_long ? 0u : nch;
                longs |= is_long ? (1u << e) : 0u;
            }
            const uint32_t incl_all = wave_scan(sum);
            const bool fits = __shfl(incl_all, 63, 64) <= (uint32_t)V2_CHUNK_CAP(WG);
            const int npass = fits ? 1 : E;
            V2_U(2);
            V2_U(3);
            for (int pass = 0; pass < npass; ++pass) {
                uint32_t mine = sum, incl = incl_all;
                if (pass > 0) fetch(base, ent, bb);
                if (!fits) {
                    mine = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) mine = e == pass ? mych_of(e) : mine;
                    incl = wave_scan(mine);
                }
                const uint32_t total = __shfl(incl, 63, 64);
                uint32_t w = incl - mine;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (!fits && e != pass) continue;
                    const uint32_t start = ent[e] & 0xFFFFu, cnt = ent[e] >> 16;
                    const uint32_t rb = (uint32_t)(base + (e * 64 + lane) * NW + wave) * (uint32_t)q.S, p0 = rb + start, e0 = p0 + cnt;
                    const uint32_t mc = mych_of(e);
                    for (uint32_t k = 0; k < mc; ++k) {
                        cseg[wave][w + k] = make_uint2(p0 + 8u * k, e0);
                        if constexpr (REC == 4) cbase[wave][w + k] = bb[e];
                    }
                    w += mc;
                }
                V2_U(4);
                V2_U(5);
                rounds(unit_tag, total);
            }
            V2_U(6);
            // long segments: listed in the (consumed) chunk list, then streamed one after the other
            uint32_t nlong = 0;
            if (__ballot(longs != 0u)) {
                uint32_t ent2[E], bb2[E];
                fetch(base, ent2, bb2);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool il = (longs >> e) & 1u;
                    const uint64_t m = __ballot(il);
                    if (il) {
                        const uint32_t st = ent2[e] & 0xFFFFu, cn = ent2[e] >> 16;
                        const uint32_t rb = (uint32_t)(base + (e * 64 + lane) * NW + wave) * (uint32_t)q.S;
                        const uint32_t at = nlong + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                        cseg[wave][at] = make_uint2(rb + st, rb + st + cn);
                        if constexpr (REC == 4) cbase[wave][at] = bb2[e];
                    }
                    nlong += (uint32_t)__builtin_popcountll(m);
                }
            }
            for (uint32_t i = 0; i < nlong; ++i) {
                const uint2 sg = cseg[wave][i];
                uint32_t b2b = 0u;
                if constexpr (REC == 4) b2b = cbase[wave][i];
                stream_segment(unit_tag, sg.x, sg.y, b2b);
            }
        }
    } else {
        // Entries go to the threads in equal batches, INTERLEAVED over the waves (slot = lane * NW + wave): a short range --
        // the last batch of a tile, or one of the many parts of a hot tile -- then still gives every wave its share instead
        // of filling wave 0 first (1221 sub-chunks: 3 batches of 408 = 51 entries per wave, not 64, 64, 64 / 64, 5, 0 ...).
        const int nbatch = (range + WG - 1) / WG;
        const int bsz = nbatch ? ((range + nbatch - 1) / nbatch + NW - 1) / NW * NW : NW;  // <= WG, a multiple of NW
        const int slot = lane * NW + wave;
        uint32_t ent_next = 0, bb_next = 0;
        {
            const int my = sc_lo + slot;
            if (slot < bsz && my < sc_hi) {
                ent_next = col[(int64_t)my * q.nt_pad];
                if constexpr (REC == 4) bb_next = bases[my];
            }
        }
        for (int base = sc_lo; base < sc_hi; base += bsz) {
            const uint32_t ent = ent_next, bb = bb_next;
            {   // next batch's entries: in flight while this batch is processed
                const int my = base + bsz + slot;
                const bool have = slot < bsz && my < sc_hi;
                ent_next = have ? col[(int64_t)my * q.nt_pad] : 0u;
                if constexpr (REC == 4) bb_next = have ? bases[my] : 0u;
            }
            if (V2_ABLATE_B < 1) {
                if (ent == 0xFFFFFFFFu) acc[0] = 1.0;
                continue;
            }
            const uint32_t start = ent & 0xFFFFu, cnt = ent >> 16;
            const uint32_t nch = (cnt + 7u) >> 3;
            // A wave lists ALL chunks of its segments, in as many passes over groups of its lanes as its list (V2_CHUNK_CAP
            // entries) needs: the pieces of a hot tile are few entries of ~25 chunks each, and through the list they get the
            // pipelined rounds and full groups of lanes.  A segment is LONG -- streamed by the whole wave, one dependent round
            // trip after the other -- only when even a quarter of the lanes overflows the list (> 7 chunks then).
            const uint32_t incl_full = wave_scan(nch);
            const uint32_t total_full = __shfl(incl_full, 63, 64);
            uint32_t npass = (total_full + (uint32_t)V2_CHUNK_CAP(WG) - 1u) / (uint32_t)V2_CHUNK_CAP(WG);   // (wave-uniform)
            npass = npass < 1u ? 1u : (npass > 4u ? 4u : npass);
            V2_U(2);
            __syncthreads();  // (a) accumulators are zero before the first adds; (b) the previous batch's list is consumed
            V2_U(3);
            for (uint32_t pass = 0; pass < npass; ++pass) {
                const bool mine = npass == 1u || ((uint32_t)lane * npass) / 64u == pass;
                uint32_t mych = mine ? nch : 0u, incl = incl_full;
                bool is_long = false;
                if (npass > 1u) {
                    incl = wave_scan(mych);
                    if (__shfl(incl, 63, 64) > (uint32_t)V2_CHUNK_CAP(WG)) {
                        is_long = mine && nch > (uint32_t)V2_MAX_CHUNKS(WG);
                        mych = (mine && !is_long) ? nch : 0u;
                        incl = wave_scan(mych);
                    }
                }
                const uint32_t total = __shfl(incl, 63, 64), excl = incl - mych;
                {
                    const uint32_t rb = (uint32_t)(base + slot) * (uint32_t)q.S, p0 = rb + start, e0 = p0 + cnt;
                    for (uint32_t k = 0; k < mych; ++k) {
                        cseg[wave][excl + k] = make_uint2(p0 + 8u * k, e0);
                        if constexpr (REC == 4) cbase[wave][excl + k] = bb;
                    }
                }
                V2_U(4);
                // The two workgroup barriers per batch are kept on purpose (the second one in the first pass, which every
                // wave runs): with wave-private entry ranges and no barrier the kernel ran at 50 us instead of 39 -- all
                // tiles walking the runs in step keeps each run L2-hot while its 600 segments are pulled
                if (pass == 0u) __syncthreads();
                V2_U(5);
                rounds(unit_tag, total);
                V2_U(6);
                uint64_t m = __ballot(is_long);
                while (m) {
                    const int s = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint32_t e2 = __shfl(ent, s, 64), b2b = __shfl(bb, s, 64);
                    const uint32_t rb = (uint32_t)(base + s * NW + wave) * (uint32_t)q.S;  // lane s's sub-chunk
                    stream_segment(unit_tag, rb + (e2 & 0xFFFFu), rb + (e2 & 0xFFFFu) + (e2 >> 16), b2b);
                }
            }
        }
    }
    };
    if constexpr (COUNTING) {
        if (unit) batches(std::true_type{});
        else batches(std::false_type{});
    } else {
        batches(std::false_type{});
    }
    V2_U(7);
    __syncthreads();
    V2_U(8);
    const int64_t plane = (int64_t)g.dom_h * g.dom_w;
    auto split_cell = [&](int c, int &b, int &row, int &col) {   // dense cell c = (plane, row, column) of the tile
        b = (int)div_magic((uint32_t)c, g.mp);
        const int l = c - b * tpix;
        row = (int)div_magic((uint32_t)l, g.mx);
        col = l - row * tw;
    };
    auto flush = [&](auto value_of) {
        for (int c = threadIdx.x; c < NB * tpix; c += WG) {
            int b, row, col;
            split_cell(c, b, row, col);
            const int X = tx0 + col, Y = ty0 + row;
            if (X < g.dom_w && Y < g.dom_h) {
                float *o = vox + b * plane + (int64_t)Y * g.dom_w + X;
                const float v = value_of(c);
                *o = overwrite ? v : *o + v;
            }
        }
    };
    auto lds_cell = [&](int c) -> float {   // dense cell c -> padded LDS layout
        int b, row, col;
        split_cell(c, b, row, col);
        const int l = row * tpitch + col;
        if (unit) {   // exact integer combination, one rounding to float32
            if ((poison[l >> 5] >> (l & 31)) & 1u) return __uint_as_float(0x7FC00000u);
            const long long v = ((long long)s0[b * ppix + l] << 31) - (long long)gq[(b + 1) * ppix + l] + (long long)gq[b * ppix + l];
            return (float)((double)v * (1.0 / G_ONE));
        }
        const acc_t a = acc[b * ppix + l];
        if constexpr (FIXED) return (float)((double)__builtin_bit_cast(long long, a) * (1.0 / V2_FIXED_ONE));
        return (float)a;
    };
    if (nparts == 1) {
        flush(lds_cell);
        V2_U(9);
        V2_UEND();
        return;
    }
    // split (hot) tile: as k_voxel_tiled -- partial tiles to staging, the last part to arrive sums them in part order
    const int cells = NB * tpix;
    const int64_t stride = v2_staging_stride(cells);   // whole 128-byte lines per item: no line is shared by two items
    float *mine = staging + (int64_t)item * stride;
    // The partial tile goes to the staging buffer with AGENT-SCOPE stores and is read back with agent-scope loads: such
    // accesses are performed at the level all XCDs share, complete (vmcnt) only when they are, and never hit a stale line of
    // this XCD's L2 -- so the hand-over to the last part needs no release / acquire FENCE, which on this chip is a write-back
    // (and an invalidate) of the whole L2 with everything the other workgroups have flushed into it (blob scene: tile kernel
    // 48.7 -> 46.3 us).  Every wave drains its own stores before the barrier; one lane then takes the ticket.
    for (int c = threadIdx.x; c < cells; c += WG)
        __hip_atomic_store(mine + c, lds_cell(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int is_last;
    if (threadIdx.x == 0) {
        uint32_t *counter = index + V2_COUNTER(ntiles) + tile;
        const uint32_t prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == nparts - 1);
        if (is_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!is_last) return;
    const float *parts = staging + (int64_t)first_item * stride;
    flush([&](int c) {
        float sum = 0.0f;
        for (uint32_t p = 0; p < nparts; ++p)
            sum += __hip_atomic_load(parts + (int64_t)p * stride + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return sum;
    });
}

// ---- host-side geometry -----------------------------------------------------------------------------------------
// Partition geometry = threads x events per thread, ONE workgroup per CU.  The library ships the two the default dispatch
// reaches: 1024 x 8 (sub-chunks of 8 K events: 105 registers, 68 KB of LDS -- room for the workgroups of another kernel,
// e.g. an overlapped RCCL collective) and 1024 x 12 (12 K events: longer segments for the tile kernel, taken when there
// are more than 680 tiles and the call need not share its CUs).  Measured and rejected (compiled
// only with -DEVK_EXPERIMENTS, tools/exp_build.sh, and selected with EVK_V2_PART): 512x32 / 1024x16 (16 K events: the
// whole register file, 67 / 74 us), two workgroups per CU (round 2: 1024x8 / 512x16 / 768x12, spills, 75-150 us; round 3,
// without spills: 512x16 56.0 and 768x8 56.8 against 48 us -- the kernel moves its 240 MB at 5 TB/s, a second workgroup per
// CU only makes the sub-chunks shorter).
struct V2Config {
    int threads, ept;
};
#ifdef EVK_EXPERIMENTS
#define V2_GEOMETRIES(X) X(1024, 8) X(1024, 12) X(1024, 16) X(512, 16)
#else
#define V2_GEOMETRIES(X) X(1024, 8) X(1024, 12)
#endif
static V2Config v2_config_env() {
    V2Config c{0, 0};
    const char *geo = getenv("EVK_V2_PART");
    int t = 0, e = 0;
    if (geo && sscanf(geo, "%dx%d", &t, &e) == 2) {
#define X(T, E) if (t == T && e == E) c = V2Config{T, E};
        V2_GEOMETRIES(X)
#undef X
    }
    return c;
}
static const V2Config &v2_config(bool share = false, int ntiles = 0) {
    static const V2Config forced = v2_config_env();
    static const V2Config small{1024, 8}, large{1024, 12};
    if (share) return small;
    if (forced.threads) return forced;
    return ntiles > 680 ? large : small;
}
#define V2_MIN_SUBCHUNK 8192
#define V2_LDS_LIMIT (160 * 1024 - 512)   // (the partition kernel also has a few bytes of static LDS)
// LDS of the partition kernel: sorted records | counts | cursors | totals | scan scratch
static size_t v2_part_lds(int threads, int ept, int rec, int ntiles) {
    return (size_t)threads * ept * rec + 16 + 3 * (size_t)((ntiles + 4) & ~3) * 4 + 68 * 4 + 16;
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
// EVK_V2_SPLIT="at,part" overrides (measurements).
struct V2Split {
    double at, part;
};
static const V2Split &v2_split() {
    static const V2Split f = [] {
        V2Split v{2.5, 1.6};
        const char *s = getenv("EVK_V2_SPLIT");
        double a = 0, p = 0;
        if (s && sscanf(s, "%lf,%lf", &a, &p) == 2 && a >= 1.0 && p >= 0.25 && p <= a) v = V2Split{a, p};
        return v;
    }();
    return f;
}
// (the three switches below are read on EVERY call -- a getenv costs nothing next to a launch --, so that one process can
// run the variants side by side: the tests do)
static bool v2_count_enabled() {   // EVK_V2_COUNT=0: no unit-polarity counting in the tile kernel (A/B measurements, tests)
    const char *s = getenv("EVK_V2_COUNT");
    return !(s && s[0] == '0');
}
static int v2_tiles_wg() {   // EVK_V2_TILES_WG=512 keeps the 512-thread tile workgroups everywhere (A/B measurements)
    const char *s = getenv("EVK_V2_TILES_WG");
    return s ? atoi(s) : 0;
}
static int64_t v2_mean(int64_t n, int ntiles) { return n / (ntiles > 0 ? ntiles : 1); }
static int64_t v2_cap(int64_t n, int ntiles) {   // a tile with more events than this is cut ...
    const int64_t c = (int64_t)(v2_split().at * (double)v2_mean(n, ntiles));
    return c > 16384 ? c : 16384;
}
static int64_t v2_part(int64_t n, int ntiles) {   // ... into pieces of at most this many
    const int64_t c = (int64_t)(v2_split().part * (double)v2_mean(n, ntiles));
    return c > 8192 ? c : 8192;
}
static int v2_max_items(int64_t n, int ntiles) { return ntiles + (int)(n / v2_part(n, ntiles)) + 1; }

// Record size of a call.  4-byte compact records (k_part_sorted) once the call's streams no longer fit the 256 MB Infinity
// Cache (more than 16 M events): there the partition is bound by HBM bytes and 20 instead of 24 B/event make it 15 % faster
// (50 M events: 245 -> 208 us, whole call 0.397 -> 0.366 ms, same box).  Below, the streams are cache-resident and the delta /
// code / escape arithmetic costs what the bytes save, the tile kernel's extra decode 6 % (10 M events: 0.0838 against
// 0.0792 ms): 8-byte records.  EVK_V2_REC=4|8 forces one (measurements, tests).
static int v2_rec_bytes(int64_t n) {
    const char *s = getenv("EVK_V2_REC");
    const int forced = s ? atoi(s) : 0;
    if (forced == 4 || forced == 8) return forced;
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
    L.total = L.staging + al256((int64_t)v2_max_items(n, ntiles) * v2_staging_stride((int64_t)planes * tw * th) * 4);
    return L;
}

}  // namespace evk

using namespace evk;

extern "C" int64_t evk_voxel2_index_len(int ntiles, int64_t n) {
    if (ntiles <= 0 || ntiles > V2_MAX_TILES || n < 0) return 0;
    return (int64_t)V2_ITEM(ntiles) + v2_max_items(n, ntiles);
}

extern "C" int64_t evk_voxel2_scratch_bytes(int ntiles, int64_t n, int planes, int tile_w, int tile_h) {
    if (ntiles <= 0 || n < 0 || planes <= 0 || tile_w <= 0 || tile_h <= 0) return 0;
    const int64_t a = v2_layout(ntiles, n, planes, tile_w, tile_h, false).total;
    const int64_t b = v2_layout(ntiles, n, planes, tile_w, tile_h, true).total;
    return a > b ? a : b;
}

// tiles of tile_w x tile_h pixels covering an (h, wd) image, or 0 when the one-pass path cannot take that tiling
extern "C" int evk_voxel2_num_tiles(int h, int wd, int tile_w, int tile_h) {
    TileGridG g;
    if (make_grid_g(g, h, wd, tile_w, tile_h) != EVK_OK) return 0;
    return g.tiles_x * g.tiles_y;
}

// largest tile count the partition kernel's LDS holds (sorted records + one uint32 per tile)
extern "C" int evk_voxel2_max_tiles(void) {
    const int64_t budget = (int64_t)V2_LDS_LIMIT - (int64_t)1024 * 12 * 8 - 1024;   // the larger shipped geometry
    const int64_t t = budget / 12;   // a counter, a cursor and a total per tile
    return (int)(t < V2_MAX_TILES ? (t > 0 ? t : 0) : V2_MAX_TILES);
}


template <int THREADS, int EPT, int REC, typename C>
static void launch_part(const C &c, int64_t n, const TileGridG &g, int ntiles, const Part2 &q, float t_first, float t_last,
                        float bm1, int t_from_events, void *rec, void *pw, uint32_t *bases, uint32_t *table, uint32_t *index,
                        uint32_t *oob, uint32_t *host_report, uint32_t seq, hipStream_t s) {
    const size_t lds = v2_part_lds(THREADS, EPT, REC, ntiles);
    static std::once_flag once[64];   // per device and instantiation: the attribute belongs to the loaded code object
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [] {   // (the kernel also has a few bytes of static LDS: ask for less than the full 160 KiB)
        (void)hipFuncSetAttribute((const void *)k_part_sorted<THREADS, EPT, REC, C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024 - 256);
    });
    k_part_sorted<THREADS, EPT, REC, C><<<q.nblk, THREADS, lds, s>>>(c, n, g, ntiles, q, t_first, t_last, bm1, t_from_events,
                                                                    rec, pw, bases, table, index, (uint32_t)v2_cap(n, ntiles), (uint32_t)v2_part(n, ntiles), oob,
                                                                    host_report, seq);
}

// Tile kernel: 768 or 512 threads (chosen in voxel2() below), 2 chunk loads per lane in flight (-DV2_U8 / -DV2_U4 for
// measurements: 1, 3 and 4 are level or slower).  `lds_dyn` = the accumulators of the mode the launch
// may run in (the chunk lists are static).
template <int WG, int U, bool SPLIT, bool FIXED, int REC>
static void launch_tiles(int items, size_t lds_dyn, hipStream_t s, const void *rec, const void *pw, const uint32_t *bases,
                         const uint32_t *table, uint32_t *index, const TileGridG &g, const Part2 &q, int B, int kf, float *vox,
                         float *staging) {
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [] {
        (void)hipFuncSetAttribute((const void *)k_voxel_tiles2<WG, U, SPLIT, FIXED, REC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024 - (REC == 4 ? 12 : 8) * (WG / 64) * V2_CHUNK_CAP(WG) - 256);
    });
    k_voxel_tiles2<WG, U, SPLIT, FIXED, REC><<<items, WG, lds_dyn, s>>>(rec, pw, bases, table, index, g, q, B, kf, vox, staging);
}

template <typename C>
static int voxel2(const C &c, int64_t n, int h, int wd, int tile_w, int tile_h, float t_first, float t_last, int B,
                  int flags, float *vox, uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob,
                  uint32_t *host_report, uint32_t seq, void *stream) {
    TileGridG g;
    const int known = EVK_VOXEL_OVERWRITE | EVK_VOXEL_SPLIT_POLARITY | EVK_VOXEL_T_FROM_EVENTS | EVK_VOXEL2_PARTITION_ONLY |
                      EVK_VOXEL2_TILES_ONLY | EVK_VOXEL2_NO_XCD_ORDER | EVK_VOXEL2_SHARE_CU | EVK_VOXEL_DETERMINISTIC;
    if (make_grid_g(g, h, wd, tile_w, tile_h) != EVK_OK || B <= 0 || !vox || !index || !scratch || n <= 0 ||
        n > (int64_t)4000000000LL || (flags & ~known))
        return EVK_EINVAL;
    if (host_report && ((uintptr_t)host_report & 7u)) return EVK_EALIGN;   // written with one 8-byte store
    const int ntiles = g.tiles_x * g.tiles_y;
    if (ntiles > evk_voxel2_max_tiles()) return EVK_EINVAL;
    const int planes = (flags & EVK_VOXEL_SPLIT_POLARITY) ? 2 * B : B;
    const size_t lds_acc = (size_t)planes * sizeof(acc_t) * g.pitch * g.th;  // odd row pitch
    const size_t lds_static = 12 * 8 * V2_CHUNK_CAP(512) + 64;             // the tile kernel's chunk lists (512 threads, either record size)
    if (lds_acc + lds_static > 150 * 1024) return EVK_EINVAL;
    const bool share = flags & EVK_VOXEL2_SHARE_CU;
    const V2Layout L = v2_layout(ntiles, n, planes, tile_w, tile_h, share);
    if (scratch_bytes < L.total) return EVK_ESCRATCH;
    if (!aligned16(scratch)) return EVK_EALIGN;
    const Part2 q = v2_geometry(n, ntiles, share);
    char *sb = (char *)scratch;
    uint32_t *table = (uint32_t *)(sb + L.table), *bases = (uint32_t *)(sb + L.bases);
    void *rec = sb + L.rec, *pw = sb + L.pw;
    float *staging = (float *)(sb + L.staging);
    hipStream_t s = (hipStream_t)stream;
    const V2Config &cfg = v2_config(share, ntiles);
    const float bm1 = (float)(B - 1);
    const int tfe = (flags & EVK_VOXEL_T_FROM_EVENTS) ? 1 : 0;
    const int recb = v2_rec_bytes(n);
    if (!(flags & EVK_VOXEL2_TILES_ONLY)) {
#define X(T, E)                                                                                                              \
    if (cfg.threads == T && cfg.ept == E) {                                                                                  \
        if (recb == 4)                                                                                                       \
            launch_part<T, E, 4>(c, n, g, ntiles, q, t_first, t_last, bm1, tfe, rec, pw, bases, table, index, oob, host_report, seq, s); \
        else                                                                                                                 \
            launch_part<T, E, 8>(c, n, g, ntiles, q, t_first, t_last, bm1, tfe, rec, pw, bases, table, index, oob, host_report, seq, s); \
    }
        V2_GEOMETRIES(X)
#undef X
    }
    if (!(flags & EVK_VOXEL2_PARTITION_ONLY)) {
        const int items = v2_max_items(n, ntiles);
        int kf = flags & (EVK_VOXEL_OVERWRITE | EVK_VOXEL_SPLIT_POLARITY | EVK_VOXEL2_NO_XCD_ORDER);
        // (512 tiles on 768 workgroup slots: the dispatcher spreads them evenly by itself -- asking for more LDS than the
        // accumulators need, so that a CU holds exactly its share, changed nothing for uniform events and cost the blob scene
        // 81 against 68 us, its many pieces then waiting for slots)
        const bool sp = flags & EVK_VOXEL_SPLIT_POLARITY, fx = flags & EVK_VOXEL_DETERMINISTIC;
        // LDS of a tile workgroup: its accumulators (dynamic) + one chunk list per wave (static).  The counting mode
        // (k_voxel_tiles2: B + 1 float64 and B int32 planes instead of B float64 planes) and the 768-thread workgroups are taken
        // while TWO workgroups still fit a CU -- that is what the 512 tiles of a VGA call need.
        const size_t lds_count = ((size_t)(B + 1) * sizeof(acc_t) + (size_t)B * 4) * g.pitch * g.th;
        auto two_fit = [](size_t acc_bytes, int wg, int rec) {
            return 2 * (acc_bytes + (size_t)(rec == 4 ? 12 : 8) * (wg / 64) * V2_CHUNK_CAP(wg) + 256) <= (size_t)160 * 1024;
        };
        const bool may_count = !sp && v2_count_enabled() && n < ((int64_t)1 << 31);   // (int32 counts)
#ifndef V2_U4
#define V2_U4 2   // chunk loads per lane in flight, 4-byte records
#endif
#ifndef V2_U8
#define V2_U8 2   // ... 8-byte records
#endif
#define V2_UU(R) ((R) == 4 ? V2_U4 : V2_U8)
#define V2_TILES(W, R)                                                                                                     \
    do {                                                                                                                   \
        if (sp && fx) launch_tiles<W, V2_UU(R), true, true, R>(items, lds_dyn, s, rec, pw, bases, table, index, g, q, B, kf, vox, staging);   \
        else if (sp) launch_tiles<W, V2_UU(R), true, false, R>(items, lds_dyn, s, rec, pw, bases, table, index, g, q, B, kf, vox, staging);   \
        else if (fx) launch_tiles<W, V2_UU(R), false, true, R>(items, lds_dyn, s, rec, pw, bases, table, index, g, q, B, kf, vox, staging);   \
        else launch_tiles<W, V2_UU(R), false, false, R>(items, lds_dyn, s, rec, pw, bases, table, index, g, q, B, kf, vox, staging);          \
    } while (0)
        // Threads per tile workgroup.  8-byte records (cache-resident calls): 768 -- twelve waves per tile -- while TWO such
        // workgroups fit a CU's LDS (accumulators + 12 chunk lists <= 80 KB: VGA at 5 bins; not split polarities or 720p
        // tiles).  Uniform events do not care (29.5 us either way, 512 tiles in one generation); the pieces of a hot tile are
        // done sooner: blob scene 57 -> 49 us, i.e. 1.25 x the uniform call end to end.  4-byte records (HBM-resident calls, three
        // table entries per lane, 128 registers): 512 (768: 151-158 against 132-142 us at 50 M events).
        // (a call that shares its CUs with a collective's workgroups keeps the 512-thread workgroups: two of them with the
        // counting mode's accumulators leave ~19 KB of every CU's LDS free, two 768-thread ones 3 KB)
        const bool may_wide = recb == 8 && v2_tiles_wg() != 512 && !share;
        int wg = 512;
        bool count = false;
        if (may_wide && may_count && two_fit(lds_count, 768, recb)) wg = 768, count = true;
        else if (may_count && two_fit(lds_count, 512, recb)) count = true;
        else if (may_wide && two_fit(lds_acc, 768, recb)) wg = 768;
        const size_t lds_dyn = count ? lds_count : lds_acc;
        if (count) kf |= EVK_VOXEL2_COUNT;
        const bool wide_wg = wg == 768;
        if (recb == 4) V2_TILES(512, 4);
        else if (wide_wg) V2_TILES(768, 8);
        else V2_TILES(512, 8);
#undef V2_TILES
    }
    return launch_status();
}

// t_norm of voxel_grid.py:134 exactly as the partition kernel computes it (time_norm, evk_part.h) -- the parity hook of
// its division shortcut, and the events' normalised time stamps for callers that want them
__global__ void k_normalise_time(const float *__restrict__ t, int64_t n, float t_first, float t_last, float bm1,
                                 float *__restrict__ out) {
    const TimeNorm k = make_time_norm(t_first, t_last, bm1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = time_norm(t[i], k);
}
extern "C" int evk_normalise_time_f32(const float *t, int64_t n, float t_first, float t_last, int B, float *out, void *stream) {
    if (n < 0 || B <= 0 || (n > 0 && (!t || !out))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_normalise_time<<<stream_grid(n), EVK_BLOCK, 0, (hipStream_t)stream>>>(t, n, t_first, t_last, (float)(B - 1), out);
    return launch_status();
}

#ifdef V2_PHASE_TIMING
extern "C" int evk_debug_phase_cycles(unsigned long long *host16) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(v2_phase_cycles), sizeof(z)) != hipSuccess) return EVK_EINVAL;
    return hipMemcpyToSymbol(HIP_SYMBOL(v2_phase_cycles), z, sizeof(z)) == hipSuccess ? EVK_OK : EVK_EINVAL;
}
extern "C" int evk_debug_tile_cycles(unsigned long long *host16) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(v2_tile_cycles), sizeof(z)) != hipSuccess) return EVK_EINVAL;
    return hipMemcpyToSymbol(HIP_SYMBOL(v2_tile_cycles), z, sizeof(z)) == hipSuccess ? EVK_OK : EVK_EINVAL;
}
#endif

extern "C" int evk_voxel2_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, int h, int wd,
                              int tile_w, int tile_h, float t_first, float t_last, int B, int flags, float *vox,
                              uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report,
                              uint32_t seq, void *stream) {
    if (n > 0 && (!x || !y || !t || !p)) return EVK_EINVAL;
    if (!(aligned16(x) && aligned16(y) && aligned16(t) && aligned16(p))) return EVK_EALIGN;
    const SrcF32 c{x, y, t, p};
    return voxel2(c, n, h, wd, tile_w, tile_h, t_first, t_last, B, flags, vox, index, scratch, scratch_bytes, oob, host_report,
                  seq, stream);
}

extern "C" int evk_voxel2_native_f32(const int16_t *x, const int16_t *y, int xy_stride, const void *t, int t_kind,
                                     double t_offset, const void *p, int p_kind, int64_t n, int h, int wd, int tile_w,
                                     int tile_h, float t_first, float t_last, int B, int flags, float *vox,
                                     uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob,
                                     uint32_t *host_report, uint32_t seq, void *stream) {
    ColsNative v;
    const int rc = native_cols(v, x, y, xy_stride, t, t_kind, t_offset, p, p_kind, n);
    if (rc != EVK_OK) return rc;
    if (!(aligned16(x) && (xy_stride == 2 || aligned16(y)) && aligned16(t) && aligned16(p))) return EVK_EALIGN;
    if (t_kind == EVK_T_F64) {   // two events per lane and load: every instruction contiguous over the wave (evk_part.h)
        const SrcNative<true> c{v.x, v.y, v.t, v.p, v.t_offset, v.xy_stride, v.p_kind};
        return voxel2(c, n, h, wd, tile_w, tile_h, t_first, t_last, B, flags, vox, index, scratch, scratch_bytes, oob, host_report,
                      seq, stream);
    }
    const SrcNative<false> c{v.x, v.y, v.t, v.p, v.t_offset, v.xy_stride, v.p_kind};
    return voxel2(c, n, h, wd, tile_w, tile_h, t_first, t_last, B, flags, vox, index, scratch, scratch_bytes, oob, host_report,
                  seq, stream);
}
