// One-pass partition for the voxel grid, round 3 (events_to_voxel_torch, voxel_grid.py:114-153): 4-byte records.
//
// As in round 2 every partition workgroup sorts SUB-CHUNKS of consecutive events by tile entirely in LDS and writes each
// sorted sub-chunk back as ONE contiguous, fully coalesced run, plus one 4-byte (start, count) entry per (sub-chunk,
// tile); the tile kernel walks its table column and pulls its segments out of the runs.  What changed is the record:
// the call moved 336 MB for 166 MB of algorithmic bytes (profiles/r02_pmc_traffic.json) because an 8-byte record was
// written and read back for every 16-byte event.  A record is now ONE 32-bit word,
//
//     [31:12] t_norm as a bit-pattern delta from the sub-chunk's base   [11:10] polarity code   [9:0] pixel in the tile
//
// * t_norm (the float32 of voxel_grid.py:134, computed here with the reference's arithmetic) is carried EXACTLY: the
//   events of a sub-chunk are consecutive in a time-sorted stream, so their t_norm values lie within a few thousand
//   float32 steps of the first one's.  base = the bit pattern of the sub-chunk's first event's t_norm (one word per
//   sub-chunk in `bases`), delta = bits(t_norm) - base < 2^20.
// * polarity code 0 / 1 / 2 = +1.0 / -1.0 / +0.0 -- what the reference's loaders and the bool / uint8 files produce.
// * anything else (another polarity value, a delta out of range: unsorted or sparse streams, the first events of a
//   stream where float32 steps are tiny, NaN from dt == 0) ESCAPES: code 3, and the delta field holds the index of an
//   8-byte {t_norm bits, polarity bits} entry in the sub-chunk's slice of a side array.  Exact for any float32 input.
// * every (sub-chunk, tile) segment starts on a 16-byte boundary of its run and is padded to whole 16-byte QUADS with
//   null records (code 3, delta field all ones): the tile kernel needs no begin / end tests, one lane = one quad.
//
// 16 B read + 4 B written by the partition, 4 B read by the tile kernel: 24 B/event instead of 32.  Half-size records
// also mean a 64 KB LDS buffer sorts 16 K events instead of 8 K (and the (key, t, p) triple of an event is compressed
// to its record word BEFORE the next sub-chunk's x, y loads are issued, so 16 events per thread fit the register file):
// the tile kernel pulls half as many segments of the same ~110 bytes.
#ifdef EVK_EXPERIMENTS   // measured, not adopted: tools/exp_build.sh builds it, EVK_VOXEL_PATH=v3 runs it
#include <cstdio>
#include <cstring>
#include <mutex>

#include "evk_part.h"
#include "evk_experiments.h"

namespace evk {

#define V3_LB 10                          // bits of the pixel-in-tile field: tiles of <= 1024 pixels
#define V3_LOCAL_MASK 0x3FFu
#define V3_CODE_SHIFT 10
#define V3_DELTA_SHIFT 12
#define V3_DELTA_LIMIT (1u << 20)
#define V3_NULL 0xFFFFFC00u               // code 3, delta field all ones, pixel 0
#define V3_NO_TILE 0xFFFFu
// ablation builds (tools/v3_ablate.sh; timing only, results are wrong): the partition kernel stops a sub-chunk after stage
// V3_ABLATE_P (1 loads + keys, 2 + histogram + scan, 3 + compression, 4 + placement; 9 = everything) / the tile kernel
// leaves out V3_ABLATE_T (1 the LDS atomics, 2 the record loads, 3 the chunk rounds altogether, 4 the table walk too)
#ifndef V3_ABLATE_P
#define V3_ABLATE_P 9
#endif
#ifndef V3_ABLATE_T
#define V3_ABLATE_T 0
#endif
#ifndef V3_TILES_MIN_WAVES
#define V3_TILES_MIN_WAVES 6  // waves per SIMD the tile kernel must fit (<= 80 registers): 3 workgroups of 8 waves per CU
#endif

struct Part3 {
    int S;          // events per sub-chunk (% 4 == 0, <= THREADS * EPT)
    int spad;       // record slots per run: S + 3 per tile of padding, % 4 == 0
    int per_block;  // consecutive sub-chunks per partition block
    int nsc;        // sub-chunks in the stream
    int nt_pad;     // table row stride: table[sub-chunk][tile]
    int nblk;
};

template <int THREADS, int EPT, int SCHED, typename C>
__global__ void __launch_bounds__(THREADS, 4) k_part3(const C c, int64_t n, TileGridG g, int ntiles, Part3 q, float t_first,
                                                                 float t_last, float bm1, int t_from_events,
                                                                 uint32_t *__restrict__ rec, uint2 *__restrict__ wide,
                                                                 uint32_t *__restrict__ table, uint32_t *__restrict__ bases,
                                                                 uint32_t *__restrict__ index, uint32_t cap, uint32_t *oob,
                                                                 uint32_t *host_report, uint32_t seq) {
    constexpr int G = C::G, NG = EPT / G;
    static_assert(EPT % G == 0 && EPT % 2 == 0, "events per thread");
    constexpr int PER_MAX = (VP_MAX_TILES + THREADS - 1) / THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *sorted = reinterpret_cast<uint32_t *>(smem);             // [THREADS * EPT + 3 * ntiles + 4]
    uint32_t *hist = sorted + ((THREADS * EPT + 3 * ntiles + 7) & ~3);  // [ntiles] counts, then cursors; [ntiles] = escape counter
    uint32_t *tmp = hist + ((ntiles + 4) & ~3);                         // [65] scan scratch
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int per = (ntiles + THREADS - 1) / THREADS;
    const int i0 = tid * per, i1 = (i0 + per < ntiles) ? i0 + per : ntiles;
    uint32_t mytot[PER_MAX];
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) mytot[k] = 0;
    uint32_t dropped = 0, nesc = 0;
    if (t_from_events) t_first = c.t1(0), t_last = c.t1(n - 1);   // ts[0], ts[-1] (voxel_grid.py:133)
    const TimeNorm tnorm = make_time_norm(t_first, t_last, bm1);

    // Group k of sub-chunk sc = G consecutive events of thread tid.  A group that is only partly inside the stream is
    // loaded whole (the over-read stays inside an aligned block; the extra events are ignored); a group entirely outside
    // is not loaded.  There is no scalar tail path: it would put branches -- and the compiler's waits -- between the loads.
    // (tl = the thread index as the LOOP BODY sees it: re-materialised through an empty asm in every iteration, so that the
    // dozen per-group offsets and predicates derived from it are recomputed -- a few VALU instructions -- instead of being
    // hoisted out of the loop into registers that stay live across it)
    int tl_ = tid;
    auto valid_in = [&](int sc, int k) -> int {  // events of group k inside the stream (<= 0: none)
        const int64_t lo = (int64_t)sc * q.S;
        const int64_t hi = (lo + q.S < n) ? lo + q.S : n;
        return (int)(hi - lo) - G * (tl_ + k * THREADS);
    };
    // Software pipeline over the two halves of an event: x, y are needed first (tile key, histogram), t, p only for the
    // record.  t, p of sub-chunk j are loaded after its keys and land during its histogram + scan; x, y of j + 1 are loaded
    // once (key, t, p) of j have been compressed to record words, and land during the placement + write-out.
    const int sc0 = blockIdx.x * q.per_block;
    const int sc_end = (sc0 + q.per_block < q.nsc) ? sc0 + q.per_block : q.nsc;
    uint32_t xyr[NG * C::XYW], tpr[NG * C::TPW];   // RAW loaded words: decoded where they are used (evk_part.h)
    float tb = 0.0f;
    // (the loads are unconditional -- a group entirely outside the stream re-reads the sub-chunk's first group -- so that
    // the loop body is straight-line code: branches between the loads cost waits and registers)
    // Addresses are (uniform base of the row of THREADS groups) + (one 32-bit lane offset): scalar registers and the
    // saddr form of the load, not a 64-bit VGPR pair per load.
    auto row_base = [&](int sc, int k) -> int64_t {   // first event of group row k; a row entirely outside: the first row
        const int64_t lo = (int64_t)sc * q.S, hi = (lo + q.S < n) ? lo + q.S : n, r = lo + (int64_t)G * k * THREADS;
        return r < hi ? r : lo;
    };
    auto load_xy = [&](int sc) {
#pragma unroll
        for (int k = 0; k < NG; ++k)
            c.load_xy(row_base(sc, k), valid_in(sc, k) > 0 ? (uint32_t)tl_ : 0u, xyr + C::XYW * k);
    };
    auto load_tp = [&](int sc) {
#pragma unroll
        for (int k = 0; k < NG; ++k)
            c.load_tp(row_base(sc, k), valid_in(sc, k) > 0 ? (uint32_t)tl_ : 0u, tpr + C::TPW * k);
        tb = c.t1((int64_t)sc * q.S);   // the sub-chunk's first event: base of the t_norm deltas (same address in every lane)
    };
    // Two schedules of one sub-chunk's phases (SCHED), measured against each other on the GPU (tools/v3_sweep.sh):
    //   0: keys | t, p loads | histogram, scan | compress | x, y loads of j + 1 | placement | wait | write-out
    //      -- a load burst overlaps the two phases after it; 3.5 registers per event (keys OR records + one column pair)
    //   1: keys, compress | x, y, t, p loads of j + 1 | histogram, scan, placement | wait | write-out
    //      -- ONE burst per sub-chunk that overlaps everything but the keys and the compression; 5.5 registers per event
    uint32_t kl[EPT], w[EPT], tl[EPT / 2];
    uint32_t kept = 0, bbits = 0;
    uint32_t tval[PER_MAX];   // this thread's table entries: stored with the write-out (no store before the next loads)
    auto tile_at = [&](int s) -> uint32_t { return (tl[s >> 1] >> (16 * (s & 1))) & 0xFFFFu; };
    auto do_keys = [&](int sc) {   // tile key + pixel in tile of every event
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int nv = valid_in(sc, k);
#pragma unroll
            for (int e = 0; e < G; ++e) {
                uint32_t local = 0;
                const int key = c.key_of(xyr + C::XYW * k, e, g, local);   // (of stale words beyond the stream)
                kl[G * k + e] = ((key >= 0) & (e < nv)) ? (((uint32_t)key << V3_LB) | local) : 0xFFFFFFFFu;
                dropped += ((key < 0) & (e < nv)) ? 1u : 0u;
                asm volatile("" : "+v"(dropped));   // counted HERE: sunk to the end of the loop body it kept a second copy of every key alive
                // one event at a time: GCN issues dependent VALU instructions back to back, while interleaving the EPT
                // independent chains (what the scheduler does for ILP) keeps ~4 temporaries per event live at once
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int s = 0; s < EPT; ++s) asm volatile("" : "+v"(kl[s])::"memory");  // keys first, any load after
    };
    auto do_hist = [&]() {   // no-return LDS atomics
#pragma unroll
        for (int s = 0; s < EPT; ++s) {
            const uint32_t tile = SCHED == 0 ? (kl[s] == 0xFFFFFFFFu ? V3_NO_TILE : kl[s] >> V3_LB) : tile_at(s);
            if (tile != V3_NO_TILE) __hip_atomic_fetch_add(&hist[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto do_scan = [&]() {   // exclusive scan of the PADDED tile counts (thread tid owns tiles [i0, i1)) -> cursors; table row; null padding
        uint32_t cnts[PER_MAX];
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < PER_MAX; ++k) {
            const int i = i0 + k;
            cnts[k] = (k < per && i < i1) ? hist[i] : 0u;
            mine += (cnts[k] + 3u) & ~3u;
        }
        uint32_t run = wg_excl_scan<THREADS>(mine, tmp, kept);   // kept = slots of the run, padding included (% 4 == 0)
#pragma unroll
        for (int k = 0; k < PER_MAX; ++k) {
            const int i = i0 + k;
            tval[k] = 0;
            if (k < per && i < i1) {
                const uint32_t cnt = cnts[k], pc = (cnt + 3u) & ~3u;
                hist[i] = run;
                tval[k] = (run >> 2) | ((pc >> 2) << 16);   // first quad, quads
                for (uint32_t j = cnt; j < pc; ++j) sorted[run + j] = V3_NULL;
                mytot[k] += cnt;
                run += pc;
            }
        }
    };
    auto do_compress = [&](int64_t slot0) {   // (key, t, p) -> record word + 16-bit tile
        bbits = __float_as_uint(time_norm(tb, tnorm));
#pragma unroll
        for (int s = 0; s < EPT / 2; ++s) tl[s] = 0;
#pragma unroll
        for (int s = 0; s < EPT; ++s) {
            const float tn = time_norm(c.t_of(tpr + C::TPW * (s / G), s % G), tnorm);  // voxel_grid.py:134, bit-identical
            const uint32_t pb = __float_as_uint(c.p_of(tpr + C::TPW * (s / G), s % G));
            const uint32_t d = __float_as_uint(tn) - bbits;
            // +1.0 -> 0, -1.0 -> 1, +0.0 -> 2, anything else -> 3 (written so that it stays two selects: a chain of equality
            // tests on one value becomes a switch with divergent branches)
            const uint32_t code = (pb & 0x7FFFFFFFu) == 0x3F800000u ? pb >> 31 : 3u - (uint32_t)(pb == 0u);
            const bool live = kl[s] != 0xFFFFFFFFu;
            const bool esc = live & ((code == 3u) | (d >= V3_DELTA_LIMIT));
            uint32_t word = (d << V3_DELTA_SHIFT) | (code << V3_CODE_SHIFT) | (kl[s] & V3_LOCAL_MASK);
            if (__any(esc)) {   // rare, wave-uniform test: exact {t_norm, polarity} to the side array, its index into the record
                if (esc) {
                    const uint32_t e = atomicAdd(&hist[ntiles], 1u);
                    wide[slot0 + e] = make_uint2(__float_as_uint(tn), pb);
                    word = (e << V3_DELTA_SHIFT) | (3u << V3_CODE_SHIFT) | (kl[s] & V3_LOCAL_MASK);
                    ++nesc;
                }
            }
            w[s] = word;
            tl[s >> 1] |= (live ? (kl[s] >> V3_LB) : V3_NO_TILE) << (16 * (s & 1));
            __builtin_amdgcn_sched_barrier(0);   // one division at a time: interleaved they cost ~5 registers of temporaries each
        }
    };
    auto fence = [&]() {   // for the compiler: loads hoisted above a compute phase keep their 2 * EPT registers live through it
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto do_place = [&]() {   // a returning LDS atomic on the tile's cursor hands every event its slot of the sorted buffer
#pragma unroll
        for (int s = 0; s < EPT; ++s) {
            const uint32_t tile = tile_at(s);
            if (tile != V3_NO_TILE) {
                const uint32_t pos = atomicAdd(&hist[tile], 1u);
                sorted[pos] = w[s];
            }
        }
    };
    auto do_writeout = [&](int sc, int64_t slot0) {   // one contiguous, coalesced run of `kept` record slots; the table row
        const uint4 *src = reinterpret_cast<const uint4 *>(sorted);
        uint4 *dst = reinterpret_cast<uint4 *>(rec + slot0);
        const int n16 = (int)(kept >> 2);
        for (int i = tid; i < n16; i += THREADS) dst[i] = src[i];
        uint32_t *trow = table + (int64_t)sc * q.nt_pad;
#pragma unroll
        for (int k = 0; k < PER_MAX; ++k)
            if (k < per && i0 + k < i1) trow[i0 + k] = tval[k];
        if (tid == 0) bases[sc] = bbits;
    };
    for (int i = tid; i <= ntiles; i += THREADS) hist[i] = 0;   // (every pass leaves it zero again)
    if (sc0 < sc_end) {
        load_xy(sc0);
        if (SCHED == 1) load_tp(sc0);
    }
    // vmcnt(0) as a BUILTIN, so that the compiler's wait-count pass knows the loads are in: their first use, at the top of the
    // loop, would otherwise get an `s_waitcnt vmcnt(0)` of its own -- and that one also waits for the write-out stores of the
    // previous sub-chunk (gfx9 has ONE counter for loads and stores)
    EVK_WAIT_VM0();
    lds_only_barrier();
    for (int sc = sc0; sc < sc_end; ++sc) {
        asm volatile("" : "+v"(tl_));
        const int64_t slot0 = (int64_t)sc * q.spad;
        do_keys(sc);
        if constexpr (SCHED == 0) {
            load_tp(sc);         // land during the histogram and the scan
            if (V3_ABLATE_P >= 2) {
                do_hist();
                lds_only_barrier();  // histogram complete
                do_scan();
                lds_only_barrier();  // cursors complete (also frees tmp)
            }
            if (V3_ABLATE_P >= 3) do_compress(slot0);
            else {
                uint32_t sink = 0;
#pragma unroll
                for (int s = 0; s < EPT; ++s) sink += kl[s] ^ tpr[s % (NG * C::TPW)], w[s] = sink;
#pragma unroll
                for (int s = 0; s < EPT / 2; ++s) tl[s] = sink == 0x12345u ? 0u : 0xFFFFFFFFu;
            }
            EVK_WAIT_VM0();      // nothing outstanding (t, p are in; the previous stores are a histogram and a scan old)
            fence();
            if (sc + 1 < sc_end) load_xy(sc + 1);  // in flight during the placement
            fence();
        } else {
            do_compress(slot0);
            fence();
            if (sc + 1 < sc_end) load_xy(sc + 1), load_tp(sc + 1);  // in flight during the histogram, the scan and the placement
            fence();
            do_hist();
            lds_only_barrier();
            do_scan();
            lds_only_barrier();
        }
        if (V3_ABLATE_P >= 4) do_place();
        lds_only_barrier();
        EVK_WAIT_VM0();   // the next sub-chunk's columns have landed: the stores below then never sit between a load and its use
        if (V3_ABLATE_P >= 9) do_writeout(sc, slot0);
        for (int i = tid; i <= ntiles; i += THREADS) hist[i] = 0;   // cursors are dead: zero for the next pass
        lds_only_barrier();  // sorted / hist are rewritten by the next sub-chunk
    }
    if (dropped && oob) atomicAdd(oob, dropped);
    // ---- totals -> global; the last block to arrive builds the work-item plan
    uint32_t *gidx = index;
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) {
        const int i = i0 + k;
        if (k < per && i < i1 && mytot[k])
            __hip_atomic_fetch_add(gidx + VP_TOTALS + i, mytot[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (nesc) __hip_atomic_fetch_add(gidx + 3, nesc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0 && tid == 0 && n > 0) {
        __hip_atomic_store(gidx + 0, __float_as_uint(c.t1(0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gidx + 1, __float_as_uint(c.t1(n - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t prev = __hip_atomic_fetch_add(gidx + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == gridDim.x - 1);
        if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!is_last) return;
    // ---- plan: part_start, per-tile combine counters, item -> tile; totals / ticket back to 0.  A tile with more than
    //      `cap` events is split over several workgroups, by sub-chunk range.
    uint32_t *part_start = index + VP_PART, *counters = index + VP_COUNTER(ntiles), *item_tile = index + VP_ITEM(ntiles);
    uint32_t tot[PER_MAX];
    uint32_t np = 0;
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) {
        const int i = i0 + k;
        tot[k] = 0;
        if (k < per && i < i1) {
            tot[k] = __hip_atomic_load(gidx + VP_TOTALS + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gidx + VP_TOTALS + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            np += tot[k] > cap ? (tot[k] + cap - 1) / cap : 1u;
        }
    }
    uint32_t total_parts;
    uint32_t prun = wg_excl_scan<THREADS>(np, tmp, total_parts);
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) {
        const int i = i0 + k;
        if (k < per && i < i1) {
            const uint32_t parts = tot[k] > cap ? (tot[k] + cap - 1) / cap : 1u;
            part_start[i] = prun;
            counters[i] = 0;
            for (uint32_t jj = 0; jj < parts; ++jj) item_tile[prun + jj] = (uint32_t)i;
            prun += parts;
        }
    }
    if (tid == 0) {
        part_start[ntiles] = total_parts;
        __hip_atomic_store(gidx + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (host_report) {  // every workgroup's dropped-event count is in *oob (added before its ticket): tell the host,
                            // in pinned memory, so that a deferred error check costs no copy and no event on the stream
            const uint32_t cnt = oob ? __hip_atomic_load(oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            __hip_atomic_store(host_report + 1, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_report, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Voxel tiles from the sorted runs: one workgroup per work item (tile, or part of a hot tile = a range of sub-chunks).
// A tile's records sit in one padded segment per sub-chunk (whole 16-byte quads).  Every thread fetches the table entry
// and the base of one sub-chunk; each wave then cuts its 64 segments into CHUNKS of <= 4 quads (64 bytes), lists the
// chunks in LDS (wave scan of the chunk counts) and hands them out to groups of 4 lanes, one quad = 4 records per lane:
// all lanes stay busy whatever the segment lengths are, U chunk loads per lane in flight.  Segments longer than
// V3_MAX_CHUNKS chunks (clustered scenes) are streamed by the whole wave instead.
#define V3_MAX_CHUNKS 7
#define V3_CHUNK_CAP (64 * V3_MAX_CHUNKS)  // list entries per wave: 28 KB for 8 waves
#define V3_FIXED_ONE 4294967296.0           // deterministic mode: cells are int64 multiples of 2^-32
template <int WG, int U, bool SPLIT, bool FIXED>
__global__ void __launch_bounds__(WG, V3_TILES_MIN_WAVES) k_voxel_tiles3(const uint32_t *__restrict__ rec, const uint2 *__restrict__ wide,
                                                     const uint32_t *__restrict__ table, const uint32_t *__restrict__ bases,
                                                     uint32_t *__restrict__ index, TileGridG g, Part3 q, int B, int flags,
                                                     float *__restrict__ vox, float *__restrict__ staging) {
    constexpr int NW = WG / 64;
    const int overwrite = flags & EVK_VOXEL_OVERWRITE;
    constexpr bool split = SPLIT;
    const int NB = split ? 2 * B : B;
    const float bm1 = (float)(B - 1);
    extern __shared__ __attribute__((aligned(16))) acc_t acc[];
    // chunk list of every wave: {(first quad of the chunk << 2) | (quads - 1), base of the sub-chunk}: ONE LDS read gives a
    // lane group everything it needs for its load and its decode
    __shared__ uint2 cseg[NW][V3_CHUNK_CAP];
    const int ntiles = g.tiles_x * g.tiles_y;
    const uint32_t *part_start = index + VP_PART, *item_tile = index + VP_ITEM(ntiles);
    const uint32_t nitems = part_start[ntiles];
    if (blockIdx.x >= nitems) return;
    // XCD-aware work-item order: workgroup b runs on XCD b % 8, and the segments of NEIGHBOURING tiles are neighbours in
    // every run (and their table entries share a cache line), so XCD k takes a contiguous range of the tile-ordered items
    uint32_t item = blockIdx.x;
    if (!(flags & EVK_VOXEL2_NO_XCD_ORDER)) {
        const uint32_t k = blockIdx.x & 7u, j = blockIdx.x >> 3, q8 = nitems >> 3, r8 = nitems & 7u;
        item = k * q8 + (k < r8 ? k : r8) + j;
    }
    const int tw = g.tw, th = g.th, tpix = tw * th;
    // accumulator rows of tw + 1 cells: an ODD pitch (the events of a real scene sit on edges -- same column, different
    // rows -- and with 32 cells = 64 dwords per row every row of a column would share one pair of banks)
    const int tpitch = g.pitch, ppix = tpitch * th;
    const int tile = (int)item_tile[item];
    const uint32_t first_item = part_start[tile], nparts = part_start[tile + 1] - first_item;
    const uint32_t part_id = item - first_item;
    const int tx0 = (tile % g.tiles_x) * tw, ty0 = (tile / g.tiles_x) * th;
    for (int i = threadIdx.x; i < NB * ppix; i += WG) acc[i] = 0.0;   // (+0.0 and int64 0 are the same bits)
    const int sc_lo = (int)(((int64_t)q.nsc * part_id) / nparts), sc_hi = (int)(((int64_t)q.nsc * (part_id + 1)) / nparts);
    const uint32_t *col = table + tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 3, grp = lane >> 2;
    const uint32_t spad4 = (uint32_t)q.spad >> 2;
    auto add = [&](acc_t *a, float v) {
        if constexpr (FIXED) {
            // order-free accumulation: int64 multiples of 2^-32.  |v| < 2^30 keeps 2^33 such adds inside int64.
            const double s = (double)v * V3_FIXED_ONE;
            if (!(fabs(s) < 4.0e18)) {   // NaN, infinity or a weight beyond 2^30: not representable -- counted, the wrapper raises
                __hip_atomic_fetch_add(index + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a), (unsigned long long)__double2ll_rn(s), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            lds_add(a, v);
        }
    };
    auto bins_general = [&](acc_t *base, int local, float tn, float p) {   // any t_norm (voxel_bins_lds)
        if (tn != tn) {  // dt == 0 (Q9): NaN in every bin of the pixel
            for (int b = 0; b < B; ++b) add(base + b * ppix + local, tn * p);
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
    auto one = [&](uint32_t word, uint32_t bbits, uint32_t slot) {
        const uint32_t code = (word >> V3_CODE_SHIFT) & 3u;
        const int local = (int)(word & V3_LOCAL_MASK);   // row * pitch + column
        float tn, p;
        if (code == 3u) {
            if ((word >> V3_DELTA_SHIFT) == 0xFFFFFu) return;   // padding
            // rare: the exact {t_norm, polarity} from the side array.  The wait stays INSIDE the branch (builtin: the
            // compiler's scoreboard sees it) -- at the join it would be a vmcnt(0) on every record
            const uint2 e = wide[(uint64_t)(slot / (uint32_t)q.spad) * (uint32_t)q.spad + (word >> V3_DELTA_SHIFT)];
            EVK_WAIT_VM0();
            tn = __uint_as_float(e.x), p = __uint_as_float(e.y);
        } else {
            tn = __uint_as_float(bbits + (word >> V3_DELTA_SHIFT));
            p = __uint_as_float(code == 2u ? 0u : (0x3F800000u | (code << 31)));
        }
        if (__builtin_expect(tn >= 0.0f && tn <= bm1, 1)) {
            // the common case, straight-line: t inside [ts[0], ts[-1]].  Bins b0 = floor(t_norm) and b0 + 1 with the
            // weights of voxel_grid.py:138 -- 1 - |t_norm - b| evaluated exactly as there (for b0 the absolute value is
            // the identity; max(0, .) cannot bind for these two bins).  A zero weight is added like any other (x + 0 = x;
            // the reference's index_put_ adds it too).
            acc_t *a = acc + local;
            float wgt = p;
            if constexpr (split) {
                if (!(p > 0.0f) && !(p <= 0.0f)) return;  // a NaN polarity is in neither grid
                a += p > 0.0f ? 0 : B * ppix;
                wgt = 1.0f;
            }
            const int b0 = (int)tn;
            const float v0 = wgt * (1.0f - (tn - (float)b0)), v1 = wgt * (1.0f - fabsf(tn - (float)(b0 + 1)));
            a += b0 * ppix;
            add(a, v0);
            if (b0 + 1 < B) add(a + ppix, v1);
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
    // The 4 records of quad q4 (`mine` false: this lane has no quad in this round).  ~99.9 % of all rounds hold nothing but
    // ordinary records -- a polarity code, t_norm inside [0, B - 1] -- and padding: those take STRAIGHT-LINE code, one
    // wave-uniform branch per quad.  (Per-record branches for the escape, the padding, the out-of-range bins and the last
    // bin were ~14 s_cbranch and ~230 instructions of text per record: the tile kernel was bound by instruction issue,
    // not by its LDS atomics or its record loads.)
    auto quad = [&](const uint4 &v, uint32_t bbits, uint32_t q4, bool mine) {
        const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
        float tn[4];
        bool real[4], bad = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tn[i] = __uint_as_float(bbits + (wd[i] >> V3_DELTA_SHIFT));
            real[i] = mine & (wd[i] != V3_NULL);
            bad |= real[i] & ((((wd[i] >> V3_CODE_SHIFT) & 3u) == 3u) | !(tn[i] >= 0.0f) | !(tn[i] <= bm1));
        }
        if (__builtin_expect(__any(bad), 0)) {
            if (mine) one(v.x, bbits, 4u * q4), one(v.y, bbits, 4u * q4 + 1u), one(v.z, bbits, 4u * q4 + 2u), one(v.w, bbits, 4u * q4 + 3u);
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // bins b0 = floor(t_norm) and b0 + 1 with the weights of voxel_grid.py:138 -- 1 - |t_norm - b| evaluated exactly as
            // there (for b0 the absolute value is the identity; max(0, .) cannot bind for these two bins).  A zero weight is
            // added like any other (x + 0 = x; the reference's index_put_ adds it too); at t_norm == B - 1 the upper bin does
            // not exist and its weight is exactly 0: it is added to the lower bin's cell instead of being branched around.
            const uint32_t code = (wd[i] >> V3_CODE_SHIFT) & 3u;
            const int b0 = (int)tn[i];
            acc_t *a = acc + (int)(wd[i] & V3_LOCAL_MASK) + b0 * ppix;   // row * pitch + column, plane b0
            float wgt = __uint_as_float(code == 2u ? 0u : (0x3F800000u | (code << 31)));
            if constexpr (split) {
                a += code == 0u ? 0 : B * ppix;   // +1 -> the grid of the positive events; -1, 0 -> the other one
                wgt = 1.0f;
            }
            const float v0 = wgt * (1.0f - (tn[i] - (float)b0)), v1 = wgt * (1.0f - fabsf(tn[i] - (float)(b0 + 1)));
            if (V3_ABLATE_T == 1) {
                if (real[i] && v0 + v1 == 1.2345e-30f) a[0] = 1.0;
                continue;
            }
            if (real[i]) {
                if constexpr (FIXED) {   // |v| <= 1 here: no range test
                    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a), (unsigned long long)__double2ll_rn((double)v0 * V3_FIXED_ONE),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a + (b0 + 1 < B ? ppix : 0)),
                                           (unsigned long long)__double2ll_rn((double)v1 * V3_FIXED_ONE), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    lds_add(a, v0);
                    lds_add(a + (b0 + 1 < B ? ppix : 0), v1);
                }
            }
        }
    };
    // Entries go to the threads in equal batches, INTERLEAVED over the waves (slot = lane * NW + wave): a short range -- the
    // last batch of a tile, or one of the many parts of a hot tile -- then still gives every wave its share
    const int range = sc_hi - sc_lo, nbatch = (range + WG - 1) / WG;
    const int bsz = nbatch ? ((range + nbatch - 1) / nbatch + NW - 1) / NW * NW : NW;  // <= WG, a multiple of NW
    const int slot = lane * NW + wave;
    uint32_t ent_next = 0, base_next = 0;
    {
        const int my = sc_lo + slot;
        if (slot < bsz && my < sc_hi) ent_next = col[(int64_t)my * q.nt_pad], base_next = bases[my];
    }
    const uint4 *rec4 = reinterpret_cast<const uint4 *>(rec);
    for (int base = sc_lo; base < (V3_ABLATE_T >= 4 ? sc_lo : sc_hi); base += bsz) {
        const uint32_t ent = ent_next, bb = base_next;
        {   // next batch's entries: in flight while this batch is processed
            const int my = base + bsz + slot;
            const bool have = slot < bsz && my < sc_hi;
            ent_next = have ? col[(int64_t)my * q.nt_pad] : 0u;
            base_next = have ? bases[my] : 0u;
        }
        const uint32_t startq = ent & 0xFFFFu, nq = ent >> 16;
        const uint32_t nch = (nq + 3u) >> 2;
        const bool is_long = nch > (uint32_t)V3_MAX_CHUNKS;
        const uint32_t mych = is_long ? 0u : nch;
        uint32_t incl = mych;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        const uint32_t total = __shfl(incl, 63, 64), excl = incl - mych;
        __syncthreads();  // (a) accumulators are zero before the first adds; (b) the previous batch's list is consumed
        const uint32_t q0 = (uint32_t)(base + slot) * spad4 + startq;   // first quad of this thread's segment
        for (uint32_t k = 0; k < mych; ++k) {
            const uint32_t left = nq - 4u * k;
            cseg[wave][excl + k] = make_uint2(((q0 + 4u * k) << 2) | ((left < 4u ? left : 4u) - 1u), bb);
        }
        __syncthreads();
        // Chunk rounds, software-pipelined in two stages: list entries + record loads of round r + 1 | accumulation of round r
        // (a third stage -- list entries two rounds ahead, as in round 2 -- costs 4 registers the straight-line accumulate
        // needs more).  The loads are UNCONDITIONAL -- a lane without a quad reads the head of the
        // record buffer -- so that nothing but arithmetic sits between them and the compiler can wait for the older round
        // alone (`vmcnt(U)`).  The two workgroup barriers per batch are kept on purpose: all tiles walking the runs in step
        // keeps each run L2-hot while its segments are pulled.
        auto meta = [&](uint32_t j0, uint2(&cs)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t j = j0 + 16u * u + grp;
                cs[u] = j < total ? cseg[wave][j] : make_uint2(0xFFFFFFFFu, 0u);
            }
        };
        auto mine_q = [&](const uint2 &cs) -> bool { return cs.x != 0xFFFFFFFFu && (uint32_t)sub <= (cs.x & 3u); };
        auto fire = [&](const uint2(&cs)[U], uint4(&v)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = rec4[(V3_ABLATE_T != 2 && mine_q(cs[u])) ? (cs[u].x >> 2) + sub : (uint32_t)sub];
        };
        auto eat = [&](const uint2(&cs)[U], const uint4(&v)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) quad(v[u], cs[u].y, (cs[u].x >> 2) + sub, mine_q(cs[u]));
        };
        constexpr uint32_t step = 16u * U;
        if (V3_ABLATE_T < 3) {
            uint2 ca[U], cb[U];
            uint4 va[U], vb[U];
            meta(0u, ca);
            fire(ca, va);
            for (uint32_t j0 = 0; j0 < total; j0 += 2u * step) {
                meta(j0 + step, cb);
                fire(cb, vb);               // round j0 + step in flight ...
                eat(ca, va);                // ... while round j0 is accumulated
                meta(j0 + 2u * step, ca);
                fire(ca, va);
                eat(cb, vb);
            }
        }
        // long segments: the whole wave streams each of them, one quad per lane
        uint64_t m = __ballot(is_long);
        while (m) {
            const int s = __builtin_ctzll(m);
            m &= m - 1;
            const uint32_t e2 = __shfl(ent, s, 64), b2 = __shfl(bb, s, 64);
            const uint32_t qb = (uint32_t)(base + s * NW + wave) * spad4 + (e2 & 0xFFFFu), qe = qb + (e2 >> 16);  // lane s's sub-chunk
            for (uint32_t q0w = qb; q0w < qe; q0w += 64u) {   // (wave-uniform trip count: quad() votes across the wave)
                const uint32_t q4 = q0w + lane;
                const bool in = q4 < qe;
                const uint4 v = rec4[in ? q4 : qb];
                quad(v, b2, q4, in);
            }
        }
    }
    __syncthreads();
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
        const acc_t a = acc[b * ppix + row * tpitch + col];
        if constexpr (FIXED) return (float)((double)__builtin_bit_cast(long long, a) * (1.0 / V3_FIXED_ONE));
        return (float)a;
    };
    if (nparts == 1) {
        flush(lds_cell);
        return;
    }
    // split (hot) tile: partial tiles to staging, the last part to arrive sums them in part order
    const int cells = NB * tpix;
    float *mine = staging + (int64_t)item * cells;
    for (int c = threadIdx.x; c < cells; c += WG) mine[c] = lds_cell(c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int is_last;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t *counter = index + VP_COUNTER(ntiles) + tile;
        const uint32_t prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == nparts - 1);
        if (is_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!is_last) return;
    const float *parts = staging + (int64_t)first_item * cells;
    flush([&](int c) {
        float sum = 0.0f;
        for (uint32_t p = 0; p < nparts; ++p) sum += parts[(int64_t)p * cells + c];
        return sum;
    });
}

// ---- host-side geometry -----------------------------------------------------------------------------------------
#define V3_MIN_SUBCHUNK 8192
// Partition geometry: threads per workgroup x events per thread x schedule (k_part3).  Default 1024 x 16 x 0; a call that
// has to share its CUs with another kernel's workgroups (EVK_VOXEL2_SHARE_CU: multi-rank jobs, the collective's channels)
// takes 8 events per thread (half the LDS).  EVK_V3_GEO=TxExS selects another compiled geometry (experiments).
struct V3Geo {
    int threads, ept, sched;
};
// geometries compiled into the library: the two the default dispatch reaches; more under -DEVK_EXPERIMENTS (tools builds)
#ifdef EVK_EXPERIMENTS
#define V3_GEOMETRIES(X) X(1024, 16, 0) X(1024, 8, 0) X(1024, 12, 0) X(1024, 8, 1) X(1024, 12, 1) X(512, 16, 0) X(512, 16, 1) X(512, 12, 1) X(512, 8, 1)
#else
#define V3_GEOMETRIES(X) X(1024, 16, 0) X(1024, 8, 0)
#endif
static bool v3_geo_compiled(const V3Geo &g) {
#define X(T, E, S) if (g.threads == T && g.ept == E && g.sched == S) return true;
    V3_GEOMETRIES(X)
#undef X
    return false;
}

static V3Geo v3_geo(bool share) {
    static const V3Geo forced = [] {
        V3Geo g{0, 0, 0};
        const char *s = getenv("EVK_V3_GEO");
        if (s && sscanf(s, "%dx%dx%d", &g.threads, &g.ept, &g.sched) == 3 && v3_geo_compiled(g)) return g;
        return V3Geo{0, 0, 0};
    }();
    if (forced.threads) return forced;
    return share ? V3Geo{1024, 8, 0} : V3Geo{1024, 16, 0};
}

static Part3 v3_geometry(int64_t n, int ntiles, bool share) {
    const V3Geo geo = v3_geo(share);
    const int64_t smax = (int64_t)geo.threads * geo.ept;
    int64_t nblk = (n + V3_MIN_SUBCHUNK - 1) / V3_MIN_SUBCHUNK;
    const int64_t max_blk = (int64_t)EVK_NUM_CU * (1024 / geo.threads);
    if (nblk > max_blk) nblk = max_blk;
    if (nblk < 1) nblk = 1;
    int64_t per_block = (n + nblk * smax - 1) / (nblk * smax);
    if (per_block < 1) per_block = 1;
    int64_t S = (n + nblk * per_block - 1) / (nblk * per_block);
    S = (S + 3) & ~(int64_t)3;
    if (S < 4) S = 4;
    Part3 q;
    q.S = (int)S, q.per_block = (int)per_block, q.nblk = (int)nblk;
    q.spad = (int)((S + 3 * (int64_t)ntiles + 3) & ~(int64_t)3);
    q.nsc = (int)((n + S - 1) / S);
    if (q.nsc < 1) q.nsc = 1;
    q.nt_pad = (ntiles + 15) & ~15;
    return q;
}
static inline int64_t v3_al256(int64_t b) { return (b + 255) & ~(int64_t)255; }

struct V3Layout {
    int64_t table, bases, rec, wide, staging, total;
};
static V3Layout v3_layout(int ntiles, int64_t n, int planes, int tw, int th, bool share) {
    const Part3 q = v3_geometry(n, ntiles, share);
    const int64_t slots = (int64_t)q.nsc * q.spad;
    V3Layout L;
    L.table = 0;
    L.bases = v3_al256((int64_t)q.nsc * q.nt_pad * 4);
    L.rec = L.bases + v3_al256((int64_t)q.nsc * 4);
    L.wide = L.rec + v3_al256(slots * 4);
    L.staging = L.wide + v3_al256(slots * 8);
    L.total = L.staging + v3_al256((int64_t)bucket_max_items(n, ntiles) * ((int64_t)planes * tw * th) * 4);
    return L;
}

static size_t v3_part_lds(int threads, int ept, int ntiles) {
    return (size_t)((threads * ept + 3 * ntiles + 7) & ~3) * 4 + (size_t)((ntiles + 4) & ~3) * 4 + 65 * 4 + 16;
}

}  // namespace evk

using namespace evk;

extern "C" int64_t evk_voxel3_index_len(int ntiles, int64_t n) {
    if (ntiles <= 0 || ntiles > VP_MAX_TILES || n < 0) return 0;
    return (int64_t)VP_ITEM(ntiles) + bucket_max_items(n, ntiles);
}

extern "C" int64_t evk_voxel3_scratch_bytes(int ntiles, int64_t n, int planes, int tile_w, int tile_h) {
    if (ntiles <= 0 || n < 0 || planes <= 0 || tile_w <= 0 || tile_h <= 0) return 0;
    const int64_t a = v3_layout(ntiles, n, planes, tile_w, tile_h, false).total;
    const int64_t b = v3_layout(ntiles, n, planes, tile_w, tile_h, true).total;
    return a > b ? a : b;
}

// largest tile count the partition kernel's LDS holds
extern "C" int evk_voxel3_max_tiles(void) {
    int t = VP_MAX_TILES;
    while (t > 0 && v3_part_lds(1024, 16, t) > 160 * 1024 - 512) t -= 16;
    return t;
}

template <int THREADS, int EPT, int SCHED, typename C>
static void launch_part3(const C &c, int64_t n, const TileGridG &g, int ntiles, const Part3 &q, float t_first, float t_last,
                         float bm1, int tfe, uint32_t *rec, uint2 *wide, uint32_t *table, uint32_t *bases, uint32_t *index,
                         uint32_t *oob, uint32_t *host_report, uint32_t seq, hipStream_t s) {
    static std::once_flag once[64];   // per device: the attribute is a property of the loaded code object
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [] {
        (void)hipFuncSetAttribute((const void *)k_part3<THREADS, EPT, SCHED, C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024 - 256);
    });
    k_part3<THREADS, EPT, SCHED, C><<<q.nblk, THREADS, v3_part_lds(THREADS, EPT, ntiles), s>>>(
        c, n, g, ntiles, q, t_first, t_last, bm1, tfe, rec, wide, table, bases, index, (uint32_t)bucket_cap(n, ntiles), oob, host_report, seq);
}

template <bool SPLIT, bool FIXED>
static void launch_tiles3(int items, size_t lds_acc, hipStream_t s, const uint32_t *rec, const uint2 *wide, const uint32_t *table,
                          const uint32_t *bases, uint32_t *index, const TileGridG &g, const Part3 &q, int B, int kf, float *vox,
                          float *staging) {
    if (lds_acc > 32 * 1024) {
        static std::once_flag once[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::call_once(once[dev & 63], [] {
            (void)hipFuncSetAttribute((const void *)k_voxel_tiles3<512, 2, SPLIT, FIXED>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      128 * 1024);
        });
    }
    k_voxel_tiles3<512, 2, SPLIT, FIXED><<<items, 512, lds_acc, s>>>(rec, wide, table, bases, index, g, q, B, kf, vox, staging);
}

template <typename C>
static int voxel3(const C &c, int64_t n, int h, int wd, int tile_w, int tile_h, float t_first, float t_last, int B,
                  int flags, float *vox, uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob,
                  uint32_t *host_report, uint32_t seq, void *stream) {
    TileGridG g;
    const int known = EVK_VOXEL_OVERWRITE | EVK_VOXEL_SPLIT_POLARITY | EVK_VOXEL_T_FROM_EVENTS | EVK_VOXEL2_PARTITION_ONLY |
                      EVK_VOXEL2_TILES_ONLY | EVK_VOXEL2_NO_XCD_ORDER | EVK_VOXEL2_SHARE_CU | EVK_VOXEL_DETERMINISTIC;
    if (make_grid_g(g, h, wd, tile_w, tile_h) != EVK_OK || B <= 0 || !vox || !index || !scratch || n <= 0 || (flags & ~known))
        return EVK_EINVAL;
    const int ntiles = g.tiles_x * g.tiles_y;
    if (ntiles > evk_voxel3_max_tiles()) return EVK_EINVAL;
    const int planes = (flags & EVK_VOXEL_SPLIT_POLARITY) ? 2 * B : B;
    const size_t lds_acc = (size_t)planes * sizeof(acc_t) * g.pitch * g.th;  // odd row pitch
    if (lds_acc > 96 * 1024) return EVK_EINVAL;
    const bool share = flags & EVK_VOXEL2_SHARE_CU;
    const Part3 q = v3_geometry(n, ntiles, share);
    if ((int64_t)q.nsc * q.spad >= ((int64_t)1 << 32) - 64) return EVK_EINVAL;   // quad indices are 30 bits
    const V3Layout L = v3_layout(ntiles, n, planes, tile_w, tile_h, share);
    if (scratch_bytes < L.total) return EVK_ESCRATCH;
    if (!aligned16(scratch)) return EVK_EALIGN;
    char *sb = (char *)scratch;
    uint32_t *table = (uint32_t *)(sb + L.table), *bases = (uint32_t *)(sb + L.bases), *rec = (uint32_t *)(sb + L.rec);
    uint2 *wide = (uint2 *)(sb + L.wide);
    float *staging = (float *)(sb + L.staging);
    hipStream_t s = (hipStream_t)stream;
    const float bm1 = (float)(B - 1);
    const int tfe = (flags & EVK_VOXEL_T_FROM_EVENTS) ? 1 : 0;
    if (!(flags & EVK_VOXEL2_TILES_ONLY)) {
        const V3Geo geo = v3_geo(share);
#define X(T, E, S)                                                                                                          \
    if (geo.threads == T && geo.ept == E && geo.sched == S)                                                                  \
        launch_part3<T, E, S>(c, n, g, ntiles, q, t_first, t_last, bm1, tfe, rec, wide, table, bases, index, oob, host_report, seq, s);
        V3_GEOMETRIES(X)
#undef X
    }
    if (!(flags & EVK_VOXEL2_PARTITION_ONLY)) {
        const int items = bucket_max_items(n, ntiles);
        const int kf = flags & (EVK_VOXEL_OVERWRITE | EVK_VOXEL_SPLIT_POLARITY | EVK_VOXEL2_NO_XCD_ORDER);
        const bool sp = flags & EVK_VOXEL_SPLIT_POLARITY, fx = flags & EVK_VOXEL_DETERMINISTIC;
        if (sp && fx) launch_tiles3<true, true>(items, lds_acc, s, rec, wide, table, bases, index, g, q, B, kf, vox, staging);
        else if (sp) launch_tiles3<true, false>(items, lds_acc, s, rec, wide, table, bases, index, g, q, B, kf, vox, staging);
        else if (fx) launch_tiles3<false, true>(items, lds_acc, s, rec, wide, table, bases, index, g, q, B, kf, vox, staging);
        else launch_tiles3<false, false>(items, lds_acc, s, rec, wide, table, bases, index, g, q, B, kf, vox, staging);
    }
    return launch_status();
}

extern "C" int evk_voxel3_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, int h, int wd,
                              int tile_w, int tile_h, float t_first, float t_last, int B, int flags, float *vox,
                              uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report,
                              uint32_t seq, void *stream) {
    if (n > 0 && (!x || !y || !t || !p)) return EVK_EINVAL;
    if (!(aligned16(x) && aligned16(y) && aligned16(t) && aligned16(p))) return EVK_EALIGN;
    const SrcF32 c{x, y, t, p};
    return voxel3(c, n, h, wd, tile_w, tile_h, t_first, t_last, B, flags, vox, index, scratch, scratch_bytes, oob, host_report,
                  seq, stream);
}

extern "C" int evk_voxel3_native_f32(const int16_t *x, const int16_t *y, int xy_stride, const void *t, int t_kind,
                                     double t_offset, const void *p, int p_kind, int64_t n, int h, int wd, int tile_w,
                                     int tile_h, float t_first, float t_last, int B, int flags, float *vox,
                                     uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob,
                                     uint32_t *host_report, uint32_t seq, void *stream) {
    ColsNative v;
    const int rc = native_cols(v, x, y, xy_stride, t, t_kind, t_offset, p, p_kind, n);
    if (rc != EVK_OK) return rc;
    if (!(aligned16(x) && (xy_stride == 2 || aligned16(y)) && aligned16(t) && aligned16(p))) return EVK_EALIGN;
    if (t_kind == EVK_T_F64) {
        const SrcNative<true> c{v.x, v.y, v.t, v.p, v.t_offset, v.xy_stride, v.p_kind};
        return voxel3(c, n, h, wd, tile_w, tile_h, t_first, t_last, B, flags, vox, index, scratch, scratch_bytes, oob,
                      host_report, seq, stream);
    }
    const SrcNative<false> c{v.x, v.y, v.t, v.p, v.t_offset, v.xy_stride, v.p_kind};
    return voxel3(c, n, h, wd, tile_w, tile_h, t_first, t_last, B, flags, vox, index, scratch, scratch_bytes, oob, host_report,
                  seq, stream);
}
#endif  // EVK_EXPERIMENTS
