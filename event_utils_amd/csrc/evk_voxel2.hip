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
#include <atomic>

#include <hip/hip_ext.h>

#include "evk_part2.h"
#include "evk_voxel_live.h"

namespace evk {

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
#ifndef V2_CHUNK_CAP
#define V2_CHUNK_CAP(WG) (64 * V2_MAX_CHUNKS(WG))  // per wave; 28 KB for 8 waves: with the padded accumulators (21 KB at VGA) three
                                           // workgroups still fit a CU's 160 KB
#endif
// FIXED (EVK_VOXEL_DETERMINISTIC): the cells are int64 multiples of 2^-32 instead of float64 -- integer adds commute, so the
// grid is bit-identical from run to run and for any order of the events.  |contribution| < 2^30 and finite, else it is
// counted in index[4] and left out (the wrapper raises).
#define V2_FIXED_ONE 4294967296.0
// A ROW BAND of the grid (round 4; evk_voxel2_band_f32): only the tiles [tile_lo, tile_hi) are accumulated, and written to a
// (planes, rows, dom_w) buffer of their own whose first row is image row y_lo -- contiguous, so that an event-sharded run can
// all-reduce band k while band k + 1 is still being accumulated.  tile_hi == 0: the whole grid, as ever.
struct Band {
    int tile_lo, tile_hi, y_lo, rows;
};
template <int WG, int U, bool SPLIT, bool FIXED, int REC>
__global__ void __launch_bounds__(WG, V2_TILES_WAVES(REC)) k_voxel_tiles2(const void *__restrict__ rec_, const void *__restrict__ side_,
                                                     const uint32_t *__restrict__ bases,
                                                     const uint32_t *__restrict__ table, uint32_t *__restrict__ index,
                                                     TileGridG g, Part2 q, int B, int flags, float *__restrict__ vox,
                                                     float *__restrict__ staging, Band band, uint32_t *live_status = nullptr,
                                                     uint32_t live_epoch = 0) {
    constexpr int NW = WG / 64, E = V2_ENT(REC);
    // REC 8: a lane takes 16 bytes = 2 records {t_norm, polarity | cell}; REC 4: 8 bytes = 2 one-word records (k_part_sorted),
    // decoded with the base of their sub-chunk, which travels with the chunk list
    typedef typename std::conditional<REC == 8, uint4, uint2>::type Pair;
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
    // (entry [CAP] of every list is a ZERO entry that is never overwritten: a lane group without a chunk reads it -- one v_min
    // and an unconditional LDS read instead of a compare, two zero moves and two exec-masked reads per list access)
    __shared__ uint2 cseg[NW][V2_CHUNK_CAP(WG) + 1];
    __shared__ uint32_t cbase[REC == 4 ? NW : 1][REC == 4 ? V2_CHUNK_CAP(WG) + 1 : 1];   // REC 4: t_norm base of the chunk's sub-chunk
    const int ntiles = g.tiles_x * g.tiles_y;
    const uint32_t *part_start = index + V2_PART, *item_tile = index + V2_ITEM(ntiles);
    V2_T0();
    const uint32_t nitems_all = part_start[ntiles];
    uint32_t item_lo = 0, nitems = nitems_all;
    if (band.tile_hi > 0) item_lo = part_start[band.tile_lo], nitems = part_start[band.tile_hi] - item_lo;
    if (blockIdx.x >= nitems) return;
    // XCD-aware work-item order: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md), and the segments of NEIGHBOURING
    // tiles are neighbours in every run (and their table entries share a cache line), so XCD k takes a contiguous
    // range of the tile-ordered work items: the 128-byte lines two adjacent tiles share are then fetched into ONE L2
    // once instead of into two L2s
    uint32_t item = blockIdx.x;
    // (not when tiles were cut: the pieces of a hot tile are neighbours in the item order, a contiguous range would hand most
    // of a blob to two or three XCDs -- blob scene 68 -> 63 us in plain order, uniform events 29 -> 34 us)
    if (!(flags & EVK_VOXEL2_NO_XCD_ORDER) && nitems_all == (uint32_t)ntiles) {
        const uint32_t k = blockIdx.x & 7u, j = blockIdx.x >> 3, q8 = nitems >> 3, r8 = nitems & 7u;
        item = k * q8 + (k < r8 ? k : r8) + j;
    }
    item += item_lo;
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
    if (nitems_all != (uint32_t)ntiles) {
        tile = (int)item_tile[item];
        first_item = part_start[tile], nparts = part_start[tile + 1] - first_item;
    }
    const uint32_t part_id = item - first_item;
    if (live_status) {
        // LIVE call (evk_voxel_live.h): the consumer kernel on the second stream has accumulated this tile while the
        // partition was sorting -- then there is nothing to do here -- unless it LEFT it (a polarity that is not a unit, a hot
        // tile, a round that did not arrive in time) or never came: then the tile is accumulated here, as ever
        __shared__ int live_mine;
        if (threadIdx.x == 0) live_mine = v2l_tile_is_mine(live_status + tile, live_epoch) ? 1 : 0;
        __syncthreads();
        if (!live_mine) return;
    }
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
    // UNIT-POLARITY COUNTING IN B PLANES (round 5, EVK_VOXEL2_COUNT2): the same integer sums with TWO int64 atomics per event --
    // p 2^31 - fx to bin b0 and fx to bin b0 + 1, fx = (int)(p f 2^31), i.e. S0[b0] 2^31 - G[b0] formed per event -- in the B
    // planes the float64 mode has: no extra LDS, so the tiles of a 1280x720 call (38x24 pixels: 37 KB, where the planes above
    // would need 64 KB and leave ONE workgroup per CU) count in integers as well: ds_add_u64 is the faster LDS atomic (4.1
    // against 2.9 lane-operations per clock and CU), no float64 multiply-adds, and the grid is bit-reproducible and equal,
    // bit for bit, to the other counting mode's.
    const bool unit2 = COUNTING && !unit && (flags & EVK_VOXEL2_COUNT2) && index[7] == 0u;
    int *const s0 = reinterpret_cast<int *>(acc + (B + 1) * ppix);   // unit mode: acc = G[-1 .. B-1], then S0[0 .. B-1]
    V2_U(0);
    unsigned long long *const gq = reinterpret_cast<unsigned long long *>(acc);   // unit mode: G as int64
    if (threadIdx.x < NW) {
        cseg[threadIdx.x][V2_CHUNK_CAP(WG)] = make_uint2(0u, 0u);
        if constexpr (REC == 4) cbase[threadIdx.x][V2_CHUNK_CAP(WG)] = 0u;
    }
    if (unit) {
        for (int i = threadIdx.x; i < (B + 1) * ppix; i += WG) acc[i] = 0.0;
        for (int i = threadIdx.x; i < B * ppix; i += WG) s0[i] = 0;
        if (threadIdx.x < (1 << V2_LB) / 32) poison[threadIdx.x] = 0u;
    } else {
        for (int i = threadIdx.x; i < NB * ppix; i += WG) acc[i] = 0.0;
        if ((FIXED || unit2) && threadIdx.x < (1 << V2_LB) / 32) poison[threadIdx.x] = 0u;
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
            if constexpr (FIXED) {   // no NaN among integers: one bit per cell, as in the counting mode (both grids of a split call)
                __hip_atomic_fetch_or(poison + (local >> 5), 1u << (local & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return;
            }
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
    // (B planes: bin 0 takes what G[-1] takes above, bin B - 1 what -G[B - 1] takes)
    auto unit2_general = [&](int local, float tn, float p) {
        if (tn != tn) {
            __hip_atomic_fetch_or(poison + (local >> 5), 1u << (local & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
        const float edge = tn < 0.0f ? 0.0f : bm1;
        const float val = p * fmaxf(0.0f, 1.0f - fabsf(tn - edge));
        if (val != 0.0f)
            __hip_atomic_fetch_add(gq + (tn < 0.0f ? 0 : (B - 1) * ppix) + local, (unsigned long long)__double2ll_rn((double)val * G_ONE),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto one = [&](auto unit_tag, uint32_t lo_w, uint32_t hi_w, uint32_t ridx) {
        constexpr int UNIT = decltype(unit_tag)::value;   // 0: float64 / fixed-point cells, 1: S0 + G planes, 2: B int64 planes
#ifndef V2_ENT_PREFETCH
#define V2_ENT_PREFETCH 0   // (A/B) measured: 92.3 against 88.6 us at 50 M events / 720p with the prefetch (six more live registers in the rounds)
#endif
#ifndef V2_FAST_UNIT
#define V2_FAST_UNIT 1   // (A/B)
#endif
        if constexpr (UNIT >= 1 && V2_FAST_UNIT && V2_ABLATE_B >= 4) {
            // THE HOT PATH of the counting modes (round 6): a record with polarity +1 or -1 and t_norm in [0, B - 1) -- all but a
            // handful of a call's records -- in ~24 vector instructions and ONE branch.  The SQ counters say this kernel is bound
            // by instruction issue (52 M wave-level VALU instructions for 50 M events, profiles/r03_voxel_sq_counters.txt: ~96 us
            // of a CU's four SIMDs), and the general code below spends ~38 VALU + ~10 scalar instructions and three taken
            // branches per record on cases that do not occur here: the counting modes run only when EVERY polarity of the call
            // is a unit (index[7]), so p is a sign -- p (2^31 - fi) and p fi (fi = (int)(f 2^31), f = t_norm - b0) are
            // conditional negations of 32-bit integers, not float products and 64-bit shifts and subtractions; a record is
            // never wide; and t_norm == B - 1 exactly (the stream's last time stamp) goes below with everything else that is rare
            // -- zero polarity, escaped 4-byte records, NaN or out-of-range times --, so bin b0 + 1 always exists and its
            // address is an addition.  Bit-identical to the general code: p f 2^31 is exact in float32 and truncation is odd.
            uint32_t tbits, sgn;   // sgn = 0 (p = +1) or 0xFFFFFFFF (p = -1)
            uint32_t bad;          // bit 31 set: not a unit polarity (or wide / escaped)
            int cell;
            if constexpr (REC == 8) {
                cell = (int)(hi_w & V2_LOCAL_MASK);
                tbits = lo_w;
                bad = (hi_w & (0x7FFFFFFFu & ~V2_LOCAL_MASK)) == 0x3F800000u ? 0u : 0x80000000u;   // |p| == 1 and not wide
                sgn = (uint32_t)((int32_t)hi_w >> 31);
            } else {
                cell = (int)(lo_w & V2_LOCAL_MASK);
                tbits = hi_w + (lo_w >> V2_DELTA_SHIFT);
                bad = lo_w << (30 - V2_CODE_SHIFT);                              // code 0 / 1 = +1 / -1 (2: zero, 3: escaped): its high bit
                sgn = (uint32_t)((int32_t)(lo_w << (31 - V2_CODE_SHIFT)) >> 31);
            }
            const float tf = __uint_as_float(tbits);
            // t_norm in [0, B - 1) as ONE unsigned compare of the bit patterns: non-negative floats order like their bits, and a
            // negative value, -0.0 or a NaN has a pattern above that of B - 1 (they all take the general path below)
            const bool ok = ((tbits | (bad & 0x80000000u)) < __float_as_uint(bm1));
            if (__builtin_expect(ok, 1)) {
                const int b0 = (int)tf;
                const uint32_t fi = (uint32_t)(int)((tf - (float)b0) * 2147483648.0f);
                const int off = __mul24(b0, ppix) + cell;
                const uint32_t lo2 = (fi ^ sgn) - sgn;                           // p fi
                const unsigned long long v2 = (unsigned long long)lo2 | ((unsigned long long)(uint32_t)((int32_t)lo2 >> 31) << 32);
                if constexpr (UNIT == 2) {
                    const uint32_t lo1 = ((0x80000000u - fi) ^ sgn) - sgn;      // p (2^31 - fi): magnitude in [1, 2^31], high word = sign
                    __hip_atomic_fetch_add(gq + off, (unsigned long long)lo1 | ((unsigned long long)sgn << 32), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(gq + off + ppix, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    __hip_atomic_fetch_add(gq + ppix + off, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(s0 + off, (int)(sgn | 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                return;
            }
        }
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
        if constexpr (UNIT == 2) {
            if (__builtin_expect(tr >= 0.0f && tr <= bm1, 1)) {
                const int b0 = (int)tn;
                unsigned long long *cell = gq + __mul24(b0, ppix) + local;
                const int fx = (int)((p * (tn - (float)b0)) * 2147483648.0f);
                if (V2_ABLATE_B < 3) {   // (timing builds) weights and addresses, no atomics
                    if (fx == 0x12345677 && cell == gq + 1) acc[0] = 1.0;
                    return;
                }
                __hip_atomic_fetch_add(cell, (unsigned long long)(((long long)(int)p << 31) - (long long)fx), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                if (V2_ABLATE_B < 4) return;   // (timing builds) one atomic per event
                // (b0 + 1 == B only for t_norm == B - 1, where fx is 0)
                __hip_atomic_fetch_add(cell + (b0 + 1 < B ? ppix : 0), (unsigned long long)(long long)fx, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                unit2_general(local, tn, p);
            }
            return;
        }
        if constexpr (UNIT == 1) {
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
    struct PairU {         // NOT packed: the compiler may assume the natural alignment it does not have (the hardware does not care)
        Pair v;
    };
#ifndef V2_ABLATE_LOADS
#define V2_ABLATE_LOADS 0   // (timing builds, results wrong) 1: every record load lands in the first 1 MB of the runs (L2-resident); 2: in its first 16 KB (L1)
#endif
    auto load_pair = [&](uint32_t pos) -> Pair {
        if (V2_ABLATE_LOADS == 1) pos &= 0x3FFFFu;
        if (V2_ABLATE_LOADS == 2) pos &= 0xFFFu;
        return reinterpret_cast<const PairU *>(static_cast<const Rec1 *>(rec_) + pos)->v;
    };
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
                uint32_t j = j0 + 16u * u + grp;
                j = j < total ? j : (uint32_t)V2_CHUNK_CAP(WG);   // (the zero entry)
                cs[u] = cseg[wave][j];
                if constexpr (REC == 4) cb_[u] = cbase[wave][j];
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
        // (round 6: the NEXT batch's entries are fetched ahead of the last pass's rounds -- 4096 uncoalesced 4-byte loads per tile,
        // ~15 % of the kernel's wave cycles at 50 M events / 720p when every batch began by waiting for them; entries are still
        // fetched again where a rare path needs them after the rounds)
        uint32_t ent[E], bb[E];
        fetch(sc_lo, ent, bb);
        __syncthreads();  // the accumulators are zero before the first adds
        for (int base = sc_lo; base < sc_hi; base += bsz) {
            uint32_t sum = 0, longs = 0, packed = 0;   // packed: chunks of entry e in bits [4e, 4e + 4)
            auto mych_of = [&](int e) { return (packed >> (4 * e)) & 15u; };
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t cnt = ent[e] >> 16;
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
                if (V2_ENT_PREFETCH && pass == npass - 1) fetch(base + bsz, ent, bb);   // (beyond sc_hi: zeros)
                rounds(unit_tag, total);
            }
            if (!V2_ENT_PREFETCH) fetch(base + bsz, ent, bb);
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
        if (unit) batches(std::integral_constant<int, 1>{});
        else if (unit2) batches(std::integral_constant<int, 2>{});
        else batches(std::integral_constant<int, 0>{});
    } else {
        batches(std::integral_constant<int, 0>{});
    }
    V2_U(7);
    __syncthreads();
    V2_U(8);
    const int64_t plane = (int64_t)(band.tile_hi > 0 ? band.rows : g.dom_h) * g.dom_w;
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
                float *o = vox + b * plane + (int64_t)(Y - band.y_lo) * g.dom_w + X;
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
        if (unit2) {
            if ((poison[l >> 5] >> (l & 31)) & 1u) return __uint_as_float(0x7FC00000u);
            return (float)((double)(long long)gq[b * ppix + l] * (1.0 / G_ONE));
        }
        const acc_t a = acc[b * ppix + l];
        if constexpr (FIXED) {
            if ((poison[l >> 5] >> (l & 31)) & 1u) return __uint_as_float(0x7FC00000u);
            return (float)((double)__builtin_bit_cast(long long, a) * (1.0 / V2_FIXED_ONE));
        }
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
    // The partial tile goes to the staging buffer with AGENT-SCOPE stores and is read back with agent-scope loads: such
    // accesses are performed at the level all XCDs share, complete (vmcnt) only when they are, and never hit a stale line of
    // this XCD's L2 -- so the hand-over to the last part needs no release / acquire FENCE, which on this chip is a write-back
    // (and an invalidate) of the whole L2 with everything the other workgroups have flushed into it (blob scene: tile kernel
    // 48.7 -> 46.3 us).  Every wave drains its own stores before the barrier; one lane then takes the ticket.
    //
    // INTEGER accumulators (the counting mode, EVK_VOXEL_DETERMINISTIC) hand over the EXACT int64 value of every cell and the
    // last part adds integers, with ONE rounding to float32 at the end (round 4): the grid then does not depend on where the
    // tiles were cut, i.e. it is bit-identical for any order of the events also on scenes with hot tiles.  (Round 3 staged
    // float32 partial tiles for every mode: a permuted stream moved events between the pieces and changed their roundings.)
    // LLONG_MIN marks a poisoned cell (NaN t_norm).
    const bool exactq = unit || unit2 || FIXED;
    const double qscale = (unit || unit2) ? 1.0 / G_ONE : 1.0 / V2_FIXED_ONE;
    auto lds_cell_q = [&](int c) -> long long {
        int b, row, col;
        split_cell(c, b, row, col);
        const int l = row * tpitch + col;
        if (unit) {
            if ((poison[l >> 5] >> (l & 31)) & 1u) return (long long)0x8000000000000000ull;
            return ((long long)s0[b * ppix + l] << 31) - (long long)gq[(b + 1) * ppix + l] + (long long)gq[b * ppix + l];
        }
        if ((FIXED || unit2) && ((poison[l >> 5] >> (l & 31)) & 1u)) return (long long)0x8000000000000000ull;
        return __builtin_bit_cast(long long, acc[b * ppix + l]);
    };
    float *mine = staging + 2 * (int64_t)item * stride;
    if (exactq) {
        unsigned long long *mq = reinterpret_cast<unsigned long long *>(mine);
        for (int c = threadIdx.x; c < cells; c += WG)
            __hip_atomic_store(mq + c, (unsigned long long)lds_cell_q(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        for (int c = threadIdx.x; c < cells; c += WG)
            __hip_atomic_store(mine + c, lds_cell(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    EVK_HANDOVER_DRAIN();
    __shared__ int is_last;
    if (threadIdx.x == 0) {
        uint32_t *counter = index + V2_COUNTER(ntiles) + tile;
        const uint32_t prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == nparts - 1);
        if (is_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!is_last) return;
    EVK_HANDOVER_ACQUIRE();
    const float *parts = staging + 2 * (int64_t)first_item * stride;
    if (exactq) {
        const unsigned long long *pq = reinterpret_cast<const unsigned long long *>(parts);
        flush([&](int c) {
            long long sum = 0;
            bool nan = false;
            for (uint32_t p = 0; p < nparts; ++p) {
                const long long v = (long long)__hip_atomic_load(pq + (int64_t)p * stride + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                nan |= v == (long long)0x8000000000000000ull;
                sum += v;
            }
            return nan ? __uint_as_float(0x7FC00000u) : (float)((double)sum * qscale);
        });
        return;
    }
    flush([&](int c) {
        float sum = 0.0f;
        for (uint32_t p = 0; p < nparts; ++p)
            sum += __hip_atomic_load(parts + 2 * (int64_t)p * stride + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return sum;
    });
}

}  // namespace evk

using namespace evk;

extern "C" int64_t evk_voxel2_index_len(int ntiles, int64_t n) {
    if (ntiles <= 0 || ntiles > V2_MAX_TILES || n < 0) return 0;
    return (int64_t)V2_ITEM(ntiles) + v2_max_items(n, ntiles);   // (header, totals, the live words, then the plan)
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
    // the partition kernel's LDS with its largest sorted buffer (12 K events x 8 bytes, or the bilinear image format's
    // 8 K x 12): the tile counters must fit beside it
    if (v2_part_lds(1024, 12, 8, g.tiles_x * g.tiles_y) > (size_t)V2_LDS_LIMIT) return 0;
    return g.tiles_x * g.tiles_y;
}

// LDS of a tile workgroup with `planes` float64 accumulator planes: the accumulators + the chunk lists (512 threads, either
// record size); the limit voxel2() enforces
static size_t v2_tiles_lds(const TileGridG &g, int planes) {
    return (size_t)planes * sizeof(acc_t) * g.pitch * g.th + 12 * 8 * (V2_CHUNK_CAP(512) + 1) + 64;
}
#define V2_TILES_LDS_LIMIT (150 * 1024)

// 1 when evk_voxel2_f32 takes this tiling with `planes` accumulator planes (B, or 2 B with EVK_VOXEL_SPLIT_POLARITY): the
// grid is valid, the partition's and the tile kernel's LDS fit, the tile count is within evk_voxel2_max_tiles().  What a
// caller's tile search asks instead of restating the kernels' constants.
extern "C" int evk_voxel2_fits(int h, int wd, int tile_w, int tile_h, int planes) {
    TileGridG g;
    if (planes <= 0 || make_grid_g(g, h, wd, tile_w, tile_h) != EVK_OK) return 0;
    const int ntiles = evk_voxel2_num_tiles(h, wd, tile_w, tile_h);
    if (ntiles <= 0 || ntiles > evk_voxel2_max_tiles()) return 0;
    return v2_tiles_lds(g, planes) <= (size_t)V2_TILES_LDS_LIMIT ? 1 : 0;
}
extern "C" int evk_num_cu(void) { return EVK_NUM_CU; }

// largest tile count the partition kernel's LDS holds (sorted records + one uint32 per tile)
extern "C" int evk_voxel2_max_tiles(void) {
    const int64_t budget = (int64_t)V2_LDS_LIMIT - (int64_t)1024 * 12 * 8 - 1024;   // the larger shipped geometry
    const int64_t t = budget / 12;   // a counter, a cursor and a total per tile
    return (int)(t < V2_MAX_TILES ? (t > 0 ? t : 0) : V2_MAX_TILES);
}



// Tile kernel: 768 or 512 threads (chosen in voxel2() below), 2 chunk loads per lane in flight (-DV2_U8 / -DV2_U4 for
// measurements: 1, 3 and 4 are level or slower).  `lds_dyn` = the accumulators of the mode the launch
// may run in (the chunk lists are static).
template <int WG, int U, bool SPLIT, bool FIXED, int REC>
static void launch_tiles(int items, size_t lds_dyn, hipStream_t s, const void *rec, const void *pw, const uint32_t *bases,
                         const uint32_t *table, uint32_t *index, const TileGridG &g, const Part2 &q, int B, int kf, float *vox,
                         float *staging, const Band &band, uint32_t *live_status = nullptr, uint32_t live_epoch = 0) {
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [] {
        (void)hipFuncSetAttribute((const void *)k_voxel_tiles2<WG, U, SPLIT, FIXED, REC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024 - (REC == 4 ? 12 : 8) * (WG / 64) * (V2_CHUNK_CAP(WG) + 1) - 256);
    });
    k_voxel_tiles2<WG, U, SPLIT, FIXED, REC><<<items, WG, lds_dyn, s>>>(rec, pw, bases, table, index, g, q, B, kf, vox, staging, band,
                                                                        live_status, live_epoch);
}

// ---- LIVE calls (evk_voxel_live.h): the second stream, the epoch counter, what a call must look like --------------------
#ifndef V2L_U
#define V2L_U 2   // chunk loads per lane in flight in a stage of the consumer's pipeline (two stages: 4 x 16 bytes per lane)
#endif
static hipStream_t v2_live_stream() {
    static std::once_flag once[64];
    static hipStream_t side[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [dev] {
        side[dev & 63] = nullptr;
        if (hipStreamCreateWithFlags(&side[dev & 63], hipStreamNonBlocking) != hipSuccess) side[dev & 63] = nullptr;
        (void)hipFuncSetAttribute((const void *)k_voxel_live<V2L_U>, hipFuncAttributeMaxDynamicSharedMemorySize, V2L_LDS_REQUEST);
    });
    return side[dev & 63];
}
// the event the caller's stream waits on for the consumer kernel's end (per device; re-recorded by every live call: a wait
// already enqueued keeps the record it was issued against)
static hipEvent_t v2_live_event() {
    static std::once_flag once[64];
    static hipEvent_t ev[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [dev] {
        ev[dev & 63] = nullptr;
        if (hipEventCreateWithFlags(&ev[dev & 63], hipEventDisableTiming) != hipSuccess) ev[dev & 63] = nullptr;
    });
    return ev[dev & 63];
}
static uint32_t v2_live_epoch() {
    static std::atomic<uint32_t> counter{0};
    uint32_t e;
    do e = (counter.fetch_add(1, std::memory_order_relaxed) + 1u) & V2L_EPOCH_MASK; while (e == 0u);
    return e;
}
// bytes of static LDS of k_voxel_live (chunk lists, poison bits, a few words)
#define V2L_STATIC_LDS (V2L_NW * V2L_CAP * 8 + 2 * ((1 << V2_LB) / 32) * 4 + 64)
// can this call run live?  (the geometry the consumer kernel is written for: <= 2 x 256 tiles, 8-byte records in the 8 K-event
// partition geometry, <= 255 runs per partition workgroup, the accumulators of two tiles beside the chunk lists, and the
// partition's and the consumer's LDS together on one CU)
static bool v2_live_fits(const TileGridG &g, int ntiles, const Part2 &q, int64_t n, int B, int recb, const V2Config &cfg) {
    const size_t acc = (size_t)2 * B * g.pitch * g.th * 8;
    return recb == 8 && cfg.threads == 1024 && cfg.ept == 8 && ntiles <= 2 * EVK_NUM_CU && q.nblk <= V2L_PROGRESS_WORDS &&
           q.per_block >= 2 && q.per_block <= 255 && n < ((int64_t)1 << 28) && (int64_t)q.nsc * q.S * 8 < ((int64_t)1 << 32) &&
           acc + V2L_STATIC_LDS <= (size_t)V2L_LDS_REQUEST &&
           v2_part_lds(1024, 8, 8, ntiles, true) + 256 + V2L_LDS_REQUEST <= (size_t)160 * 1024;
}

template <typename C>
static int voxel2(const C &c, int64_t n, int h, int wd, int tile_w, int tile_h, float t_first, float t_last, int B,
                  int flags, float *vox, uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob,
                  uint32_t *host_report, uint32_t seq, void *stream, const Band &band = Band{0, 0, 0, 0}) {
    TileGridG g;
    const int known = EVK_VOXEL_OVERWRITE | EVK_VOXEL_SPLIT_POLARITY | EVK_VOXEL_T_FROM_EVENTS | EVK_VOXEL2_PARTITION_ONLY |
                      EVK_VOXEL2_TILES_ONLY | EVK_VOXEL2_NO_XCD_ORDER | EVK_VOXEL2_SHARE_CU | EVK_VOXEL_DETERMINISTIC |
                      EVK_VOXEL2_REC4 | EVK_VOXEL2_REC8 | EVK_VOXEL2_NO_COUNT | EVK_VOXEL2_WG512 | EVK_VOXEL2_LIVE | EVK_VOXEL2_NO_COUNT2;
    if (make_grid_g(g, h, wd, tile_w, tile_h) != EVK_OK || B <= 0 || !vox || !index || !scratch || n <= 0 ||
        n > (int64_t)4000000000LL || (flags & ~known))
        return EVK_EINVAL;
    if (host_report && ((uintptr_t)host_report & 7u)) return EVK_EALIGN;   // written with one 8-byte store
    const int ntiles = g.tiles_x * g.tiles_y;
    if (ntiles > evk_voxel2_max_tiles() || !evk_voxel2_num_tiles(h, wd, tile_w, tile_h)) return EVK_EINVAL;
    const int planes = (flags & EVK_VOXEL_SPLIT_POLARITY) ? 2 * B : B;
    const size_t lds_acc = (size_t)planes * sizeof(acc_t) * g.pitch * g.th;  // odd row pitch
    if (v2_tiles_lds(g, planes) > (size_t)V2_TILES_LDS_LIMIT) return EVK_EINVAL;
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
    const int recb = v2_rec_bytes(n, flags);
    // LIVE (evk_voxel_live.h): partition on this stream, the consumer kernel on the library's second stream, then the tile
    // kernel proper here for whatever the consumers leave.  A request, not a demand: calls the consumer kernel is not written
    // for (other geometries, split polarities, shared CUs, single stages, a stream that is being captured) run as ever.
    bool live = false;
    hipStream_t s2 = nullptr;
    hipEvent_t live_done = nullptr;
    uint32_t epoch = 0;
    if constexpr (std::is_same<C, SrcF32>::value) {
        if ((flags & EVK_VOXEL2_LIVE) && band.tile_hi == 0 &&
            !(flags & (EVK_VOXEL2_PARTITION_ONLY | EVK_VOXEL2_TILES_ONLY | EVK_VOXEL_SPLIT_POLARITY | EVK_VOXEL2_SHARE_CU |
                       EVK_VOXEL2_NO_COUNT)) &&
            v2_live_fits(g, ntiles, q, n, B, recb, cfg)) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone && (s2 = v2_live_stream()) != nullptr)
                live = true, epoch = v2_live_epoch();
            else
                (void)hipGetLastError();
        }
    }
    uint32_t *const live_progress = index + V2_LIVE_PROGRESS, *const live_status = index + V2_LIVE_STATUS;
    if (live) {
        if constexpr (std::is_same<C, SrcF32>::value) {
            launch_part<1024, 8, 8, C, true>(c, n, g, ntiles, q, t_first, t_last, bm1, tfe, rec, pw, bases, table, index, oob,
                                             host_report, seq, s, live_progress, epoch);
            LiveArgs la{live_progress, live_status, epoch, (uint32_t)v2_cap(n, ntiles), 400u};
            k_voxel_live<V2L_U><<<(ntiles + 1) / 2, V2L_WG, V2L_LDS_REQUEST - V2L_STATIC_LDS, s2>>>(
                rec, (uint32_t)((int64_t)q.nsc * q.S * 8), table, g, q, B, flags & EVK_VOXEL_OVERWRITE, vox, la);
            live_done = v2_live_event();
            if (!live_done || hipEventRecord(live_done, s2) != hipSuccess) live_done = nullptr, (void)hipGetLastError();
        }
    } else if (!(flags & EVK_VOXEL2_TILES_ONLY)) {
        // (8-byte records in the 8 K-event geometry of a call that has its CUs to itself: the exact polarities are staged in
        // LDS and wide ones leave as a dense run -- V2_FMT_VOX8W, evk_part2.h)
#ifndef V2_USE_VOX8W
#define V2_USE_VOX8W 1   // (A/B)
#endif
        if (V2_USE_VOX8W && recb == 8 && cfg.threads == 1024 && cfg.ept == 8 && !share) {
            launch_part<1024, 8, V2_FMT_VOX8W>(c, n, g, ntiles, q, t_first, t_last, bm1, tfe, rec, pw, bases, table, index, oob,
                                               host_report, seq, s);
        } else {
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
            return 2 * (acc_bytes + (size_t)(rec == 4 ? 12 : 8) * (wg / 64) * (V2_CHUNK_CAP(wg) + 1) + 256) <= (size_t)160 * 1024;
        };
        const bool may_count = !sp && !(flags & EVK_VOXEL2_NO_COUNT) && n < ((int64_t)1 << 31);   // (int32 counts)
#ifndef V2_U4
#define V2_U4 2   // chunk loads per lane in flight, 4-byte records
#endif
#ifndef V2_U8
#define V2_U8 2   // ... 8-byte records
#endif
#define V2_UU(R) ((R) == 4 ? V2_U4 : V2_U8)
#define V2_TILES(W, R)                                                                                                     \
    do {                                                                                                                   \
        if (sp && fx) launch_tiles<W, V2_UU(R), true, true, R>(items, lds_dyn, s, rec, pw, bases, table, index, g, q, B, kf, vox, staging, band, lst, epoch);   \
        else if (sp) launch_tiles<W, V2_UU(R), true, false, R>(items, lds_dyn, s, rec, pw, bases, table, index, g, q, B, kf, vox, staging, band, lst, epoch);   \
        else if (fx) launch_tiles<W, V2_UU(R), false, true, R>(items, lds_dyn, s, rec, pw, bases, table, index, g, q, B, kf, vox, staging, band, lst, epoch);   \
        else launch_tiles<W, V2_UU(R), false, false, R>(items, lds_dyn, s, rec, pw, bases, table, index, g, q, B, kf, vox, staging, band, lst, epoch);          \
    } while (0)
        uint32_t *const lst = live ? live_status : nullptr;
        // Threads per tile workgroup.  8-byte records (cache-resident calls): 768 -- twelve waves per tile -- while TWO such
        // workgroups fit a CU's LDS (accumulators + 12 chunk lists <= 80 KB: VGA at 5 bins; not split polarities or 720p
        // tiles).  Uniform events do not care (29.5 us either way, 512 tiles in one generation); the pieces of a hot tile are
        // done sooner: blob scene 57 -> 49 us, i.e. 1.25 x the uniform call end to end.  4-byte records (HBM-resident calls, three
        // table entries per lane, 128 registers): 512 (768: 151-158 against 132-142 us at 50 M events).
        // (a call that shares its CUs with a collective's workgroups keeps the 512-thread workgroups: two of them with the
        // counting mode's accumulators leave ~19 KB of every CU's LDS free, two 768-thread ones 3 KB)
        const bool may_wide = recb == 8 && !(flags & EVK_VOXEL2_WG512) && !share;
        int wg = 512;
        bool count = false;
        if (may_wide && may_count && two_fit(lds_count, 768, recb)) wg = 768, count = true;
        else if (may_count && two_fit(lds_count, 512, recb)) count = true;
        else if (may_wide && two_fit(lds_acc, 768, recb)) wg = 768;
        const size_t lds_dyn = count ? lds_count : lds_acc;
        if (count) kf |= EVK_VOXEL2_COUNT;
        else if (may_count && !(flags & EVK_VOXEL2_NO_COUNT2)) kf |= EVK_VOXEL2_COUNT2;   // the same integers in the float64 mode's B planes
        const bool wide_wg = wg == 768;
#ifndef V2_WG4
#define V2_WG4 512   // (A/B) threads of a tile workgroup with 4-byte records
#endif
        if (recb == 4) V2_TILES(V2_WG4, 4);
        else if (wide_wg) V2_TILES(768, 8);
        else V2_TILES(512, 8);
#undef V2_TILES
    }
    // The consumer kernel STARTED unordered (that is the overlap), but it must not outlive the call: a consumer that left its
    // tiles or lost a take-over keeps polling `progress`, loading `table` / `rec` and finally CASes a word of `status` -- in
    // buffers whose lifetime evk.h ties to `stream`, not to the library's side stream.  So the caller's stream waits for the
    // consumer's end behind the tile kernel (bounded: every wait inside the consumer is, LiveArgs.wait_us): whatever follows
    // on `stream` -- the next call's partition rewriting the records, a free of the scratch -- is ordered behind it.
    if (live) {
        if (!live_done) (void)hipStreamSynchronize(s2);   // (no event: the slow, safe form)
        else if (hipStreamWaitEvent(s, live_done, 0) != hipSuccess) (void)hipGetLastError(), (void)hipStreamSynchronize(s2);
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
    if (!(column_ok(x, flags) && column_ok(y, flags) && column_ok(t, flags) && column_ok(p, flags))) return EVK_EALIGN;
    const SrcF32 c{x, y, t, p};
    return voxel2(c, n, h, wd, tile_w, tile_h, t_first, t_last, B, flags & ~EVK_COLUMNS_UNALIGNED, vox, index, scratch, scratch_bytes,
                  oob, host_report, seq, stream);
}

extern "C" int evk_voxel2_band_f32(int64_t n, int h, int wd, int tile_w, int tile_h, int B, int flags, int tile_row_lo,
                                   int tile_row_hi, float *band, uint32_t *index, void *scratch, int64_t scratch_bytes,
                                   void *stream) {
    TileGridG g;
    if (make_grid_g(g, h, wd, tile_w, tile_h) != EVK_OK || tile_row_lo < 0 || tile_row_hi <= tile_row_lo ||
        tile_row_hi > g.tiles_y || !band)
        return EVK_EINVAL;
    const int y_lo = tile_row_lo * tile_h, y_hi = tile_row_hi * tile_h < h ? tile_row_hi * tile_h : h;
    const Band b{tile_row_lo * g.tiles_x, tile_row_hi * g.tiles_x, y_lo, y_hi - y_lo};
    const SrcF32 none{nullptr, nullptr, nullptr, nullptr};
    // the tile kernel alone (the records of the preceding EVK_VOXEL2_PARTITION_ONLY call are in `scratch`), over the band's
    // tiles, always overwriting: every cell of the band buffer is written
    return voxel2(none, n, h, wd, tile_w, tile_h, 0.0f, 0.0f, B,
                  (flags & ~(EVK_VOXEL2_PARTITION_ONLY | EVK_VOXEL_T_FROM_EVENTS)) | EVK_VOXEL2_TILES_ONLY | EVK_VOXEL_OVERWRITE, band,
                  index, scratch, scratch_bytes, nullptr, nullptr, 0, stream, b);
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
