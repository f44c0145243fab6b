// The LIVE voxel tile kernel (round 5): the tiles are accumulated WHILE the partition kernel is still sorting.
//
// The one-pass voxel call was two kernels back to back -- k_part_sorted (48 us at 10 M events from HBM: bound by the 240 MB it
// moves) and k_voxel_tiles2 (23 us: VALU / LDS-atomic work on 80 MB of records, half of it fixed cost of a launch whose 512
// workgroups all start and end together).  The two use different resources of a CU, and the partition leaves room beside its
// one 1024-thread workgroup per CU (80 registers x 4 waves per SIMD, 74 KB of LDS).  So the tile work moves INTO the
// partition's run time:
//   * k_voxel_live runs on a SECOND stream (its own hardware queue: kernels of one HIP stream never overlap -- the
//     any-order launch flag is refused on gfx9, tools/anyorder_probe.hip -- while two streams pair their workgroups up one
//     of each kind on all 256 CUs, same probe), launched right behind the partition, with no event between the streams;
//   * a workgroup (512 threads, <= 96 registers) owns TWO neighbouring tiles for the whole call -- B planes of int64 cells
//     each, 49 KB at 640x480 with 5 bins -- and asks for > 80 KB of LDS, so that no CU ever holds two of them and every CU has
//     room for its partition workgroup (74 + 82 KB);
//   * the partition (k_part_sorted<..., LIVE>) writes every run and its table row through (sc1) and, once all its waves have
//     drained those stores, publishes {epoch, runs out} per workgroup; the consumers poll the 256 progress words (one wave,
//     one 16-byte agent-scope load per lane, s_sleep in between) and consume ROUND r -- run r of every partition workgroup
//     -- as soon as it is out: their loads (agent scope, like every hand-over in this library) and LDS atomics overlap the
//     partition's streaming of round r + 1 ... r + 2;
//   * accumulation is the counting mode's integer arithmetic in two int64 atomics per event (p 2^31 - fx to bin b0, fx to bin
//     b0 + 1, fx = (int)(p f 2^31)): the SAME sums as k_voxel_tiles2's unit mode (S0[b] 2^31 - G[b] + G[b-1]), so the grid is
//     bit-identical to the two-launch path's and does not depend on the order of the events -- at 60 % of the LDS.
// What the consumers cannot take -- a polarity that is not +1 / -1 / +0, a tile that turns out hot (a segment beyond 56
// records, or 2.5 x the mean tile population: the plan will cut it) -- they LEAVE: the tile's status word says so and the tile
// kernel proper, launched behind the partition on the caller's stream as ever, accumulates exactly those tiles with its
// general code (plan, cut tiles, float64 or fixed-point cells) and returns at once for the others.
//
// Two streams and no event between them means nothing orders the consumer kernel against the caller's stream, so every
// hand-over is a word in memory and every wait is bounded:
//   progress[b] = epoch << 8 | runs out         one writer (partition workgroup b); a stale word has another epoch
//   status[t]   = epoch << 2 | state            pending (an OLDER epoch) -> FLUSHING -> DONE      (consumer, by CAS)
//                                               pending -> LEFT                                   (consumer: not mine)
//                                               pending -> LEFT                                   (tile kernel: timeout)
//   * a consumer that waits longer than `wait_us` for a round leaves its tiles (the partition may be queued behind minutes of
//     other work on the caller's stream: the consumers must not sit on a third of every CU until then);
//   * the tile kernel waits for a pending tile at most V2L_TAKEOVER_US, then claims it itself; a consumer flushes only after
//     winning the CAS pending -> FLUSHING, so a tile is written by exactly one of the two whatever the timing;
//   * the consumer's grid stores are write-through and drained before DONE: when the tile kernel has seen DONE (and returns),
//     the grid is in memory for whatever follows on the caller's stream.
// Epochs grow by one per live call (process-wide); "older" is decided modulo 2^30, so a consumer that starts absurdly late
// -- after the NEXT call has begun -- finds newer epochs everywhere and touches nothing.
#pragma once
#include "evk_part2.h"

namespace evk {

#define V2L_WG 512
#define V2L_NW (V2L_WG / 64)
#define V2L_MAXCH 7                    // chunks (of 8 records) of a listed segment; a longer segment makes its tile hot
#define V2L_CAP (64 * V2L_MAXCH)       // chunk list entries per wave: every lane's segment fits, no passes
#define V2L_LDS_REQUEST (82 * 1024)    // > 80 KB: two consumers never share a CU; + the partition's 74 KB <= 160 KB
#define V2L_TAKEOVER_US 150            // tile kernel: how long a pending tile is waited for before it is taken over
#define V2L_EPOCH_MASK 0x3FFFFFFFu
#define V2L_FLUSHING 1u
#define V2L_DONE 2u
#define V2L_LEFT 3u
#define V2L_PROGRESS_WORDS 256         // index[V2_LIVE_PROGRESS ...]: one word per partition workgroup (<= 256 of them)

// the state of tile word v in epoch e: 0 = pending (v belongs to an older call), else V2L_*; 4 = v belongs to a NEWER call
__device__ __forceinline__ uint32_t v2l_state(uint32_t v, uint32_t e) {
    const uint32_t d = (e - (v >> 2)) & V2L_EPOCH_MASK;     // how far v's epoch lies behind e
    return d == 0u ? (v & 3u) : (d < (1u << 29) ? 0u : 4u);
}

// Tile kernel side (k_voxel_tiles2, first thing a workgroup does in a live call): does this tile still need me?
// true = accumulate it (the consumer left it, or never came); false = the consumer has written it.
__device__ __forceinline__ bool v2l_tile_is_mine(uint32_t *status_word, uint32_t epoch) {
    const unsigned long long t0 = wall_clock64();   // 100 MHz
    for (;;) {
        const uint32_t v = __hip_atomic_load(status_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t st = v2l_state(v, epoch);
        if (st == V2L_DONE) return false;
        if (st == V2L_LEFT || st == 4u) return true;
        if (st == 0u && wall_clock64() - t0 > (unsigned long long)V2L_TAKEOVER_US * 100ull) {
            uint32_t expect = v;
            if (__hip_atomic_compare_exchange_strong(status_word, &expect, (epoch << 2) | V2L_LEFT, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT))
                return true;
            continue;
        }
        __builtin_amdgcn_s_sleep(8);   // pending or FLUSHING: the consumer is about to finish
    }
}

struct LiveArgs {
    uint32_t *progress, *status;
    uint32_t epoch;
    uint32_t hot_total;    // a tile with more events than this is left (the plan cuts it: v2_cap)
    uint32_t wait_us;      // bound on the wait for one round
};

template <int U>
__global__ void __launch_bounds__(V2L_WG, 5) k_voxel_live(const void *__restrict__ rec_, uint32_t rec_bytes, const uint32_t *__restrict__ table,
                                                          TileGridG g, Part2 q, int B, int flags, float *__restrict__ vox, LiveArgs a) {
    constexpr int NW = V2L_NW;
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned long long accq[];   // [2 tiles][B][ppix] int64, multiples of 2^-31
    __shared__ uint2 cseg[NW][V2L_CAP];        // {first record of the chunk, end of its segment | tile (bit 31)}
    __shared__ uint32_t poison[2][(1 << V2_LB) / 32];
    __shared__ uint32_t sh_tot[2], sh_v0[2], sh_leave, sh_mine[2], sh_go;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 3, grp = lane >> 2;
    const int ntiles = g.tiles_x * g.tiles_y;
    const int tile0 = 2 * (int)blockIdx.x;
    const int ntl = ntiles - tile0 < 2 ? ntiles - tile0 : 2;
    const int ppix = g.pitch * g.th, tcells = B * ppix;
    const float bm1 = (float)(B - 1);
    const uint32_t epoch = a.epoch & V2L_EPOCH_MASK;
    for (int i = tid; i < 2 * tcells; i += V2L_WG) accq[i] = 0ull;
    if (tid < 2 * (1 << V2_LB) / 32) (&poison[0][0])[tid] = 0u;
    if (tid < 2) {
        uint32_t v = 0, mine = 0;
        if (tid < ntl) {
            v = __hip_atomic_load(a.status + tile0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mine = v2l_state(v, epoch) == 0u ? 1u : 0u;   // anything else: taken over already, or this call is long over
        }
        sh_v0[tid] = v, sh_mine[tid] = mine, sh_tot[tid] = 0u;
    }
    if (tid == 0) sh_leave = 0u;
    __syncthreads();
    // leave bits: 1, 2 = tile 0 / 1 (hot, or a polarity that is not a unit); 4 = everything (a round did not arrive in time)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(rec_), 0, (int)rec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)a.progress, 0, V2L_PROGRESS_WORDS * 4, 0x00020000);
    auto load_pair = [&](uint32_t pos) -> u4v {   // records pos, pos + 1 (any 8-byte boundary), agent scope (sc1)
        return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(pos * 8u), 0, 16);
    };
    auto add64 = [&](unsigned long long *p, long long v) {
        __hip_atomic_fetch_add(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto one = [&](uint32_t lo_w, uint32_t hi_w, uint32_t tsel) {
#ifdef V2L_ABLATE   // (timing builds, results wrong) 1: record loads only; 2: + decode and weights, one int32 atomic per event
        if (V2L_ABLATE == 1) {
            if ((lo_w ^ hi_w) == 0x12345u) sh_leave = 8u;
            return;
        }
#endif
        const int local = (int)(hi_w & V2_LOCAL_MASK);
        const uint32_t pb = hi_w & V2_P_MASK;
        // +1.0, -1.0, +0.0 carried by the record itself; everything else (wide, other values, -0.0) is not for this kernel.
        // A cell outside the tile's plane cannot come from this call's partition: the word is not a record of this call (a
        // straggling consumer reading a buffer the NEXT call is rewriting) -- never form an LDS address from it
        const bool unit = !(hi_w & V2_WIDE) & (((pb & 0x7FFFFFFFu) == 0x3F800000u) | (pb == 0u)) & (local < ppix);
        if (__builtin_expect(!unit, 0)) {
            __hip_atomic_fetch_or(&sh_leave, 1u << tsel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
        const float p = __uint_as_float(pb), tn = __uint_as_float(lo_w);
        unsigned long long *cell = accq + __mul24((int)tsel, tcells) + local;
        if (__builtin_expect(tn >= 0.0f && tn <= bm1, 1)) {
            // bins b0 and b0 + 1 with p (1 - f) and p f, f = t_norm - b0, as integers in 2^-31 steps -- k_voxel_tiles2's
            // counting mode, with S0[b0] 2^31 - G[b0] formed per event (same integer sums; b0 + 1 == B only for f == 0)
            const int b0 = (int)tn;
            const int fx = (int)((p * (tn - (float)b0)) * 2147483648.0f);
            cell += __mul24(b0, ppix);
            add64(cell, ((long long)(int)p << 31) - (long long)fx);
            add64(cell + (b0 + 1 < B ? ppix : 0), (long long)fx);
        } else if (tn != tn) {   // dt == 0 (Q9): NaN in every bin of the pixel
            __hip_atomic_fetch_or(&poison[tsel][local >> 5], 1u << (local & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {                 // outside [ts[0], ts[-1]] (unsorted streams): an edge bin only
            const float edge = tn < 0.0f ? 0.0f : bm1;
            const float val = p * fmaxf(0.0f, 1.0f - fabsf(tn - edge));
            if (val != 0.0f) add64(cell + (tn < 0.0f ? 0 : (B - 1) * ppix), __double2ll_rn((double)val * 2147483648.0));
        }
    };
    auto pair = [&](const u4v &v, uint32_t pos, uint32_t endw) {
        const uint32_t end = endw & 0x7FFFFFFFu, tsel = endw >> 31;
        one(v.x, v.y, tsel);
        if (pos + 1 < end) one(v.z, v.w, tsel);
    };
    // chunk rounds over a wave's list, three stages in flight (list entries | record loads | accumulation): k_voxel_tiles2's
    auto rounds = [&](const uint32_t total) {
        auto meta = [&](uint32_t j0, uint2(&cs)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t j = j0 + 16u * u + grp;
                cs[u] = j < total ? cseg[wave][j] : make_uint2(0u, 0u);
            }
        };
        auto fire = [&](const uint2(&cs)[U], u4v(&v)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pos = cs[u].x + 2u * sub;
                v[u] = load_pair(pos < (cs[u].y & 0x7FFFFFFFu) ? pos : 2u * sub);
            }
        };
        auto eat = [&](const uint2(&cs)[U], const u4v(&v)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pos = cs[u].x + 2u * sub;
                if (pos < (cs[u].y & 0x7FFFFFFFu)) pair(v[u], pos, cs[u].y);
            }
        };
        constexpr uint32_t step = 16u * U;
        uint2 ca[U], cb[U], cn[U];
        u4v va[U], vb[U];
        meta(0u, ca);
        fire(ca, va);
        meta(step, cb);
        for (uint32_t j0 = 0; j0 < total; j0 += 2u * step) {
            fire(cb, vb);
            meta(j0 + 2u * step, cn);
            eat(ca, va);
            fire(cn, va);
#pragma unroll
            for (int u = 0; u < U; ++u) ca[u] = cn[u];
            meta(j0 + 3u * step, cn);
            eat(cb, vb);
#pragma unroll
            for (int u = 0; u < U; ++u) cb[u] = cn[u];
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
    const bool any_mine = (sh_mine[0] | sh_mine[1]) != 0u;
    const int pb_ = q.per_block;
    // entry of this thread in every round: partition workgroup b's run, tile tile0 + j
    const int eb = tid >> 1, ej = tid & 1;
    for (int r = 0; any_mine && r < pb_; ++r) {
        // partition workgroups that have a run number r: b * per_block + r < nsc
        int cnt = q.nsc > r ? (q.nsc - r + pb_ - 1) / pb_ : 0;
        cnt = cnt < q.nblk ? cnt : q.nblk;
        if (cnt <= 0) break;
        if (wave == 0) {
            // all of them have published run r?  progress[b] = epoch << 8 | runs out: 4 words per lane, one 16-byte load
            const unsigned long long t0 = wall_clock64();
            const uint32_t want = (epoch & 0xFFFFFFu) << 8;
            for (;;) {
                const u4v w = __builtin_amdgcn_raw_buffer_load_b128(rp, lane * 16, 0, 16);
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    ok &= (4 * lane + k >= cnt) | (((ww[k] & 0xFFFFFF00u) == want) & ((ww[k] & 0xFFu) > (uint32_t)r));
                if (__all(ok)) break;
                if (wall_clock64() - t0 > (unsigned long long)a.wait_us * 100ull) {
                    if (lane == 0) __hip_atomic_fetch_or(&sh_leave, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
        }
        __syncthreads();   // the round is out (or given up); every wave has issued its atomics of the previous round
        // ONE thread decides which tiles go on (between two barriers: the leave bits and totals do not move here, and every
        // wave must take the same way out of this loop -- there are barriers in it)
        if (tid == 0) {
            const uint32_t lv = sh_leave;
            uint32_t go = 0u;
            if (!(lv & 4u)) {
                go |= (sh_mine[0] && !(lv & 1u) && sh_tot[0] <= a.hot_total) ? 1u : 0u;
                go |= (sh_mine[1] && !(lv & 2u) && sh_tot[1] <= a.hot_total) ? 2u : 0u;
            }
            sh_go = go;
        }
        __syncthreads();
        const uint32_t go = sh_go;
        if (!go) break;
        const bool t0_on = go & 1u, t1_on = go & 2u;
        const bool have = eb < cnt && ej < ntl && (ej ? t1_on : t0_on);
        const uint32_t sc = (uint32_t)(eb * pb_ + r);
        uint32_t ent = 0u;
        if (have) ent = __hip_atomic_load(table + (int64_t)sc * q.nt_pad + tile0 + ej, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t start = ent & 0xFFFFu, ecnt = ent >> 16;
        uint32_t nch = (ecnt + 7u) >> 3;
        if (nch > (uint32_t)V2L_MAXCH) {   // a hot tile: the tile kernel proper will take it, in pieces
            __hip_atomic_fetch_or(&sh_leave, 1u << ej, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nch = 0u;
        }
        {   // the tiles' running totals: lanes of equal parity, then one LDS atomic per wave and tile
            uint32_t s = ecnt;
#pragma unroll
            for (int off = 2; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
            if (lane < 2 && s) __hip_atomic_fetch_add(&sh_tot[lane], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const uint32_t incl = wave_scan(nch);
        const uint32_t total = __shfl(incl, 63, 64), excl = incl - nch;
        {
            const uint32_t p0 = sc * (uint32_t)q.S + start, e0 = (p0 + ecnt) | ((uint32_t)ej << 31);
            for (uint32_t k = 0; k < nch; ++k) cseg[wave][excl + k] = make_uint2(p0 + 8u * k, e0);
        }
        // (the lists are wave-private and a wave's LDS operations execute in order: no barrier between build and use)
        rounds(total);
    }
    __syncthreads();
    // ---- who writes the tiles: the consumer only after winning pending -> FLUSHING
    if (tid < 2) {
        uint32_t won = 0u;
        if (tid < ntl && sh_mine[tid]) {
            const bool leave = (sh_leave & (4u | (1u << tid))) || sh_tot[tid] > a.hot_total;
            uint32_t expect = sh_v0[tid];
            won = __hip_atomic_compare_exchange_strong(a.status + tile0 + tid, &expect, (epoch << 2) | (leave ? V2L_LEFT : V2L_FLUSHING),
                                                       __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) && !leave ? 1u : 0u;
        }
        sh_mine[tid] = won;
    }
    __syncthreads();
    const int tw = g.tw, th = g.th, tpix = tw * th;
    const int64_t plane = (int64_t)g.dom_h * g.dom_w;
    const int overwrite = flags & EVK_VOXEL_OVERWRITE;
#pragma unroll 1
    for (int j = 0; j < ntl; ++j) {
        if (!sh_mine[j]) continue;
        const int tile = tile0 + j;
        const int tx0 = (tile % g.tiles_x) * tw, ty0 = (tile / g.tiles_x) * th;
        for (int c = tid; c < B * tpix; c += V2L_WG) {
            const int b = (int)div_magic((uint32_t)c, g.mp);
            const int l = c - b * tpix;
            const int row = (int)div_magic((uint32_t)l, g.mx), col = l - row * tw;
            const int X = tx0 + col, Y = ty0 + row;
            if (X < g.dom_w && Y < g.dom_h) {
                const int cell = row * g.pitch + col;
                float v = (float)((double)(long long)accq[j * tcells + b * ppix + cell] * (1.0 / 2147483648.0));
                if ((poison[j][cell >> 5] >> (cell & 31)) & 1u) v = __uint_as_float(0x7FC00000u);
                float *o = vox + b * plane + (int64_t)Y * g.dom_w + X;
                if (!overwrite) v += __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // written through: see the header
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < ntl && sh_mine[tid])
        __hip_atomic_store(a.status + tile0 + tid, (epoch << 2) | V2L_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace evk
