// Pieces of the one-pass voxel path (evk_voxel2.hip): the tile grid with tiles of any size and the nearest-pixel key, the
// normalised time, the column sources (float32 SoA / the reference's on-disk dtypes, raw words decoded at use), an LDS-only
// workgroup barrier and a workgroup scan.
#pragma once
#include "evk_tiles.h"

namespace evk {

// vmcnt(0), expcnt / lgkmcnt untouched -- as a BUILTIN, so that the compiler's wait-count pass knows the loads are in
#define EVK_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)

// Tile grid of the one-pass voxel paths: tiles of ANY width and height (pixels), not only powers of two, so that the
// tile COUNT can be chosen: the tile kernel runs one workgroup per tile, all resident at once, and a launch lasts as long
// as the busiest CU -- 600 tiles of 32x16 on 256 CUs are 3 workgroups on 88 CUs and 2 on the others (measured: the same
// 10 M events on a 512x512 sensor, 512 tiles, take 17 % less), 512 tiles of 20x30 are 2 everywhere.
// Inside a tile a pixel is addressed as row * pitch + column with an ODD pitch (tile_w | 1): that is directly the index of
// its LDS accumulator cell (the events of a real scene sit on edges -- same column, different rows -- and an even pitch
// would put every row of a column on the same pair of banks).  Divisions by the tile size are multiplications by
// ceil(2^32 / d) (exact for operands below 2^16).
struct TileGridG {
    int tw, th, pitch;     // tile size in pixels; accumulator row pitch in cells
    int tiles_x, tiles_y;
    int dom_w, dom_h;      // key domain in pixels
    float ix, iy;          // 1 / tw, 1 / th (float32, rounded): tile of pixel u = floor((u + 0.5) * ix), see tile_of()
    uint32_t mx, mp;       // ceil(2^32 / tw), ceil(2^32 / (tw * th)): the flush's divisions (a handful per thread)
};
__host__ __device__ inline uint32_t magic_div(uint32_t d) { return (uint32_t)((((uint64_t)1 << 32) + d - 1) / d); }
__device__ __forceinline__ uint32_t div_magic(uint32_t x, uint32_t m) { return __umulhi(x, m); }   // x / d for x < 2^16
#define EVK_GRIDG_MAX_CELLS 1024   // pitch * th: the pixel field of a record has 10 bits

// u / d for 0 <= u < 2^16, 4 <= d <= 256 with full-rate instructions (32-bit integer multiplies run at quarter rate, and the
// key of every event needs two divisions): (u + 0.5) / d is never an integer and lies at least 0.5 / d away from one, while
// the float32 product (u + 0.5) * RN(1 / d) is off by less than 2^-22 * 2^16 / d = 2^-6 / d -- the floor is exact.
// (measured against the multiply-high form, same box: partition kernel 58.4 -> 54.8 us at 10 M events)
__device__ __forceinline__ int tile_of(int u, float inv) { return (int)(((float)u + 0.5f) * inv); }

static inline int make_grid_g(TileGridG &g, int dom_h, int dom_w, int tw, int th) {
    if (dom_h <= 0 || dom_w <= 0 || dom_h > 65535 || dom_w > 65535 || tw < 4 || th < 4 || tw > 256 || th > 256 ||
        (tw | 1) * th > EVK_GRIDG_MAX_CELLS)
        return EVK_EINVAL;
    g.tw = tw, g.th = th, g.pitch = tw | 1;
    g.dom_w = dom_w, g.dom_h = dom_h;
    g.tiles_x = (dom_w + tw - 1) / tw;
    g.tiles_y = (dom_h + th - 1) / th;
    g.ix = 1.0f / (float)tw, g.iy = 1.0f / (float)th;
    g.mx = magic_div((uint32_t)tw), g.mp = magic_div((uint32_t)(tw * th));
    return EVK_OK;
}

// Nearest-pixel key (EVK_KEY_NEAREST of tile_key): the tile of an event, and the accumulator cell inside the tile.
// Branch-free: sixteen of these per thread with early returns became thirty-two divergent branches and the register
// allocator spilled across them.  24-bit multiplies (full rate): every operand is below 2^16.
template <bool WRAP = true>   // WRAP: negative indices wrap once, as index_put_ does (false: np.ravel_multi_index rejects them)
__device__ __forceinline__ int nearest_key_cell_int(int xi, int yi, bool finite, const TileGridG &g, uint32_t &cell) {
    if constexpr (WRAP) {
        xi += xi < 0 ? g.dom_w : 0;
        yi += yi < 0 ? g.dom_h : 0;
    }
    const bool ok = finite & ((uint32_t)xi < (uint32_t)g.dom_w) & ((uint32_t)yi < (uint32_t)g.dom_h);
    const int ux = ok ? xi : 0, uy = ok ? yi : 0;
    const int tx = tile_of(ux, g.ix), ty = tile_of(uy, g.iy);
    cell = (uint32_t)(__mul24(uy - __mul24(ty, g.th), g.pitch) + (ux - __mul24(tx, g.tw)));
    return ok ? __mul24(ty, g.tiles_x) + tx : -1;
}
__device__ __forceinline__ int nearest_key_cell(float x, float y, const TileGridG &g, uint32_t &cell) {
    // .long() truncation (saturating v_cvt_i32_f32; NaN -> 0, rejected explicitly: torch gives INT64_MIN -> IndexError)
    return nearest_key_cell_int((int)x, (int)y, (x == x) & (y == y), g, cell);
}

// Normalised time of voxel_grid.py:134, (t - t_first) / dt * (B - 1), in float32 with the reference's operation order and an
// IEEE division (v_div_scale / v_rcp / fma ... / v_div_fixup): bit-identical to numpy / torch float32 arithmetic
// (tests/test_gpu_parity.py compares evk_normalise_time_f32 with numpy on 10^8 samples, bit for bit).
// (Measured alternative: dt is the same for every event, so y = RN(1 / dt) once and q = a * y corrected by two
// fma(-dt, q, a) / fma(r, y, q) steps -- Markstein's division, the same bits on all 10^8 samples -- is 7 instructions
// instead of 13, but with the range tests its premises need it ran 2 us SLOWER per 10 M events; tools/ab.sh.)
struct TimeNorm {
    float t_first, dt, bm1;
};
__host__ __device__ inline TimeNorm make_time_norm(float t_first, float t_last, float bm1) {
    return TimeNorm{t_first, t_last - t_first, bm1};
}
__device__ __forceinline__ float time_norm(float t, const TimeNorm &k) { return (t - k.t_first) / k.dt * k.bm1; }

// ---- column sources of the one-pass partition: G consecutive events per lane -------------------------------------------
// A load returns RAW 32-bit words; the event's values are decoded where they are USED (x_of ... p_of).  Converting at the
// load site -- int16 -> float, (float)(t - t_offset) in float64, {0,1} -> -1/+1, as round 2 did -- makes the compiler wait
// for the data right behind the load instruction: the loads then overlap nothing, which is why the 13 B/event path was
// slower than the 16 B/event one.
// NT: nontemporal loads.  A call reads its event columns ONCE; loaded with the nontemporal hint they do not displace the
// records the partition writes (and the tile kernel reads back) in L2 / MALL: 10 M events from HBM 0.0730 -> 0.0711 ms, 50 M events
// 0.340 -> 0.335 ms (tools/ab.sh, ROTATE=1).  Only a loop that re-reads ONE cache-sized stream call after call -- rounds 1-3's
// bench -- is slower with it (0.0686 -> 0.0715 ms: the columns no longer survive in the Infinity Cache from call to call).
// (round 6) The 16 bytes are declared DWORD-aligned: gfx950 takes a global_load_dwordx4 at any dword boundary, and the instruction
// is the same one (checked in the ISA: no kernel changed) -- so a column may start anywhere, e.g. a device slice xs[a:b]
// (EVK_COLUMNS_UNALIGNED: the caller then guarantees that the 12 bytes behind the column's last event are readable, because a
// group that is only partly inside the stream is loaded whole).
typedef uint32_t evk_u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) evk_quad4 {
    evk_u32x4 v;
};
template <bool NT>
__device__ __forceinline__ uint4 load_col16(const float *col, int64_t ev0, uint32_t gl) {
    const evk_quad4 *q = reinterpret_cast<const evk_quad4 *>(col + ev0) + gl;
    evk_u32x4 v;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Waddress-of-packed-member"
    if constexpr (NT) v = __builtin_nontemporal_load(&q->v);
    else v = q->v;
#pragma clang diagnostic pop
    return make_uint4(v.x, v.y, v.z, v.w);
}
struct SrcF32 {  // four float32 SoA columns, 16 B / event
    static constexpr int G = 4, XYW = 8, TPW = 8;
    const float *x, *y, *t, *p;
    template <bool NT = false>
    __device__ __forceinline__ void load_xy(int64_t ev0, uint32_t gl, uint32_t *r) const {
        const uint4 a = load_col16<NT>(x, ev0, gl), b = load_col16<NT>(y, ev0, gl);
        r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
    }
    template <bool NT = false>
    __device__ __forceinline__ void load_tp(int64_t ev0, uint32_t gl, uint32_t *r) const {
        const uint4 a = load_col16<NT>(t, ev0, gl), b = load_col16<NT>(p, ev0, gl);
        r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
    }
    __device__ __forceinline__ int key_of(const uint32_t *r, int e, const TileGridG &g, uint32_t &cell) const {
        return nearest_key_cell(__uint_as_float(r[e]), __uint_as_float(r[4 + e]), g, cell);
    }
    __device__ __forceinline__ float t_of(const uint32_t *r, int e) const { return __uint_as_float(r[e]); }
    __device__ __forceinline__ float p_of(const uint32_t *r, int e) const { return __uint_as_float(r[4 + e]); }
    __device__ __forceinline__ float t1(int64_t i) const { return t[i]; }
};

// The on-disk dtypes of the reference's event files (event_packagers.py:90-93: xs, ys int16, ts float64, ps bool;
// h5_to_memmap.py:119-121: xy int16 (N, 2), t float64, p uint8): 13 B / event.  xy_stride 1: separate x / y columns;
// 2: one interleaved (N, 2) array.  t float64 (T64) or float32; the record holds (float)(t - t_offset), the subtraction in
// float64 (the loaders' (ts - ts_0).float(), base_dataset.py:306).  p: EVK_P_U8_PM1 uint8 / bool {0,1} -> 2p - 1 (what
// the loaders' get_events does, hdf5_dataset.py:22, memmap_dataset.py:23), EVK_P_U8 uint8 as is, EVK_P_I8 int8 as is.
template <bool T64>
struct SrcNative {
    static constexpr int G = 4, XYW = 4, TPW = T64 ? 9 : 5;
    const int16_t *x, *y;
    const void *t;
    const uint8_t *p;
    double t_offset;
    int xy_stride, p_kind;
    template <bool NT = false>
    __device__ __forceinline__ void load_xy(int64_t ev0, uint32_t gl, uint32_t *r) const {
        if (xy_stride == 2) {   // x0 y0 | x1 y1 | x2 y2 | x3 y3
            const uint4 w = reinterpret_cast<const uint4 *>(x + 2 * ev0)[gl];
            r[0] = w.x, r[1] = w.y, r[2] = w.z, r[3] = w.w;
        } else {                // x0 x1 | x2 x3 ; y0 y1 | y2 y3
            const uint2 a = reinterpret_cast<const uint2 *>(x + ev0)[gl], b = reinterpret_cast<const uint2 *>(y + ev0)[gl];
            r[0] = a.x, r[1] = a.y, r[2] = b.x, r[3] = b.y;
        }
    }
    template <bool NT = false>
    __device__ __forceinline__ void load_tp(int64_t ev0, uint32_t gl, uint32_t *r) const {
        if constexpr (T64) {    // two 16-byte loads, issued together
            const uint4 a = reinterpret_cast<const uint4 *>(static_cast<const double *>(t) + ev0)[2 * gl];
            const uint4 b = reinterpret_cast<const uint4 *>(static_cast<const double *>(t) + ev0)[2 * gl + 1];
            r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
        } else {
            const uint4 a = reinterpret_cast<const uint4 *>(static_cast<const float *>(t) + ev0)[gl];
            r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w;
        }
        r[TPW - 1] = reinterpret_cast<const uint32_t *>(p + ev0)[gl];
    }
    __device__ __forceinline__ int key_of(const uint32_t *r, int e, const TileGridG &g, uint32_t &cell) const {
        // integer pixel coordinates straight from the int16 (no float round trip); the int16 is moved to the high half and
        // shifted back arithmetically
        const uint32_t wx = xy_stride == 2 ? r[e] << 16 : (e & 1 ? r[e >> 1] : r[e >> 1] << 16);
        const uint32_t wy = xy_stride == 2 ? r[e] : (e & 1 ? r[2 + (e >> 1)] : r[2 + (e >> 1)] << 16);
        return nearest_key_cell_int((int32_t)wx >> 16, (int32_t)wy >> 16, true, g, cell);
    }
    __device__ __forceinline__ float t_of(const uint32_t *r, int e) const {
        if constexpr (T64) {
            const double d = __hiloint2double((int)r[2 * e + 1], (int)r[2 * e]);
            return (float)(d - t_offset);
        } else {
            return (float)((double)__uint_as_float(r[e]) - t_offset);
        }
    }
    __device__ __forceinline__ float p_of(const uint32_t *r, int e) const {
        const uint32_t b = (r[TPW - 1] >> (8 * e)) & 0xffu;
        return p_kind == EVK_P_U8_PM1 ? (float)(2 * (int)b - 1) : (p_kind == EVK_P_I8 ? (float)(int8_t)b : (float)b);
    }
    __device__ __forceinline__ float t1(int64_t i) const {
        return (float)((T64 ? static_cast<const double *>(t)[i] : (double)static_cast<const float *>(t)[i]) - t_offset);
    }
};

}  // namespace evk
