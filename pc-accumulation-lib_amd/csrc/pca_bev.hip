// pca_bev.hip -- BEV rasteriser for gfx950: two-level counting sort with LDS-staged atomics, no global atomics.
//
//   level 1 (tiles of 8x8 cells, T tiles)
//     bev_tile_bin      every workgroup bins a contiguous chunk of the window (rotate / translate / crop / height /
//                       floor -> key = tile<<7 | cell_in_tile<<1 | set; applies and writes back an owed re-transform),
//                       scans its own LDS histogram over the tiles and leaves the chunk's kept records -- 16 bytes:
//                       {z, intensity | set, rgb | flags | cell} -- SORTED BY TILE in its own segment of the record
//                       buffer, with the per-tile counts / offsets bh, boff [tile][workgroup].  No global scan, no
//                       second pass over the window: ~10 000 points per chunk stay in registers between the passes.
//   level 2 (one workgroup per tile)
//     bev_tile_cells    gathers its tile from the G segments (RecMap: an LDS prefix table over the workgroups' counts,
//                       binary search per record); tiles above heavy_min (<= RGB_CAP) records are pushed to the heavy
//                       queue, larger size classes first.  pass 1: per (cell,set) counts / exact
//                       integer intensity sums / min z (per-thread runs of equal keys, then LDS atomics); pass 2: LDS
//                       counting sort of the colours by (cell,set); exact medians per cell (n <= 64: bit-sliced radix
//                       select on bit planes obtained by a cross-lane transpose, one lane per target; else a per-wave
//                       256-bin histogram); closed-form maps, fp16, tile written back.
//     bev_tile_cells_heavy  the queued tiles, drawn by one resident 1024-thread workgroup per CU: 256-bin histograms
//                       of 32 cells at a time filled straight from the record stream (two passes, no sort).
// 'full' = present (+) future is formed per cell (counts add, min of mins, median of the union).
#include "pca_bev_common.h"
#include "pca_k1_body.h"
#include <cstdlib>
#include <mutex>

#define KEY_INVALID 0xffffffffu
#define TS 8                      // tile side [cells]
#define TCELLS (TS * TS)
#define NFK (2 * TCELLS)          // fine keys per tile: cell_in_tile*2 + set
#ifndef AB_THREADS
#define AB_THREADS 1024           // workgroup size of the hist / scatter kernels
#endif
#define MAX_G 512                 // workgroups of the hist / scatter kernels
#ifndef BIN_RUNR
#define BIN_RUNR 4                // ... of the register path
#endif
#ifndef BIN_WPE
#define BIN_WPE 4                 // waves per SIMD level 1 is compiled for (4: 128 VGPRs, one 1024-thread workgroup per CU)
#endif
#ifndef BIN_UNR
#define BIN_UNR 4
#endif
#ifndef BIN_REG_P
#define BIN_REG_P 12              // points per thread of level 1 that stay in registers between its passes (a multiple of 4)
#endif
#ifndef C_THREADS
#define C_THREADS 256             // workgroup size of the tile kernel
#endif
#define C_WAVES (C_THREADS / 64)
#define C_HIST_WAVES 4            // waves that own a per-wave histogram (cells with more than 64 values)
#define RGB_CAP 4096              // colour records resident in LDS
#define CONTIG_MIN 2560           // tiles above this many records are read thread-contiguously (runs form)
#define HEAVY_MIN_DEFAULT 2560    // tiles above this many records go to bev_tile_cells_heavy (PCA_BEV_HEAVY_MIN; the
                                  // uniform benchmark's tiles hold ~2000, a dense tile costs the light kernel 4x its neighbours)
#define FLAG_ROAD (1u << 24)
#define FLAG_DYNOBJ (1u << 25)

// tile-ordered records: one packed stream so that a (workgroup, tile) run is one contiguous write.
// RecF is exactly one 16-byte access (a 20-byte record cost five strided dword loads per lane: the texture
// addresser, not HBM, then bounds the per-tile passes):
//   iw = f32 intensity bits (sign bit is free: intensities are >= 0) | set << 31
//   cw = r | g<<8 | b<<16 | FLAG_ROAD | FLAG_DYNOBJ | cell_in_tile << 26
struct __attribute__((aligned(16))) RecF { double z; uint32_t iw; uint32_t cw; };                     // 16 B
struct RecD { double z; double inten; uint32_t c; uint32_t fk; };                                      // 24 B (f64 intensities)

struct alignas(16) BevArgs {                             // (16: pca_fetch_block moves it in 16-byte words)
    pca_store st;
    const double *intensity64;
    const int64_t *frame_off;
    int slot_begin, slot_split, slot_end;
    int64_t max_points;
    pca_bev_params prm;
    int n_pend;           // owed re-transforms, oldest first: transform k is owed by slots [slot_begin, pend_slot_end[k])
    int write_back;       // apply them to the store (else to this raster only: they stay owed)
    int pend_slot_end[PCA_BEV_MAX_CHAIN];   // ascending
    Mat34 pend_T[PCA_BEV_MAX_CHAIN];
    int tx, T, G;
    int Gr, Gp;           // counter tables: row length (>= G) and workgroups per XCD column block (table_pos)
    int tile_mult;        // bev_tile_cells: workgroup -> tile permutation (coprime to its period: T / 4 or T)
    int heavy_min;        // tiles with more records than this are bev_tile_cells_heavy's (<= RGB_CAP)
    int stagger;          // bev_tile_cells: start offset between the workgroups that share a CU, in units of s_sleep(16) (PCA_BEV_STAGGER)
    uint32_t *key;        // [max_points]
    uint32_t *bh;         // [T][G] kept records per (tile, workgroup)
    uint32_t *boff;       // [T][G] exclusive scan of bh in that order
    int Gk;               // the first Gk of the G pieces are the tiles of a K1 that rides in level 1's launch (0: none), see bev_tile_bin_k1
    int bin_first, bin_end;   // the slots level 1 bins: the window, or what the caller proved to be all that can reach the view
                          // (pca_bev_bin_range; never with a write-back: a skipped frame could not receive the transforms it owes)
    int k1_slot, k1_n;    // its slot (the window's last: the pieces behind Gk cover the window up to it) and its input points
    uint32_t *bh0;        // [T][G] with `split`: how many of a (tile, workgroup) piece's records belong to the tile's cells 0..31
    int split;            // level 1 orders every piece by HALF of the tile (cells 0..31, then 32..63), so that an item (tile, half)
                          // of bev_tile_cells_heavy walks its own records only: 0 never, 1 if the window is small enough for
                          // level 1's register path (bev_split: decided by the kernels, which know the window), 2 always
    uint32_t *heavy_hint; // host-visible word: the heavy count of this call, read by the host before the next one
    uint32_t heavy_hint_known;   // its value when this call was made
    uint32_t *span_hint;  // host-visible word: the points (in units of 1024) this call's level 1 binned, written when the call came
    uint32_t span_hint_known;    // with a bin range; the host sizes the NEXT such launch by it (any piece count gives the same result)
    int heavy_launched;   // a bev_tile_cells_heavy launch follows (else the light kernel's last workgroup drains the queue)
    uint32_t *heavy;      // [64 + 32 T]: [0..32) tiles per size class (class 0 = largest), [32] item cursor, then the
                          // classes' tile ids [class][T] -- the queue of bev_tile_cells_heavy, filled by bev_tile_cells
    void *recs;           // RecF / RecD [max_points], tile-ordered; c = r | g<<8 | b<<16 | FLAG_*
    double *planes;
    uint16_t *planes_f16;
    double *extra;        // [3 sets][PCA_BEV_EXTRA_PLANES][px][px] or NULL
    uint32_t *status;     // context status word (PCA_STATUS_* bits)
    int dbg;
};
#define HQ_CLASSES 32
#define HQ_CURSOR 32
#define HQ_DONE 33              // workgroups of bev_tile_cells that have finished (cells_drain)
#define HQ_IDS 64

struct Window { int64_t lo, hi, sp, c_lo, c_hi; };

// Workgroup b of level 1 takes chunk b.  (Other orders were measured, the result does not depend on it -- segments, counters
// and offsets are indexed by the chunk: the chunks around the present frame first -- the ones with points inside the view,
// the expensive ones -- and outwards from there: 62.9 us against 61.0; a golden-ratio permutation that mixes expensive and
// cheap chunks: 67.3 us.  Neighbouring chunks running at the same time is worth more than an even tail.)
__device__ __forceinline__ int chunk_index(const BevArgs &a, int64_t lo, int64_t sp, int64_t chunk)
{
    (void)a; (void)lo; (void)sp; (void)chunk;
    return (int)blockIdx.x;
}
#define K1_SEG 4096               // record slots of a K1 piece (one 1024 x 4 tile of k1_body keeps at most that many points)
#define K1_RIDE 32                // K1 tiles that may ride in a level-1 launch (frames of up to 131 072 points): they come ON TOP of
                                  // the window's G pieces, so the counter tables and the record buffer are sized for G + K1_RIDE
// the window level 1 reads from the store: up to the frame whose K1 rides along (its points come out of K1's registers), else all
__device__ __forceinline__ int64_t window_end(const BevArgs &a)
{
    const int e = a.Gk ? a.k1_slot : a.slot_end;
    return a.frame_off[a.bin_end < e ? a.bin_end : e];
}
__device__ __forceinline__ int64_t window_begin(const BevArgs &a) { return a.frame_off[a.bin_first]; }
__device__ __forceinline__ Window chunk_of(const BevArgs &a, int &g)
{
    Window w;
    w.lo = window_begin(a);
    const int64_t hi0 = window_end(a);
    // (with a K1 in the launch frame_off[k1_slot + 1] is not written yet: a split behind that frame = everything the store holds)
    w.sp = (a.Gk && a.slot_split > a.k1_slot) ? hi0 : a.frame_off[a.slot_split];
    w.hi = (hi0 - w.lo > a.max_points) ? w.lo + a.max_points : hi0;
    const int G = a.G - a.Gk;                               // chunks of the store's part of the window
    const int64_t chunk = (w.hi - w.lo + G - 1) / G;
    g = chunk_index(a, w.lo, w.sp, chunk);                  // the piece: K1's tiles come first
    const int c = g - a.Gk;
    w.c_lo = w.lo + (int64_t)c * chunk;
    w.c_hi = w.c_lo + chunk < w.hi ? w.c_lo + chunk : w.hi;
    if (w.c_lo > w.hi) w.c_lo = w.c_hi = w.hi;
    return w;
}
__host__ __device__ __forceinline__ int64_t seg_stride(int64_t n, int G) { return (n + G - 1) / G; }
// The per-(tile, workgroup) counter tables bh / boff are [tile][Gr] with workgroup g at position (g mod 8) Gp + g / 8:
// workgroups are dealt to the eight XCDs round-robin, so the entries that share a cache line are written by workgroups of
// ONE XCD -- the 32 of a round fill a whole 128-byte line in that XCD's L2, which then leaves it as one full-line write.
// In plain [tile][g] order a line held the 4-byte stores of eight different L2s: 1 M partial-line writes per call, 100 MB
// of write traffic for 75 MB of payload.  Level 2 reads a tile's row as one range either way.  Gp = 0: plain order.
// Are the pieces of THIS call ordered by half of the tile (a.split: 1 = if the window stays on level 1's register path -- the
// host only knows an upper bound of the window, the kernels read its size --, 2 = always)?  The same answer in every kernel.
__device__ __forceinline__ bool bev_split(const BevArgs &a, int64_t window_points)
{
    return a.split == 2 || (a.split == 1 && window_points <= (int64_t)(a.G - a.Gk) * BIN_REG_P * AB_THREADS);
}
__device__ __forceinline__ int table_pos(const BevArgs &a, int g) { return a.Gp ? (g & 7) * a.Gp + (g >> 3) : g; }
__device__ __forceinline__ int table_group(const BevArgs &a, int p)
{
    if (!a.Gp) return p;
    const int x = p / a.Gp;
    return (p - x * a.Gp) * 8 + x;                          // may be >= G: an unused place of the row
}

// ---------------------------------------------------------------------------------------------
// level 1: bev_tile_bin -- every workgroup bins its contiguous chunk of the window into the T tiles and leaves the
// chunk's kept records SORTED BY TILE in its own segment of the record buffer (recs[g * chunk ...]), together with the
// per-tile counts and segment-local offsets bh / boff [tile][workgroup].  Level 2 gathers a tile from the G segments.
//   pass A  subtract origin, 3x3 chain, crop, height, floor -> key = tile<<7 | cell<<1 | set; LDS histogram over the
//           tiles; applies and writes back the owed re-transform (K2 fused)
//   scan    exclusive scan of the histogram in LDS -> the chunk's tile offsets (no global scan, no second kernel: a
//           global scan over T x G counters plus a scatter into tile-major order cost 6 + 31 us -- the scatter's ~5-record
//           runs were partial-line writes all over a 33 MB buffer; here a workgroup's records fill one dense 64 KB window)
//   pass B  re-reads the keys and z it has just written (L2), gathers intensity and colour of the kept points, packs
//           the 16-byte record, LDS cursor per tile
// ---------------------------------------------------------------------------------------------
// One point of pass A: owed re-transform (returns the stored coordinates), BEV-frame key.  KEY_INVALID = not in the view.
struct BinPoint { double x, y, z; uint32_t key; };
struct PendHi { int64_t v[PCA_BEV_MAX_CHAIN]; };           // first point index that does NOT owe transform k
// the owed re-transforms of point p, oldest first, each a separate fma chain (the roundings of one K2 pass per
// transform); returns whether the point owed any
__device__ __forceinline__ bool apply_owed(const BevArgs &a, const PendHi &pend_hi, int64_t p, double &X, double &Y, double &Z)
{
    bool moved = false;
#pragma unroll 1                                            // (not unrolled: the coefficients are fetched when their turn
    for (int k = 0; k < a.n_pend; ++k) {                    //  comes -- unrolled, all 48 are preloaded into scalar registers
        const int64_t hi = k == 0 ? pend_hi.v[0] : k == 1 ? pend_hi.v[1] : k == 2 ? pend_hi.v[2] : pend_hi.v[3];   // and spill)
        if (p < hi) {
            const Mat34 &T = a.pend_T[k];
            const double nx = row4(T.m + 0, X, Y, Z), ny = row4(T.m + 4, X, Y, Z), nz = row4(T.m + 8, X, Y, Z);
            X = nx; Y = ny; Z = nz;
            moved = true;
        }
    }
    return moved;
}
// BEV-frame key of one point (stored coordinates X, Y, Z after the owed transforms): KEY_INVALID = not in the view
__device__ __forceinline__ uint32_t view_key(const BevArgs &a, const Window &w, int64_t p, double X, double Y, double Z, uint8_t D)
{
    const pca_bev_params &q = a.prm;
    const double v = q.view, vlo = -0.5 * v, vhi = 0.5 * v, pxd = (double)q.px, half_px = 0.5 * pxd;
    const bool use_h = !(q.height_filter != q.height_filter);
    const double x = X - q.origin[0];
    const double y = Y - q.origin[1];
    const double z = Z - q.origin[2];
    double ax = q.R[0] * x; ax = fma(q.R[1], y, ax); ax = fma(q.R[2], z, ax);
    double ay = q.R[3] * x; ay = fma(q.R[4], y, ay); ay = fma(q.R[5], z, ay);
    double az = q.R[6] * x; az = fma(q.R[7], y, az); az = fma(q.R[8], z, az);
    ax += q.dx;
    ay += q.dy;
    bool keep = (ax > vlo) && (ax < vhi) && (ay > vlo) && (ay < vhi);
    if (use_h) keep = keep && (az < q.height_filter);
    keep = keep && (D != 1);
    uint32_t key = KEY_INVALID;
    if (keep) {
        int i = (int)floor(ax / v * pxd + half_px);
        int j = (int)floor(ay / v * pxd + half_px);
        i = i > q.px - 1 ? q.px - 1 : (i < 0 ? 0 : i);
        j = j > q.px - 1 ? q.px - 1 : (j < 0 ? 0 : j);
        const int row = q.px - 1 - j, col = i;
        const uint32_t tile = (uint32_t)((row / TS) * a.tx + (col / TS));
        const uint32_t fk = (uint32_t)(((row % TS) * TS + (col % TS)) * 2) + (p >= w.sp ? 1u : 0u);
        key = (tile << 7) | fk;
    }
    return key;
}
// One point of pass A (memory path): owed re-transform (returns the stored coordinates), BEV-frame key.
struct ViewConst;
__device__ __forceinline__ uint32_t view_key_lean(const ViewConst &c, double X, double Y, double Z, bool live, uint32_t set);
__device__ __forceinline__ BinPoint bin_point(const BevArgs &a, const ViewConst &vc, const Window &w, const PendHi &pend_hi, int64_t p,
                                              double X, double Y, double Z, uint8_t D)
{
    BinPoint r;
    const bool moved = apply_owed(a, pend_hi, p, X, Y, Z);
    if (moved && a.write_back) { a.st.x[p] = X; a.st.y[p] = Y; a.st.z[p] = Z; }
    r.x = X; r.y = Y; r.z = Z;
    r.key = view_key_lean(vc, X, Y, Z, D != 1, p >= w.sp ? 1u : 0u);
    return r;
}
// packs and stores one kept record (pass B).  dynbits: the 256-bit set of 'dynamic object' classes as eight dwords in LDS
// (the kernel's copy of prm.dynobj_mask): indexed by the point's class straight out of the kernel arguments it was a
// VECTOR load from the argument segment per record -- a memory round trip, waited for, in front of every record store of
// pass B (twelve in a row per thread).
template <bool I64>
__device__ __forceinline__ void bin_store(const BevArgs &a, const uint32_t *dynbits, uint32_t pos, uint32_t key, uint32_t rgbs, double z_store,
                                          double iv, float iv32)
{
    const pca_bev_params &q = a.prm;
    const unsigned sem = rgbs >> 24;
    uint32_t c = rgbs & 0xffffffu;
    if ((int)sem == q.road_class) c |= FLAG_ROAD;
    if ((dynbits[sem >> 5] >> (sem & 31u)) & 1u) c |= FLAG_DYNOBJ;
    const double z = z_store - q.origin[2];                 // rotation about z: row 3 of R is (0,0,1)
    if (I64) {
        RecD r; r.z = z; r.inten = iv; r.c = c; r.fk = key & 127u;
        reinterpret_cast<RecD *>(a.recs)[pos] = r;
    } else {
        const float fi = iv32;
        if (fi < 0.0f) pca_raise(a.status, PCA_STATUS_NEGATIVE_INTENSITY);
        RecF r; r.z = z;
        r.iw = (__float_as_uint(fi) & 0x7fffffffu) | ((key & 1u) << 31);
        r.cw = c | (((key & 127u) >> 1) << 26);
        reinterpret_cast<RecF *>(a.recs)[pos] = r;
    }
}

// The view test and cell key of pass A's register path, written for instruction count (the pass is bound by vector
// issue, not by memory: ~270 instructions per point before, a third of them v_readlane of spilled scalars).  Same values
// as view_key: R is a rotation about z (checked by the host), so R[2] z, R[5] z, R[6] x and R[7] y are exact zeros and
// R[8] z is z -- for FINITE z; a point whose z is not finite is dropped here, as the reference drops it (0 * inf = NaN
// poisons its x and y).
struct ViewConst { double ox, oy, oz, r0, r1, r3, r4, dx, dy, vlo, vhi, v, rv, pxd, half_px, hf; int px, tx; bool use_h; };
__device__ __forceinline__ ViewConst view_const(const BevArgs &a)
{
    const pca_bev_params &q = a.prm;
    ViewConst c;
    c.ox = q.origin[0]; c.oy = q.origin[1]; c.oz = q.origin[2];
    c.r0 = q.R[0]; c.r1 = q.R[1]; c.r3 = q.R[3]; c.r4 = q.R[4];
    c.dx = q.dx; c.dy = q.dy;
    c.v = q.view; c.rv = 1.0 / q.view; c.vlo = -0.5 * q.view; c.vhi = 0.5 * q.view; c.pxd = (double)q.px; c.half_px = 0.5 * c.pxd;
    // (handing 1 / view, (double)px and px / 2 over from the host instead was measured: 57.7-57.8 us against 56.8-57.2)
    c.hf = q.height_filter; c.use_h = !(q.height_filter != q.height_filter);
    c.px = q.px; c.tx = a.tx;
    return c;
}
__device__ __forceinline__ uint32_t view_key_lean(const ViewConst &c, double X, double Y, double Z, bool live, uint32_t set)
{
    const double x = X - c.ox, y = Y - c.oy;
    const double ax = fma(c.r1, y, c.r0 * x) + c.dx;
    const double ay = fma(c.r4, y, c.r3 * x) + c.dy;
    bool keep = live && (ax > c.vlo) && (ax < c.vhi) && (ay > c.vlo) && (ay < c.vhi) && (fabs(Z) < __builtin_huge_val());
    if (c.use_h) keep = keep && (Z - c.oz < c.hf);
    uint32_t key = KEY_INVALID;
    if (keep) {
        // floor(a / view * px + px / 2), the reference's expression.  The IEEE division (~22 instructions) is replaced by
        // the quotient estimate  q = a rv,  q += fma(-q, view, a) rv  (within one ulp of the rounded quotient): the floor
        // can only differ if the sum lands within a few ulps of an integer, and a sum within 1e-9 of one is recomputed
        // with the real division (about two points in a billion).
        const double qx0 = ax * c.rv, qy0 = ay * c.rv;
        const double qx = fma(fma(-qx0, c.v, ax), c.rv, qx0), qy = fma(fma(-qy0, c.v, ay), c.rv, qy0);
        double tx = qx * c.pxd + c.half_px, ty = qy * c.pxd + c.half_px;
        double fx = floor(tx), fy = floor(ty);
        if ((tx - fx < 1e-9) | (fx + 1.0 - tx < 1e-9) | (ty - fy < 1e-9) | (fy + 1.0 - ty < 1e-9)) {
            fx = floor(ax / c.v * c.pxd + c.half_px);
            fy = floor(ay / c.v * c.pxd + c.half_px);
        }
        int i = (int)fx;
        int j = (int)fy;
        i = i > c.px - 1 ? c.px - 1 : (i < 0 ? 0 : i);
        j = j > c.px - 1 ? c.px - 1 : (j < 0 ? 0 : j);
        const uint32_t row = (uint32_t)(c.px - 1 - j), col = (uint32_t)i;
        const uint32_t tile = (row / TS) * (uint32_t)c.tx + (col / TS);
        key = (tile << 7) | ((((row % TS) * TS + (col % TS)) * 2u) + set);
    }
    return key;
}

// The first REG_P * 1024 points of a chunk stay in REGISTERS between the passes (key and stored z; colour and intensity of
// the kept points are gathered at the start of pass B, all of a lane's gathers back to back), so pass B is LDS cursors and
// 16-byte stores.  A 200-frame KITTI window is ~10 000 points per chunk: all of it.  Whatever a chunk holds beyond that
// (giant windows) takes the memory path: keys to the key buffer, re-read with z / intensity / colour in pass B.
__device__ unsigned long long g_dbg_stamps[1024][8];   // PCA_BEV_DBG=8|16|32: per-tile / per-chunk phase stamps (diagnostics)
#define BIN_STAMP(slot) do { if ((a.dbg & 32) && threadIdx.x == 0 && blockIdx.x < 1024) g_dbg_stamps[blockIdx.x][slot] = wall_clock64(); } while (0)
// exclusive scan of a level-1 workgroup's histogram over the tiles (thread t owns `per` consecutive tiles): LDS cursors and the
// workgroup's column of the tile-major tables.  All AB_THREADS threads; one barrier inside.
__device__ __forceinline__ void bin_scan_tables(const BevArgs &a, const uint32_t *s_h, uint32_t *s_cur, uint32_t *s_wsum, bool split, int g)
{
{
    const int per = (a.T + AB_THREADS - 1) / AB_THREADS;
    const int t0 = threadIdx.x * per;
    const int gp = table_pos(a, g);
    uint32_t sum = 0;
    if (split) { for (int k = 0; k < per; ++k) sum += t0 + k < a.T ? s_h[2 * (t0 + k)] + s_h[2 * (t0 + k) + 1] : 0u; }
    else { for (int k = 0; k < per; ++k) sum += t0 + k < a.T ? s_h[t0 + k] : 0u; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan_add(sum);
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int k = 0; k < wave; ++k) before += s_wsum[k];
    uint32_t run = before + inc - sum;
    for (int k = 0; k < per; ++k) {
        const int t = t0 + k;
        if (t >= a.T) break;
        // tile-major tables [tile][place of the workgroup]: a tile's workgroup of level 2 reads its counters as one range
        if (split) {                                    // the piece = the tile's first half, then its second
            const uint32_t c0 = s_h[2 * t], c1 = s_h[2 * t + 1];
            s_cur[2 * t] = run;
            s_cur[2 * t + 1] = run + c0;
            a.bh[(int64_t)t * a.Gr + gp] = c0 + c1;
            a.boff[(int64_t)t * a.Gr + gp] = run;
            a.bh0[(int64_t)t * a.Gr + gp] = c0;
            run += c0 + c1;
        } else {
            const uint32_t c = s_h[t];
            s_cur[t] = run;
            a.bh[(int64_t)t * a.Gr + gp] = c;
            a.boff[(int64_t)t * a.Gr + gp] = run;
            run += c;
        }
    }
}
}

template <bool I64>
__device__ __forceinline__ void bev_tile_bin_body(const BevArgs &a)
{
    constexpr int REG_P = I64 ? 0 : BIN_REG_P;
    constexpr int UNR = BIN_RUNR;   // independent points per thread and iteration of the register path
    extern __shared__ uint32_t s_lds[];                     // [T] histogram, [T] cursors
    __shared__ uint32_t s_wsum[AB_THREADS / 64];
    __shared__ uint32_t s_dyn[8];                           // prm.dynobj_mask as dwords (see bin_store)
    uint32_t *s_h = s_lds;
    if (threadIdx.x < 8) {                                  // (a select chain over scalar registers, no indexed read)
        uint32_t w = (uint32_t)a.prm.dynobj_mask[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) w = (int)threadIdx.x == i ? (uint32_t)(a.prm.dynobj_mask[i >> 1] >> (32 * (i & 1))) : w;
        s_dyn[threadIdx.x] = w;
    }
    BIN_STAMP(0);
    int g;                                                  // this workgroup's chunk
    const Window w = chunk_of(a, g);
    const int64_t chunk = seg_stride(w.hi - w.lo, a.G - a.Gk);
    // split: one histogram entry and one cursor per (tile, half of its cells): key >> 6 = tile << 1 | cell >> 5
    const bool split = bev_split(a, w.hi - w.lo);
    const int hs = split ? 6 : 7;
    const int n_hist = split ? 2 * a.T : a.T;
    uint32_t *s_cur = s_lds + n_hist;
    if ((int)blockIdx.x == a.Gk) {                          // (the first workgroup that reads the store)
        // (with a K1 in the launch the window's end is not written yet: its input points bound it)
        const int64_t end_ub = a.Gk ? a.frame_off[a.k1_slot] + a.k1_n : a.frame_off[a.slot_end];
        if (threadIdx.x == 0 && end_ub - a.frame_off[a.slot_begin] > a.max_points + a.k1_n) pca_raise(a.status, PCA_STATUS_STORE_OVERFLOW);   // (a.max_points: the store's part)
        if (threadIdx.x < HQ_IDS) a.heavy[threadIdx.x] = 0;  // the heavy queue of this call starts empty
        if (threadIdx.x == 0 && a.span_hint) {
            const uint32_t kp = (uint32_t)((w.hi - w.lo + 1023) >> 10);
            if (kp != a.span_hint_known) *a.span_hint = kp;
        }
    }
    for (int t = threadIdx.x; t < n_hist; t += AB_THREADS) s_h[t] = 0;
    __syncthreads();
    PendHi pend_hi;
#pragma unroll
    for (int k = 0; k < PCA_BEV_MAX_CHAIN; ++k)
        pend_hi.v[k] = (k < a.n_pend && a.pend_slot_end[k] > a.slot_begin) ? a.frame_off[a.pend_slot_end[k]] : w.lo;
    const int64_t reg_hi = w.c_lo + (int64_t)REG_P * AB_THREADS < w.c_hi ? w.c_lo + (int64_t)REG_P * AB_THREADS : w.c_hi;
    // ---- pass A, register part ----
    // Every load is issued by every lane, in straight-line code: a lane past the end of the chunk reads the chunk's first
    // point instead.  (Loads inside divergent branches leave the compiler without a count of what is in flight; it then
    // waits for EVERYTHING before the next use, which serialised the four points of a batch into four memory round trips.)
    // Addresses are a workgroup-uniform base (the chunk's first point: scalar registers) + a 32-bit lane offset; everything
    // per point is relative to the chunk: `i < n` compares of 32-bit numbers.
    uint32_t rkey[REG_P > 0 ? REG_P : 1], rrgb[REG_P > 0 ? REG_P : 1];
    float rinten[REG_P > 0 ? REG_P : 1];
    double rz[REG_P > 0 ? REG_P : 1];
    const uint32_t n_reg = w.c_lo < reg_hi ? (uint32_t)(reg_hi - w.c_lo) : 0u;
    auto rel = [&](int64_t p) -> uint32_t {                 // position p relative to the chunk, clamped to [0, n_reg]
        const int64_t d = p - w.c_lo;
        return d <= 0 ? 0u : (d < (int64_t)n_reg ? (uint32_t)d : n_reg);
    };
    const uint32_t sp_rel = rel(w.sp);                      // lanes i >= sp_rel are 'future' points
    if (REG_P > 0 && n_reg > 0) {
        const double *xb = a.st.x + w.c_lo, *yb = a.st.y + w.c_lo, *zb = a.st.z + w.c_lo;
        const uint8_t *db = a.st.dyn + w.c_lo;
        const ViewConst vc = view_const(a);
        // (Issuing the loads of batch b + 1 before batch b is worked on -- two register sets -- was tried twice: with batches
        // of four 128 VGPRs and spills, 6 us slower; with batches of 2 / 3 / 4 and no spills 62.1 / 62.0 / 63.1 against 60.6 us:
        // the pass does not wait for its point loads.  So was dealing the window to the workgroups in round-robin blocks of 1024 points instead of
        // contiguous chunks, to even out their run times: every workgroup then pays the kept-point work, +4 us.)
        double XA[UNR], YA[UNR], ZA[UNR];
        uint32_t DA[UNR];
        auto issue = [&](int j0, double (&X)[UNR], double (&Y)[UNR], double (&Z)[UNR], uint32_t (&D)[UNR]) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const uint32_t i = (uint32_t)(j0 + u) * AB_THREADS + threadIdx.x, ic = i < n_reg ? i : 0u;
                X[u] = xb[ic];
                Y[u] = yb[ic];
                Z[u] = zb[ic];
                D[u] = db[ic];
            }
        };
        auto work = [&](int j0, double (&X)[UNR], double (&Y)[UNR], double (&Z)[UNR], uint32_t (&D)[UNR]) {
            // the owed re-transforms, oldest first, each a separate fma chain (the roundings of one K2 pass per transform).
            // Transform k's twelve coefficients are fetched (scalar loads) when its turn comes instead of all forty-eight
            // living in scalar registers through the whole pass; applied branch-free: all but the newest frames' points
            // owe every pending transform, and UNR independent chains interleave.
            uint32_t n_moved = 0;                           // lanes i < n_moved were moved by some transform
#pragma unroll 1
            for (int k = 0; k < a.n_pend; ++k) {
                const Mat34 &Tk = a.pend_T[k];
                const uint32_t n_owe = rel(k == 0 ? pend_hi.v[0] : k == 1 ? pend_hi.v[1] : k == 2 ? pend_hi.v[2] : pend_hi.v[3]);
                if (n_owe == 0) continue;                   // (uniform)
                n_moved = n_owe > n_moved ? n_owe : n_moved;
                const double t0 = Tk.m[0], t1 = Tk.m[1], t2 = Tk.m[2], t3 = Tk.m[3], t4 = Tk.m[4], t5 = Tk.m[5], t6 = Tk.m[6],
                             t7 = Tk.m[7], t8 = Tk.m[8], t9 = Tk.m[9], t10 = Tk.m[10], t11 = Tk.m[11];
                // (uniform) every point of this batch owes it -- all but the batches that reach into the newest frames: no selects
                const uint32_t batch_end = (uint32_t)(j0 + UNR) * AB_THREADS < n_reg ? (uint32_t)(j0 + UNR) * AB_THREADS : n_reg;
                const bool all_owe = n_owe >= batch_end;
                if (all_owe) {
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const double nx = fma(t2, Z[u], fma(t1, Y[u], t0 * X[u])) + t3;
                        const double ny = fma(t6, Z[u], fma(t5, Y[u], t4 * X[u])) + t7;
                        const double nz = fma(t10, Z[u], fma(t9, Y[u], t8 * X[u])) + t11;
                        X[u] = nx; Y[u] = ny; Z[u] = nz;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const uint32_t i = (uint32_t)(j0 + u) * AB_THREADS + threadIdx.x;
                        const bool owed = i < n_owe;
                        const double nx = fma(t2, Z[u], fma(t1, Y[u], t0 * X[u])) + t3;
                        const double ny = fma(t6, Z[u], fma(t5, Y[u], t4 * X[u])) + t7;
                        const double nz = fma(t10, Z[u], fma(t9, Y[u], t8 * X[u])) + t11;
                        X[u] = owed ? nx : X[u]; Y[u] = owed ? ny : Y[u]; Z[u] = owed ? nz : Z[u];
                    }
                }
            }
            if (a.write_back && n_moved > 0) {
                double *xw = a.st.x + w.c_lo, *yw = a.st.y + w.c_lo, *zw = a.st.z + w.c_lo;
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const uint32_t i = (uint32_t)(j0 + u) * AB_THREADS + threadIdx.x;
                    if (i < n_moved) { xw[i] = X[u]; yw[i] = Y[u]; zw[i] = Z[u]; }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const uint32_t i = (uint32_t)(j0 + u) * AB_THREADS + threadIdx.x;
                rkey[j0 + u] = view_key_lean(vc, X[u], Y[u], Z[u], i < n_reg && D[u] != 1u, i >= sp_rel ? 1u : 0u);
                rz[j0 + u] = Z[u];
                if (rkey[j0 + u] != KEY_INVALID) atomicAdd(&s_h[rkey[j0 + u] >> hs], 1u);
            }
        };
#pragma unroll
        for (int j0 = 0; j0 < REG_P; j0 += UNR) {
            issue(j0, XA, YA, ZA, DA);
            work(j0, XA, YA, ZA, DA);
        }
    } else {
#pragma unroll
        for (int j = 0; j < REG_P; ++j) { rkey[j] = KEY_INVALID; rz[j] = 0.0; }
    }
    BIN_STAMP(1);
    // ---- pass A, memory part (what the chunk holds beyond the registers) ----
    constexpr int MUNR = BIN_UNR;
    const ViewConst vcm = view_const(a);
    for (int64_t base = reg_hi + threadIdx.x; base < w.c_hi; base += MUNR * AB_THREADS) {
        double X[MUNR], Y[MUNR], Z[MUNR];
        uint8_t D[MUNR];
#pragma unroll
        for (int u = 0; u < MUNR; ++u) {
            const int64_t p = base + u * AB_THREADS;
            const bool in = p < w.c_hi;
            X[u] = in ? a.st.x[p] : 0.0;
            Y[u] = in ? a.st.y[p] : 0.0;
            Z[u] = in ? a.st.z[p] : 0.0;
            D[u] = in ? a.st.dyn[p] : (uint8_t)1;
        }
#pragma unroll
        for (int u = 0; u < MUNR; ++u) {
            const int64_t p = base + u * AB_THREADS;
            if (p >= w.c_hi) continue;
            const BinPoint b = bin_point(a, vcm, w, pend_hi, p, X[u], Y[u], Z[u], D[u]);
            if (b.key != KEY_INVALID) atomicAdd(&s_h[b.key >> hs], 1u);
            a.key[p - w.lo] = b.key;
        }
    }
    __syncthreads();
    BIN_STAMP(2);
    bin_scan_tables(a, s_h, s_cur, s_wsum, split, g);
    __syncthreads();
    BIN_STAMP(3);
    // ---- pass B: the chunk's records into its segment, tile by tile ----
    const uint32_t seg = (uint32_t)((int64_t)a.Gk * K1_SEG + (int64_t)(g - a.Gk) * chunk);
    if (REG_P > 0 && n_reg > 0) {
        // colour and intensity of the kept points: all of a lane's gathers issued back to back, before the first is used
        // (a point outside the view re-reads the chunk's first point: no branch around a load).  Issuing them before the
        // barrier, to fly during the scan, costs 24 more live registers: spills, slower.
        const uint32_t *cb = a.st.rgbs + w.c_lo;
        const float *ib = a.st.intensity + w.c_lo;
#pragma unroll
        for (int j = 0; j < REG_P; ++j) {
            const uint32_t i = (uint32_t)j * AB_THREADS + threadIdx.x, ic = rkey[j] != KEY_INVALID ? i : 0u;
            rrgb[j] = cb[ic];
            rinten[j] = ib[ic];
        }
        // Every gathered word is touched HERE, once: behind the divergent `continue`s below the compiler cannot count what is
        // still in flight and waited for EVERYTHING (vmcnt(0)) in front of each record -- i.e. also for the previous record's
        // store.  After this point no load is pending and the stores follow one another without a wait.  (Measured: within
        // the noise, 56.8-57.0 against 56.8-57.3 us -- the other waves of the SIMD cover the waits; kept because it is free.)
        {
            uint32_t touch = 0;
#pragma unroll
            for (int j = 0; j < REG_P; ++j) touch |= rrgb[j] ^ __float_as_uint(rinten[j]);
            asm volatile("" ::"v"(touch));
        }
        // (all cursor atomics first, then all stores: no gain, 62.2 against 61.5 us.  Round 4, same box, alternating runs: the
        // counter tables stored at the very end of the kernel instead of in front of these gathers 57.1-57.3 against 56.8-57.2;
        // origin z / road class / record pointer / segment base parked in vector registers for the loop instead of being
        // re-read as scalars per record 57.8-58.0 against 57.2-57.3: neither kept.)
#pragma unroll
        for (int j = 0; j < REG_P; ++j) {
            if (rkey[j] == KEY_INVALID) continue;
            const uint32_t pos = seg + atomicAdd(&s_cur[rkey[j] >> hs], 1u);
            bin_store<I64>(a, s_dyn, pos, rkey[j], rrgb[j], rz[j], 0.0, rinten[j]);
        }
    }
    BIN_STAMP(4);
    const bool stale = a.n_pend > 0 && !a.write_back;
    for (int64_t base = reg_hi + threadIdx.x; base < w.c_hi; base += MUNR * AB_THREADS) {
        uint32_t key[MUNR], pos[MUNR], rgbs[MUNR];
        double zz[MUNR], iv[MUNR], xs[MUNR], ys[MUNR];
#pragma unroll
        for (int u = 0; u < MUNR; ++u) {
            const int64_t p = base + u * AB_THREADS;
            key[u] = p < w.c_hi ? a.key[p - w.lo] : KEY_INVALID;
        }
        // every load of the kept points first, nothing consumed in between (the owed chain used to be applied inside this loop:
        // it waited for x, y, z of point u before the loads of point u + 1 were issued -- four round trips per iteration)
#pragma unroll
        for (int u = 0; u < MUNR; ++u) {
            const int64_t p = base + u * AB_THREADS;
            const bool ok = key[u] != KEY_INVALID;
            rgbs[u] = ok ? a.st.rgbs[p] : 0u;
            zz[u] = ok ? a.st.z[p] : 0.0;
            xs[u] = (ok && stale) ? a.st.x[p] : 0.0;        // transforms still owed and not written back by pass A:
            ys[u] = (ok && stale) ? a.st.y[p] : 0.0;        // the store holds the coordinates from before them
            if (I64) iv[u] = ok ? a.intensity64[p] : 0.0;
            else iv[u] = ok ? (double)a.st.intensity[p] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < MUNR; ++u) {
            const int64_t p = base + u * AB_THREADS;
            const bool ok = key[u] != KEY_INVALID;
            if (ok && stale) apply_owed(a, pend_hi, p, xs[u], ys[u], zz[u]);
            pos[u] = ok ? seg + atomicAdd(&s_cur[key[u] >> hs], 1u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < MUNR; ++u)
            if (key[u] != KEY_INVALID) bin_store<I64>(a, s_dyn, pos[u], key[u], rgbs[u], zz[u], iv[u], (float)iv[u]);
    }
    BIN_STAMP(5);
    if ((a.dbg & 32) && threadIdx.x == 0 && blockIdx.x < 1024) { g_dbg_stamps[blockIdx.x][6] = (unsigned long long)a.n_pend * 2 + a.write_back; g_dbg_stamps[blockIdx.x][7] = __smid(); }
}

// A tile's records lie in up to G pieces, one per level-1 workgroup.  RecMap (LDS) turns the tile-local record number
// into its place in the record buffer: pre[g] = records of the pieces before piece g, base[g] = where piece g starts.
struct RecMap {
    uint32_t pre[1024 + 1];
    uint32_t base[1024];
    uint32_t wsum[16];
};
// builds the map of `tile` (all `nthreads` threads of the workgroup, G <= 1024); returns the tile's record count.
// Ends with a barrier.
// half = 0 / 1 (only with a.split): the map of the records of the tile's cells 0..31 / 32..63 alone -- level 1 laid every piece
// out as [first half][second half] and left the first half's count in bh0.
__device__ __forceinline__ uint32_t recmap_build(RecMap &M, const BevArgs &a, int tile, int nthreads, uint16_t *owner = nullptr, int half = -1)
{
    // (places of the row, not workgroups: see table_pos; a place beyond the last workgroup counts as an empty piece)
    const int per = (a.Gr + nthreads - 1) / nthreads;
    const int g0 = threadIdx.x * per;
    const uint32_t *cnt = a.bh + (int64_t)tile * a.Gr, *off = a.boff + (int64_t)tile * a.Gr;
    uint32_t c[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0}, c0[4] = {0, 0, 0, 0}, sum = 0;  // per <= 4: Gr <= 1024, nthreads >= 256
    int grp[4] = {0, 0, 0, 0};
    // (the counters' loads are issued before the window's size is waited for: two memory round trips side by side)
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < per && g0 + k < a.Gr) {
            grp[k] = table_group(a, g0 + k);
            if (grp[k] < a.G) {
                c[k] = cnt[g0 + k]; o[k] = off[g0 + k];
                if (half >= 0) c0[k] = a.bh0[(int64_t)tile * a.Gr + g0 + k];     // (uniform; read before it is known to be valid: never used then)
            }
        }
    const int64_t lo = window_begin(a), hi0 = window_end(a);
    const int64_t n = hi0 - lo > a.max_points ? a.max_points : hi0 - lo;
    if (half >= 0 && bev_split(a, n)) {                     // (uniform) the item's own half of every piece
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (half == 0) c[k] = c0[k]; else { o[k] += c0[k]; c[k] -= c0[k]; } }
    }
    const uint32_t chunk = (uint32_t)seg_stride(n, a.G - a.Gk);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < per && g0 + k < a.Gr) {
            M.base[g0 + k] = (grp[k] < a.Gk ? (uint32_t)grp[k] * K1_SEG : (uint32_t)a.Gk * K1_SEG + (uint32_t)(grp[k] - a.Gk) * chunk) + o[k];
            sum += c[k];
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan_add(sum);
    if (lane == 63) M.wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int k = 0; k < wave; ++k) run += M.wsum[k];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < per && g0 + k < a.Gr) {
            M.pre[g0 + k] = run;
            // the light tile kernel's lookup table: owner[i] = the piece record i lies in (a piece holds ~3 records of a
            // tile), so that a record costs three LDS reads instead of the ten-step search (50 of the ~200 vector
            // instructions a record cost).  Only for tiles that fit the table (RGB_CAP records): the caller checks.
            if (owner && run + c[k] <= RGB_CAP)
                for (uint32_t j = 0; j < c[k]; ++j) owner[run + j] = (uint16_t)(g0 + k);
            run += c[k];
        }
    if ((int)threadIdx.x == nthreads - 1) M.pre[a.Gr] = run;
    __syncthreads();
    return M.pre[a.Gr];
}
// place of the tile's record number i (< the tile's count): the last piece g with pre[g] <= i.  Ten fixed steps, no
// branches: the lookups of a thread's 16 records are independent chains of LDS reads that the compiler interleaves (a
// data-dependent loop per record cost the light kernel 7 us).
__device__ __forceinline__ uint32_t recmap_at(const RecMap &M, int G, uint32_t i)
{
    int lo = 0;
#pragma unroll
    for (int step = 512; step >= 1; step >>= 1) {
        const int m = lo + step;
        const uint32_t pm = M.pre[m < G ? m : G];          // pre[G] = the tile's count > i
        lo = pm <= i ? m : lo;
    }
    return M.base[lo] + (i - M.pre[lo]);
}
// A thread that reads CONSECUTIVE records of a tile (the heavy kernel: a contiguous chunk per thread) searches once and
// walks from there: a piece of a dense tile holds tens of records, so the ten-step search per record (50 vector
// instructions, a quarter of the kernel's) becomes an increment and a compare.
struct RecWalk { uint32_t i, end, at; };                   // record i lies at `at`; its piece ends before record `end`
__device__ __forceinline__ void recwalk_seek(RecWalk &w, const RecMap &M, int G, uint32_t i)
{
    int lo = 0;
#pragma unroll
    for (int step = 512; step >= 1; step >>= 1) {
        const int m = lo + step;
        const uint32_t pm = M.pre[m < G ? m : G];
        lo = pm <= i ? m : lo;
    }
    w.i = i; w.end = M.pre[lo + 1]; w.at = M.base[lo] + (i - M.pre[lo]);
}
// place of the current record; moves on to the next one (a new search at the end of a piece: it skips the empty pieces --
// a dense tile of a real accumulation is fed by a few dozen of the 512).  Not to be called past the tile's last record.
__device__ __forceinline__ uint32_t recwalk_next(RecWalk &w, const RecMap &M, int G, uint32_t n_records)
{
    const uint32_t at = w.at;
    ++w.i; ++w.at;
    if (w.i == w.end && w.i < n_records) recwalk_seek(w, M, G, w.i);
    return at;
}
// the light tile kernel's lookup (owner[] filled by recmap_build)
__device__ __forceinline__ uint32_t recmap_at(const RecMap &M, const uint16_t *owner, uint32_t i)
{
    const uint32_t g = owner[i];
    return M.base[g] + (i - M.pre[g]);
}
// the tile joins the queue of bev_tile_cells_heavy, larger size classes first (classes of 1024 records): its workgroups
// draw items one at a time and the kernel ends with the last item to finish
__device__ __forceinline__ void heavy_push(const BevArgs &a, int tile, uint32_t count)
{
    const uint32_t c = count >> 10, cls = (HQ_CLASSES - 1) - (c > HQ_CLASSES - 1 ? HQ_CLASSES - 1 : c);
    a.heavy[HQ_IDS + (size_t)cls * a.T + atomicAdd(&a.heavy[cls], 1u)] = (uint32_t)tile;
}

// ---------------------------------------------------------------------------------------------
// level 2: one workgroup per tile.  Tiles whose records fit the LDS colour buffer take bev_tile_cells (256
// threads, 4 workgroups per CU); the others -- the cells next to the driven path of a real accumulation hold
// hundreds to thousands of points -- take bev_tile_cells_heavy (1024 threads, direct 256-bin histograms).
// ---------------------------------------------------------------------------------------------
struct TileStats {
    uint32_t cnt[NFK], road[NFK], dyn[NFK];
    unsigned long long ihi[NFK], ilo[NFK], zk[NFK];
    unsigned long long zmaxk[NFK], zhi[NFK], zlo[NFK];      // extra reducers (max z, exact sum of z): only with a.extra
    uint32_t med2[3][TCELLS][3];                            // [present, future, full][cell][channel]
};

__device__ __forceinline__ void stats_init(TileStats &S, int nthreads)
{
    for (int k = threadIdx.x; k < NFK; k += nthreads) {
        S.cnt[k] = 0; S.road[k] = 0; S.dyn[k] = 0; S.ihi[k] = 0; S.ilo[k] = 0; S.zk[k] = ~0ull;
        S.zmaxk[k] = 0; S.zhi[k] = 0; S.zlo[k] = 0;
    }
    for (int k = threadIdx.x; k < 3 * TCELLS * 3; k += nthreads) (&S.med2[0][0][0])[k] = 0;
}

// Per (cell,set) statistics are accumulated by every THREAD over a run of consecutive records with the same key and
// reach LDS once per run.  Records that arrive together are neighbours in space (consecutive lidar returns), so in a
// real accumulation a thread's contiguous chunk of a dense tile lies in one or two cells; LDS atomics on one address
// cost 2 cycles per lane on the CU's only LDS pipe (tools/experiments/lds_atomics.hip), a register add costs nothing.
// (With random keys every run has length one and this degenerates to one set of atomics per record.)
struct Run {
    uint32_t key, cnt, road, dyn;
    unsigned long long zmin, zmax;
    long long ihi, ilo, zhi, zlo;
};
#define RUN_NONE 0xffffffffu
__device__ __forceinline__ void run_reset(Run &r, uint32_t key)
{
    r.key = key; r.cnt = 0; r.road = 0; r.dyn = 0; r.zmin = ~0ull; r.zmax = 0; r.ihi = 0; r.ilo = 0; r.zhi = 0; r.zlo = 0;
}
// v = hi * 2^-20 + lo * 2^-60 exactly (hi = floor(v 2^20), lo = rint(frac 2^40)): the two terms of the fixed-point sums.
// The conversions double -> int64 go through the 1.5 * 2^52 trick (one add and a 64-bit subtract instead of the ~8
// instructions of the emulated cast; the add rounds to nearest-even like rint) while |v 2^20| < 2^51.
__device__ __forceinline__ void fx_split(double v, long long &hi, long long &lo)
{
    const double sc = v * FX_HI, fl = floor(sc);
    constexpr double MAGIC = 6755399441055744.0;            // 1.5 * 2^52 = 0x4338000000000000
    if (fabs(sc) < 2251799813685248.0) {                    // 2^51
        hi = __double_as_longlong(fl + MAGIC) - 0x4338000000000000ll;
        lo = __double_as_longlong((sc - fl) * FX_LO + MAGIC) - 0x4338000000000000ll;
    } else {
        hi = (long long)fl;
        lo = (long long)rint((sc - fl) * FX_LO);
    }
}
// (iv: the record's intensity as stored; div255 -- NuScenes' i / 255. -- is applied HERE, to the road points that use it:
// formed when the record was loaded, the IEEE division (12 vector instructions and a 16-cycle reciprocal) ran for every
// record of every tile, KITTI's included, where it is never wanted)
__device__ __forceinline__ double road_intensity(double iv, bool div255)
{
    if (div255) {                                           // (uniform)
        asm volatile("" ::: "memory");                      // keeps this a BRANCH: written as a select the division runs regardless
        iv = iv / 255.0;
    }
    return iv;
}
__device__ __forceinline__ void run_add(Run &r, bool extra, uint32_t c, double z, double iv, bool div255)
{
    const unsigned long long zkey = f64_order_key(z);
    r.cnt++;
    r.zmin = zkey < r.zmin ? zkey : r.zmin;
    if (extra) {
        long long hi, lo;
        fx_split(z, hi, lo);
        r.zmax = zkey > r.zmax ? zkey : r.zmax;
        r.zhi += hi;
        r.zlo += lo;
    }
    if (c & FLAG_DYNOBJ) r.dyn++;
    if (c & FLAG_ROAD) {
        long long hi, lo;
        fx_split(road_intensity(iv, div255), hi, lo);
        r.road++;
        r.ihi += hi;
        r.ilo += lo;
    }
}
// one record on its own (a run of length one without the bookkeeping): returns its rank inside its (cell,set)
__device__ __forceinline__ uint32_t rec_add(TileStats &S, bool extra, uint32_t key, uint32_t c, double z, double iv, bool div255)
{
    const unsigned long long zkey = f64_order_key(z);
    const uint32_t rank = atomicAdd(&S.cnt[key], 1u);
    atomicMin(&S.zk[key], zkey);
    if (extra) {
        long long hi, lo;
        fx_split(z, hi, lo);
        atomicMax(&S.zmaxk[key], zkey);
        atomicAdd(&S.zhi[key], (unsigned long long)hi);
        atomicAdd(&S.zlo[key], (unsigned long long)lo);
    }
    if (c & FLAG_DYNOBJ) atomicAdd(&S.dyn[key], 1u);
    if (c & FLAG_ROAD) {
        long long hi, lo;
        fx_split(road_intensity(iv, div255), hi, lo);
        atomicAdd(&S.road[key], 1u);
        atomicAdd(&S.ihi[key], (unsigned long long)hi);
        atomicAdd(&S.ilo[key], (unsigned long long)lo);
    }
    return rank;
}
// returns the rank of the run's first record inside its (cell,set)
__device__ __forceinline__ uint32_t run_flush(TileStats &S, bool extra, const Run &r)
{
    const uint32_t base = atomicAdd(&S.cnt[r.key], r.cnt);
    atomicMin(&S.zk[r.key], r.zmin);
    if (extra) {
        atomicMax(&S.zmaxk[r.key], r.zmax);
        atomicAdd(&S.zhi[r.key], (unsigned long long)r.zhi);
        atomicAdd(&S.zlo[r.key], (unsigned long long)r.zlo);
    }
    if (r.dyn) atomicAdd(&S.dyn[r.key], r.dyn);
    if (r.road) {
        atomicAdd(&S.road[r.key], r.road);
        atomicAdd(&S.ihi[r.key], (unsigned long long)r.ihi);
        atomicAdd(&S.ilo[r.key], (unsigned long long)r.ilo);
    }
    return base;
}

template <bool I64>
__device__ __forceinline__ void load_rec(const BevArgs &a, uint32_t r, uint32_t &k, uint32_t &c, double &z, double &iv)
{
    if (I64) {
        const RecD rec = reinterpret_cast<const RecD *>(a.recs)[r];
        k = rec.fk; c = rec.c; z = rec.z; iv = rec.inten;
    } else {
        const RecF rec = reinterpret_cast<const RecF *>(a.recs)[r];
        k = ((rec.cw >> 26) << 1) | (rec.iw >> 31);
        c = rec.cw & 0x03ffffffu;
        z = rec.z;
        iv = (double)__uint_as_float(rec.iw & 0x7fffffffu);                   // (/ 255: see run_add)
    }
}
// A RecF as one 16-byte word, and its fields.  The tile kernels load the records of a round with EVERY lane issuing EVERY load
// (a lane past the tile's end re-reads a record that exists) and decode afterwards: a load inside `if (r < r_hi)` is consumed
// inside that branch, so the compiler waited for each record before it asked for the next -- a chain of sixteen memory round
// trips per thread was what "pass 1" of a tile consisted of (ISA: global_load_dwordx4; s_waitcnt vmcnt(0), per record).
__device__ __forceinline__ uint4 recf_raw(const BevArgs &a, uint32_t pos)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(reinterpret_cast<uintptr_t>(a.recs) + (size_t)pos * 16);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void recf_fields(const uint4 w, uint32_t &k, uint32_t &c, double &z, double &iv)
{
    k = ((w.w >> 26) << 1) | (w.z >> 31);
    c = w.w & 0x03ffffffu;
    z = __hiloint2double((int)w.y, (int)w.x);
    iv = (double)__uint_as_float(w.z & 0x7fffffffu);
}
// fine key and colour only
template <bool I64>
__device__ __forceinline__ void load_rec_key_colour(const BevArgs &a, uint32_t r, uint32_t &k, uint32_t &c)
{
    if (I64) {
        const RecD *rec = reinterpret_cast<const RecD *>(a.recs) + r;
        k = rec->fk; c = rec->c & 0xffffffu;
    } else {
        const uint2 w = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(a.recs) + (size_t)r * 16 + 8);
        k = ((w.y >> 26) << 1) | (w.x >> 31);
        c = w.y & 0xffffffu;
    }
}

// closed-form maps of the tile (thread -> (set, cell)), staged in LDS, then written row by row
__device__ __forceinline__ void tile_finalize_write(const BevArgs &a, const TileStats &S, double (*s_out)[TCELLS], int tile,
                                                    int nthreads, int cell_lo = 0, int cell_hi = TCELLS)
{
    const pca_bev_params &q = a.prm;
    const bool extra = a.extra != nullptr;
    if (threadIdx.x < 3 * TCELLS) {
        const int s = threadIdx.x / TCELLS, cell = threadIdx.x % TCELLS;
        const int kp = 2 * cell, kf = 2 * cell + 1;
        uint32_t n, n_r, n_d;
        long long ihi, ilo, zhi, zlo;
        unsigned long long zk, zmaxk;
        if (s < 2) {
            const int k = s ? kf : kp;
            n = S.cnt[k]; n_r = S.road[k]; n_d = S.dyn[k]; ihi = (long long)S.ihi[k]; ilo = (long long)S.ilo[k];
            zk = S.zk[k]; zmaxk = S.zmaxk[k]; zhi = (long long)S.zhi[k]; zlo = (long long)S.zlo[k];
        } else {
            n = S.cnt[kp] + S.cnt[kf]; n_r = S.road[kp] + S.road[kf]; n_d = S.dyn[kp] + S.dyn[kf];
            ihi = (long long)S.ihi[kp] + (long long)S.ihi[kf];
            ilo = (long long)S.ilo[kp] + (long long)S.ilo[kf];
            zk = S.zk[kp] < S.zk[kf] ? S.zk[kp] : S.zk[kf];
            zmaxk = S.zmaxk[kp] > S.zmaxk[kf] ? S.zmaxk[kp] : S.zmaxk[kf];
            zhi = (long long)S.zhi[kp] + (long long)S.zhi[kf];
            zlo = (long long)S.zlo[kp] + (long long)S.zlo[kf];
        }
        double o[7];
        finalize_cell(q, n, n_r, n_d, ihi, ilo, n ? f64_from_order_key(zk) : 0.0, S.med2[s][cell], o);
#pragma unroll
        for (int k = 0; k < 7; ++k) s_out[7 * s + k][cell] = o[k];
        if (extra) {
            const double isum = (double)ihi * FX_HI_INV + (double)ilo * FX_LO_INV;
            const double zsum = (double)zhi * FX_HI_INV + (double)zlo * FX_LO_INV;
            s_out[21 + 3 * s + 0][cell] = n ? f64_from_order_key(zmaxk) : 0.0;
            s_out[21 + 3 * s + 1][cell] = n ? zsum / (double)n : 0.0;
            s_out[21 + 3 * s + 2][cell] = n_r ? isum / (double)n_r : 0.0;
        }
    }
    __syncthreads();
    const int row0 = (tile / a.tx) * TS, col0 = (tile % a.tx) * TS;
    const int64_t ncell = (int64_t)q.px * q.px;
    const int n_planes = extra ? 21 + PCA_BEV_EXTRA_PLANES * 3 : 21;
    // fp16 planes only, whole tile inside the grid: one 16-byte store per (plane, row of the tile) instead of eight 2-byte ones
    static_assert(TS == 8, "a tile row of fp16 cells is one uint4");
    if (!a.planes && a.planes_f16 && !extra && cell_lo == 0 && cell_hi == TCELLS && (q.px % TS) == 0 &&
        (reinterpret_cast<uintptr_t>(a.planes_f16) & 15) == 0) {
        if (threadIdx.x < 21 * TS) {
            const int plane = threadIdx.x / TS, r = threadIdx.x % TS;
            const double *src = &s_out[plane][r * TS];
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = (uint32_t)f64_to_f16_bits(src[2 * k]) | ((uint32_t)f64_to_f16_bits(src[2 * k + 1]) << 16);
            *reinterpret_cast<uint4 *>(a.planes_f16 + (int64_t)plane * ncell + (int64_t)(row0 + r) * q.px + col0) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return;
    }
    for (int idx = threadIdx.x; idx < n_planes * TCELLS; idx += nthreads) {
        const int plane = idx / TCELLS, lc = idx % TCELLS;
        const int row = row0 + lc / TS, col = col0 + lc % TS;
        if (row >= q.px || col >= q.px || lc < cell_lo || lc >= cell_hi) continue;
        const double v = s_out[plane][lc];
        if (plane < 21) {
            const int64_t o = (int64_t)plane * ncell + (int64_t)row * q.px + col;
            if (a.planes) a.planes[o] = v;
            if (a.planes_f16) a.planes_f16[o] = f64_to_f16_bits(v);
        } else {
            a.extra[(int64_t)(plane - 21) * ncell + (int64_t)row * q.px + col] = v;
        }
    }
}
#define OUT_STAGE_BYTES ((21 + 3 * PCA_BEV_EXTRA_PLANES) * TCELLS * 8)

// ---- medians from 16-bit-packed 256-bin histograms (bin b = half b&1 of dword b>>1), one wave ----------------
__device__ __forceinline__ uint4 hist16_bins(const uint32_t *row)
{
    const uint2 w = reinterpret_cast<const uint2 *>(row)[threadIdx.x & 63];
    return make_uint4(w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16);
}
__device__ __forceinline__ void hist16_add(uint32_t *row, unsigned val, uint32_t times)
{
    atomicAdd(&row[val >> 1], times << (16 * (val & 1)));
}
// present / future / full medians of one (cell, channel) from its two packed histograms
__device__ __forceinline__ void hist16_medians(TileStats &S, const uint32_t *row_p, const uint32_t *row_f, int cell, int ch,
                                               uint32_t n_p, uint32_t n_f)
{
    const uint4 p = hist16_bins(row_p), f = hist16_bins(row_f);
    const uint4 u = make_uint4(p.x + f.x, p.y + f.y, p.z + f.z, p.w + f.w);
    // two scans serve three histograms: the prefix of the union is the sum of the prefixes
    const uint32_t s_p = p.x + p.y + p.z + p.w, s_f = f.x + f.y + f.z + f.w;
    const uint32_t e_p = wave_incl_scan_add(s_p) - s_p, e_f = wave_incl_scan_add(s_f) - s_f;
    const uint32_t m_p = hist_med2_scanned(p, e_p, s_p, n_p), m_f = hist_med2_scanned(f, e_f, s_f, n_f),
                   m_u = hist_med2_scanned(u, e_p + e_f, s_p + s_f, n_p + n_f);
    if ((threadIdx.x & 63) == 0) { S.med2[0][cell][ch] = m_p; S.med2[1][cell][ch] = m_f; S.med2[2][cell][ch] = m_u; }
}

struct TileLds {
    TileStats S;
    uint32_t off[NFK + 1];
    uint32_t n_big;                                         // cells of the tile with more than 64 values
    union {
        unsigned long long bits[TCELLS][24];                // cells with <= 64 values: one 64-lane mask per colour bit
        uint32_t whist[C_HIST_WAVES][2][3][128];            // cells with more: per-wave packed histograms [set][channel]
        uint16_t owner[RGB_CAP];                            // pass 1: record -> piece (recmap_expand)
    };
};

// ---- medians of cells with at most 64 values: bit-sliced radix select -------------------------------------
// Phase 1 (per cell, whole wave): the cell's values sit one per lane (present lanes first); a cross-lane bit
// transpose gives the 64-bit membership mask of every colour bit, parked in LDS.
// Phase 2 (per wave): one LANE per (cell, set, channel, lower|upper middle) target -- 18 per cell -- walks the 8
// bit planes from the top: zeros = cand & ~plane; rank < popcount(zeros) ? keep zeros : (rank -= ..., keep ones,
// set the bit).  That is ~12 VALU per plane for 64 targets at once instead of a 21-stage sort per channel.
__device__ __forceinline__ void small_cells_bitplanes(TileLds &L, const uint32_t *s_rgb)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER_WAVE = TCELLS / C_WAVES;              // this wave's cells: C_WAVES * i + wave
    static_assert(PER_WAVE <= 64, "one lane per cell of the wave");
    uint32_t n_mine = 0;
    if (lane < PER_WAVE) n_mine = L.S.cnt[2 * (C_WAVES * lane + wave)] + L.S.cnt[2 * (C_WAVES * lane + wave) + 1];
    // The transpose works on the two halves of the wave independently: two cells of at most 32 values share one (the
    // uniform 200-frame window averages 25 values per cell), lanes 0..31 holding the first cell's values, 32..63 the second's.
    unsigned long long half = __ballot(n_mine >= 1 && n_mine <= 32), full = __ballot(n_mine > 32 && n_mine <= 64);
    while (half) {
        const int ia = __ffsll(half) - 1;
        half &= half - 1;
        const int ib = half ? __ffsll(half) - 1 : -1;
        half &= half - 1;                                   // (0 & anything stays 0)
        const int i = lane < 32 ? ia : ib;
        const int l = lane & 31;
        uint32_t v = 0;
        int cell = 0;
        if (i >= 0) {
            cell = C_WAVES * i + wave;
            const uint32_t n = L.S.cnt[2 * cell] + L.S.cnt[2 * cell + 1];
            if ((uint32_t)l < n) v = s_rgb[L.off[2 * cell] + l];
        }
        const uint32_t t = wave_bit_transpose32(v);
        if (i >= 0 && l < 24) L.bits[cell][l] = (unsigned long long)t;
    }
    while (full) {
        const int i = __ffsll(full) - 1;
        full &= full - 1;
        const int cell = C_WAVES * i + wave;
        const uint32_t n = L.S.cnt[2 * cell] + L.S.cnt[2 * cell + 1];
        const uint32_t v = (uint32_t)lane < n ? s_rgb[L.off[2 * cell] + lane] : 0u;
        // 64 values x 24 bits -> 24 masks of 64 lanes: a bit-matrix transpose across lanes (30 VALU; 24 ballots with
        // their select chains were 130)
        const uint32_t t = wave_bit_transpose32(v);
        if ((lane & 31) < 24) reinterpret_cast<uint32_t *>(&L.bits[cell][lane & 31])[lane >> 5] = t;
    }
}

// the walk down the eight bit planes of one target; W = uint32_t when no cell of the wave's batch holds more than 32 values
template <typename W>
__device__ __forceinline__ uint32_t bitplane_select(const unsigned long long *planes, W cand, uint32_t k)
{
    uint32_t val = 0;
#pragma unroll
    for (int b = 7; b >= 0; --b) {
        const W B = (W)planes[b];
        const W zeros = cand & ~B;
        const uint32_t cz = sizeof(W) == 8 ? (uint32_t)__popcll((unsigned long long)zeros) : (uint32_t)__popc((uint32_t)zeros);
        const bool take0 = k < cz;
        cand = take0 ? zeros : (W)(cand & B);
        k = take0 ? k : k - cz;
        val |= take0 ? 0u : (1u << b);
    }
    return val;
}

__device__ __forceinline__ void small_cells_select(TileLds &L)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NT = (TCELLS / C_WAVES) * 18;             // targets of one wave
    for (int t0 = 0; t0 < NT; t0 += 64) {
        const int t = t0 + lane;
        const int i = t / 18, qq = t - 18 * i;
        const int cell = C_WAVES * i + wave;
        const int set = qq / 6, ch = (qq % 6) >> 1, upper = qq & 1;
        bool ok = t < NT;
        uint32_t n_p = 0, n_f = 0;
        if (ok) { n_p = L.S.cnt[2 * cell]; n_f = L.S.cnt[2 * cell + 1]; }
        const uint32_t n = n_p + n_f;
        ok = ok && n > 0 && n <= 64;
        const unsigned long long m_p = n_p >= 64 ? ~0ull : ((1ull << n_p) - 1ull);
        const unsigned long long m_all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
        const unsigned long long cand = set == 0 ? m_p : (set == 1 ? (m_all & ~m_p) : m_all);
        const uint32_t n_s = (uint32_t)__popcll(cand);
        ok = ok && n_s > 0;
        const uint32_t k = upper ? (n_s >> 1) : ((n_s - 1) >> 1);
        uint32_t val = 0;
        const unsigned long long *planes = &L.bits[ok ? cell : 0][8 * ch];
        if (__any(ok && n > 32)) {
            if (ok) val = bitplane_select<unsigned long long>(planes, cand, k);
        } else {
            if (ok) val = bitplane_select<uint32_t>(planes, (uint32_t)cand, k);
        }
        // the partner target (lower <-> upper middle) is the adjacent lane: quad_perm [1,0,3,2]
        const uint32_t other = dpp_or<0xb1>(0u, val);
        if (ok && !upper) L.S.med2[set][cell][ch] = val + other;
    }
}

// ---- cells with more than 64 values in a tile whose colours sit sorted in LDS: every wave takes cells on its
// own (no workgroup barrier), histogramming into 16-bit counters (a fitting tile holds <= RGB_CAP values)
__device__ __forceinline__ void wave_cells_hist(TileLds &L, const uint32_t *s_rgb)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= C_HIST_WAVES) return;                       // (the histograms' LDS is sized for four waves)
    uint32_t *h = &L.whist[wave][0][0][0];
    for (int i = 0; i < TCELLS / C_HIST_WAVES; ++i) {
        const int cell = C_HIST_WAVES * i + wave;
        const uint32_t n_p = L.S.cnt[2 * cell], n_f = L.S.cnt[2 * cell + 1], n = n_p + n_f;
        if (n <= 64) continue;
        for (int k = lane; k < 2 * 3 * 128 / 4; k += 64) reinterpret_cast<uint4 *>(h)[k] = make_uint4(0, 0, 0, 0);
        const uint32_t base = L.off[2 * cell];
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t idx = i0 + lane;
            const bool act = idx < n;
            const uint32_t v = act ? s_rgb[base + idx] : 0u;
            const uint32_t set = idx >= n_p ? 1u : 0u;
            // a wave of identical colours in one set (e.g. rgb == 0 with GT semantics) adds once per channel
            const uint32_t tag = act ? (v | (set << 24)) : 0xffffffffu;
            const uint32_t first = __builtin_amdgcn_readfirstlane(tag);
            const uint64_t actm = __ballot(act);
            if (__ballot(tag == first) == actm && first != 0xffffffffu) {
                if (lane == (int)__ffsll((unsigned long long)actm) - 1) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch)
                        hist16_add(&L.whist[wave][set][ch][0], (v >> (8 * ch)) & 255u, (uint32_t)__popcll(actm));
                }
            } else if (act) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) hist16_add(&L.whist[wave][set][ch][0], (v >> (8 * ch)) & 255u, 1u);
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) hist16_medians(L.S, &L.whist[wave][0][ch][0], &L.whist[wave][1][ch][0], cell, ch, n_p, n_f);
    }
}

template <bool I64, int NT>
__device__ __forceinline__ void tile_cell_medians32(TileStats &S, uint32_t (*hist)[256], const RecMap &M, const BevArgs &a, int cell,
                                                    uint32_t r_lo, uint32_t r_hi);

// When the previous call queued no heavy tile the host launches no bev_tile_cells_heavy (an empty launch is 4-6 us of
// every step on uniform data).  Should tiles be queued nonetheless -- the first dense tile of a sequence -- the LAST
// workgroup of the light kernel to finish works them off, slowly but exactly: statistics in one pass, then 32-bit
// histograms cell by cell.  It also tells the host, so that the next call launches the heavy kernel.
template <bool I64>
__device__ __forceinline__ void cells_drain(const BevArgs &a, TileLds &L, unsigned char *s_buf, bool pushed)
{
    TileStats &S = L.S;
    if (a.heavy_launched) return;
    __shared__ uint32_t s_drain_n;
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        if (pushed) __threadfence();                        // this workgroup's queue entry, before it counts as done
        if (__hip_atomic_fetch_add(&a.heavy[HQ_DONE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __threadfence();
            for (int c = 0; c < HQ_CLASSES; ++c) n += __hip_atomic_load(&a.heavy[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.heavy_hint && n != a.heavy_hint_known) *a.heavy_hint = n;
        }
        s_drain_n = n;
    }
    __syncthreads();
    if (s_drain_n == 0) return;
    __threadfence();
    const bool extra = a.extra != nullptr;
    const bool div255 = !I64 && a.prm.intensity_div255;
    RecMap &M = *reinterpret_cast<RecMap *>(s_buf);
    uint32_t (*hist)[256] = reinterpret_cast<uint32_t (*)[256]>(s_buf + ((sizeof(RecMap) + 15) & ~(size_t)15));
    static_assert(((sizeof(RecMap) + 15) & ~(size_t)15) + 3 * 256 * 4 <= RGB_CAP * 4, "RecMap + histograms live in the colour buffer");
    for (int cls = 0; cls < HQ_CLASSES; ++cls) {
        const uint32_t n_cls = __hip_atomic_load(&a.heavy[cls], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t k = 0; k < n_cls; ++k) {
            const int tile = (int)__hip_atomic_load(&a.heavy[HQ_IDS + (size_t)cls * a.T + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();                                // the staging area of the tile before
            stats_init(S, C_THREADS);
            const uint32_t r_hi = recmap_build(M, a, tile, C_THREADS);
            Run run;
            run_reset(run, RUN_NONE);
            const uint32_t per_thread = (r_hi + C_THREADS - 1) / C_THREADS;
            const uint32_t t_lo = threadIdx.x * per_thread, t_hi = t_lo + per_thread < r_hi ? t_lo + per_thread : r_hi;
            for (uint32_t r = t_lo; r < t_hi; ++r) {
                uint32_t kk, c;
                double z, iv;
                load_rec<I64>(a, recmap_at(M, a.Gr, r), kk, c, z, iv);
                if (kk != run.key) {
                    if (run.key != RUN_NONE) run_flush(S, extra, run);
                    run_reset(run, kk);
                }
                run_add(run, extra, c, z, iv, div255);
            }
            if (run.key != RUN_NONE) run_flush(S, extra, run);
            __syncthreads();
            // medians: four cells (eight fine keys) per pass over the tile's records, 16-bit packed histograms in the
            // light path's per-wave histogram area; a (cell,set) beyond 65 535 values takes the 32-bit path on its own
            uint32_t *h16 = &L.whist[0][0][0][0];           // [8 keys][3 channels][128 dwords]
            for (int c0 = 0; c0 < TCELLS; c0 += 4) {
                uint32_t any = 0;
                for (int c = c0; c < c0 + 4; ++c) any += S.cnt[2 * c] + S.cnt[2 * c + 1];
                if (any == 0) continue;                     // uniform: LDS values
                for (int k = threadIdx.x; k < 8 * 3 * 128 / 4; k += C_THREADS) reinterpret_cast<uint4 *>(h16)[k] = make_uint4(0, 0, 0, 0);
                __syncthreads();
                for (uint32_t r = threadIdx.x; r < r_hi; r += C_THREADS) {
                    uint32_t kk, v;
                    load_rec_key_colour<I64>(a, recmap_at(M, a.Gr, r), kk, v);
                    const uint32_t rel = kk - 2u * (uint32_t)c0;
                    if (rel >= 8u) continue;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) hist16_add(h16 + (rel * 3 + ch) * 128, (v >> (8 * ch)) & 255u, 1u);
                }
                __syncthreads();
                const int wave = threadIdx.x >> 6;
                for (int job = wave; job < 4 * 3; job += C_THREADS / 64) {
                    const int cl = job / 3, ch = job % 3, cell = c0 + cl;
                    const uint32_t n_p = S.cnt[2 * cell], n_f = S.cnt[2 * cell + 1];
                    if (n_p + n_f == 0 || n_p > 0xffffu || n_f > 0xffffu) continue;
                    hist16_medians(S, h16 + ((2 * cl) * 3 + ch) * 128, h16 + ((2 * cl + 1) * 3 + ch) * 128, cell, ch, n_p, n_f);
                }
                __syncthreads();
                for (int cell = c0; cell < c0 + 4; ++cell)
                    if (S.cnt[2 * cell] > 0xffffu || S.cnt[2 * cell + 1] > 0xffffu)
                        tile_cell_medians32<I64, C_THREADS>(S, hist, M, a, cell, 0u, r_hi);
            }
            __syncthreads();
            tile_finalize_write(a, S, reinterpret_cast<double(*)[TCELLS]>(s_buf), tile, C_THREADS);
        }
    }
}

#define DBG_STAMP(bit, slot) do { if ((a.dbg & (bit)) && threadIdx.x == 0 && tile < 1024) g_dbg_stamps[tile][slot] = wall_clock64(); } while (0)
template <bool I64>
__device__ __forceinline__ void bev_tile_cells_body(const BevArgs &a)
{
    __shared__ TileLds L;
    __shared__ __align__(16) unsigned char s_buf[RGB_CAP * 4 > OUT_STAGE_BYTES ? RGB_CAP * 4 : OUT_STAGE_BYTES];
    uint32_t *s_rgb = reinterpret_cast<uint32_t *>(s_buf);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool extra = a.extra != nullptr;
    // Dense tiles are neighbours in the grid (they follow the driven path), and all tiles are resident at once (four per CU at
    // 256 x 256): the kernel lasts as long as the CU with the most records.  Workgroups b, b + T/4, b + T/2, b + 3T/4 share a CU
    // (round-robin dispatch); they take tiles a quarter of the grid apart in BOTH directions -- a band of rows each, the
    // columns rotated by a quarter per band -- so that a CU samples the density at four spread-out places instead of four
    // tiles of one column (the CU totals then ranged 3 800..8 400 records, the kernel's span 30 us against a mean tile life of
    // 23).  Inside a band, workgroup i takes place i * tile_mult mod T/4 (tile_mult coprime to T/4), which interleaves dense
    // and empty tiles in dispatch order.  Grids whose tile count per side is not a multiple of four: one permutation of all.
    int tile;
    if ((a.tx & 3) == 0) {
        const int Tq = a.T >> 2, bw = a.tx >> 2;
        const int q = (int)blockIdx.x / Tq, i = (int)blockIdx.x - q * Tq;
        const int i2 = (int)(((int64_t)i * a.tile_mult) % Tq);
        const int row = q * bw + i2 / a.tx, col = (i2 % a.tx + q * bw) % a.tx;
        tile = row * a.tx + col;
    } else {
        tile = (int)(((int64_t)blockIdx.x * a.tile_mult) % a.T);
    }
    // PCA_BEV_STAGGER = d (default 0: off): the four workgroups of a CU (blocks b, b + T/4, ...) start d x 0.43 us apart.  Left
    // alone they run in lockstep on the uniform window: all 1024 tiles are resident from the first microsecond, every one asks
    // for its counters and then its records at the same moment (44 MB in ~3 us) and computes while the memory system idles.
    // Measured (profiles/r04_experiments/bev_cells_stagger.txt): steps of 0.9-1.7 us take the kernel from 34.5 to 33.0 us on the
    // headline although a CU's last workgroup starts 4-5 us late -- but cost the ring model 1.4 us (its tiles are unequal and out
    // of step anyway) and the many-sample launch 7 % (every workgroup of every round sleeps): a knob for A/B, not a default.
    if (a.stagger > 0) {
        const int slot = (int)blockIdx.x / ((a.T + 3) >> 2);
        for (int k = 0; k < slot * a.stagger; ++k) __builtin_amdgcn_s_sleep(16);
    }
    const unsigned long long t_begin = wall_clock64();
    RecMap &M = *reinterpret_cast<RecMap *>(s_buf);         // lives in the colour buffer until pass 2 fills that
    static_assert(sizeof(RecMap) <= sizeof(s_buf), "RecMap aliases the colour buffer");
    stats_init(L.S, C_THREADS);
    const uint32_t r_lo = 0, r_hi = recmap_build(M, a, tile, C_THREADS, L.owner);
    if (r_hi > (uint32_t)a.heavy_min) {                     // bev_tile_cells_heavy's
        if (threadIdx.x == 0) heavy_push(a, tile, r_hi);
        cells_drain<I64>(a, L, s_buf, true);
        return;
    }

    // ---- pass 1: per (cell,set) statistics; every thread keeps its <= RPT records' (key, rank, colour) in registers
    // -- the rank is the run's base (returned by the counting atomic) + the position in the run -- so that pass 2 is
    // a pure LDS scatter.  Sparse tiles are read coalesced (record u*C_THREADS + t); dense ones give every thread a
    // contiguous chunk so that runs form (16 records = two cache lines per lane).
    constexpr int RPT = RGB_CAP / C_THREADS;
    constexpr int HALF = RPT / 2;                           // two rounds of loads: bounds the registers in flight
    const bool contig = (r_hi - r_lo) > CONTIG_MIN;
    const bool div255 = !I64 && a.prm.intensity_div255;
    uint32_t kr[RPT], cc[RPT];
    // the HALF records of one round of thread-record numbers rec(u): all loads first (see recf_raw), then the fields
    auto load_round = [&](auto rec, uint32_t (&k)[HALF], uint32_t (&c)[HALF], double (&z)[HALF], double (&iv)[HALF]) {
        if constexpr (I64) {
#pragma unroll
            for (int u = 0; u < HALF; ++u) {
                const uint32_t r = rec(u);
                k[u] = RUN_NONE; c[u] = 0; z[u] = 0; iv[u] = 0;
                if (r < r_hi) load_rec<I64>(a, recmap_at(M, L.owner, r), k[u], c[u], z[u], iv[u]);
            }
        } else {
            uint4 w[HALF];
#pragma unroll
            for (int u = 0; u < HALF; ++u) {
                const uint32_t r = rec(u);
                w[u] = recf_raw(a, recmap_at(M, L.owner, r < r_hi ? r : r_lo));      // (r_lo exists: the tile is not empty)
            }
#pragma unroll
            for (int u = 0; u < HALF; ++u) {
                recf_fields(w[u], k[u], c[u], z[u], iv[u]);
                if (!(rec(u) < r_hi)) { k[u] = RUN_NONE; c[u] = 0; }
            }
        }
    };
    if (r_hi == r_lo) {                                     // (uniform) an empty tile: nothing to load, nothing to count
#pragma unroll
        for (int uu = 0; uu < RPT; ++uu) { kr[uu] = RUN_NONE; cc[uu] = 0; }
    } else if (!contig) {
        // records a workgroup-width apart share a cell only by chance: every record is counted on its own
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t k[HALF], c[HALF];
            double z[HALF], iv[HALF];
            load_round([&](int u) { return r_lo + (uint32_t)(h * HALF + u) * C_THREADS + threadIdx.x; }, k, c, z, iv);
#pragma unroll
            for (int u = 0; u < HALF; ++u) {
                const int uu = h * HALF + u;
                kr[uu] = RUN_NONE; cc[uu] = c[u] & 0xffffffu;
                if (k[u] != RUN_NONE) kr[uu] = k[u] | (rec_add(L.S, extra, k[u], c[u], z[u], iv[u], div255) << 8);
            }
        }
    } else {
        uint32_t end_base[RPT];
        uint32_t is_end = 0;
        Run run;
        run_reset(run, RUN_NONE);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t k[HALF], c[HALF];
            double z[HALF], iv[HALF];
            load_round([&](int u) { return r_lo + threadIdx.x * RPT + (uint32_t)(h * HALF + u); }, k, c, z, iv);
#pragma unroll
            for (int u = 0; u < HALF; ++u) {
                const int uu = h * HALF + u;
                kr[uu] = RUN_NONE; cc[uu] = c[u] & 0xffffffu; end_base[uu] = 0;
                if (k[u] == RUN_NONE) continue;
                if (k[u] != run.key) {
                    if (run.key != RUN_NONE && uu > 0) { end_base[uu - 1] = run_flush(L.S, extra, run); is_end |= 1u << (uu - 1); }
                    run_reset(run, k[u]);
                }
                kr[uu] = k[u] | (run.cnt << 8);             // position in the run, for now
                run_add(run, extra, c[u], z[u], iv[u], div255);
            }
        }
        uint32_t b = run.key != RUN_NONE ? run_flush(L.S, extra, run) : 0u;
#pragma unroll
        for (int uu = RPT - 1; uu >= 0; --uu) {
            if ((is_end >> uu) & 1u) b = end_base[uu];
            if (kr[uu] != RUN_NONE) kr[uu] += b << 8;
        }
    }
    __syncthreads();
    DBG_STAMP(16, 3);
    // ---- offsets of the (cell,set) segments inside the tile ----
    if (wave == 0) {
        const uint32_t c0 = L.S.cnt[2 * lane], c1 = L.S.cnt[2 * lane + 1];
        const uint32_t inc = wave_incl_scan_add(c0 + c1);
        const uint32_t excl = inc - (c0 + c1);
        L.off[2 * lane] = excl;
        L.off[2 * lane + 1] = excl + c0;
        if (lane == 63) L.off[NFK] = inc;
        const unsigned long long big = __ballot(c0 + c1 > 64);  // (lane = cell)
        if (lane == 0) L.n_big = (uint32_t)__popcll(big);
    }
    __syncthreads();
    // ---- pass 2: colours sorted by (cell,set) in LDS, then the medians ----
#pragma unroll
    for (int u = 0; u < RPT; ++u)
        if (kr[u] != 0xffffffffu) s_rgb[L.off[kr[u] & 127u] + (kr[u] >> 8)] = cc[u];
    __syncthreads();
    DBG_STAMP(16, 4);
    small_cells_bitplanes(L, s_rgb);
    __syncthreads();
    small_cells_select(L);
    __syncthreads();                                        // the bit planes' LDS becomes the per-wave histograms
    DBG_STAMP(16, 5);
    if (L.n_big) {                                          // (uniform; no cell above 64 values: nothing to do, no barrier)
        wave_cells_hist(L, s_rgb);
        __syncthreads();
    }
    DBG_STAMP(16, 6);
    tile_finalize_write(a, L.S, reinterpret_cast<double(*)[TCELLS]>(s_buf), tile, C_THREADS);
    cells_drain<I64>(a, L, s_buf, false);
    if ((a.dbg & 16) && threadIdx.x == 0 && tile < 1024) {
        g_dbg_stamps[tile][0] = t_begin; g_dbg_stamps[tile][1] = wall_clock64(); g_dbg_stamps[tile][2] = r_hi - r_lo;
        uint32_t big = 0;
        for (int c = 0; c < TCELLS; ++c) big += (L.S.cnt[2 * c] + L.S.cnt[2 * c + 1]) > 64;
        g_dbg_stamps[tile][7] = __smid() | ((unsigned long long)big << 32);
    }
}

// ---- heavy tiles: 1024 threads, 16-bit-packed 256-bin histograms per (cell, set, channel) for half of the
// tile's cells at a time, filled straight from the record stream (two passes, no sort) ------------------------
#define H_THREADS 1024
#define H_CELLS 32                                          // cells per pass
#define H_HIST_DWORDS (H_CELLS * 2 * 3 * 128)               // 96 KiB
struct HeavyLds {
    TileStats S;
    RecMap M;
    uint32_t hist[3][256];                                  // 32-bit fallback for a (cell,set) of 65 536 values or more
    uint32_t overflow;                                      // some (cell,set) of this tile needs it
};
#define HEAVY_LDS_BYTES (H_HIST_DWORDS * 4 + sizeof(HeavyLds))

// 32-bit 256-bin histograms hist[3][256] of the colours of one fine key (cell,set), whole workgroup of NT threads
template <bool I64, int NT>
__device__ __forceinline__ void tile_hist32(uint32_t (*hist)[256], const RecMap &M, const BevArgs &a, uint32_t fk, uint32_t r_lo,
                                            uint32_t r_hi)
{
    const int lane = threadIdx.x & 63;
    for (uint32_t r0 = r_lo; r0 < r_hi; r0 += NT) {
        const uint32_t r = r0 + threadIdx.x;
        bool act = r < r_hi;
        uint32_t v = 0;
        if (act) {
            uint32_t k;
            load_rec_key_colour<I64>(a, recmap_at(M, a.Gr, r), k, v);
            act = k == fk;
        }
        const uint32_t tag = act ? v : 0xffffffffu;
        const uint32_t first = __builtin_amdgcn_readfirstlane(tag);
        const uint64_t actm = __ballot(act);
        if (actm == 0) continue;
        if (__ballot(tag == first) == actm && first != 0xffffffffu) {
            if (lane == (int)__ffsll((unsigned long long)actm) - 1) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) atomicAdd(&hist[ch][(v >> (8 * ch)) & 255u], (uint32_t)__popcll(actm));
            }
        } else if (act) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) atomicAdd(&hist[ch][(v >> (8 * ch)) & 255u], 1u);
        }
    }
}
// present / future / full medians of one cell from 32-bit histograms (three passes over the tile's records)
template <bool I64, int NT>
__device__ __forceinline__ void tile_cell_medians32(TileStats &S, uint32_t (*hist)[256], const RecMap &M, const BevArgs &a, int cell,
                                                    uint32_t r_lo, uint32_t r_hi)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n_p = S.cnt[2 * cell], n_f = S.cnt[2 * cell + 1];
    for (int round = 0; round < 2; ++round) {               // round 0: present, then + future = full; round 1: future
        for (int i = threadIdx.x; i < 3 * 256; i += NT) (&hist[0][0])[i] = 0;
        __syncthreads();
        if (round == 0) {
            tile_hist32<I64, NT>(hist, M, a, 2 * cell, r_lo, r_hi);
            __syncthreads();
            if (wave < 3) { const uint32_t m = hist_med2(hist[wave], n_p); if (lane == 0) S.med2[0][cell][wave] = m; }
            __syncthreads();
        }
        tile_hist32<I64, NT>(hist, M, a, 2 * cell + 1, r_lo, r_hi);
        __syncthreads();
        if (wave < 3) {
            const uint32_t m = hist_med2(hist[wave], round == 0 ? n_p + n_f : n_f);
            if (lane == 0) S.med2[round == 0 ? 2 : 1][cell][wave] = m;
        }
        __syncthreads();
    }
}

// EXTRA (the opt-in reducers) and OWN (an item walks its own half's records: a.split) are template parameters: the kernel
// sits at the 128-VGPR limit of a 1024-thread workgroup and spilled 4-9 registers with them as run-time flags.
template <bool I64, bool EXTRA, bool OWN>
__device__ __forceinline__ void bev_tile_cells_heavy_body(const BevArgs &a)
{
    unsigned long long t_begin = wall_clock64();
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);                    // [H_CELLS][2][3][128]
    HeavyLds &L = *reinterpret_cast<HeavyLds *>(smem + (size_t)H_HIST_DWORDS * 4);
    __shared__ uint32_t s_next;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr bool extra = EXTRA;
    const bool div255 = !I64 && a.prm.intensity_div255;
    // the queue: HQ_CLASSES lists of tile ids, larger tiles first; item -> (class, place) by the classes' running counts
    __shared__ uint32_t s_cls[HQ_CLASSES + 1];
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int c = 0; c < HQ_CLASSES; ++c) { s_cls[c] = run; run += a.heavy[c]; }
        s_cls[HQ_CLASSES] = run;
    }
    __syncthreads();
    const uint32_t n_heavy = s_cls[HQ_CLASSES];
    // the count goes to host-visible memory only when it differs from what the host already knows (a write over PCIe
    // holds the kernel's end back by microseconds; in steady state nothing changes)
    if (a.heavy_hint && blockIdx.x == 0 && threadIdx.x == 0 && n_heavy != a.heavy_hint_known) *a.heavy_hint = n_heavy;
    // Work items are (tile, half of its cells): the two halves of a tile are independent (every statistic is per cell),
    // so a 12 000-record tile is two items of half the time each, drawn one at a time by whichever workgroup is free --
    // the kernel ends with its slowest ITEM, and most CUs would otherwise idle behind the few densest tiles.
    constexpr int HALVES = TCELLS / H_CELLS;
    for (;;) {
    if (threadIdx.x == 0) s_next = atomicAdd(&a.heavy[HQ_CURSOR], 1u);
    __syncthreads();
    const uint32_t item = s_next;
    if (item >= n_heavy * HALVES) break;
    int tile;
    {
        const uint32_t ti = item / HALVES;
        int cls = 0;
        while (cls + 1 < HQ_CLASSES && s_cls[cls + 1] <= ti) ++cls;
        tile = (int)a.heavy[HQ_IDS + (size_t)cls * a.T + (ti - s_cls[cls])];
    }
    const int half = (int)(item % HALVES);
    if (item >= (uint32_t)gridDim.x) t_begin = wall_clock64();
    stats_init(L.S, H_THREADS);
    // (split: the item's own records only -- round 4 let both items of a tile stream ALL its records and drop the other half's:
    // 2 x 16 B read per record of a heavy tile, 3.2 GB for 1.6 GB of records on BASELINE config 4)
    bool own = false;                                       // (uniform) OWN: compiled for it; in effect if this window's pieces are ordered
    if (OWN) {
        const int64_t wn = window_end(a) - window_begin(a);                // (what level 1 decided on)
        own = bev_split(a, wn > a.max_points ? a.max_points : wn);
    }
    const uint32_t r_lo = 0, r_hi = recmap_build(L.M, a, tile, H_THREADS, nullptr, OWN ? half : -1);
    if (threadIdx.x == 0) L.overflow = 0;
    {
        for (int k = threadIdx.x; k < H_HIST_DWORDS / 4; k += H_THREADS) reinterpret_cast<uint4 *>(hist)[k] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        // Every thread takes UNR consecutive records at a time, the wave 64 * UNR consecutive ones: a wave's loads cover whole
        // cache lines (one contiguous chunk per THREAD made every load instruction touch 64 lines of which the L1 kept none
        // until the thread's next visit); runs of up to UNR records still form (see Run).
        constexpr int UNR = 4;
        Run run;
        run_reset(run, RUN_NONE);
        for (uint32_t r0 = r_lo + threadIdx.x * UNR; r0 < r_hi; r0 += H_THREADS * UNR) {
            uint32_t k[UNR], c[UNR];
            double z[UNR], iv[UNR];
            RecWalk walk;
            recwalk_seek(walk, L.M, a.Gr, r0);
            if constexpr (I64) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    k[u] = RUN_NONE; c[u] = 0; z[u] = 0; iv[u] = 0;
                    if (r0 + u < r_hi) load_rec<I64>(a, recwalk_next(walk, L.M, a.Gr, r_hi), k[u], c[u], z[u], iv[u]);
                }
            } else {                                        // places first, then every load by every lane (see recf_raw)
                uint32_t at[UNR];
                uint4 w[UNR];
                const uint32_t at0 = walk.at;
#pragma unroll
                for (int u = 0; u < UNR; ++u) at[u] = r0 + u < r_hi ? recwalk_next(walk, L.M, a.Gr, r_hi) : at0;
#pragma unroll
                for (int u = 0; u < UNR; ++u) w[u] = recf_raw(a, at[u]);
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    recf_fields(w[u], k[u], c[u], z[u], iv[u]);
                    if (!(r0 + u < r_hi)) { k[u] = RUN_NONE; c[u] = 0; }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (k[u] == RUN_NONE || (!own && (int)(k[u] >> 1) / H_CELLS != half)) continue;
                if (k[u] != run.key) {
                    if (run.key != RUN_NONE) run_flush(L.S, extra, run);
                    run_reset(run, k[u]);
                }
                run_add(run, extra, c[u], z[u], iv[u], div255);
                uint32_t *row = hist + ((((k[u] >> 1) % H_CELLS) * 2 + (k[u] & 1u)) * 3) * 128;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) hist16_add(row + ch * 128, (c[u] >> (8 * ch)) & 255u, 1u);
            }
        }
        if (run.key != RUN_NONE) run_flush(L.S, extra, run);
        __syncthreads();
        if ((a.dbg & 8) && threadIdx.x == 0 && tile < 1024) g_dbg_stamps[tile][3 + 2 * half] = wall_clock64();
        for (int job = wave; job < H_CELLS * 3; job += H_THREADS / 64) {
            const int cl = job / 3, ch = job % 3, cell = half * H_CELLS + cl;
            const uint32_t n_p = L.S.cnt[2 * cell], n_f = L.S.cnt[2 * cell + 1];
            if (n_p > 0xffffu || n_f > 0xffffu) { if (lane == 0) L.overflow = 1; continue; }
            if (n_p + n_f == 0) continue;
            hist16_medians(L.S, hist + ((cl * 2 + 0) * 3 + ch) * 128, hist + ((cl * 2 + 1) * 3 + ch) * 128, cell, ch, n_p, n_f);
        }
        __syncthreads();
        if ((a.dbg & 8) && threadIdx.x == 0 && tile < 1024) g_dbg_stamps[tile][4 + 2 * half] = wall_clock64();
    }
    // (cell,set)s too large for 16-bit counters: 32-bit histograms, whole workgroup, re-reading the tile
    for (int cell = half * H_CELLS; L.overflow && cell < (half + 1) * H_CELLS; ++cell) {
        const uint32_t n_p = L.S.cnt[2 * cell], n_f = L.S.cnt[2 * cell + 1];
        if (n_p <= 0xffffu && n_f <= 0xffffu) continue;
        tile_cell_medians32<I64, H_THREADS>(L.S, L.hist, L.M, a, cell, r_lo, r_hi);
    }
    __syncthreads();
    tile_finalize_write(a, L.S, reinterpret_cast<double(*)[TCELLS]>(smem), tile, H_THREADS, half * H_CELLS, (half + 1) * H_CELLS);
    __syncthreads();                                        // the staging area is the next item's histogram
    if ((a.dbg & 8) && threadIdx.x == 0 && tile < 1024) {
        g_dbg_stamps[tile][0] = t_begin; g_dbg_stamps[tile][1] = wall_clock64(); g_dbg_stamps[tile][2] = r_hi - r_lo;
        g_dbg_stamps[tile][7] = __smid();
    }
    }
}

// ---- level 1 with the K1 of the newest frame in the same launch (pca_kitti_integrate deferred it: pca_common.h, K1Pending) ----
// Workgroups [0, Gk) are K1's tiles (k1_body, FUSED form, 1024 x 4): project / sample / filter / compact, look-back among
// themselves, append to the store, close the frame -- and then bin the points they kept straight out of their registers: same
// key, same record as level 1 would make of them after reading them back (f32 -> f64 is exact; the newest frame owes no
// re-transform; dyn = 0), into piece g = the workgroup's number.  Workgroups [Gk, G) are level 1 over the window up to that frame.
struct BinK1Tail {
    const BevArgs &a;
    template <int PPT>
    __device__ __forceinline__ void operator()(const float4 (&p)[PPT], const uint32_t (&packed)[PPT], const uint64_t (&km)[PPT]) const
    {
        extern __shared__ uint32_t s_lds[];
        __shared__ uint32_t s_wsum[AB_THREADS / 64];
        __shared__ uint32_t s_dyn[8];
        const int64_t lo = window_begin(a), hi0 = window_end(a);
        const bool split = bev_split(a, hi0 - lo > a.max_points ? a.max_points : hi0 - lo);
        const int hs = split ? 6 : 7, n_hist = split ? 2 * a.T : a.T;
        uint32_t *s_h = s_lds, *s_cur = s_lds + n_hist;
        if (threadIdx.x < 8) {
            uint32_t w = (uint32_t)a.prm.dynobj_mask[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) w = (int)threadIdx.x == i ? (uint32_t)(a.prm.dynobj_mask[i >> 1] >> (32 * (i & 1))) : w;
            s_dyn[threadIdx.x] = w;
        }
        for (int t = threadIdx.x; t < n_hist; t += AB_THREADS) s_h[t] = 0;
        __syncthreads();
        const ViewConst vc = view_const(a);
        const uint32_t set = a.k1_slot >= a.slot_split ? 1u : 0u;
        const int lane = threadIdx.x & 63;
        uint32_t rkey[PPT];
#pragma unroll
        for (int r = 0; r < PPT; ++r) {
            const bool kept = (km[r] >> lane) & 1ull;
            rkey[r] = view_key_lean(vc, (double)p[r].x, (double)p[r].y, (double)p[r].z, kept, set);
            if (rkey[r] != KEY_INVALID) atomicAdd(&s_h[rkey[r] >> hs], 1u);
        }
        __syncthreads();
        bin_scan_tables(a, s_h, s_cur, s_wsum, split, (int)blockIdx.x);
        __syncthreads();
        const uint32_t seg = (uint32_t)blockIdx.x * K1_SEG;
#pragma unroll
        for (int r = 0; r < PPT; ++r) {
            if (rkey[r] == KEY_INVALID) continue;
            const uint32_t pos = seg + atomicAdd(&s_cur[rkey[r] >> hs], 1u);
            bin_store<false>(a, s_dyn, pos, rkey[r], packed[r], (double)p[r].z, 0.0, p[r].w);
        }
    }
};
static_assert(K1_SEG == 1024 * 4, "a K1 piece holds one 1024 x 4 tile");
__global__ __launch_bounds__(AB_THREADS) __attribute__((amdgpu_waves_per_eu(BIN_WPE, BIN_WPE))) void bev_tile_bin_k1(const BevArgs a, const K1Args k)
{
    if ((int)blockIdx.x < a.Gk) k1_body<AB_THREADS, 4, false, false, false>(k, BinK1Tail{a});
    else bev_tile_bin_body<false>(a);
}

// ---- the kernels: one sample per launch (arguments in the kernel-argument segment), or many (pca_bev_generate_many:
// blockIdx.y = sample, its arguments read from a device array -- a NuScenes-size window is 108 level-1 workgroups and
// three dependent launches of pure latency; S samples in one launch of each kernel take little longer than one) ----
template <bool I64>
__global__ __launch_bounds__(AB_THREADS) __attribute__((amdgpu_waves_per_eu(BIN_WPE, BIN_WPE))) void bev_tile_bin(const BevArgs a) { bev_tile_bin_body<I64>(a); }
#if C_THREADS == 512
#define C_OCC __attribute__((amdgpu_waves_per_eu(6, 6)))
#else
#define C_OCC
#endif
template <bool I64>
__global__ __launch_bounds__(C_THREADS) C_OCC void bev_tile_cells(const BevArgs a) { bev_tile_cells_body<I64>(a); }
template <bool I64, bool EXTRA, bool OWN>
__global__ __launch_bounds__(H_THREADS) void bev_tile_cells_heavy(const BevArgs a) { bev_tile_cells_heavy_body<I64, EXTRA, OWN>(a); }
// (the samples' argument blocks live in CONSTANT memory: the kernels read them through the scalar cache exactly as they read
// kernel arguments.  Copies out of a global array ended up in scratch -- the blocks are indexed dynamically -- and made
// level 1 three times slower than a launch per sample.)
#define PCA_BEV_MANY_MAX 48
__constant__ BevArgs g_bev_many[PCA_BEV_MANY_MAX];
template <bool I64>
__global__ __launch_bounds__(AB_THREADS) __attribute__((amdgpu_waves_per_eu(BIN_WPE, BIN_WPE))) void bev_tile_bin_many()
{
    const BevArgs &a = g_bev_many[blockIdx.y];
    if ((int)blockIdx.x < a.G) bev_tile_bin_body<I64>(a);
}
template <bool I64>
__global__ __launch_bounds__(C_THREADS) C_OCC void bev_tile_cells_many()
{
    bev_tile_cells_body<I64>(g_bev_many[blockIdx.y]);
}
template <bool I64>
__global__ __launch_bounds__(H_THREADS) void bev_tile_cells_heavy_many()
{
    bev_tile_cells_heavy_body<I64, false, false>(g_bev_many[blockIdx.y]);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
int pca_k1_prepare_pending(pca_ctx *ctx, K1Args *out, int *n_tiles, hipStream_t s);    // pca_k1.hip
void pca_k1_pending_launched(pca_ctx *ctx, hipStream_t s);
static inline int64_t align256(int64_t v) { return (v + 255) & ~255ll; }
static inline int tiles_x(int px) { return (px + TS - 1) / TS; }
// PCA_BEV_G / PCA_BEV_CHUNK: tuning overrides, read on every call (tests switch them to force the memory path)
static inline int max_groups()
{
    const char *e = getenv("PCA_BEV_G");
    const int max_g = e ? atoi(e) : MAX_G;
    return (max_g < 1 || max_g > 1024) ? MAX_G : max_g;
}
static inline int n_groups(int64_t max_points)
{
    const char *c = getenv("PCA_BEV_CHUNK");
    const int max_g = max_groups();
    int per_g = c ? atoi(c) : 8192;
    if (per_g < 1024) per_g = 8192;
    int64_t g = (max_points + per_g - 1) / per_g;
    return (int)(g < 1 ? 1 : (g > max_g ? max_g : g));
}

extern "C" {

int64_t pca_bev_workspace_bytes(int64_t max_points, int px)
{
    if (max_points < 1) max_points = 1;
    const int64_t T = (int64_t)tiles_x(px) * tiles_x(px), G = n_groups(max_points);
    return align256(max_points * 4) + 3 * align256((G + K1_RIDE + 8) * T * 4) + align256((HQ_IDS + HQ_CLASSES * T) * 4) +
           align256((max_points + G + K1_RIDE + K1_SEG) * 24) + 512;
}

int pca_bev_generate(pca_ctx *ctx, const pca_store *store, const double *intensity64, const int64_t *frame_off,
                     int slot_begin, int slot_split, int slot_end, int64_t max_points, const pca_bev_params *prm,
                     const double *pending_T, int pending_slot_end, void *workspace, int64_t workspace_bytes,
                     double *planes, uint16_t *planes_f16, void *stream)
{
    return pca_bev_generate_ex(ctx, store, intensity64, frame_off, slot_begin, slot_split, slot_end, max_points, prm,
                               pending_T, pending_slot_end, workspace, workspace_bytes, planes, planes_f16, nullptr, stream);
}

int pca_bev_generate_ex(pca_ctx *ctx, const pca_store *store, const double *intensity64, const int64_t *frame_off,
                        int slot_begin, int slot_split, int slot_end, int64_t max_points, const pca_bev_params *prm,
                        const double *pending_T, int pending_slot_end, void *workspace, int64_t workspace_bytes,
                        double *planes, uint16_t *planes_f16, double *extra_planes, void *stream)
{
    const int n = (pending_T && pending_slot_end > slot_begin) ? 1 : 0;
    return pca_bev_generate_chain(ctx, store, intensity64, frame_off, slot_begin, slot_split, slot_end, max_points, prm,
                                  pending_T, &pending_slot_end, n, 1, workspace, workspace_bytes, planes, planes_f16,
                                  extra_planes, stream);
}

// Checks one raster's arguments and fills the kernels' argument block (everything but the heavy-tile bookkeeping)
static void bev_table_order(BevArgs &a)
{   // PCA_BEV_XCD_TABLES=0: the counter tables in plain [tile][workgroup] order (A/B)
    static int xt = -1;
    if (xt < 0) { const char *e = getenv("PCA_BEV_XCD_TABLES"); xt = e ? atoi(e) : 1; }
    a.Gp = (xt && a.G >= 16 && a.G <= 1016) ? (a.G + 7) / 8 : 0;   // (RecMap holds 1024 places)
    a.Gr = a.Gp ? 8 * a.Gp : a.G;
}
static int bev_prepare(pca_ctx *ctx, const pca_store *store, const double *intensity64, const int64_t *frame_off,
                       int slot_begin, int slot_split, int slot_end, int64_t max_points, const pca_bev_params *prm,
                       const double *pending_Ts, const int *pending_slot_ends, int n_pending, int write_back,
                       void *workspace, int64_t workspace_bytes, double *planes, uint16_t *planes_f16,
                       double *extra_planes, BevArgs &a)
{
    if (n_pending < 0 || n_pending > PCA_BEV_MAX_CHAIN || (n_pending > 0 && (!pending_Ts || !pending_slot_ends))) {
        ctx->err = "bev: bad chain of owed transforms";
        return -1;
    }
    if (!store || !frame_off || !prm || !workspace || (!planes && !planes_f16)) { ctx->err = "bev: bad arguments"; return -1; }
    if (prm->px < 1 || prm->px > 1024) { ctx->err = "bev: px must be in 1..1024"; return -1; }
    // the elevation plane is min(z - origin_z): the rotation has to leave z alone (rotation_matrix_3d of the reference)
    if (!(prm->R[6] == 0.0 && prm->R[7] == 0.0 && prm->R[8] == 1.0 && prm->R[2] == 0.0 && prm->R[5] == 0.0)) {
        ctx->err = "bev: R must be a rotation about the z axis (R[2] = R[5] = R[6] = R[7] = 0, R[8] = 1)";
        return -1;
    }
    if (!(slot_begin <= slot_split && slot_split <= slot_end)) { ctx->err = "bev: need slot_begin <= slot_split <= slot_end"; return -1; }
    if (max_points >= (1ll << 32)) { ctx->err = "bev: window too large for 32-bit positions"; return -1; }
    if (workspace_bytes < pca_bev_workspace_bytes(max_points, prm->px)) { ctx->err = "bev: workspace too small"; return -1; }
    a.st = *store;
    a.intensity64 = intensity64;
    a.frame_off = frame_off;
    a.slot_begin = slot_begin; a.slot_split = slot_split; a.slot_end = slot_end;
    a.max_points = max_points;
    a.prm = *prm;
    a.n_pend = n_pending;
    a.write_back = write_back ? 1 : 0;
    for (int k = 0; k < PCA_BEV_MAX_CHAIN; ++k) {
        a.pend_slot_end[k] = slot_begin;
        for (int i = 0; i < 12; ++i) a.pend_T[k].m[i] = 0.0;
        if (k >= n_pending) continue;
        if (pending_slot_ends[k] > slot_end) { ctx->err = "bev: a pending slot end lies beyond the window"; return -1; }
        if (k > 0 && pending_slot_ends[k] < pending_slot_ends[k - 1]) { ctx->err = "bev: pending slot ends must ascend"; return -1; }
        a.pend_slot_end[k] = pending_slot_ends[k];
        for (int i = 0; i < 12; ++i) a.pend_T[k].m[i] = pending_Ts[16 * k + i];
    }
    a.tx = tiles_x(prm->px);
    a.T = a.tx * a.tx;
    a.G = n_groups(max_points);
    bev_table_order(a);
    char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    a.key = reinterpret_cast<uint32_t *>(w); w += align256(max_points * 4);
    a.bh = reinterpret_cast<uint32_t *>(w); w += align256((int64_t)(a.G + K1_RIDE + 8) * a.T * 4);
    a.boff = reinterpret_cast<uint32_t *>(w); w += align256((int64_t)(a.G + K1_RIDE + 8) * a.T * 4);
    a.bh0 = reinterpret_cast<uint32_t *>(w); w += align256((int64_t)(a.G + K1_RIDE + 8) * a.T * 4);
    a.split = 0;
    a.Gk = 0; a.k1_slot = -1; a.k1_n = 0;
    a.bin_first = slot_begin; a.bin_end = slot_end;
    a.heavy = reinterpret_cast<uint32_t *>(w); w += align256((int64_t)(HQ_IDS + HQ_CLASSES * (int64_t)a.T) * 4);
    a.recs = w;
    a.planes = planes;
    a.planes_f16 = planes_f16;
    a.extra = extra_planes;
    { static int hm = -1; if (hm < 0) { const char *e = getenv("PCA_BEV_HEAVY_MIN"); hm = e ? atoi(e) : HEAVY_MIN_DEFAULT; if (hm < 1 || hm > RGB_CAP) hm = RGB_CAP; } a.heavy_min = hm; }
    { static int dbg = -1; if (dbg < 0) { const char *e = getenv("PCA_BEV_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }
    { static int sg = -1; if (sg < 0) { const char *e = getenv("PCA_BEV_STAGGER"); sg = e ? atoi(e) : 0; if (sg < 0 || sg > 64) sg = 0; } a.stagger = sg; }
    a.status = ctx->ticket + 1;
    {
        auto gcd = [](int x, int y) { while (y) { const int t = x % y; x = y; y = t; } return x; };
        const int Tp = (a.tx & 3) == 0 ? a.T / 4 : a.T;     // the permutation's period (bev_tile_cells_body)
        int m = (int)(Tp * 0.6180339887) | 1;
        while (m > 1 && gcd(m, Tp) != 1) m -= 2;
        a.tile_mult = m < 1 ? 1 : m;
    }
    a.heavy_hint = nullptr; a.heavy_hint_known = 0; a.heavy_launched = 1;
    a.span_hint = nullptr; a.span_hint_known = 0;
    return 0;
}

static int bev_set_lds_attributes(pca_ctx *ctx)
{
    static bool lds_set = false;                            // > 64 KiB of dynamic LDS has to be asked for once
    if (lds_set) return 0;
#define PCA_BEV_LDS_ATTR(k, bytes) PCA_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&(k)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))
#define PCA_BEV_HEAVY_ATTR(I) PCA_BEV_LDS_ATTR((bev_tile_cells_heavy<I, false, false>), HEAVY_LDS_BYTES); PCA_BEV_LDS_ATTR((bev_tile_cells_heavy<I, false, true>), HEAVY_LDS_BYTES); \
    PCA_BEV_LDS_ATTR((bev_tile_cells_heavy<I, true, false>), HEAVY_LDS_BYTES); PCA_BEV_LDS_ATTR((bev_tile_cells_heavy<I, true, true>), HEAVY_LDS_BYTES)
    PCA_BEV_HEAVY_ATTR(false);
    PCA_BEV_HEAVY_ATTR(true);
#undef PCA_BEV_HEAVY_ATTR
    PCA_BEV_LDS_ATTR(bev_tile_cells_heavy_many<false>, HEAVY_LDS_BYTES);
    PCA_BEV_LDS_ATTR(bev_tile_cells_heavy_many<true>, HEAVY_LDS_BYTES);
    PCA_BEV_LDS_ATTR(bev_tile_bin_k1, 80 * 1024);
    PCA_BEV_LDS_ATTR(bev_tile_bin<false>, 128 * 1024);
    PCA_BEV_LDS_ATTR(bev_tile_bin<true>, 128 * 1024);
    PCA_BEV_LDS_ATTR(bev_tile_bin_many<false>, 128 * 1024);
    PCA_BEV_LDS_ATTR(bev_tile_bin_many<true>, 128 * 1024);
#undef PCA_BEV_LDS_ATTR
    lds_set = true;
    return 0;
}

int pca_bev_generate_chain(pca_ctx *ctx, const pca_store *store, const double *intensity64, const int64_t *frame_off,
                           int slot_begin, int slot_split, int slot_end, int64_t max_points, const pca_bev_params *prm,
                           const double *pending_Ts, const int *pending_slot_ends, int n_pending, int write_back,
                           void *workspace, int64_t workspace_bytes, double *planes, uint16_t *planes_f16,
                           double *extra_planes, void *stream)
{
    if (!ctx) return -1;
    const bool bin_given = ctx->bin_valid;                  // (pca_bev_bin_range: for this call, whatever becomes of it)
    const int bin_first = ctx->bin_first, bin_end = ctx->bin_end;
    ctx->bin_valid = false;
    if (max_points < 1) max_points = 1;
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    BevArgs a;
    if (bev_prepare(ctx, store, intensity64, frame_off, slot_begin, slot_split, slot_end, max_points, prm, pending_Ts,
                    pending_slot_ends, n_pending, write_back, workspace, workspace_bytes, planes, planes_f16, extra_planes, a))
        return -1;
    if (bev_set_lds_attributes(ctx)) return -1;
    // one resident workgroup per CU draws from the queue -- when the previous call had no heavy tile (uniform data)
    // only a few are launched: any number of them drains the queue, and 256 idle 110-KiB workgroups cost ~5 us
    const int heavy_grid = a.T < ctx->n_cu ? a.T : ctx->n_cu;
    a.heavy_hint = ctx->heavy_hint_dev;
    a.heavy_hint_known = *ctx->heavy_hint;
    // the heavy kernel is launched while heavy tiles were seen in one of the last 64 calls (a tile hovering around the
    // threshold must not pay the light kernel's slow path every other call); PCA_BEV_HEAVY_ALWAYS=1: regardless (A/B)
    if (a.heavy_hint_known != 0) ctx->heavy_cooldown = 64;
    else if (ctx->heavy_cooldown > 0) --ctx->heavy_cooldown;
    a.heavy_launched = a.heavy_hint_known != 0 || ctx->heavy_cooldown > 0;
    { static int always = -1; if (always < 0) { const char *e = getenv("PCA_BEV_HEAVY_ALWAYS"); always = e ? atoi(e) : 0; } if (always) a.heavy_launched = 1; }
    // Pieces ordered by half of the tile when the heavy kernel follows (its items are halves), level 1's LDS holds 2 T histogram
    // entries + 2 T cursors (px <= 720) AND the window stays on level 1's register path.  On the memory path of giant windows
    // (BASELINE config 4: 1e8 points, 512^2) it was measured and LOSES: the heavy kernel 1485 -> 1261 us, but level 1 2980 ->
    // 3420 us -- its scattered 16-byte record stores then feed 8192 open ranges per workgroup instead of 4096, a 128-byte line
    // takes twice as many iterations to fill and is evicted half-written more often (profiles/r05_experiments/bev_split_halves.txt).
    // Ring model (register path): heavy kernel 44.3 -> 40.9 us, level 1 +0.7.  PCA_BEV_SPLIT=0 / 2: off / also on the memory path (A/B).
    // (the host knows only an upper bound of the window: it allows the split -- and sizes level 1's LDS for it -- and the
    // kernels decide on the window's real size, all of them alike: bev_split)
    { static int sp = -1; if (sp < 0) { const char *e = getenv("PCA_BEV_SPLIT"); sp = e ? atoi(e) : 1; }
      a.split = (sp && a.heavy_launched && !intensity64 && (size_t)a.T * 16 <= 128 * 1024) ? (sp == 2 ? 2 : 1) : 0; }
    const size_t lds = (size_t)a.T * 8 * (a.split ? 2 : 1);  // bev_tile_bin: histogram + cursors
    // The caller's proof of which slots can reach the view (pca_bev_bin_range): good for this call only, and not when this call
    // writes owed transforms back (a skipped frame could not receive them).  PCA_BEV_CULL=0: ignored (A/B; same results).
    // With a bin range, ONE round of workgroups if the last such call binned few enough points for level 1's register path at one
    // piece per CU: a piece costs ~15 us before its first point, two rounds of half-size pieces take almost twice as long as
    // one (profiles/r05_experiments/bev_cull_frames_out_of_view.txt).  The count comes from the device through a host-visible
    // word and only sizes the launch: a window that turns out larger goes through the memory path as ever.
    int one_round = 0;
    if (bin_given) {
        static int cu = -1;
        static int64_t pts = -1;
        if (cu < 0) { const char *e = getenv("PCA_BEV_CULL"); cu = e ? atoi(e) : 1; }
        if (pts < 0) { const char *e = getenv("PCA_BEV_ONE_ROUND_PTS"); pts = e ? atoll(e) : 14500; }
        if (cu && !(n_pending > 0 && write_back)) {
            int f = bin_first > slot_begin ? bin_first : slot_begin, e = bin_end < slot_end ? bin_end : slot_end;
            if (e < f) e = f;
            a.bin_first = f; a.bin_end = e;
            a.span_hint = ctx->heavy_hint_dev + 1;
            a.span_hint_known = ctx->heavy_hint[1];
            if (pts > 0 && a.span_hint_known != 0 && (int64_t)a.span_hint_known * 1024 <= pts * ctx->n_cu && a.G > ctx->n_cu) one_round = ctx->n_cu;
        }
    }
    // A K1 that pca_kitti_integrate left for this raster (the window's last frame, same store, same stream) rides in level 1's
    // launch as its first workgroups; any other deferred K1 runs now, on its own.  PCA_FUSE_K1=0: always on its own (A/B).
    bool fuse = false;
    K1Args k1a;
    if (ctx->k1_pend.valid) {
        static int fk = -1;
        if (fk < 0) { const char *e = getenv("PCA_FUSE_K1"); fk = e ? atoi(e) : 1; }
        const pca_ctx::K1Pending &pd = ctx->k1_pend;
        int nt = 0;
        bool owed_ok = true;                                // (a transform owed by the riding frame itself: its end is not written yet)
        for (int k = 0; k < n_pending; ++k) owed_ok = owed_ok && pending_slot_ends[k] <= pd.slot;
        fuse = fk && owed_ok && !intensity64 && pd.stream == s && pd.slot == slot_end - 1 && pd.slot >= slot_begin && pd.frame_off == frame_off &&
               pd.store.x == store->x && lds <= 80 * 1024 && pd.fr.n <= K1_RIDE * K1_SEG && pd.fr.n <= max_points &&
               pca_k1_prepare_pending(ctx, &k1a, &nt, s) == 0 && nt <= K1_RIDE;
        { static int fd = -1; if (fd < 0) { const char *e = getenv("PCA_FUSE_DBG"); fd = e ? atoi(e) : 0; }       // diagnostics: why a noted K1 ran on its own
          if (fd && !fuse)
              fprintf(stderr, "bev: K1 of slot %d not taken along: switch %d owed_ok %d i64 %d stream %d slot %d (window %d..%d) frame_off %d store %d lds %zu n %d max_points %lld nt %d\n",
                      pd.slot, fk, (int)owed_ok, intensity64 != nullptr, pd.stream == s, pd.slot == slot_end - 1, slot_begin, slot_end,
                      pd.frame_off == frame_off, pd.store.x == store->x, lds, pd.fr.n, (long long)max_points, nt); }
        // K1's tiles are pieces 0 .. nt-1, the window's pieces follow: G of them on top (the workspace is sized for G + K1_RIDE) --
        // unless that exceeds the cap, 512 = two rounds of one workgroup per CU: then the launch stays within those two rounds
        // and the nt CUs that run a K1 tile first take one window piece less (cap - 2 nt window pieces).  Measured on the
        // headline, one box, us of level 1 for window pieces + K1 tiles: 512 + 30 63.3 (the 30 start a third round), 482 + 30
        // 58.0, 450 + 30 56.0, 418 + 30 56.3, 386 + 30 60.4; ring model flat from 354 to 482 (tools/experiments/bev_g_sweep.sh).
        if (fuse) {
            const int cap = max_groups(), G0 = a.G;
            if (one_round) one_round -= nt;                     // (K1's tiles hold nt of the CUs when the launch starts)
            a.G = one_round > 0 ? one_round + nt : (G0 + nt <= cap || (cap - 2 * nt) * 4 < 3 * G0) ? G0 + nt : cap - nt;
            a.Gk = nt; bev_table_order(a); a.k1_slot = pd.slot; a.k1_n = pd.fr.n;
            a.max_points = max_points - pd.fr.n;            // the store's part of the window: the riding frame's records come on top
        }
        else if (pca_k1_flush_pending(ctx)) return -1;
    }
    if (one_round > 0 && !fuse) { a.G = one_round; bev_table_order(a); }
    if (ctx->profiling == 2) pca_prof_begin(ctx, PCA_K_BEV_UNIT, s);
    // (Running the two tile kernels side by side was tried: a second stream with fork / join events costs ~20 us per
    // call, and hipExtAnyOrderLaunch is not honoured on gfx9 -- see DESIGN.md.)
#define PCA_BEV_HEAVY_LAUNCH(I, E, O) PCA_LAUNCH_SHM(ctx, PCA_K_BEV_CELLS_HEAVY, (bev_tile_cells_heavy<I, E, O>), dim3(heavy_grid), dim3(H_THREADS), HEAVY_LDS_BYTES, s, a)
#define PCA_BEV_HEAVY_PICK(I)                                                                                          \
    do {                                                                                                               \
        if (a.extra) { if (a.split) PCA_BEV_HEAVY_LAUNCH(I, true, true); else PCA_BEV_HEAVY_LAUNCH(I, true, false); }  \
        else { if (a.split) PCA_BEV_HEAVY_LAUNCH(I, false, true); else PCA_BEV_HEAVY_LAUNCH(I, false, false); }        \
    } while (0)
    if (intensity64) {
        PCA_LAUNCH_SHM(ctx, PCA_K_BEV_BIN, bev_tile_bin<true>, dim3(a.G), dim3(AB_THREADS), lds, s, a);
        PCA_LAUNCH(ctx, PCA_K_BEV_CELLS, bev_tile_cells<true>, dim3(a.T), dim3(C_THREADS), s, a);
        if (a.heavy_launched) PCA_BEV_HEAVY_PICK(true);
    } else {
        if (fuse) {
            PCA_LAUNCH_SHM(ctx, PCA_K_BEV_BIN, bev_tile_bin_k1, dim3(a.G), dim3(AB_THREADS), lds, s, a, k1a);
            pca_k1_pending_launched(ctx, s);
        } else {
            PCA_LAUNCH_SHM(ctx, PCA_K_BEV_BIN, bev_tile_bin<false>, dim3(a.G), dim3(AB_THREADS), lds, s, a);
        }
        PCA_LAUNCH(ctx, PCA_K_BEV_CELLS, bev_tile_cells<false>, dim3(a.T), dim3(C_THREADS), s, a);
        if (a.heavy_launched) PCA_BEV_HEAVY_PICK(false);
    }
#undef PCA_BEV_HEAVY_PICK
#undef PCA_BEV_HEAVY_LAUNCH
    if (ctx->profiling == 2) pca_prof_end(ctx, s);
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

// Several rasters of ONE store in one launch of each kernel (blockIdx.y = sample): the sweep over present_idx of a
// finished scene (run_nuscenes_bev_gen.py:245-271), or the bev_num augmented samples of one window
// (kitti360_sem_pc_accum.py:236-241).  No owed transforms (the caller flushes them first); every sample has its own
// workspace slice of pca_bev_workspace_bytes(max_points, px) bytes and its own output planes.
int pca_bev_generate_many(pca_ctx *ctx, const pca_store *store, const double *intensity64, const int64_t *frame_off,
                          const pca_bev_job *jobs, int n_jobs, int64_t max_points, void *workspace,
                          int64_t workspace_bytes, void *stream)
{
    if (!ctx) return -1;
    if (pca_k1_flush_pending(ctx)) return -1;               // a deferred K1 of this context comes first
    ctx->bin_valid = false;                                 // (a bin range is for a single raster)
    if (!jobs || n_jobs < 1) { ctx->err = "bev: bad job list"; return -1; }
    if (n_jobs > 65535) { ctx->err = "bev: at most 65535 rasters per call"; return -1; }
    if (max_points < 1) max_points = 1;
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    const int px = jobs[0].prm.px;
    const int64_t per = (pca_bev_workspace_bytes(max_points, px) + 255) & ~255ll;
    if (workspace_bytes < per * n_jobs + 256) { ctx->err = "bev: workspace too small for this many rasters"; return -1; }
    // argument blocks: built in pinned memory, copied to the constant array PCA_BEV_MANY_MAX samples at a time
    const int64_t up_bytes = (int64_t)sizeof(BevArgs) * n_jobs;
    if (ctx->bevm_busy) { PCA_CHECK(ctx, hipEventSynchronize(ctx->bevm_ev)); ctx->bevm_busy = false; }
    if (up_bytes > ctx->bevm_cap) {
        if (ctx->bevm_pin) PCA_CHECK(ctx, hipHostFree(ctx->bevm_pin));
        ctx->bevm_pin = nullptr; ctx->bevm_cap = 0;
        PCA_CHECK(ctx, hipHostMalloc(&ctx->bevm_pin, (size_t)(2 * up_bytes), hipHostMallocMapped));
        ctx->bevm_cap = 2 * up_bytes;
    }
    if (!ctx->bevm_ev) PCA_CHECK(ctx, hipEventCreateWithFlags(&ctx->bevm_ev, hipEventDisableTiming));
    BevArgs *ha = reinterpret_cast<BevArgs *>(ctx->bevm_pin);
    char *ws = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    int G = 0, T = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const pca_bev_job &j = jobs[k];
        if (j.prm.px != px) { ctx->err = "bev: the rasters of one call share the grid size"; return -1; }
        if (bev_prepare(ctx, store, intensity64, frame_off, j.slot_begin, j.slot_split, j.slot_end, max_points, &j.prm, nullptr,
                        nullptr, 0, 1, ws + per * k, per, j.planes, j.planes_f16, nullptr, ha[k]))
            return -1;
        G = ha[k].G; T = ha[k].T;
    }
    if (bev_set_lds_attributes(ctx)) return -1;
    const size_t lds = (size_t)T * 8;
    // a few resident workgroups per sample drain whatever heavy tiles there are (any number of them does)
    const int heavy_grid = T < 8 ? T : 8;
    // The argument blocks live in ONE constant array per device (g_bev_many): a call on another stream or from another
    // context of this device must not overwrite it while this call's kernels may still read it.  Calls take turns on the
    // host (mutex) and on the device (every call first waits for the event the previous one recorded behind its kernels).
    static std::mutex many_mutex;
    static hipEvent_t many_last[64] = {};
    std::lock_guard<std::mutex> many_lock(many_mutex);
    hipEvent_t &last = many_last[ctx->device & 63];
    if (last) PCA_CHECK(ctx, hipStreamWaitEvent(s, last, 0));
    else PCA_CHECK(ctx, hipEventCreateWithFlags(&last, hipEventDisableTiming));
    if (ctx->profiling) pca_prof_begin(ctx, PCA_K_BEV_UNIT, s);
    for (int k0 = 0; k0 < n_jobs; k0 += PCA_BEV_MANY_MAX) {
        const int nk = n_jobs - k0 < PCA_BEV_MANY_MAX ? n_jobs - k0 : PCA_BEV_MANY_MAX;
        // (the argument blocks, 1.3 KB per sample: fetched by a kernel from the mapped host block instead of a copy command,
        // see pca_fetch_block; PCA_SMALL_COPY=1 restores the copy for A/B)
        static int small_copy = -1;
        static void *many_dev[64] = {};
        if (small_copy < 0) { const char *e = getenv("PCA_SMALL_COPY"); small_copy = e ? atoi(e) : 0; }
        if (!many_dev[ctx->device & 63]) PCA_CHECK(ctx, hipGetSymbolAddress(&many_dev[ctx->device & 63], HIP_SYMBOL(g_bev_many)));
        if (small_copy) PCA_CHECK(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(g_bev_many), ha + k0, sizeof(BevArgs) * (size_t)nk, 0, hipMemcpyHostToDevice, s));
        else if (pca_fetch_block(ctx, ha, (int64_t)sizeof(BevArgs) * k0, many_dev[ctx->device & 63], (int64_t)sizeof(BevArgs) * nk, s)) return -1;
        if (intensity64) {
            hipLaunchKernelGGL(bev_tile_bin_many<true>, dim3(G, nk), dim3(AB_THREADS), lds, s);
            hipLaunchKernelGGL(bev_tile_cells_many<true>, dim3(T, nk), dim3(C_THREADS), 0, s);
            hipLaunchKernelGGL(bev_tile_cells_heavy_many<true>, dim3(heavy_grid, nk), dim3(H_THREADS), HEAVY_LDS_BYTES, s);
        } else {
            hipLaunchKernelGGL(bev_tile_bin_many<false>, dim3(G, nk), dim3(AB_THREADS), lds, s);
            hipLaunchKernelGGL(bev_tile_cells_many<false>, dim3(T, nk), dim3(C_THREADS), 0, s);
            hipLaunchKernelGGL(bev_tile_cells_heavy_many<false>, dim3(heavy_grid, nk), dim3(H_THREADS), HEAVY_LDS_BYTES, s);
        }
    }
    PCA_CHECK(ctx, hipEventRecord(ctx->bevm_ev, s));
    ctx->bevm_busy = true;
    PCA_CHECK(ctx, hipEventRecord(last, s));
    if (ctx->profiling) pca_prof_end(ctx, s);
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Polynomial warp of finished planes (bev_generator.py:482-525 of the reference, --bev_do_warp augmentation):
//   out[n][jw][iw] = in[n][clamp(rint(b1 jw + b2 jw^2))][clamp(rint(a1 iw + a2 iw^2))]
// (the reference's loop writes B[:, j_warp, i_warp] = A[:, j, i]: warped row index from the b pair, column from the a
// pair).  Source indices are evaluated as numpy does -- un-fused f64 products and sum, rint, clamp -- so they are the
// reference's integers; a gather commutes with the element-wise f64 -> f16 cast, so the warp runs on the fp16 planes.
// ---------------------------------------------------------------------------------------------------------------
struct WarpArgs {
    const uint16_t *in;
    uint16_t *out;
    int n_planes, px;
    double a1, a2, b1, b2;
};

__device__ __forceinline__ int warp_src(double c1, double c2, int k, int n)
{
    const double kd = (double)k;
    double v = rint(c1 * kd + c2 * (kd * kd));
    v = v < 0.0 ? 0.0 : (v > (double)(n - 1) ? (double)(n - 1) : v);      // NaN -> n - 1 never happens for finite c
    return (int)v;
}

__global__ __launch_bounds__(256) void bev_warp(const WarpArgs a)
{
    const int iw = blockIdx.x * 256 + threadIdx.x, jw = blockIdx.y;
    if (iw >= a.px) return;
    const int i = warp_src(a.a1, a.a2, iw, a.px), j = warp_src(a.b1, a.b2, jw, a.px);
    const size_t plane = (size_t)a.px * a.px;
    for (int n = 0; n < a.n_planes; ++n) a.out[n * plane + (size_t)jw * a.px + iw] = a.in[n * plane + (size_t)j * a.px + i];
}

int pca_bev_warp(pca_ctx *ctx, const uint16_t *planes_f16, uint16_t *out_f16, int n_planes, int px, double a_1, double a_2,
                 double b_1, double b_2, void *stream)
{
    if (!ctx) return -1;
    if (!planes_f16 || !out_f16 || planes_f16 == out_f16 || n_planes < 1 || px < 1 || px > 65535) { ctx->err = "warp: bad arguments"; return -1; }
    if (!(a_1 == a_1 && a_2 == a_2 && b_1 == b_1 && b_2 == b_2)) { ctx->err = "warp: NaN coefficient"; return -1; }
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    WarpArgs a;
    a.in = planes_f16; a.out = out_f16; a.n_planes = n_planes; a.px = px;
    a.a1 = a_1; a.a2 = a_2; a.b1 = b_1; a.b2 = b_2;
    hipLaunchKernelGGL(bev_warp, dim3((px + 255) / 256, px), dim3(256), 0, s, a);
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

int pca_debug_bev_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_stamps), sizeof(unsigned long long) * 8192);
}

}  // extern "C"
