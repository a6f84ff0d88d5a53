// K6/K7: resampling kernels.
//
// K6 nearest: the warp every source read performs (raster/sources.py:119-149 hands it
//   to GDAL's GRA_NearestNeighbour) for axis-aligned same-CRS requests, and SciPy's
//   affine_transform(order=0) of Smooth's zoom branch (raster/spatial.py:296-305).
//   The separable destination->source index maps are evaluated inside the gather kernel in
//   float64 (same operation order as the oracle, FMA contraction off; until round 3 two
//   small kernels wrote them into int32 tables first); the gather kernel is output-stationary:
//   a lane owns 4 consecutive destination columns (one 16-byte store) and walks a strip of rows.
//
// K7 bilinear (our spec -- the reference has none): pixel-centre aligned,
//   edge-clamped, evaluated in float64 and rounded once.  A workgroup owns a
//   TILE_H x TILE_W block of destination pixels; the source rows/cols it touches
//   (block/zoom + 1 px apron) are staged in LDS once, so every source pixel is read
//   from HBM about once (5 B per output px for a 2x upsample of float32).
#include "gm_common.h"

namespace {

// ---------------------------------------------------------------------------------
// index maps
// ---------------------------------------------------------------------------------
// mode 0 (warp):   idx = floor(((o_dst + (k + 0.5) * s_dst) - o_src) / s_src + eps)
// mode 1 (affine): idx = floor(k * z + off + 0.5)            params = {z, off, -, -}
// Evaluated inside the gather kernel (round 3; two small table kernels and a dependent memory round trip before): a lane
// needs its 4 columns and one of the strip's rows -- five float64 divisions per 32 pixels.
struct AxisMap {
  double p0, p1, p2, p3;
};
__device__ __forceinline__ int nearest_index(int k, int n_src, int mode, const AxisMap& m, double eps) {
  double pix;
  if (mode == 0) {
    const double world = m.p0 + ((double)k + 0.5) * m.p1;
    pix = (world - m.p2) / m.p3 + eps;
  } else {
    pix = ((double)k * m.p0 + m.p1) + 0.5;
  }
  const double f = floor(pix);
  int i = -1;
  if (f >= 0.0 && f < (double)n_src) i = (int)f;
  return i;
}

template <typename T>
__device__ __forceinline__ bool is_finite_t(T v) {
  if constexpr (std::is_floating_point<T>::value) return isfinite(v);
  else return true;
}

// grid: x = groups of 4 destination columns (256 lanes -> 1024 columns), y = row strips, z = band.
// Two things keep the gather close to a copy: (1) when the 4 source columns of a lane are consecutive
// (no resampling along x: windows, identity grids) they are fetched as ONE 16-byte load; (2) when the
// next destination row maps to the source row just read (up-sampling along y) the registers are
// reused instead of gathering again.
template <typename T, int ROWS>
__global__ __launch_bounds__(256) void nearest_gather_kernel(const T* __restrict__ src, T* __restrict__ dst, int hs,
                                                             int ws, int hd, int wd, int mode, AxisMap row_map,
                                                             AxisMap col_map, double eps, T fill, int finite_to_fill) {
  typedef T VT __attribute__((ext_vector_type(4)));
  typedef T VTU __attribute__((ext_vector_type(4), aligned(sizeof(T))));  // element-aligned 4-vector
  const int band = blockIdx.z;
  const int c0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;  // (lanes past the last column stay: they hold row indices)
  const T* sband = src + (size_t)band * hs * ws;
  T* dband = dst + (size_t)band * hd * wd;
  int ci[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ci[k] = (c0 + k < wd) ? nearest_index(c0 + k, ws, mode, col_map, eps) : -1;
  const bool run = (ci[0] >= 0) & (ci[1] == ci[0] + 1) & (ci[2] == ci[0] + 2) & (ci[3] == ci[0] + 3);
  const int r_begin = blockIdx.y * ROWS;
  // (destination rows that do not start on 16 bytes -- a width that is not a multiple of 4 -- keep their 16-byte stores,
  //  element-aligned: see bilinear_kernel)
  const bool vec_ok = c0 + 3 < wd;
  const bool vec_aligned = ((wd & 3) == 0) && ((((uintptr_t)dband) % (4 * sizeof(T))) == 0);
  // all ROWS source rows of the strip are requested before the first store (ROWS independent gathers in flight per lane
  // instead of one load-store round trip per row).  Every load is UNCONDITIONAL -- its address clamped into the raster, what
  // it must not deliver selected away afterwards -- and the one choice left, 16-byte loads or four gathers, is made for the
  // whole wave: with per-lane branches around the loads the compiler waited for every row's loads before it issued the next
  // row's (round 3: eight memory round trips per workgroup, 2x up-sampling 278 us; see MEASUREMENTS K6).
  // the strip's source rows (wave-uniform): lane l works out row l mod ROWS, everybody reads them off the first lanes
  static_assert((ROWS & (ROWS - 1)) == 0 && ROWS <= 64, "ROWS");
  const int my_row = nearest_index(min(r_begin + (int)(threadIdx.x & (ROWS - 1)), hd - 1), hs, mode, row_map, eps);
  int rows[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) rows[i] = __builtin_amdgcn_readlane(my_row, i);
  const bool idle = c0 >= wd;  // (takes part in the wave's choices and loads -- from a clamped address -- and stores nothing)
  const bool all_run = __all(run || idle) && ws >= 4;
  const int first_col = min(max(ci[0], 0), max(ws - 4, 0));
  int off[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) off[k] = ci[k] - ci[0];
  const bool span = (ci[0] >= 0) & (ci[0] + 3 < ws) & (off[1] >= 0) & (off[2] >= off[1]) & (off[3] >= off[2]) & (off[3] <= 3);
  int cc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) cc[k] = min(max(ci[k], 0), ws - 1);
  T v[ROWS][4];
  // (a row that repeats the one before it -- up-sampling along y, wave-uniform -- is not loaded again; its registers are
  // filled from the row before once ALL loads are on their way: a copy in between would wait for the load it copies)
  if (__builtin_amdgcn_readfirstlane(all_run ? 1 : 0)) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      if (i > 0 && rows[i] == rows[i - 1]) continue;
      const VTU x = *(const VTU*)(sband + (size_t)max(rows[i], 0) * ws + first_col);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[i][k] = x[k];
    }
  } else if (__builtin_amdgcn_readfirstlane((__all(span || idle) && ws >= 4) ? 1 : 0)) {
    // up-sampling along x: the lane's 4 source columns lie within 4 consecutive ones (2x: j j j+1 j+1): one 16-byte load
    // and a pick per column instead of four 4-byte gathers
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      if (i > 0 && rows[i] == rows[i - 1]) continue;
      const VTU x = *(const VTU*)(sband + (size_t)max(rows[i], 0) * ws + first_col);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[i][k] = off[k] == 0 ? x[0] : (off[k] == 1 ? x[1] : (off[k] == 2 ? x[2] : x[3]));
    }
  } else {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      if (i > 0 && rows[i] == rows[i - 1]) continue;
      const T* srow = sband + (size_t)max(rows[i], 0) * ws;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[i][k] = srow[cc[k]];
    }
  }
#pragma unroll
  for (int i = 1; i < ROWS; ++i) {
    if (rows[i] == rows[i - 1]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[i][k] = v[i - 1][k];
    }
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const bool void_row = rows[i] < 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (void_row || ci[k] < 0) v[i][k] = fill;
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int r = r_begin + i;
    if (r >= hd) break;
    if (finite_to_fill) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[i][k] = is_finite_t<T>(v[i][k]) ? v[i][k] : fill;
    }
    T* out = dband + (size_t)r * wd + c0;
    if (vec_ok) {
      VT pack;
      pack[0] = v[i][0]; pack[1] = v[i][1]; pack[2] = v[i][2]; pack[3] = v[i][3];
      if (vec_aligned) __builtin_nontemporal_store(pack, (VT*)out);
      else __builtin_nontemporal_store(pack, (VTU*)out);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c0 + k < wd) out[k] = v[i][k];
    }
  }
}

// ---------------------------------------------------------------------------------
// bilinear
// ---------------------------------------------------------------------------------
// A workgroup owns BL_TILE_H x BL_TILE_W destination pixels: lane (tx, wave) owns the 4
// consecutive columns 4*tx..4*tx+3 (one 16-byte non-temporal store per row: a wave writes 1 KiB
// contiguous) of the rows wave, wave+4, ...  The source window (+1 px apron) is staged in LDS once
// with coalesced loads issued before anything else; column weights live in registers for all rows
// of the lane, row weights are wave-uniform.
constexpr int BL_TILE_W = 256;  // destination columns per workgroup
constexpr int BL_LDS_BYTES = 64 * 1024;

__device__ __forceinline__ double src_coord(int k, double scale) { return ((double)k + 0.5) * scale - 0.5; }

// T: float or double.  STAGED: the source window of a block fits LDS; otherwise (strong
// down-sampling) taps are read from global memory directly.
template <typename T, bool STAGED, int BL_TILE_H, int SU>
__global__ __launch_bounds__(256) void bilinear_kernel(const T* __restrict__ src, T* __restrict__ dst, int hs, int ws,
                                                       int spitch, int hd, int wd, double sy, double sx, int has_ndv, T ndv,
                                                       int pitch, int vec_src) {
  // spitch: elements between the rows of the source (a zero-copy window of a wider resident raster: spitch > ws, one band)
  extern __shared__ __align__(16) unsigned char smem[];
  T* const tile = (T*)smem;
  const int band = blockIdx.z;
  const T* sband = src + (size_t)band * hs * ws;
  T* dband = dst + (size_t)band * hd * wd;
  // (tuning flags in vec_src: bit 1 -- staging loads through the caches instead of non-temporal; bit 2 -- inside a row of
  //  tiles every XCD takes a contiguous run, so that horizontal neighbours share an L2)
  const bool cached_staging = (vec_src & 2) != 0;
  unsigned bx = blockIdx.x;
  if ((vec_src & 4) && (gridDim.x & 7u) == 0) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
  vec_src &= 1;
  const int r0 = blockIdx.y * BL_TILE_H, c0 = (int)bx * BL_TILE_W;
  const int r1 = min(r0 + BL_TILE_H, hd), c1 = min(c0 + BL_TILE_W, wd);

  // source window of this block (clamped taps): rows [wy0, wy1], cols [wx0, wx1]
  int wy0 = (int)floor(src_coord(r0, sy)), wy1 = (int)floor(src_coord(r1 - 1, sy)) + 1;
  int wx0 = (int)floor(src_coord(c0, sx)), wx1 = (int)floor(src_coord(c1 - 1, sx)) + 1;
  wy0 = max(wy0, 0); wx0 = max(wx0, 0);
  wy1 = min(wy1, hs - 1); wx1 = min(wx1, ws - 1);
  const int win_h = wy1 - wy0 + 1, win_w = wx1 - wx0 + 1;
  // per destination row of the block: upper / lower source row and the lower row's weight, computed
  // once by one lane each and broadcast from LDS (the rows a lane walks are wave-uniform)
  __shared__ double row_fv[BL_TILE_H];
  __shared__ int row_i0[BL_TILE_H];  // the upper source row, not clamped: -1 .. hs - 1
  if (threadIdx.x < BL_TILE_H) {
    const double v = src_coord(r0 + (int)threadIdx.x, sy);
    const double i0f = floor(v);
    row_fv[threadIdx.x] = v - i0f;
    row_i0[threadIdx.x] = (int)i0f;
  }
  if constexpr (!STAGED) __syncthreads();
  // A tap that is no data or not finite voids every pixel it touches (any weight): it
  // is turned into NaN once, where it enters the block, and the arithmetic carries it to the result.
  // (without a no-data value nothing is marked: the limit becomes +inf, which only NaN itself fails)
  const T mark_limit = has_ndv ? std::numeric_limits<T>::max() : std::numeric_limits<T>::infinity();
  const T mark_ndv = has_ndv ? ndv : std::numeric_limits<T>::quiet_NaN();
  const T void_out = has_ndv ? ndv : std::numeric_limits<T>::quiet_NaN();
  auto mark = [&](T v) -> T {
    const bool good = (fabs(v) <= mark_limit) & (v != mark_ndv);
    return good ? v : std::numeric_limits<T>::quiet_NaN();
  };
  // vec_src: every source row starts 16-byte aligned and holds whole 16-byte groups
  const int stage_x0 = vec_src ? (wx0 & ~(16 / (int)sizeof(T) - 1)) : -1;
  if constexpr (STAGED) {
    // Coalesced staging, a wave per source row segment.  All loads of a lane are issued before the
    // first LDS write (a load -> wait -> write loop would pay one HBM round trip per row), and rows
    // are fetched as aligned 16-byte groups when the source allows it.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // SU = rows in flight per lane: the launcher picks ceil(window rows / 4) so that one batch covers the window
    constexpr int PER = 16 / (int)sizeof(T);
    typedef T VT __attribute__((ext_vector_type(PER)));
    if (stage_x0 >= 0) {
      // columns [stage_x0, ...) with stage_x0 = wx0 rounded down to a 16-byte group: tile column of
      // source column x is x - stage_x0
      const int groups = (wx1 - stage_x0) / PER + 1;
      for (int g = lane; g < groups; g += 64) {
        for (int yb = wave; yb < win_h; yb += 4 * SU) {
          VT v[SU];
#pragma unroll
          for (int u = 0; u < SU; ++u) {
            const int y = min(yb + 4 * u, win_h - 1);  // (unconditional: rows past the window re-read its last row)
            const VT* const at = (const VT*)(sband + (size_t)(wy0 + y) * spitch + stage_x0) + g;
            v[u] = cached_staging ? *at : __builtin_nontemporal_load(at);
          }
#pragma unroll
          for (int u = 0; u < SU; ++u) {
            const int y = yb + 4 * u;
            if (y < win_h) {
#pragma unroll
              for (int e = 0; e < PER; ++e) v[u][e] = mark(v[u][e]);
              *(VT*)(tile + y * pitch + g * PER) = v[u];
            }
          }
        }
      }
    } else {
      for (int x = lane; x < win_w; x += 64) {
        for (int yb = wave; yb < win_h; yb += 4 * SU) {
          T v[SU];
#pragma unroll
          for (int u = 0; u < SU; ++u) {
            const int y = min(yb + 4 * u, win_h - 1);
            const T* const at = sband + (size_t)(wy0 + y) * spitch + wx0 + x;
            v[u] = cached_staging ? *at : __builtin_nontemporal_load(at);
          }
#pragma unroll
          for (int u = 0; u < SU; ++u) {
            const int y = yb + 4 * u;
            if (y < win_h) tile[y * pitch + x] = mark(v[u]);
          }
        }
      }
    }
    __syncthreads();
  }
  const int tile_x0 = (STAGED && stage_x0 >= 0) ? stage_x0 : wx0;
  auto tap = [&](int y, int x) -> T {
    if constexpr (STAGED) return tile[(y - wy0) * pitch + (x - tile_x0)];
    else return mark(sband[(size_t)y * spitch + x]);
  };

  const int tx = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cb = c0 + 4 * tx;
  if (cb >= c1) return;
  int ja[4], jb[4];
  double fu[4], gu[4];  // fu = weight of the right tap, gu = 1 - fu
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = min(cb + k, wd - 1);
    const double u = src_coord(c, sx);
    const double j0f = floor(u);
    fu[k] = u - j0f;
    gu[k] = 1.0 - fu[k];
    const int j0 = (int)j0f;
    ja[k] = min(max(j0, 0), ws - 1);
    jb[k] = min(max(j0 + 1, 0), ws - 1);
  }
  // one 16-byte store per row and lane wherever the four columns exist; rows that do not start on 16 bytes (a width that is not
  // a multiple of 4, a band at an odd offset) take the same store unaligned -- element-aligned is all the hardware asks for
  // (12285-wide tiles: 323 -> 216 us at 1.25 x, 310 -> 179 at 2 x against four 4-byte stores)
  const bool vec_ok = cb + 3 < wd;
  const bool vec_aligned = ((wd & 3) == 0) && ((((uintptr_t)dband) % (4 * sizeof(T))) == 0);
  // A wave walks BL_TILE_H / 4 CONSECUTIVE destination rows, source row by source row: the horizontally interpolated values
  // of the source rows y (`upper`) and y + 1 (`lower`) stay in registers while every destination row between them is written
  // -- when up-sampling several, and the lower row then becomes the upper one: every source row is read from LDS, converted
  // and interpolated across once per lane.  (Until round 5 the walk went destination row by destination row and decided per
  // row which of its two source rows it already had: ~90 instructions per row plus ~50 per new source row, issue-bound at
  // 3 x; now ~40 + ~60.)  The rows' source rows are wave-uniform: read from the table through readfirstlane, the loops'
  // branches are scalar.
  constexpr int RPW = BL_TILE_H / 4;
  const int r_begin = r0 + wave * RPW, r_end = min(r0 + (wave + 1) * RPW, r1);
  if (r_begin >= r_end) return;
  auto horizontal = [&](int y_unclamped, double (&h)[4]) {
    const int y = min(max(y_unclamped, 0), hs - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const T A = tap(y, ja[k]), B = tap(y, jb[k]);
      h[k] = gu[k] * (double)A + fu[k] * (double)B;
    }
  };
  auto upper_row_of = [&](int r) { return __builtin_amdgcn_readfirstlane(row_i0[r - r0]); };
  const bool fast_store = __all(vec_ok) && vec_aligned;  // (wave-uniform)
  double upper[4], lower[4];
  int r = r_begin, y = upper_row_of(r_begin);
  horizontal(y, upper);
  T* out = dband + (size_t)r * wd + cb;
#pragma unroll 1
  while (true) {
    horizontal(y + 1, lower);
#pragma unroll 1
    while (r < r_end && upper_row_of(r) == y) {
      const double fv = row_fv[r - r0], gv = 1.0 - fv;  // (wave-uniform values)
      T res[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const T v = (T)(gv * upper[k] + fv * lower[k]);
        res[k] = (v != v) ? void_out : v;
      }
      typedef T VT __attribute__((ext_vector_type(4)));
      VT pack;
      pack[0] = res[0]; pack[1] = res[1]; pack[2] = res[2]; pack[3] = res[3];
      if (fast_store) {
        __builtin_nontemporal_store(pack, (VT*)out);
      } else if (vec_ok) {
        typedef T VU __attribute__((ext_vector_type(4), aligned(sizeof(T))));
        __builtin_nontemporal_store(pack, (VU*)out);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (cb + k < wd) out[k] = res[k];
      }
      out += wd;
      ++r;
    }
    if (r >= r_end) break;
    const int next = upper_row_of(r);
    if (next == y + 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) upper[k] = lower[k];
    } else {  // (down-sampling: source rows are skipped)
      horizontal(next, upper);
    }
    y = next;
  }
}

// ---------------------------------------------------------------------------------
// bilinear, exact 2x up-sampling of a float32 tile (BASELINE configs[3])
// ---------------------------------------------------------------------------------
// The general kernel above spends its time on things a 2x zoom does not need: per-pixel float64 index
// arithmetic, LDS gathers through computed columns, and -- what the PMC passes showed -- a source window
// whose two apron columns each pull in a whole 128-byte line per staged row (FETCH 1.54 x the source).
// With wd = 2 ws the spec of DESIGN section 5 collapses: destination columns 4t .. 4t+3 read the source
// columns 2t-1 .. 2t+2 with the weights (1/4, 3/4), (3/4, 1/4), (1/4, 3/4), (3/4, 1/4) -- the very
// values fu = u - floor(u) and 1 - fu take there, so the float64 products and sums below are the
// general kernel's, bit for bit -- and rows pair up the same way.
//   * a workgroup owns TH x TW destination pixels = SH x SW source pixels + a one-pixel apron.  The
//     SW interior columns start on a 16-byte boundary and are staged as whole 1 KiB rows (one wave
//     instruction per row); the two apron columns are single 4-byte loads of 2 (SH + 2) lanes;
//   * tiles are dealt to the XCDs in contiguous row-major runs (blocks b, b + 8, ... share an XCD):
//     horizontally adjacent tiles run side by side on ONE L2, so the line an apron column drags in
//     is the neighbour's interior line and is fetched from HBM once;
//   * a wave walks RPW consecutive destination rows and keeps the horizontally interpolated values
//     of the last three source rows in registers: every source row is read from LDS once per lane
//     (one ds_read_b64 + two ds_read_b32) and feeds two destination rows.
template <int TH, int TW, int THREADS, bool NT, bool NTS, bool LATE>
__global__ __launch_bounds__(THREADS) void bilinear2x_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int hs, int ws, int spitch, int has_ndv, float ndv,
                                                             unsigned tiles_x, unsigned tiles, int swizzle) {
  constexpr int SH = TH / 2, SW = TW / 2;
  constexpr int P = SW + 8;           // LDS row pitch in floats: source column x of the tile sits at 4 + (x - sx0)
  constexpr int G = SW / 4;           // 16-byte groups per staged row
  constexpr int ROWS = SH + 2;
  constexpr int NW = THREADS / 64, WPR = TW / 256, RG = NW / WPR, RPW = TH / RG;
  static_assert(TW % 256 == 0 && NW % WPR == 0 && TH % RG == 0 && RPW % 2 == 0, "tile shape");
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  extern __shared__ __align__(16) unsigned char smem[];
  float* const tile = (float*)smem;

  // blocks b, b + 8, ... share an XCD (and its L2).  swizzle 1: every XCD takes a contiguous run of the row-major tile
  // order (eight write fronts); swizzle 2: the tile order stays row-major in TIME -- the whole chip sweeps the
  // destination top to bottom, which is what the memory side wants from a write-heavy kernel -- and only inside a
  // row of tiles every XCD takes a contiguous run, so that horizontal neighbours share an L2
  unsigned pid = blockIdx.x;
  if (swizzle == 1) {
    const unsigned per = (tiles + 7u) / 8u;
    pid = (pid & 7u) * per + (pid >> 3);
    if (pid >= tiles) return;
  } else if (swizzle == 2 && (tiles_x & 7u) == 0) {
    const unsigned row = pid / tiles_x, j = pid - row * tiles_x;
    pid = row * tiles_x + (j & 7u) * (tiles_x >> 3) + (j >> 3);
  } else if (swizzle == 3 && (tiles_x & 7u) == 0) {
    // round 6: every XCD takes a vertical STRIP of the destination, tiles_x / 8 tiles wide, and walks it row by row: the tile
    // below a tile follows it tiles_x / 8 places later on the same XCD -- both are in flight together, and the two source rows
    // they share (the lower apron of the one, the first interior rows of the other) come out of that XCD's L2 the second
    // time: configs[3] fetched 1.24 x its source before, the aprons of every tile from HBM.  The chip as a whole still sweeps
    // the destination top to bottom.
    const unsigned strip = tiles_x >> 3, k = pid >> 3;
    pid = (k / strip) * tiles_x + (pid & 7u) * strip + (k % strip);
  }
  const int band = blockIdx.y;
  const int hd = 2 * hs, wd = 2 * ws;
  const float* sband = src + (size_t)band * hs * ws;  // (a window -- spitch != ws -- has one band)
  float* dband = dst + (size_t)band * hd * wd;
  const int ty = pid / tiles_x, tx = pid - ty * tiles_x;
  const int r0 = ty * TH, c0 = tx * TW;
  const int sy0 = r0 / 2, sx0 = c0 / 2;

  const float mark_limit = has_ndv ? std::numeric_limits<float>::max() : std::numeric_limits<float>::infinity();
  const float mark_ndv = has_ndv ? ndv : std::numeric_limits<float>::quiet_NaN();
  const float void_out = has_ndv ? ndv : std::numeric_limits<float>::quiet_NaN();
  auto mark = [&](float v) -> float {
    const bool good = (fabsf(v) <= mark_limit) & (v != mark_ndv);
    return good ? v : std::numeric_limits<float>::quiet_NaN();
  };
  auto load4 = [&](const float* p) -> f4 {
    if constexpr (NT) return __builtin_nontemporal_load((const f4*)p);
    else return *(const f4*)p;
  };

  // staging: every load of a lane is issued before its first LDS write.  LATE: the apron (first and last row of the
  // window, the columns left and right of it) is requested only once the interior has arrived: those lines are interior
  // lines of the neighbouring tiles, which -- dispatched alongside on the same XCD -- have requested them by then, so the
  // apron is served by the L2 instead of a second trip to HBM.
  constexpr int AROWS = LATE ? SH : ROWS, AOFF = LATE ? 1 : 0;
  constexpr int LPL = (AROWS * G + THREADS - 1) / THREADS;  // staging loads per lane
  f4 v[LPL];
#pragma unroll
  for (int k = 0; k < LPL; ++k) {
    const int idx = min((int)threadIdx.x + k * THREADS, AROWS * G - 1);
    const int row = idx / G + AOFF, g = idx % G;
    const int y = min(max(sy0 - 1 + row, 0), hs - 1);
    const int x = min(sx0 + 4 * g, ws - 4);  // (groups past a ragged right edge re-read the last one; never stored)
    v[k] = load4(sband + (size_t)y * spitch + x);
  }
  auto apron_column = [&](int i) {  // i = 2 * row + side
    const int row = i >> 1, right = i & 1;
    const int y = min(max(sy0 - 1 + row, 0), hs - 1);
    const int x = right ? min(sx0 + SW, ws - 1) : max(sx0 - 1, 0);  // edge-clamped taps
    return sband[(size_t)y * spitch + x];
  };
  auto apron_slot = [&](int i) { return (i >> 1) * P + ((i & 1) ? 4 + min(SW, ws - sx0) : 3); };
  float apron = 0.f;
  const bool early_apron = !LATE && threadIdx.x < 2 * ROWS;
  if (early_apron) apron = apron_column(threadIdx.x);
#pragma unroll
  for (int k = 0; k < LPL; ++k) {
    const int idx = (int)threadIdx.x + k * THREADS;
    const int row = idx / G + AOFF, g = idx % G;
    if (idx < AROWS * G && sx0 + 4 * g < ws) {
      f4 m;
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = mark(v[k][e]);
      *(f4*)(tile + row * P + 4 + 4 * g) = m;
    }
  }
  if (early_apron) tile[apron_slot(threadIdx.x)] = mark(apron);
  if constexpr (LATE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (keeps the compiler from hoisting the apron loads)
#pragma unroll
    for (int i = threadIdx.x; i < 2 * G + 2 * ROWS; i += THREADS) {
      if (i < 2 * G) {
        const int row = (i / G) ? ROWS - 1 : 0, g = i % G;
        const int y = min(max(sy0 - 1 + row, 0), hs - 1);
        const f4 w = load4(sband + (size_t)y * spitch + min(sx0 + 4 * g, ws - 4));
        if (sx0 + 4 * g < ws) {
          f4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = mark(w[e]);
          *(f4*)(tile + row * P + 4 + 4 * g) = m;
        }
      } else {
        tile[apron_slot(i - 2 * G)] = mark(apron_column(i - 2 * G));
      }
    }
  }
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = (wave % WPR) * 64 + lane;  // destination columns c0 + 4t .. + 3  <-  source columns sx0 + 2t - 1 .. + 2
  const int cb = c0 + 4 * t;
  if (cb >= wd) return;
  const int rg = wave / WPR;
  const int r_begin = r0 + rg * RPW;
  const float* const col = tile + (rg * (RPW / 2)) * P + 3 + 2 * t;
  auto horizontal = [&](int i, double (&h)[4]) {
    const float* p = col + i * P;
    const double a = (double)p[0];
    const f2 bc = *(const f2*)(p + 1);
    const double b = (double)bc[0], c = (double)bc[1], d = (double)p[3];
    h[0] = 0.25 * a + 0.75 * b;
    h[1] = 0.75 * b + 0.25 * c;
    h[2] = 0.25 * b + 0.75 * c;
    h[3] = 0.75 * c + 0.25 * d;
  };
  auto store_row = [&](int r, const double (&up)[4], double wu, const double (&lo)[4], double wl) {
    f4 pack;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float out = (float)(wu * up[k] + wl * lo[k]);
      pack[k] = (out != out) ? void_out : out;
    }
    if constexpr (NTS) __builtin_nontemporal_store(pack, (f4*)(dband + (size_t)r * wd + cb));
    else *(f4*)(dband + (size_t)r * wd + cb) = pack;
  };
  double hm[4], h0[4], hp[4];
  horizontal(0, hm);
  horizontal(1, h0);
#pragma unroll
  for (int p = 0; p < RPW / 2; ++p) {
    const int r = r_begin + 2 * p;
    if (r >= hd) break;  // (wave-uniform; hd is even)
    horizontal(2 + p, hp);
    store_row(r, hm, 0.25, h0, 0.75);
    store_row(r + 1, h0, 0.75, hp, 0.25);
#pragma unroll
    for (int k = 0; k < 4; ++k) { hm[k] = h0[k]; h0[k] = hp[k]; }
  }
}

struct Bl2xShape { int th, tw, threads; };

// GM_BL2X="<shape>,<swizzle>,<nt>" overrides the launch shape (tuning runs); read once.
static void bl2x_config(int* shape, int* swizzle, int* nt) {
  static int cfg[3] = {-1, 0, 0};
  if (cfg[0] < 0) {
    int s = 16, z = 0, n = 0;
    if (const char* e = getenv("GM_BL2X")) sscanf(e, "%d,%d,%d", &s, &z, &n);
    cfg[1] = z; cfg[2] = n; cfg[0] = s;
  }
  *shape = cfg[0]; *swizzle = cfg[1]; *nt = cfg[2];
}

}  // namespace

extern "C" {

int gm_resample_nearest(const void* src, int dtype, int bands, int hs, int ws, void* dst, int hd, int wd, int mode,
                        const double* row_params, const double* col_params, double eps, uint64_t fill_bits,
                        int finite_to_fill) {
  if (!src || !dst || !row_params || !col_params) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (bands < 0 || hs < 0 || ws < 0 || hd < 0 || wd < 0) return gm::fail(GM_ERR_INVALID, "negative extent");
  if (bands == 0 || hd == 0 || wd == 0) return GM_OK;
  if (bands > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "more than 65535 bands");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  gm_scalar fill;
  fill.u64 = fill_bits;
  const AxisMap row_map{row_params[0], row_params[1], row_params[2], row_params[3]};
  const AxisMap col_map{col_params[0], col_params[1], col_params[2], col_params[3]};
  // rows per workgroup: 4 streams best (identity grid: 72 % of peak, 8 rows: 66 %); when the
  // request up-samples along y, 8 rows share more source rows (2x: 61 % against 57 %)
  const int ROWS = hd > hs ? 8 : 4;
  dim3 grid((unsigned)((wd + 1023) / 1024), (unsigned)((hd + ROWS - 1) / ROWS), (unsigned)bands);
  if (grid.y > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "destination taller than %d rows", 65535 * ROWS);
#define GM_NEAREST(T, FILL)                                                                                       \
  do {                                                                                                            \
    if (ROWS == 8)                                                                                                \
      hipLaunchKernelGGL((nearest_gather_kernel<T, 8>), grid, dim3(256), 0, s, (const T*)src, (T*)dst, hs, ws,    \
                         hd, wd, mode, row_map, col_map, eps, (T)(FILL), finite_to_fill);                                    \
    else                                                                                                          \
      hipLaunchKernelGGL((nearest_gather_kernel<T, 4>), grid, dim3(256), 0, s, (const T*)src, (T*)dst, hs, ws,    \
                         hd, wd, mode, row_map, col_map, eps, (T)(FILL), finite_to_fill);                                    \
  } while (0)
  switch (dtype) {
    case GM_BOOL: case GM_U8: case GM_I8: GM_NEAREST(unsigned char, fill.i64); break;
    case GM_U16: case GM_I16: GM_NEAREST(unsigned short, fill.i64); break;
    case GM_U32: case GM_I32: GM_NEAREST(unsigned int, fill.i64); break;
    case GM_U64: case GM_I64: GM_NEAREST(unsigned long long, fill.u64); break;
    case GM_F32: GM_NEAREST(float, fill.f64); break;
    case GM_F64: GM_NEAREST(double, fill.f64); break;
    default: return gm::fail(GM_ERR_INVALID, "gm_resample_nearest: dtype %d", dtype);
  }
#undef GM_NEAREST
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return gm::hip_fail(e, "gm_resample_nearest");
  return GM_OK;
}

int gm_resample_bilinear(const void* src, int dtype, int bands, int hs, int ws, void* dst, int hd, int wd,
                         int has_ndv, uint64_t ndv_bits) {
  return gm_resample_bilinear_window(src, dtype, bands, hs, ws, ws, dst, hd, wd, has_ndv, ndv_bits);
}

int gm_resample_bilinear_window(const void* src, int dtype, int bands, int hs, int ws, int src_pitch, void* dst, int hd, int wd,
                                int has_ndv, uint64_t ndv_bits) {
  if (!src || !dst) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (src_pitch < ws) return gm::fail(GM_ERR_INVALID, "gm_resample_bilinear: pitch %d of a %d-wide source", src_pitch, ws);
  if (src_pitch != ws && bands != 1) return gm::fail(GM_ERR_UNSUPPORTED, "gm_resample_bilinear: a window of %d bands", bands);
  if (dtype != GM_F32 && dtype != GM_F64) return gm::fail(GM_ERR_UNSUPPORTED, "bilinear needs a float tile");
  if (bands <= 0 || hd <= 0 || wd <= 0) return GM_OK;
  if (hs <= 0 || ws <= 0) return gm::fail(GM_ERR_INVALID, "empty source");
  if (bands > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "more than 65535 bands");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  gm_scalar ndv;
  ndv.u64 = ndv_bits;
  gm::route_hit(GM_ROUTE_BILINEAR, GM_RB_CALLS);
  if (dtype == GM_F32 && hd == 2 * hs && wd == 2 * ws && ws >= 4 && (ws % 4) == 0 && (src_pitch % 4) == 0 &&
      ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0) {
    int shape, swizzle, nt;
    bl2x_config(&shape, &swizzle, &nt);
    if (shape > 0) {
      gm::route_hit(GM_ROUTE_BILINEAR, GM_RB_TWO_X);
#define GM_BL2X_K(TH_, TW_, N_, NT_, NTS_, LATE_)                                                                     \
  do {                                                                                                               \
    if (lds2 > 65536) GM_HIP(hipFuncSetAttribute((const void*)bilinear2x_kernel<TH_, TW_, N_, NT_, NTS_, LATE_>,    \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));            \
    hipLaunchKernelGGL((bilinear2x_kernel<TH_, TW_, N_, NT_, NTS_, LATE_>), dim3(blocks, (unsigned)bands), dim3(N_), \
                       lds2, s, (const float*)src, (float*)dst, hs, ws, src_pitch, has_ndv, (float)ndv.f64, tiles_x, \
                       (unsigned)tiles, swizzle);                                                                    \
  } while (0)
#define GM_BL2X_GO(TH_, TW_, N_)                                                                                     \
  do {                                                                                                               \
    const unsigned tiles_x = (unsigned)((wd + TW_ - 1) / TW_), tiles_y = (unsigned)((hd + TH_ - 1) / TH_);           \
    const unsigned long long tiles = (unsigned long long)tiles_x * tiles_y;                                          \
    if (tiles > 0x7ffffff0ull) return gm::fail(GM_ERR_UNSUPPORTED, "gm_resample_bilinear: tile too large for one launch"); \
    const unsigned blocks = swizzle == 1 ? (unsigned)((tiles + 7) / 8 * 8) : (unsigned)tiles;                       \
    const size_t lds2 = (size_t)(TH_ / 2 + 2) * (TW_ / 2 + 8) * sizeof(float);                                       \
    switch (nt & 7) {                                                                                                \
      case 0: GM_BL2X_K(TH_, TW_, N_, false, true, false); break;                                                    \
      case 1: GM_BL2X_K(TH_, TW_, N_, true, true, false); break;                                                     \
      case 2: GM_BL2X_K(TH_, TW_, N_, false, false, false); break;                                                   \
      case 4: GM_BL2X_K(TH_, TW_, N_, false, true, true); break;                                                     \
      default: return gm::fail(GM_ERR_INVALID, "GM_BL2X: flags %d", nt);                                             \
    }                                                                                                                \
  } while (0)
      switch (shape) {
        case 4: GM_BL2X_GO(32, 512, 256); break;
        case 13: GM_BL2X_GO(8, 1024, 256); break;
        case 29: GM_BL2X_GO(32, 512, 512); break;
        default: GM_BL2X_GO(16, 512, 256); break;
      }
#undef GM_BL2X_K
#undef GM_BL2X_GO
      GM_HIP(hipGetLastError());
      return GM_OK;
    }
  }
  const double sy = (double)hs / (double)hd, sx = (double)ws / (double)wd;
  // largest source window a block can need, + apron; odd pitch (in 4-byte words) spreads the rows
  // of the window over the LDS banks
  const int itemsize = dtype == GM_F32 ? 4 : 8;
  // destination rows per workgroup (GM_BL_H: tuning runs)
  static int tile_h_cfg = 0;
  if (!tile_h_cfg) {
    const char* e = getenv("GM_BL_H");
    tile_h_cfg = e ? atoi(e) : 32;
    if (tile_h_cfg != 8 && tile_h_cfg != 16 && tile_h_cfg != 64) tile_h_cfg = 32;
  }
  const int BL_TILE_H = tile_h_cfg;
  long long win_h = (long long)ceil(BL_TILE_H * sy) + 3, win_w = (long long)ceil(BL_TILE_W * sx) + 3;
  // 16-byte staging: the window may start up to a group earlier and is padded to whole groups
  const int per = 16 / itemsize;
  const bool vec_src = (ws % per) == 0 && (src_pitch % per) == 0 && ((uintptr_t)src % 16) == 0;
  win_w = vec_src ? ((win_w + per - 1 + per - 1) / per) * per : (win_w | 1);
  // From a zoom of 3.5 on the taps are read through the caches instead (a destination row of a lane touches two source cells
  // or three, a wave ~100 consecutive ones, and every source row feeds several destination rows out of registers): no staging
  // phase, no barrier, every wave on its own -- 12288^2 results (tools/bilinear_zoom_sweep.py): 3 x 142 -> 117 us, 4 x 153 -> 101,
  // 6 x 136 -> 96; at 2.5 x the two routes meet (150 / 155), below that the staged window wins by as
  // much (1.5 x: 161 against 304).  GM_BL_STAGED=0 / 1 forces a route (tuning runs).
  static const int staged_cfg = getenv("GM_BL_STAGED") ? atoi(getenv("GM_BL_STAGED")) : -1;
  // (the threshold: first drawn at 2.5 x.  Single runs of either route differ by +- 10 % and more with the state of the part; over four
  //  sweeps, each configuration in a fresh process, the staged window takes 123-138 us at 2.5 x against 150-161, 114-127 at 3 x
  //  against 110-114, 108-118 at 4 x against 94-102 -- but inside the long run of tools/kbench.py the direct route's 3 x is 144-160
  //  and its 4 x 114-132, the staged window's 3 x 132: the window staged in LDS is the steadier of the two, the taps through the
  //  caches follow the clocks.  Hence: direct from 3.5 x on.  Its taps requested a source row ahead were slower: 3 x 112 -> 131.)
  const bool direct = staged_cfg < 0 ? (3.5 * sy <= 1.0 && 3.5 * sx <= 1.0) : staged_cfg == 0;
  const bool staged = !direct && win_h * win_w * itemsize <= BL_LDS_BYTES - 2048;
  const size_t lds = staged ? (size_t)(win_h * win_w * itemsize) : 0;
  gm::route_hit(GM_ROUTE_BILINEAR, staged ? GM_RB_STAGED : (direct ? GM_RB_DIRECT : GM_RB_UNSTAGED_FALLBACK));
  // (GM_BL_SWZ=0: tiles dealt to the XCDs in launch order -- tuning runs; with the runs per XCD 1.5 x 157 -> 151 us, 3 x 135 -> 125,
  //  12285-wide rows 192 -> 170 -- the staged route only: the direct one loses by it, 4 x 96 -> 122; GM_BL_CACHED=1: staging loads through the caches -- no effect measured)
  static const int stage_flags = (getenv("GM_BL_CACHED") && atoi(getenv("GM_BL_CACHED")) ? 2 : 0) | (getenv("GM_BL_SWZ") && !atoi(getenv("GM_BL_SWZ")) ? 0 : 4);
  dim3 grid((unsigned)((wd + BL_TILE_W - 1) / BL_TILE_W), (unsigned)((hd + BL_TILE_H - 1) / BL_TILE_H), (unsigned)bands);
  if (grid.y > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "destination taller than %d rows", 65535 * BL_TILE_H);
#define GM_BILINEAR_SU(T, STAGED, NDV, H_, SU_)                                                                           \
  hipLaunchKernelGGL((bilinear_kernel<T, STAGED, H_, SU_>), grid, dim3(256), lds, s, (const T*)src, (T*)dst, hs, ws,     \
                     src_pitch, hd, wd, sy, sx, has_ndv, (T)(NDV), (int)win_w, (vec_src ? 1 : 0) | (STAGED ? stage_flags : 0))
#define GM_BILINEAR_H(T, STAGED, NDV, H_)                                   \
  do {                                                                      \
    const int rows_per_wave = (int)((win_h + 3) / 4);                       \
    if (!STAGED || rows_per_wave > 6) GM_BILINEAR_SU(T, STAGED, NDV, H_, 8); \
    else if (rows_per_wave > 5) GM_BILINEAR_SU(T, STAGED, NDV, H_, 6);      \
    else if (rows_per_wave > 4) GM_BILINEAR_SU(T, STAGED, NDV, H_, 5);      \
    else GM_BILINEAR_SU(T, STAGED, NDV, H_, 4);                             \
  } while (0)
#define GM_BILINEAR(T, STAGED, NDV)                                   \
  do {                                                                \
    if (BL_TILE_H == 8) GM_BILINEAR_H(T, STAGED, NDV, 8);             \
    else if (BL_TILE_H == 16) GM_BILINEAR_H(T, STAGED, NDV, 16);      \
    else if (BL_TILE_H == 64) GM_BILINEAR_H(T, STAGED, NDV, 64);      \
    else GM_BILINEAR_H(T, STAGED, NDV, 32);                           \
  } while (0)
  if (dtype == GM_F32) {
    if (staged) GM_BILINEAR(float, true, ndv.f64);
    else GM_BILINEAR(float, false, ndv.f64);
  } else {
    if (staged) GM_BILINEAR(double, true, ndv.f64);
    else GM_BILINEAR(double, false, ndv.f64);
  }
#undef GM_BILINEAR
#undef GM_BILINEAR_H
#undef GM_BILINEAR_SU
  GM_HIP(hipGetLastError());
  return GM_OK;
}

}  // extern "C"
