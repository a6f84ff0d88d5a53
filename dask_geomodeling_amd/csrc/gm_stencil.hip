// N3: small-radius stencils of raster/spatial.py.  Every kernel takes the tile WITH the margin the
// Block requested (expand_request_pixels, spatial.py:27-47) and writes the cropped result, so the
// halo is recomputed per tile, never exchanged (tiles stay independent).
//
//   gm_dilate      Dilate.process     (spatial.py:143-154)  3-D 6-neighbourhood, values in order
//   gm_moving_max  MovingMax.process  (spatial.py:190-213)  circular footprint, LDS-staged window
//   gm_hillshade   HillShade.process  (spatial.py:354-417)  3x3 gradient -> shaded uint8
//
// All three are HBM-bound streams (<= 2 x itemsize + result per pixel): full grid, coalesced rows,
// non-temporal stores; the moving maximum stages its (tile + 2 r) window in LDS because a pixel is
// read by up to size^2 neighbours.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include <limits>

#include "gm_common.h"

namespace {

// ---------------------------------------------------------------------------------
// Dilate
// ---------------------------------------------------------------------------------
constexpr int DIL_MAX_VALUES = 16;
template <typename T>
struct DilateArgs {
  T values[DIL_MAX_VALUES];
  int n_values;
};

// grid: x over 256 output columns, y over output rows, z over bands; T compared by bit pattern for
// integers, by == for floats (NaN never matches, as np.ndarray.__eq__)
// Column walker: a workgroup owns 256 output columns and walks ST_ROWS rows downwards; a lane keeps
// the 3 x 3 window of its column in registers and fetches one new row (3 values) per step.  The three
// loads of a step are contiguous across the lanes (offsets 0, 1, 2 elements), so every
// wave-instruction touches one 256-byte span (a lane that gathered a whole window itself would
// spread each instruction over a 1 KiB span at 25 % use and make address processing the limit).
// (16 / 8 rows and batches of 2 were measured for the float32 walkers: no gain over 32 / 4)
constexpr int ST_ROWS = 32, ST_BATCH = 4;

// CHAIN: the values of this launch are applied on top of what earlier launches over the same tile left in `out`
// (more values than one launch carries: every group still tests the ORIGINAL tile, spatial.py:147-154).
template <typename T, bool CHAIN = false>
__global__ __launch_bounds__(256) void dilate_kernel(const T* __restrict__ in, T* out, int bands, int h,
                                                     int w_cols, int pitch, DilateArgs<T> A) {
  // pitch: elements between the rows of the input (a zero-copy window, or a tile inside a padded one: pitch > w_cols,
  // one band); the walk below addresses rows by `w`
  const int c = blockIdx.x * 256 + threadIdx.x, r0 = blockIdx.y * ST_ROWS, b = blockIdx.z;
  const int ow = w_cols - 2, oh = h - 2;
  if (c >= ow) return;
  const int w = pitch;
  const size_t plane = (size_t)h * w;
  const T* base = in + (size_t)b * plane + c;  // column c of the padded tile = west neighbour of output column c
  const bool has_up = b > 0, has_down = b + 1 < bands;
  // ST_BATCH new rows are fetched together (3 * ST_BATCH loads in flight) before their cells are
  // computed: a row-by-row walk would pay one memory round trip per row
  T rows[ST_BATCH + 2][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    rows[0][j] = base[(size_t)r0 * w + j];
    rows[1][j] = base[(size_t)(r0 + 1) * w + j];
  }
  const int r1 = min(r0 + ST_ROWS, oh);
  for (int rb = r0; rb < r1; rb += ST_BATCH) {
    T up[ST_BATCH], down[ST_BATCH];
#pragma unroll
    for (int i = 0; i < ST_BATCH; ++i) {
      const bool live = rb + i < r1;
      const T* row = base + (size_t)(rb + i + 2) * w;
#pragma unroll
      for (int j = 0; j < 3; ++j) rows[i + 2][j] = live ? row[j] : (T)0;
      const T* centre = base + (size_t)(rb + i + 1) * w + 1;
      up[i] = (live && has_up) ? centre[-(ptrdiff_t)plane] : (T)0;
      down[i] = (live && has_down) ? centre[plane] : (T)0;
    }
#pragma unroll
    for (int i = 0; i < ST_BATCH; ++i) {
      if (rb + i < r1) {
        const T* top = rows[i];
        const T* mid = rows[i + 1];
        const T* bot = rows[i + 2];
        T* const cell = out + ((size_t)b * oh + rb + i) * ow + c;
        T result = CHAIN ? *cell : mid[1];
        for (int k = 0; k < A.n_values; ++k) {  // in order: later values overwrite earlier ones
          const T v = A.values[k];
          const bool hit = (mid[1] == v) | (top[1] == v) | (bot[1] == v) | (mid[0] == v) | (mid[2] == v) |
                           (has_up & (up[i] == v)) | (has_down & (down[i] == v));
          result = hit ? v : result;
        }
        __builtin_nontemporal_store(result, cell);
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      rows[0][j] = rows[ST_BATCH][j];
      rows[1][j] = rows[ST_BATCH + 1][j];
    }
  }
}

// Staging of the LDS tiles of this file: the ROWS x (256 + HALO) window at (r0, c0) of a plane, pitch
// 256 + HALO.  A lane loads its own column (the row offsets are scalar arithmetic: ~3 vector instructions per
// cell; addressing cell by cell with `cell / PITCH` cost ~15, as much as the stencils themselves), the HALO
// columns east of the workgroup's 256 are shared out cell by cell.  All loads of a lane are in flight before
// the first LDS write; rows / columns past the plane are clamped (those cells feed no output).  `pitch`: elements
// between the rows of the plane (a zero-copy window of a wider resident raster has pitch > w).
template <typename T, int ROWS, int HALO, typename F>
__device__ __forceinline__ void stage_window(T* tile, const T* __restrict__ plane, int r0, int c0, int h, int w, int pitch, F clean) {
  constexpr int PITCH = 256 + HALO, EAST = ROWS * HALO, EAST_PER_LANE = (EAST + 255) / 256;
  const T* column = plane + min(c0 + (int)threadIdx.x, w - 1);
  T own[ROWS], east[EAST_PER_LANE];
#pragma unroll
  for (int wy = 0; wy < ROWS; ++wy) own[wy] = column[(size_t)min(r0 + wy, h - 1) * pitch];
#pragma unroll
  for (int i = 0; i < EAST_PER_LANE; ++i) {
    const int cell = min((int)threadIdx.x + 256 * i, EAST - 1);
    const int ey = cell / HALO, ex = 256 + cell - ey * HALO;
    east[i] = plane[(size_t)min(r0 + ey, h - 1) * pitch + min(c0 + ex, w - 1)];
  }
#pragma unroll
  for (int wy = 0; wy < ROWS; ++wy) tile[wy * PITCH + threadIdx.x] = clean(own[wy]);
#pragma unroll
  for (int i = 0; i < EAST_PER_LANE; ++i) {
    const int cell = (int)threadIdx.x + 256 * i;
    const int ey = cell / HALO, ex = 256 + cell - ey * HALO;
    if (cell < EAST) tile[ey * PITCH + ex] = clean(east[i]);
  }
}

// Maximum as the hardware has it: one instruction for two or three operands (the b > a ? b : a form costs a
// compare and a select each).  NaN operands are ignored -- see the note on NaN at the MovingMax kernels.
template <typename T>
__device__ __forceinline__ T hw_max(T a, T b) {
  if constexpr (std::is_same<T, float>::value) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  } else if constexpr (std::is_same<T, double>::value) {
    return __builtin_fmax(a, b);
  } else {
    return b > a ? b : a;
  }
}
template <typename T>
__device__ __forceinline__ T hw_max3(T a, T b, T c) {
  if constexpr (std::is_same<T, float>::value) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
  } else {
    return hw_max(hw_max(a, b), c);  // (integers: the compiler forms v_max3_i32 / _u32 itself)
  }
}

// Single-band tiles of 2- and 4-byte cells through the LDS tile of HillShade / MovingMax (the window
// of 16 x 256 outputs staged once, all loads of a lane in flight; lanes walk their column out of LDS):
// the column walker above waits one memory round trip per 4 rows, eight times per wave.
constexpr int DT_H = 16, DT_W = 256;
template <typename T>
__global__ __launch_bounds__(256) void dilate_tile_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w,
                                                          int pitch, int opitch, DilateArgs<T> A) {
  __shared__ T tile[(DT_H + 2) * (DT_W + 2)];
  const int c0 = blockIdx.x * DT_W, r0 = blockIdx.y * DT_H;
  const int ow = w - 2, oh = h - 2;
  constexpr int PITCH = DT_W + 2;
  stage_window<T, DT_H + 2, 2>(tile, in, r0, c0, h, w, pitch, [](T v) { return v; });
  __syncthreads();
  const int c = c0 + threadIdx.x;
  if (c >= ow) return;
  const T* col = tile + threadIdx.x;  // window columns threadIdx.x .. + 2 = west, centre, east
  T north = col[1], west = col[PITCH], centre = col[PITCH + 1], east = col[PITCH + 2];
#pragma unroll
  for (int r = 0; r < DT_H; ++r) {
    const T sw = col[(r + 2) * PITCH], south = col[(r + 2) * PITCH + 1], se = col[(r + 2) * PITCH + 2];
    T result = centre;
    for (int k = 0; k < A.n_values; ++k) {  // in order: later values overwrite earlier ones
      const T v = A.values[k];
      const bool hit = (centre == v) | (north == v) | (south == v) | (west == v) | (east == v);
      result = hit ? v : result;
    }
    if (r0 + r < oh) __builtin_nontemporal_store(result, out + (size_t)(r0 + r) * opitch + c);
    north = centre;
    west = sw;
    centre = south;
    east = se;
  }
}

// Dilate for 1-byte tiles (class rasters are uint8): a lane owns FOUR consecutive output columns and
// works on them as one 32-bit word (SWAR).  Per input row it loads 8 bytes (the 6 columns its
// outputs see), takes the three byte-shifted words (west / centre / east), and per value turns
// "byte == v" into a 0xFF / 0x00 byte mask with five integer operations -- so a row costs one load
// instead of twelve byte loads, and ~7 VALU operations per pixel and value instead of ~14.
constexpr int DIL8_MAX_VALUES = 4;
// rows a lane walks / rows fetched together; measured on 8192^2, 2 values: 32/4 92 us, 32/8 109, 16/4 79, 8/4 76,
// 8/2 73, 4/4 72 (short walks = more waves in flight; the two priming rows are re-read from L2)
constexpr int D8_ROWS = 8, D8_BATCH = 2;
struct Dilate8Args {
  unsigned splat[DIL8_MAX_VALUES];  // the value in every byte
  int n_values;
};

__device__ __forceinline__ unsigned bytes_equal(unsigned x, unsigned splat) {
  const unsigned t = x ^ splat;                                       // zero byte <=> equal
  const unsigned nz = ((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t;          // bit 7 of a byte set <=> byte != 0
  const unsigned m = ~nz & 0x80808080u;                               // 0x80 in equal bytes
  return m | (m - (m >> 7));                                          // 0xFF in equal bytes
}

// 8 bytes at p (any alignment); columns at or past `limit` read as `pad`
__device__ __forceinline__ unsigned long long load_window8(const unsigned char* p, int col, int limit, unsigned char pad) {
  unsigned long long x;
  if (col + 8 <= limit) {
    __builtin_memcpy(&x, p, 8);
  } else {
    x = 0;
    for (int j = 0; j < 8; ++j) x |= (unsigned long long)(col + j < limit ? p[j] : pad) << (8 * j);
  }
  return x;
}

__global__ __launch_bounds__(256) void dilate8_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                      int bands, int h, int w, Dilate8Args A) {
  const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4, r0 = blockIdx.y * D8_ROWS, b = blockIdx.z;
  const int ow = w - 2, oh = h - 2;
  if (c4 >= ow) return;
  const size_t plane = (size_t)h * w;
  const unsigned char* base = in + (size_t)b * plane + c4;  // input column c4 = west neighbour of output column c4
  const bool has_up = b > 0, has_down = b + 1 < bands;
  const int r1 = min(r0 + D8_ROWS, oh);
  // per value: horizontal hits of the previous input row (as a centre row), and the centre-column
  // equality of the previous two input rows (the vertical neighbours of the rows around them)
  unsigned horiz_prev[DIL8_MAX_VALUES], eq_prev[DIL8_MAX_VALUES], eq_prev2[DIL8_MAX_VALUES], band_prev[DIL8_MAX_VALUES];
  unsigned centre_prev = 0;
  auto absorb = [&](unsigned long long win, unsigned long long up, unsigned long long down, bool emit, int out_row) {
    const unsigned west = (unsigned)win, centre = (unsigned)(win >> 8), east = (unsigned)(win >> 16);
    const unsigned upc = (unsigned)(up >> 8), downc = (unsigned)(down >> 8);
    unsigned result = centre_prev;
#pragma unroll
    for (int k = 0; k < DIL8_MAX_VALUES; ++k) {
      if (k < A.n_values) {
        const unsigned v = A.splat[k];
        const unsigned eq = bytes_equal(centre, v);
        const unsigned horiz = eq | bytes_equal(west, v) | bytes_equal(east, v);
        unsigned band = 0;
        if (has_up) band |= bytes_equal(upc, v);
        if (has_down) band |= bytes_equal(downc, v);
        // the previous input row is the centre row of `out_row`: its own horizontal hits, the rows above
        // (eq_prev2) and below (eq: this row), and its band neighbours
        const unsigned hit = horiz_prev[k] | eq_prev2[k] | eq | band_prev[k];
        result = (result & ~hit) | (v & hit);  // in order: later values overwrite earlier ones
        eq_prev2[k] = eq_prev[k];
        eq_prev[k] = eq;
        horiz_prev[k] = horiz;
        band_prev[k] = band;
      }
    }
    centre_prev = centre;
    if (emit) {
      unsigned char* o = out + ((size_t)b * oh + out_row) * ow + c4;
      if (c4 + 4 <= ow) {
        typedef unsigned u32u __attribute__((aligned(1)));
        __builtin_nontemporal_store(result, (u32u*)o);
      } else {
        for (int j = 0; c4 + j < ow; ++j) o[j] = (unsigned char)(result >> (8 * j));
      }
    }
  };
#pragma unroll
  for (int k = 0; k < DIL8_MAX_VALUES; ++k) horiz_prev[k] = eq_prev[k] = eq_prev2[k] = band_prev[k] = 0;
  // (a wave whose lanes all have their 8 bytes inside the row -- every wave but the last of a row -- loads without the
  // per-lane test of load_window8: with that test around them the loads went out one memory round trip at a time)
  const bool all_inside = __all(c4 + 8 <= w);
  auto window = [&](const unsigned char* p) -> unsigned long long {
    if (all_inside) {
      unsigned long long x;
      __builtin_memcpy(&x, p, 8);
      return x;
    }
    return load_window8(p, c4, w, 0);
  };
  auto fetch = [&](int y, unsigned long long& win, unsigned long long& up, unsigned long long& down) {
    const unsigned char* row = base + (size_t)y * w;
    win = window(row);
    up = has_up ? window(row - plane) : 0ull;
    down = has_down ? window(row + plane) : 0ull;
  };
  // input rows r0, r0 + 1 prime the state (no output), then input row y + 1 completes output row y - 1 ... :
  // output row r has centre input row r + 1 and is complete once input row r + 2 was absorbed
  {
    unsigned long long w0, u0, d0, w1, u1, d1;
    fetch(r0, w0, u0, d0);
    fetch(r0 + 1, w1, u1, d1);
    absorb(w0, u0, d0, false, 0);
    absorb(w1, u1, d1, false, 0);
  }
  for (int rb = r0; rb < r1; rb += D8_BATCH) {
    unsigned long long win[D8_BATCH], up[D8_BATCH], down[D8_BATCH];
#pragma unroll
    for (int i = 0; i < D8_BATCH; ++i) {
      win[i] = up[i] = down[i] = 0;
      if (rb + i < r1) fetch(rb + i + 2, win[i], up[i], down[i]);
    }
#pragma unroll
    for (int i = 0; i < D8_BATCH; ++i)
      if (rb + i < r1) absorb(win[i], up[i], down[i], true, rb + i);
  }
}

// Single-band tiles, eight pixels per lane from an LDS tile: a workgroup stages the 18 rows x 2064 bytes its
// 16 x 2048 outputs see.  The byte compare is the expensive part of the word-parallel scheme (~8 integer
// operations per word and value, three words per output word above).  Here a lane owns 8
// consecutive output columns = input bytes [8 l, 8 l + 10) = three aligned LDS words; "byte == value" is
// computed ONCE per input word as a flag (0x80 in equal bytes: 5 operations) and the west / centre / east
// views of an output word are byte shifts of those flags (v_alignbyte); the flags are OR-ed as they are and
// widened to byte masks once per output word.  ~4 integer operations per pixel and value instead of ~7,
// 8-byte stores, 16-byte staging loads (unaligned when the row pitch is: a load that runs past the end of
// its row reads the start of the next one, which feeds no output; only the end of the tile is guarded).
__device__ __forceinline__ unsigned equal_flags(unsigned x, unsigned splat) {
  const unsigned t = x ^ splat;                                // zero byte <=> equal
  const unsigned nz = ((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t;   // bit 7 of a byte set <=> byte != 0
  return ~nz & 0x80808080u;                                    // 0x80 in equal bytes
}
constexpr int D8W_COLS = 2048, D8W_PITCH = 516;  // window: (D8W_H + 2) rows of 516 words (2064 bytes)
template <int D8W_H>
__global__ __launch_bounds__(256) void dilate8_wide_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                           int h, int w, int pitch, int opitch, Dilate8Args A) {
  __shared__ __align__(16) unsigned tile[(D8W_H + 2) * D8W_PITCH];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int c0 = blockIdx.x * D8W_COLS, r0 = blockIdx.y * D8W_H;
  const int ow = w - 2, oh = h - 2;
  const size_t left = (size_t)(h - 1) * pitch + w;  // bytes from the first cell to the end of the last row
  constexpr int QUADS = D8W_PITCH / 4, CELLS = (D8W_H + 2) * QUADS, PER_LANE = (CELLS + 255) / 256;
  // Staging: every 16-byte load is UNCONDITIONAL, from an address clamped into the raster -- what a cell past the row end must
  // hold (zeros) or the last 15 bytes of the raster need (byte loads) is put right afterwards.  With the tests around the
  // loads the compiler waited for each load before it issued the next (round 3: PER_LANE memory round trips per lane).
  u32x4 staged[PER_LANE];
  const bool roomy = left >= 16;  // (a raster of fewer than 16 bytes: everything goes byte by byte)
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    const int cell = threadIdx.x + 256 * i;
    const int wy = min(cell / QUADS, D8W_H + 1), q = cell - (cell / QUADS) * QUADS;
    const int sy = min(r0 + wy, h - 1), sx = c0 + 16 * q;
    const size_t at = (size_t)sy * pitch + sx;
    u32x4 x = {0u, 0u, 0u, 0u};
    if (roomy) __builtin_memcpy(&x, in + min(at, left - 16), 16);  // (wave-uniform test)
    staged[i] = x;
  }
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    const int cell = threadIdx.x + 256 * i;
    const int wy = min(cell / QUADS, D8W_H + 1), q = cell - (cell / QUADS) * QUADS;
    const int sy = min(r0 + wy, h - 1), sx = c0 + 16 * q;
    const size_t at = (size_t)sy * pitch + sx;
    u32x4 x = staged[i];
    if (sx >= w) {
      x = u32x4{0u, 0u, 0u, 0u};
    } else if (!roomy || at + 16 > left) {
      unsigned char tail[16] = {0};
      for (int j = 0; at + j < left; ++j) tail[j] = in[at + j];
      __builtin_memcpy(&x, tail, 16);
    }
    if (cell < CELLS) *(u32x4*)(tile + 4 * cell) = x;
  }
  __syncthreads();
  const int c8 = c0 + 8 * threadIdx.x;
  if (c8 >= ow) return;
  unsigned horiz_prev[DIL8_MAX_VALUES][2], eq_prev[DIL8_MAX_VALUES][2], eq_prev2[DIL8_MAX_VALUES][2];
#pragma unroll
  for (int k = 0; k < DIL8_MAX_VALUES; ++k)
#pragma unroll
    for (int j = 0; j < 2; ++j) horiz_prev[k][j] = eq_prev[k][j] = eq_prev2[k][j] = 0;
  unsigned centre_prev[2] = {0u, 0u};
#pragma unroll
  for (int y = 0; y < D8W_H + 2; ++y) {
    const unsigned* row = tile + y * D8W_PITCH + 2 * threadIdx.x;
    const unsigned w0 = row[0], w1 = row[1], w2 = row[2];
    const unsigned centre[2] = {__builtin_amdgcn_alignbyte(w1, w0, 1), __builtin_amdgcn_alignbyte(w2, w1, 1)};
    unsigned result[2] = {centre_prev[0], centre_prev[1]};
#pragma unroll
    for (int k = 0; k < DIL8_MAX_VALUES; ++k) {
      if (k < A.n_values) {
        const unsigned v = A.splat[k];
        const unsigned f0 = equal_flags(w0, v), f1 = equal_flags(w1, v), f2 = equal_flags(w2, v);
        const unsigned eq[2] = {__builtin_amdgcn_alignbyte(f1, f0, 1), __builtin_amdgcn_alignbyte(f2, f1, 1)};
        const unsigned east[2] = {__builtin_amdgcn_alignbyte(f1, f0, 2), __builtin_amdgcn_alignbyte(f2, f1, 2)};
        const unsigned west[2] = {f0, f1};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned horiz = eq[j] | west[j] | east[j];
          const unsigned hit = horiz_prev[k][j] | eq_prev2[k][j] | eq[j];  // centre row y - 1: its row, the rows above / below
          const unsigned mask = hit | (hit - (hit >> 7));                  // 0x80 flags -> 0xFF bytes
          result[j] = (result[j] & ~mask) | (v & mask);                    // in order: later values overwrite earlier ones
          eq_prev2[k][j] = eq_prev[k][j];
          eq_prev[k][j] = eq[j];
          horiz_prev[k][j] = horiz;
        }
      }
    }
    centre_prev[0] = centre[0];
    centre_prev[1] = centre[1];
    const int orow = r0 + y - 2;
    if (y >= 2 && orow < oh) {
      unsigned char* o = out + (size_t)orow * opitch + c8;
      if (c8 + 8 <= ow) {
        // (rows of w - 2 bytes: unless that is a multiple of 8 these stores are not 8-byte aligned, which is what
        //  separates the 8192-column tile -- 8190-byte rows, 42-46 % of peak -- from the 8194-column one a Dilate
        //  Block actually requests -- 8192-byte rows, 54-57 %.  Measured and dropped: storing in naturally aligned
        //  pieces by the row's phase (2 + 4 + 2 bytes ...: no faster than the hardware's own split) and rotating
        //  the column segment with the tile row so that an XCD does not stay on one segment (no effect: it is not
        //  the channel aliasing of 8 KiB pitches suspected in round 2).)
        // (past the caches, like the other kernels' results: 8192-wide 37.5 -> 33.8 us, the 8194-wide tile of a Block 31.6 -> 29.3)
        typedef unsigned u32x2u __attribute__((ext_vector_type(2), aligned(1)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(u32x2{result[0], result[1]}, (u32x2u*)o);
      } else {
        for (int j = 0; c8 + j < ow; ++j) o[j] = (unsigned char)(result[j >> 2] >> (8 * (j & 3)));
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// MovingMax
// ---------------------------------------------------------------------------------
// The sign of a zero maximum.  The hardware maximum of +0.0 and -0.0 is +0.0; scipy.ndimage.maximum_filter keeps whichever
// zero its scan order leaves standing (restated from measurements against SciPy 1.15):
//   * a footprint with holes (sizes 5 and up: the disk) is scanned in raster order, a later cell replaces the running
//     maximum only when it compares GREATER -- the FIRST zero of the footprint stays;
//   * the all-true 3 x 3 footprint takes SciPy's separable route, maximum_filter1d down the columns and then along the
//     rows, whose running-maximum queue takes a later cell that compares greater OR EQUAL -- the bottom-most zero of the
//     right-most column that has one.
// Both rules pick the zero that comes first in an order of the CELLS that does not depend on the window (raster order; columns
// from the right, rows from below): a maximum over (value, rank of the cell) pairs.  A zero has 31 idle bits, so the pair fits
// the cell: the kernels note a -0.0 while they stage their window (a compare per staged cell, a rare thing in a raster),
// and a workgroup that staged one re-writes its LDS window as sortable integer KEYS -- negative values below, zeros in the middle
// carrying their rank and their sign, positive values above -- and walks it with the integer maximum instead: the same
// walk, the same instruction count, no second pass over the raster.  Workgroups without a -0.0 never leave the float walk.
template <typename T>
__device__ __forceinline__ bool is_negative_zero(T v) {
  if constexpr (std::is_same<T, float>::value) return __float_as_uint(v) == 0x80000000u;
  else if constexpr (std::is_same<T, double>::value) return (unsigned long long)__double_as_longlong(v) == 0x8000000000000000ull;
  else return false;
}

// K = T: the window holds values.  K = the signed integer of T's width: it holds keys (float tiles of a workgroup that staged a -0.0).
template <typename T>
struct ZeroKey {
  typedef T K;
};
template <>
struct ZeroKey<float> {
  typedef int K;
  typedef unsigned U;
  static constexpr U SIGN = 0x80000000u, INF = 0x7F800000u;
  static constexpr K NAN_KEY = (K)0x80000000u;  // below every value: a NaN loses every maximum, as with the hardware maximum
};
template <>
struct ZeroKey<double> {
  typedef long long K;
  typedef unsigned long long U;
  static constexpr U SIGN = 0x8000000000000000ull, INF = 0x7FF0000000000000ull;
  static constexpr K NAN_KEY = (K)0x8000000000000000ull;
};
template <typename T>
__device__ __forceinline__ typename ZeroKey<T>::U float_bits(T v) {
  if constexpr (std::is_same<T, float>::value) return __float_as_uint(v);
  else return (unsigned long long)__double_as_longlong(v);
}
// rank: which zero wins among zeros (larger wins), 0 <= rank < zoff / 2; keys of zeros are 0 .. zoff - 1
template <typename T>
__device__ __forceinline__ typename ZeroKey<T>::K zero_key(T v, int rank, int zoff) {
  typedef ZeroKey<T> Z;
  const typename Z::U b = float_bits(v), mag = b & ~Z::SIGN;
  if (mag > Z::INF) return Z::NAN_KEY;
  if (mag == 0) return (typename Z::K)(2 * rank + ((b & Z::SIGN) ? 0 : 1));
  if (b & Z::SIGN) return (typename Z::K)(b ^ ~Z::SIGN);  // -tiny -> -2 ... -max -> far below: the order of the values
  return (typename Z::K)(b + (typename Z::U)zoff);
}
template <typename T>
__device__ __forceinline__ T zero_key_value(typename ZeroKey<T>::K k, int zoff) {
  typedef ZeroKey<T> Z;
  typename Z::U b;
  if (k < 0) b = (typename Z::U)k ^ ~Z::SIGN;
  else if (k < (typename Z::K)zoff) b = (k & 1) ? 0 : Z::SIGN;
  else b = (typename Z::U)k - (typename Z::U)zoff;
  if constexpr (std::is_same<T, float>::value) return __uint_as_float(b);
  else return __longlong_as_double((long long)b);
}
// Has any lane of the workgroup `mine`?  (One ballot per wave into its own LDS word: nothing to clear, one barrier -- the one
// the kernels need after staging anyway.)
__device__ __forceinline__ bool workgroup_any_and_sync(bool mine, unsigned* flags) {
  const unsigned long long votes = __ballot(mine);
  if ((threadIdx.x & 63) == 0) flags[threadIdx.x >> 6] = votes != 0;
  __syncthreads();
  return (flags[0] | flags[1] | flags[2] | flags[3]) != 0;
}

constexpr int MM_TILE_H = 16, MM_TILE_W = 64;
// NaN.  scipy.ndimage.maximum_filter starts from the first cell of the footprint (top row, leftmost) and
// takes every later cell that compares greater: a NaN in that first cell stays (nothing compares greater than
// it), a NaN anywhere else is never taken.  The kernels below reduce with the hardware maximum, which ignores
// NaN operands, and then apply exactly that rule: the result is NaN where the first cell of the footprint is
// (in a window of keys a NaN is the lowest key; the NaN itself is then fetched from the raster).

// The walk of the general kernel over a window of values (K = T) or of keys.
// A lane owns 4 CONSECUTIVE output rows of one column.  The circle is scanned column by column
// (wave-uniform dx): the vertical runs of the 4 outputs overlap in all but 3 rows at either end,
// so the shared core is read and reduced once -- (2 hh + 4) LDS reads per column for 4 outputs
// instead of 4 (2 hh + 1); consecutive lanes read consecutive addresses (no bank conflicts).
template <typename T, typename K>
__device__ __forceinline__ void moving_max_walk(const K* tile, int pitch, const T* __restrict__ sband, T* __restrict__ out, int b,
                                                int w, int size, int radius, int oh, int ow, int r0, int c0, int has_ndv, T ndv,
                                                T minimum, K identity, int zoff) {
  constexpr bool KEYS = !std::is_same<T, K>::value;
  const int c = threadIdx.x & 63, o0 = (threadIdx.x >> 6) * 4;
  if (c0 + c >= ow) return;
  const int s2 = size * size;
  K m[4] = {identity, identity, identity, identity};
  auto up = [](K a, K b) -> K { return hw_max(a, b); };
  for (int dx = 0; dx < size; ++dx) {
    const int xx = dx - radius;
    int hh = -1;  // half-height of the footprint's column: (x^2 + y^2) < (size/2)^2 (utils.py:536-547)
    while (4 * (xx * xx + (hh + 1) * (hh + 1)) < s2) ++hh;
    if (hh < 0) continue;
    const K* col = tile + (c + dx);
    const int first = o0 + radius - hh;  // window row of the topmost cell of output 0's run
    if (hh == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = up(m[k], col[(first + k) * pitch]);
      continue;
    }
    const K l0 = col[first * pitch], l1 = col[(first + 1) * pitch], l2 = col[(first + 2) * pitch];
    const int after = first + 2 * hh + 1;
    const K r0v = col[after * pitch], r1v = col[(after + 1) * pitch], r2v = col[(after + 2) * pitch];
    K core = identity;
    for (int y = first + 3; y < after; ++y) core = up(core, col[y * pitch]);
    m[0] = up(m[0], up(core, up(l0, up(l1, l2))));
    m[1] = up(m[1], up(core, up(l1, up(l2, r0v))));
    m[2] = up(m[2], up(core, up(l2, up(r0v, r1v))));
    m[3] = up(m[3], up(core, up(r0v, up(r1v, r2v))));
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rr = o0 + k;
    if (r0 + rr >= oh) break;
    T best;
    if constexpr (KEYS) best = zero_key_value<T>(m[k], zoff);
    else best = m[k];
    if constexpr (std::is_floating_point<T>::value) {  // (NaN: see the note above)
      int a = -1;
      while (4 * ((a + 1) * (a + 1) + radius * radius) < s2) ++a;
      const K first = tile[rr * pitch + c + radius - a];
      if constexpr (KEYS) {
        if (first == ZeroKey<T>::NAN_KEY) best = sband[(size_t)(r0 + rr) * w + (c0 + c + radius - a)];
      } else {
        if (first != first) best = first;
      }
    }
    // still the minimum on a no-data centre: no-data again (spatial.py:208-209).  Rare: only then is the
    // original centre fetched (the LDS copy cannot tell "was no data" from "is the minimum")
    if (has_ndv && best == minimum) {
      const T centre = sband[(size_t)(r0 + rr + radius) * w + (c0 + c + radius)];
      if (centre == ndv) best = ndv;
    }
    __builtin_nontemporal_store(best, out + ((size_t)b * oh + (r0 + rr)) * ow + (c0 + c));
  }
}

// RMAX: largest radius of the instantiation (bounds the staging loops at compile time so that all of
// a wave's window loads are in flight together instead of one load-wait-store round trip per row)
template <typename T, int RMAX>
__global__ __launch_bounds__(256) void moving_max_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w,
                                                         int size, int has_ndv, T ndv, T minimum) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ unsigned flags[4];
  T* const tile = (T*)smem;
  const int radius = size / 2;
  const int oh = h - 2 * radius, ow = w - 2 * radius;
  const int r0 = blockIdx.y * MM_TILE_H, c0 = blockIdx.x * MM_TILE_W, b = blockIdx.z;
  const T* sband = in + (size_t)b * h * w;
  const int win_h = min(MM_TILE_H, oh - r0) + 2 * radius, win_w = min(MM_TILE_W, ow - c0) + 2 * radius;
  const int pitch = MM_TILE_W + 2 * radius;
  // stage the window, no-data lowered to the dtype's minimum (spatial.py:199-203)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int ROWS_PER_WAVE = (MM_TILE_H + 2 * RMAX + 3) / 4, CHUNKS = (MM_TILE_W + 2 * RMAX + 63) / 64;
  T staged[ROWS_PER_WAVE][CHUNKS];
  bool negative_zero = false;
#pragma unroll
  for (int i = 0; i < ROWS_PER_WAVE; ++i) {
    const int y = wave + 4 * i;
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) {
      const int x = lane + 64 * j;
      staged[i][j] = (y < win_h && x < win_w) ? sband[(size_t)(r0 + y) * w + c0 + x] : minimum;
    }
  }
#pragma unroll
  for (int i = 0; i < ROWS_PER_WAVE; ++i) {
    const int y = wave + 4 * i;
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) {
      const int x = lane + 64 * j;
      const T v = staged[i][j];
      negative_zero |= is_negative_zero(v);
      if (y < win_h && x < win_w) tile[y * pitch + x] = (has_ndv && v == ndv) ? minimum : v;
    }
  }
  if constexpr (std::is_floating_point<T>::value) {
    if (workgroup_any_and_sync(negative_zero, flags)) {
      // a -0.0 in this window: the walk over keys (see "The sign of a zero maximum"); zeros rank in raster order, first wins
      // -- but by column, then by row, last wins, under the all-true 3 x 3 footprint (8-byte cells come here at every size)
      typedef typename ZeroKey<T>::K K;
      K* const keys = (K*)smem;
      const int rows = MM_TILE_H + 2 * radius, cells = rows * pitch, zoff = 2 * cells;
      for (int i = threadIdx.x; i < cells; i += 256) {
        const int y = i / pitch, x = i - y * pitch;
        const T v = (y < win_h && x < win_w) ? tile[i] : minimum;  // (cells nobody staged)
        keys[i] = zero_key<T>(v, size == 3 ? x * rows + y : cells - 1 - i, zoff);
      }
      __syncthreads();
      moving_max_walk<T, K>(keys, pitch, sband, out, b, w, size, radius, oh, ow, r0, c0, has_ndv, ndv, minimum,
                            zero_key<T>(minimum, 0, zoff), zoff);
      return;
    }
  } else {
    __syncthreads();
  }
  moving_max_walk<T, T>(tile, pitch, sband, out, b, w, size, radius, oh, ow, r0, c0, has_ndv, ndv, minimum, minimum, 0);
}

// The 3 x 3 footprint (size 3: every cell of the square satisfies x^2 + y^2 < 2.25) is separable: the
// maximum of each window row, then the maximum of three row maxima.  Same LDS tile as HillShade: a
// workgroup stages the 18 x 258 window of its 16 x 256 outputs (no data lowered to the dtype's minimum,
// all loads of a lane in flight), a lane walks its column keeping the row maxima of the last two rows.
constexpr int M3_H = 16, M3_W = 256;
template <typename T, typename K>
__device__ __forceinline__ void moving_max3_walk(const K* tile, const T* __restrict__ plane, T* __restrict__ out, int b, int oh, int ow,
                                                 int r0, int c0, int pitch, int opitch, int has_ndv, T ndv, T minimum, int zoff) {
  constexpr bool KEYS = !std::is_same<T, K>::value;
  constexpr int PITCH = M3_W + 2;
  const int c = c0 + threadIdx.x;
  if (c >= ow) return;
  const K* col = tile + threadIdx.x;
  auto row_max = [&](int wy) -> K { return hw_max3(col[wy * PITCH], col[wy * PITCH + 1], col[wy * PITCH + 2]); };
  K above = row_max(0), centre = row_max(1);
#pragma unroll
  for (int r = 0; r < M3_H; ++r) {
    const K below = row_max(r + 2);
    const K top = hw_max3(above, centre, below);
    above = centre;
    centre = below;
    T best;
    if constexpr (KEYS) best = zero_key_value<T>(top, zoff);
    else best = top;
    if constexpr (std::is_floating_point<T>::value) {  // (NaN: see the note above moving_max_walk)
      const K first = col[r * PITCH];
      if constexpr (KEYS) {
        if (first == ZeroKey<T>::NAN_KEY) best = plane[(size_t)(r0 + r) * pitch + c];
      } else {
        if (first != first) best = first;
      }
    }
    if (r0 + r < oh) {
      // still the minimum on a no-data centre: no-data again (spatial.py:208-209); only then is the
      // original centre fetched (the LDS copy cannot tell "was no data" from "is the minimum")
      if (has_ndv && best == minimum) {
        const T original = plane[(size_t)(r0 + r + 1) * pitch + (c + 1)];
        if (original == ndv) best = ndv;
      }
      __builtin_nontemporal_store(best, out + (size_t)b * oh * ow + (size_t)(r0 + r) * opitch + c);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void moving_max3_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w,
                                                          int pitch, int opitch, int has_ndv, T ndv, T minimum) {
  constexpr int ROWS = M3_H + 2, PITCH = M3_W + 2;
  __shared__ T tile[ROWS * PITCH];
  __shared__ unsigned flags[4];
  const int c0 = blockIdx.x * M3_W, r0 = blockIdx.y * M3_H, b = blockIdx.z;
  const int ow = w - 2, oh = h - 2;
  const T* plane = in + (size_t)b * h * w;
  bool negative_zero = false;
  stage_window<T, ROWS, 2>(tile, plane, r0, c0, h, w, pitch, [&](T v) {
    negative_zero |= is_negative_zero(v);
    return (has_ndv && v == ndv) ? minimum : v;  // spatial.py:199-203
  });
  if constexpr (std::is_same<T, float>::value) {
    if (workgroup_any_and_sync(negative_zero, flags)) {
      // a -0.0 in this window: keys; zeros rank by column, then by row -- the bottom-most zero of the right-most column wins
      int* const keys = (int*)tile;
      constexpr int zoff = 2 * ROWS * PITCH;
      for (int i = threadIdx.x; i < ROWS * PITCH; i += 256) {
        const int y = i / PITCH, x = i - y * PITCH;
        keys[i] = zero_key<float>(tile[i], x * ROWS + y, zoff);
      }
      __syncthreads();
      moving_max3_walk<T, int>(keys, plane, out, b, oh, ow, r0, c0, pitch, opitch, has_ndv, ndv, minimum, zoff);
      return;
    }
  } else {
    __syncthreads();
  }
  moving_max3_walk<T, T>(tile, plane, out, b, oh, ow, r0, c0, pitch, opitch, has_ndv, ndv, minimum, 0);
}

// Fixed small footprints (sizes 5 to 15).  The disk is a stack of horizontal runs: out(r) = max over dy of
// HM_{a(dy)}(row r + dy), HM_a(row) = maximum of the 2 a + 1 cells around the lane's column, a(dy) = the
// half-width of the footprint at that row offset.  HM_0 .. HM_R of a window row are nested (two more
// cells per step), so a lane that walks its column keeps them for the last 2 R + 1 window rows in
// registers and spends 2 R + 1 LDS reads and 2 R three-operand maxima per output -- all indices compile-time constants
// (fully unrolled walk), i.e. straight-line code without the data-dependent loops of the general kernel.
template <int R>
__host__ __device__ constexpr int disk_half_width(int dy) {
  int a = -1;
  const int s2 = (2 * R + 1) * (2 * R + 1);
  for (int dx = 0; dx <= R; ++dx)
    if (4 * (dx * dx + dy * dy) < s2) a = dx;
  return a;  // -1: the row offset is outside the footprint (cannot happen for |dy| <= R)
}

// output rows per workgroup: the walk passes TH + 2 R window rows for TH outputs
__host__ __device__ constexpr int disk_tile_rows(int R) { return R >= 7 ? 32 : 16; }
template <typename T, typename K, int R>
__device__ __forceinline__ void moving_max_disk_walk(const K* tile, const T* __restrict__ plane, T* __restrict__ out, int b, int oh,
                                                     int ow, int r0, int c0, int pitch, int opitch, int has_ndv, T ndv, T minimum,
                                                     K identity, int zoff) {
  constexpr bool KEYS = !std::is_same<T, K>::value;
  constexpr int TH = disk_tile_rows(R), TW = 256, PITCH = TW + 2 * R, ROWS = TH + 2 * R;
  const int c = c0 + threadIdx.x;
  if (c >= ow) return;
  const K* col = tile + threadIdx.x;  // window columns threadIdx.x .. threadIdx.x + 2 R = dx -R .. R
  K hm[2 * R + 1][R + 1];             // hm[k][a]: HM_a of window row y - 2 R + k after window row y was taken
#pragma unroll
  for (int k = 0; k <= 2 * R; ++k)
#pragma unroll
    for (int a = 0; a <= R; ++a) hm[k][a] = identity;
#pragma unroll
  for (int y = 0; y < ROWS; ++y) {
#pragma unroll
    for (int k = 0; k < 2 * R; ++k)
#pragma unroll
      for (int a = 0; a <= R; ++a) hm[k][a] = hm[k + 1][a];
    K v[2 * R + 1];
#pragma unroll
    for (int j = 0; j <= 2 * R; ++j) v[j] = col[y * PITCH + j];
    hm[2 * R][0] = v[R];
#pragma unroll
    for (int a = 1; a <= R; ++a) hm[2 * R][a] = hw_max3(hm[2 * R][a - 1], v[R - a], v[R + a]);
    if (y >= 2 * R) {
      const int r = y - 2 * R;
      K top = hm[0][disk_half_width<R>(R)];
#pragma unroll
      for (int k = 1; k <= 2 * R; k += 2)
        top = hw_max3(top, hm[k][disk_half_width<R>(k > R ? k - R : R - k)],
                      hm[k + 1][disk_half_width<R>(k + 1 > R ? k + 1 - R : R - k - 1)]);
      T best;
      if constexpr (KEYS) best = zero_key_value<T>(top, zoff);
      else best = top;
      if constexpr (std::is_floating_point<T>::value) {  // (NaN: see the note above moving_max_walk)
        const K first = col[r * PITCH + R - disk_half_width<R>(R)];
        if constexpr (KEYS) {
          if (first == ZeroKey<T>::NAN_KEY) best = plane[(size_t)(r0 + r) * pitch + (c + R - disk_half_width<R>(R))];
        } else {
          if (first != first) best = first;
        }
      }
      if (r0 + r < oh) {
        if (has_ndv && best == minimum) {  // no data again where the centre was no data (spatial.py:208-209)
          const T original = plane[(size_t)(r0 + r + R) * pitch + (c + R)];
          if (original == ndv) best = ndv;
        }
        __builtin_nontemporal_store(best, out + (size_t)b * oh * ow + (size_t)(r0 + r) * opitch + c);
      }
    }
  }
}

template <typename T, int R>
__global__ __launch_bounds__(256) void moving_max_disk_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w,
                                                              int pitch, int opitch, int has_ndv, T ndv, T minimum) {
  constexpr int TH = disk_tile_rows(R), TW = 256, PITCH = TW + 2 * R, ROWS = TH + 2 * R;
  __shared__ T tile[ROWS * PITCH];
  __shared__ unsigned flags[4];
  const int c0 = blockIdx.x * TW, r0 = blockIdx.y * TH, b = blockIdx.z;
  const int ow = w - 2 * R, oh = h - 2 * R;
  const T* plane = in + (size_t)b * h * w;
  bool negative_zero = false;
  stage_window<T, ROWS, 2 * R>(tile, plane, r0, c0, h, w, pitch, [&](T v) {
    negative_zero |= is_negative_zero(v);
    return (has_ndv && v == ndv) ? minimum : v;  // spatial.py:199-203
  });
  if constexpr (std::is_same<T, float>::value) {
    if (workgroup_any_and_sync(negative_zero, flags)) {
      // a -0.0 in this window: keys; zeros rank in raster order, the first of the footprint wins
      int* const keys = (int*)tile;
      constexpr int cells = ROWS * PITCH, zoff = 2 * cells;
      for (int i = threadIdx.x; i < cells; i += 256) keys[i] = zero_key<float>(tile[i], cells - 1 - i, zoff);
      __syncthreads();
      moving_max_disk_walk<T, int, R>(keys, plane, out, b, oh, ow, r0, c0, pitch, opitch, has_ndv, ndv, minimum,
                                      zero_key<float>(minimum, 0, zoff), zoff);
      return;
    }
  } else {
    __syncthreads();
  }
  moving_max_disk_walk<T, T, R>(tile, plane, out, b, oh, ow, r0, c0, pitch, opitch, has_ndv, ndv, minimum, minimum, 0);
}

// ---------------------------------------------------------------------------------
// HillShade
// ---------------------------------------------------------------------------------
struct HillArgs {
  double xres, yres;       // division of integer / float64 tiles happens in float64
  float xres32, yres32;    // float32 tiles divide by the float32-rounded resolution (NumPy weak scalar)
  // a resolution that is a power of two (1 m, 0.5 m, ...) divides exactly as a multiplication by its
  // reciprocal: 0 = divide, else multiply by the exact 1 / res
  double xinv, yinv;
  float xinv32, yinv32;
  float sin_alt, cos_alt_zsf, square_zsf, cos_az, sin_az;
  float az32;              // float32(az): what `aspect - az` subtracts (python float meets a float32 array)
};

// spatial.py:405-416 on one cell: x, y are the float32 gradients.
//
// Reference sequence (NumPy, float32 arrays, python floats as weak scalars), every step rounded to
// float32: xx = x*x + y*y; aspect = arctan2(y, x); t = aspect - az; s = sin(t); root = sqrt(xx);
// num = sin_alt - (k * root) * s; den = sqrt(1 + zsf^2 * xx); cang = num / den; result = trunc(255 * cang).
//
// Two tiers.  FAST: sin(arctan2(y, x) - az) * sqrt(xx) = y cos(az) - x sin(az), one v_rsq for the
// normalisation; its 255 * cang lies within 1.4e-4 of the reference sequence's (measured over 3e8 random
// gradients at scales 1e-2 .. 1e5 and four sun positions; the bound does not depend on the scale since
// every term is <= 1 in magnitude; worst case by error analysis 3.8e-4, reached only for a sun on the
// horizon).  The truncation can only differ when that value is within the error of
// an integer, so only cells within HS_MARGIN = 6e-4 of one (0.12 % of cells) take the EXACT tier: the
// reference sequence step by step, arctan2 and sin as the float32 roundings of the float64 functions,
// sqrt and division correctly rounded.  What then remains against NumPy is its own float32 arctan2 (SVML
// on AVX512 hosts, glibc elsewhere: the reference's output itself differs between such hosts) and sin,
// which are off the correctly rounded value by 1 ulp on 38 % / 24 % of arguments: 1-5 cells per million
// differ by 1 (tests/test_gpu_parity.py records the count).
constexpr float HS_MARGIN = 6e-4f;
__device__ __forceinline__ unsigned char shade_exact(float x, float y, const HillArgs& A) {
  const float xx_plus_yy = x * x + y * y;
  const float aspect = (float)atan2((double)y, (double)x);
  const float t = aspect - A.az32;
  const float s = (float)sin((double)t);
  // (sqrtf and / are the correctly rounded ones under hipcc's default
  //  -fhip-fp32-correctly-rounded-divide-sqrt; __fsqrt_rn is the 1-ulp native instruction in this toolchain)
  const float root = sqrtf(xx_plus_yy);
  const float num = A.sin_alt - (A.cos_alt_zsf * root) * s;
  const float den = sqrtf(1.0f + A.square_zsf * xx_plus_yy);
  const float cang = num / den;
  return (cang <= 0.0f) ? (unsigned char)0 : (unsigned char)(int)(255.0f * cang);
}
// the fast tier: true when it decided the cell (-> *out), false when the cell is for the exact tier
__device__ __forceinline__ bool shade_fast(float x, float y, const HillArgs& A, unsigned char* out) {
#pragma clang fp contract(fast)  // (this tier only has to land within HS_MARGIN: fused multiply-adds are welcome)
  const float xx_plus_yy = x * x + y * y;
  const float fast = (A.sin_alt - A.cos_alt_zsf * (y * A.cos_az - x * A.sin_az)) * __builtin_amdgcn_rsqf(1.0f + A.square_zsf * xx_plus_yy);
  const float v = 255.0f * fast;
  const float nearest = rintf(v);
  // (NaN / inf gradients compare false here and take the exact tier, which treats them like NumPy does)
  *out = (fast <= 0.0f) ? (unsigned char)0 : (unsigned char)(int)v;
  return fabsf(v - nearest) > HS_MARGIN || nearest < 0.5f;
}
__device__ __forceinline__ unsigned char shade_cell(float x, float y, const HillArgs& A) {
  unsigned char r;
  return shade_fast(x, y, A, &r) ? r : shade_exact(x, y, A);
}

// T arithmetic exactly as NumPy carries it out on a T array: integers wrap modulo 2^bits,
// float32 / float64 round after every operation (-ffp-contract=off)
template <typename T>
__device__ __forceinline__ float gradient(T a, T b2, T c, T d, T e2, T f, double res, float res32, double inv, float inv32) {
  if constexpr (std::is_integral<T>::value && sizeof(T) <= 4) {
    // modular arithmetic in 32 bits: the truncation to T equals NumPy's wrap-around at every step
    const unsigned wide = (unsigned)a + 2u * (unsigned)b2 + (unsigned)c - (unsigned)d - 2u * (unsigned)e2 - (unsigned)f;
    const T s = (T)wide;
    // a power-of-two resolution: float(float64(s) * inv) = float(s) * inv32 -- the product with a power of two is exact
    // either way and the one rounding (of s, when it has more than 24 bits) is the same; far outside float32's
    // comfortable exponent range the float64 route stays
    if (inv32 != 0.0f && fabsf(inv32) < 0x1p64f && fabsf(inv32) > 0x1p-64f) return (float)s * inv32;
    const double wide_s = (double)s;
    return (float)(inv != 0.0 ? wide_s * inv : wide_s / res);
  } else if constexpr (std::is_integral<T>::value) {
    typedef unsigned long long U;  // modular arithmetic: the truncation to T equals NumPy's wrap-around
    const U wide = (U)(long long)a + 2ull * (U)(long long)b2 + (U)(long long)c - (U)(long long)d - 2ull * (U)(long long)e2 - (U)(long long)f;
    const double s = (double)(T)wide;
    return (float)(inv != 0.0 ? s * inv : s / res);
  } else if constexpr (std::is_same<T, float>::value) {
    const float s = ((((a + 2.0f * b2) + c) - d) - 2.0f * e2) - f;
    return inv32 != 0.0f ? s * inv32 : s / res32;
  } else {
    const double s = ((((a + 2.0 * b2) + c) - d) - 2.0 * e2) - f;
    return (float)(inv != 0.0 ? s * inv : s / res);
  }
}

// Column walker like dilate_kernel: 3 x 3 window of the lane's column in registers, one new row of
// three contiguous-across-lanes loads per step.
// The shading of a cell is shade_cell() above.
template <typename T>
__global__ __launch_bounds__(256) void hillshade_kernel(const T* __restrict__ in, unsigned char* __restrict__ out, int h,
                                                        int w, int has_ndv, T ndv, T fill, HillArgs A) {
  const int c = blockIdx.x * 256 + threadIdx.x, r0 = blockIdx.y * ST_ROWS, b = blockIdx.z;
  const int ow = w - 2, oh = h - 2;
  if (c >= ow) return;
  const T* base = in + (size_t)b * h * w + c;
  auto fetch = [&](int row, T (&v)[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const T x = base[(size_t)row * w + j];
      v[j] = (has_ndv && x == ndv) ? fill : x;  // spatial.py:363-364
    }
  };
  T rows[ST_BATCH + 2][3];
  fetch(r0, rows[0]);
  fetch(r0 + 1, rows[1]);
  const int r1 = min(r0 + ST_ROWS, oh);
  for (int rb = r0; rb < r1; rb += ST_BATCH) {
#pragma unroll
    for (int i = 0; i < ST_BATCH; ++i) {  // ST_BATCH rows of loads in flight together
      if (rb + i < r1) fetch(rb + i + 2, rows[i + 2]);
    }
#pragma unroll
    for (int i = 0; i < ST_BATCH; ++i) {
      if (rb + i < r1) {
        const T* top = rows[i];
        const T* mid = rows[i + 1];
        const T* bot = rows[i + 2];
        // spatial.py:384-404 (s0..s8 = rows top/middle/bottom x columns left/middle/right)
        const float y = gradient<T>(top[0], top[1], top[2], bot[0], bot[1], bot[2], A.yres, A.yres32, A.yinv, A.yinv32);
        const float x = gradient<T>(top[0], mid[0], bot[0], top[2], mid[2], bot[2], A.xres, A.xres32, A.xinv, A.xinv32);
        out[((size_t)b * oh + rb + i) * ow + c] = shade_cell(x, y, A);
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      rows[0][j] = rows[ST_BATCH][j];
      rows[1][j] = rows[ST_BATCH + 1][j];
    }
  }
}

// HillShade through an LDS tile: a workgroup stages the (HT_H + 2) x (HT_W + 2) window of its HT_H x HT_W
// outputs once -- every load of a lane in flight before the first LDS write, no data already
// replaced by the fill -- and a lane then walks HT_H rows of its column out of LDS (the walker above
// waits one memory round trip per 4 rows, eight times per wave).  The uint8 results of a wave (HT_H rows
// x 64 columns) are transposed through LDS as well, so that a lane stores 16 consecutive pixels of one
// row instead of one byte per row.
// (8192^2 float32: 16 rows 92 us, 8 rows 105 us, 32 rows 116 us)
constexpr int HT_H = 16, HT_W = 256;
template <typename T>
__global__ __launch_bounds__(256) void hillshade_tile_kernel(const T* __restrict__ in, unsigned char* __restrict__ out,
                                                             int h, int w, int pitch, int has_ndv, T ndv, T fill, HillArgs A) {
  __shared__ T tile[(HT_H + 2) * (HT_W + 2)];
  __shared__ unsigned char shade[4][HT_H * 64];  // per wave: HT_H rows x 64 columns
  constexpr unsigned HT_PENDING = 256;
  __shared__ float pending_x[HT_PENDING], pending_y[HT_PENDING];
  __shared__ unsigned short pending_at[HT_PENDING];
  __shared__ unsigned pending_count;
  if (threadIdx.x == 0) pending_count = 0;
  const int c0 = blockIdx.x * HT_W, r0 = blockIdx.y * HT_H, b = blockIdx.z;
  const int ow = w - 2, oh = h - 2;
  const T* plane = in + (size_t)b * h * w;
  constexpr int PITCH = HT_W + 2;
  stage_window<T, HT_H + 2, 2>(tile, plane, r0, c0, h, w, pitch, [&](T v) { return (has_ndv && v == ndv) ? fill : v; });  // spatial.py:363-364
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* col = tile + threadIdx.x;  // window column of the west neighbour of output column c0 + threadIdx.x
  T top[3], mid[3], bot[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    top[j] = col[j];
    mid[j] = col[PITCH + j];
  }
  bool overflowed = false;  // this lane met a full queue
#pragma unroll
  for (int r = 0; r < HT_H; ++r) {
#pragma unroll
    for (int j = 0; j < 3; ++j) bot[j] = col[(r + 2) * PITCH + j];
    // spatial.py:384-404 (s0..s8 = rows top/middle/bottom x columns left/middle/right)
    const float y = gradient<T>(top[0], top[1], top[2], bot[0], bot[1], bot[2], A.yres, A.yres32, A.yinv, A.yinv32);
    const float x = gradient<T>(top[0], mid[0], bot[0], top[2], mid[2], bot[2], A.xres, A.xres32, A.xinv, A.xinv32);
    // cells the fast tier cannot decide are queued for the whole workgroup and shaded together after the
    // walk: a wave that met one would otherwise run the (float64 arctan2 + sin) exact tier for a single lane
    unsigned char fast_result;
    if (!shade_fast(x, y, A, &fast_result)) {
      const unsigned slot = atomicAdd(&pending_count, 1u);
      if (slot < HT_PENDING) {
        pending_x[slot] = x;
        pending_y[slot] = y;
        pending_at[slot] = (unsigned short)((wave << 10) | (r * 64 + lane));
      } else {
        overflowed = true;  // (queue full -- e.g. a plateau whose shade sits on an integer)
      }
    }
    shade[wave][r * 64 + lane] = fast_result;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      top[j] = mid[j];
      mid[j] = bot[j];
    }
  }
  if (overflowed) {
    // the lane walks its column again and shades what the fast tier leaves open itself (one copy of the
    // exact tier -- float64 arctan2 and sin, ~350 instructions -- instead of one per row of the unrolled walk)
#pragma unroll 1
    for (int r = 0; r < HT_H; ++r) {
      const T* p = col + r * PITCH;
      const float y = gradient<T>(p[0], p[1], p[2], p[2 * PITCH], p[2 * PITCH + 1], p[2 * PITCH + 2], A.yres, A.yres32, A.yinv, A.yinv32);
      const float x = gradient<T>(p[0], p[PITCH], p[2 * PITCH], p[2], p[PITCH + 2], p[2 * PITCH + 2], A.xres, A.xres32, A.xinv, A.xinv32);
      unsigned char unused;
      if (!shade_fast(x, y, A, &unused)) shade[wave][r * 64 + lane] = shade_exact(x, y, A);
    }
  }
  __syncthreads();
  {
    const unsigned n_pending = min(pending_count, HT_PENDING);
    for (unsigned i = threadIdx.x; i < n_pending; i += 256) {
      const unsigned at = pending_at[i];
      shade[at >> 10][at & 1023u] = shade_exact(pending_x[i], pending_y[i], A);
    }
  }
  __syncthreads();
  // lane l: row l / 4, columns 16 (l % 4) ..
  const int part = lane & 3;
#pragma unroll
  for (int row = lane >> 2; row < HT_H; row += 16) {
    const int orow = r0 + row, ocol = c0 + 64 * wave + 16 * part;
    if (orow < oh && ocol < ow) {
      unsigned char* dst = out + ((size_t)b * oh + orow) * ow + ocol;
      const unsigned char* srcb = shade[wave] + row * 64 + 16 * part;
      if (ocol + 16 <= ow && (((uintptr_t)dst) & 15) == 0) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(*(const u32x4*)srcb, (u32x4*)dst);
      } else {
        for (int k = 0; k < 16 && ocol + k < ow; ++k) dst[k] = srcb[k];
      }
    }
  }
}

// float32 tiles, two pixels per lane in packed arithmetic (round 2).  The tile kernel above is bound by its
// vector instructions (46 per pixel, 73 % of the issue slots): here a lane owns two adjacent columns and the
// Sobel sums, the products with the reciprocal resolution and most of the fast tier run as v_pk_*_f32
// instructions on (pixel 0, pixel 1) pairs -- the same IEEE operations in the same order per pixel, two per
// issue slot.  Per row of the window a lane reads L = (v0, v1), M = (v1, v2), R = (v2, v3); 2 M is shared by
// the pixel above and the one below.  Workgroup: HP_H rows x 512 columns.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int HP_H = 8, HP_W = 512;
// T: float, or an integer type of up to 4 bytes (USE_INV only): the Sobel sums of integers run in 32-bit modular
// arithmetic per pixel (NumPy's wrap-around), and float(s) * inv is the float32 the reference's float64 route gives for
// a power-of-two resolution (see gradient()); everything after the gradient is the same packed float32 arithmetic.
// BATCH: the tiles of one mosaic in one launch (blockIdx.z = tile, one band each): per-tile input pointer / pitch and
// output pointer / pitch -- the shades are written straight into their windows of the mosaic (round 4: a RasterTiler
// over HillShade launched and pasted its tiles one by one, ~35 us of host time each against 23 us of kernel).
constexpr int HS_BATCH = 64;
struct HillBatch {
  const void* in[HS_BATCH];
  unsigned char* out[HS_BATCH];
  int in_pitch[HS_BATCH], out_pitch[HS_BATCH];
};
struct NoBatch {};
// DIRECT (round 6, GM_HS_DIRECT=1): a lane stores its two shades of a row straight to the result as it goes -- 64 lanes x 2
// bytes are one 128-byte line per wave and row -- instead of parking the block's shades in LDS for a store phase behind a
// barrier; the exact tier patches its cells in place afterwards.
template <typename T, bool USE_INV, bool BATCH = false, bool DIRECT = false>
__global__ __launch_bounds__(256) void hillshade_pair_kernel(const T* __restrict__ in, unsigned char* __restrict__ out,
                                                             int h, int w, int pitch, int has_ndv, T ndv, T fill, HillArgs A,
                                                             typename std::conditional<BATCH, HillBatch, NoBatch>::type B = {}) {
  static_assert(std::is_same<T, float>::value || (std::is_integral<T>::value && sizeof(T) <= 4 && USE_INV), "see above");
  constexpr int PITCH = HP_W + 2;
  __shared__ __align__(16) T tile[(HP_H + 2) * PITCH];
  __shared__ __align__(16) unsigned char shade[DIRECT ? 16 : HP_H * HP_W];
  constexpr unsigned HP_PENDING = 256;
  __shared__ float pending_x[HP_PENDING], pending_y[HP_PENDING];
  __shared__ unsigned short pending_at[HP_PENDING];
  __shared__ unsigned pending_count;
  if (threadIdx.x == 0) pending_count = 0;
  const int c0 = blockIdx.x * HP_W, r0 = blockIdx.y * HP_H, b = blockIdx.z;
  const int ow = w - 2, oh = h - 2;
  const T* plane = in + (size_t)b * h * w;
  unsigned char* oplane = out + (size_t)b * oh * ow;
  int opitch = ow;
  if constexpr (BATCH) {
    plane = (const T*)B.in[b];
    pitch = B.in_pitch[b];
    oplane = B.out[b];
    opitch = B.out_pitch[b];
  }
  auto clean = [&](T v) -> T { return (has_ndv && v == ndv) ? fill : v; };  // spatial.py:363-364
  {
    // a lane stages columns t and t + 256 of the window, lanes 0..35 the two columns east of the 512
    const T* west = plane + min(c0 + (int)threadIdx.x, w - 1);  // (clamped: cells past the tile feed no output)
    const T* east = plane + min(c0 + 256 + (int)threadIdx.x, w - 1);
    T sw[HP_H + 2], se[HP_H + 2];
#pragma unroll
    for (int wy = 0; wy < HP_H + 2; ++wy) {
      const size_t row = (size_t)min(r0 + wy, h - 1) * pitch;
      sw[wy] = west[row];
      se[wy] = east[row];
    }
    const bool rim = threadIdx.x < 2 * (HP_H + 2);
    const int ry = threadIdx.x >> 1, rx = HP_W + (threadIdx.x & 1);
    T beyond = T(0);
    if (rim) beyond = plane[(size_t)min(r0 + ry, h - 1) * pitch + min(c0 + rx, w - 1)];
#pragma unroll
    for (int wy = 0; wy < HP_H + 2; ++wy) {
      tile[wy * PITCH + threadIdx.x] = clean(sw[wy]);
      tile[wy * PITCH + 256 + threadIdx.x] = clean(se[wy]);
    }
    if (rim) tile[ry * PITCH + rx] = clean(beyond);
  }
  __syncthreads();
  const T* col = tile + 2 * threadIdx.x;  // window column of the west neighbour of the lane's first pixel
  // a window row as the lane needs it: floats as the pairs L = (v0, v1), R = (v2, v3) and 2 M = 2 (v1, v2); integers as
  // the four cells in 32 bits
  typedef typename std::conditional<std::is_same<T, float>::value, float, unsigned>::type Cell;
  struct Row { Cell v[4]; f32x2 M2; };
  auto read_row = [&](int wy) -> Row {
    const T* p = col + wy * PITCH;
    Row row;
#pragma unroll
    for (int k = 0; k < 4; ++k) row.v[k] = (Cell)p[k];  // (signed integers: sign-extended, then modulo 2^32)
    if constexpr (std::is_same<T, float>::value) {
      const f32x2 M = {p[1], p[2]};
      row.M2 = M + M;
    }
    return row;
  };
  const f32x2 inv_x = {A.xinv32, A.xinv32}, inv_y = {A.yinv32, A.yinv32};
  bool overflowed = false;  // this lane met a full queue
  Row top = read_row(0), mid = read_row(1);
#pragma unroll
  for (int r = 0; r < HP_H; ++r) {
    const Row bot = read_row(r + 2);
    // spatial.py:384-404, per pixel: ((((a + 2 b) + c) - d) - 2 e) - f
    f32x2 x, y;
    if constexpr (std::is_same<T, float>::value) {
      const f32x2 tL = {top.v[0], top.v[1]}, tR = {top.v[2], top.v[3]}, mL = {mid.v[0], mid.v[1]}, mR = {mid.v[2], mid.v[3]};
      const f32x2 bL = {bot.v[0], bot.v[1]}, bR = {bot.v[2], bot.v[3]};
      const f32x2 sy = ((((tL + top.M2) + tR) - bL) - bot.M2) - bR;
      const f32x2 sx = ((((tL + (mL + mL)) + bL) - tR) - (mR + mR)) - bR;
      if (USE_INV) {
        x = sx * inv_x;
        y = sy * inv_y;
      } else {
        x = f32x2{sx.x / A.xres32, sx.y / A.xres32};
        y = f32x2{sy.x / A.yres32, sy.y / A.yres32};
      }
    } else {
      f32x2 sx, sy;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const unsigned wy_ = top.v[k] + 2u * top.v[k + 1] + top.v[k + 2] - bot.v[k] - 2u * bot.v[k + 1] - bot.v[k + 2];
        const unsigned wx_ = top.v[k] + 2u * mid.v[k] + bot.v[k] - top.v[k + 2] - 2u * mid.v[k + 2] - bot.v[k + 2];
        sy[k] = (float)(T)wy_;  // the truncation to T is NumPy's wrap-around
        sx[k] = (float)(T)wx_;
      }
      x = sx * inv_x;
      y = sy * inv_y;
    }
    unsigned packed = 0;
    {
#pragma clang fp contract(fast)  // (the fast tier only has to land within HS_MARGIN)
      const f32x2 xx_plus_yy = x * x + y * y;
      const f32x2 den = 1.0f + A.square_zsf * xx_plus_yy;
      const f32x2 num = A.sin_alt - A.cos_alt_zsf * (y * A.cos_az - x * A.sin_az);
      const f32x2 fast = num * f32x2{__builtin_amdgcn_rsqf(den.x), __builtin_amdgcn_rsqf(den.y)};
      const f32x2 v = 255.0f * fast;
      const f32x2 nearest = {rintf(v.x), rintf(v.y)};
      const f32x2 off = v - nearest;
      // (NaN / inf gradients compare false here and take the exact tier, which treats them like NumPy does)
      const bool open0 = !(fabsf(off.x) > HS_MARGIN || nearest.x < 0.5f);
      const bool open1 = !(fabsf(off.y) > HS_MARGIN || nearest.y < 0.5f);
      if (open0 | open1) {  // one test for the pair: 0.12 % of the cells are within the margin of an integer
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (k ? open1 : open0) {
            const unsigned slot = atomicAdd(&pending_count, 1u);
            if (slot < HP_PENDING) {
              pending_x[slot] = x[k];
              pending_y[slot] = y[k];
              pending_at[slot] = (unsigned short)(r * HP_W + 2 * threadIdx.x + k);
            } else {
              overflowed = true;  // (queue full -- e.g. a plateau whose shade sits on an integer)
            }
          }
        }
      }
      // trunc(max(v, 0)) as ONE instruction: v_cvt_u32_f32 truncates and clamps negative values and NaN to 0 (v < 256;
      // the C cast of a negative float to unsigned is undefined, hence the instruction by name)
      unsigned lo, hi;
      asm("v_cvt_u32_f32 %0, %1" : "=v"(lo) : "v"(v.x));
      asm("v_cvt_u32_f32 %0, %1" : "=v"(hi) : "v"(v.y));
      packed = lo | (hi << 8);
    }
    if constexpr (DIRECT) {
      const int orow = r0 + r, ocol = c0 + 2 * (int)threadIdx.x;
      if (orow < oh && ocol < ow) {
        unsigned char* dst = oplane + (size_t)orow * opitch + ocol;
        typedef unsigned short u16u __attribute__((aligned(1)));
        if (ocol + 1 < ow) __builtin_nontemporal_store((unsigned short)packed, (u16u*)dst);
        else *dst = (unsigned char)packed;
      }
    } else {
      *reinterpret_cast<unsigned short*>(shade + r * HP_W + 2 * threadIdx.x) = (unsigned short)packed;
    }
    top = mid;
    mid = bot;
  }
  if (overflowed) {
    // the lane walks its columns again and shades them itself (one copy of the exact tier -- float64
    // arctan2 and sin, ~350 instructions -- instead of one per row of the unrolled walk)
#pragma unroll 1
    for (int i = 0; i < 2 * HP_H; ++i) {
      const int r = i >> 1, k = i & 1;
      const T* p = col + r * PITCH + k;
      const float y = gradient<T>(p[0], p[1], p[2], p[2 * PITCH], p[2 * PITCH + 1], p[2 * PITCH + 2], A.yres, A.yres32, A.yinv, A.yinv32);
      const float x = gradient<T>(p[0], p[PITCH], p[2 * PITCH], p[2], p[PITCH + 2], p[2 * PITCH + 2], A.xres, A.xres32, A.xinv, A.xinv32);
      const unsigned char cell = shade_cell(x, y, A);
      if constexpr (DIRECT) {
        const int orow = r0 + r, ocol = c0 + 2 * (int)threadIdx.x + k;
        if (orow < oh && ocol < ow) oplane[(size_t)orow * opitch + ocol] = cell;
      } else {
        shade[r * HP_W + 2 * threadIdx.x + k] = cell;
      }
    }
  }
  __syncthreads();  // (also: the lanes' own stores have been acknowledged before another lane patches a cell)
  {
    const unsigned n_pending = min(pending_count, HP_PENDING);
    for (unsigned i = threadIdx.x; i < n_pending; i += 256) {
      const unsigned char cell = shade_exact(pending_x[i], pending_y[i], A);
      if constexpr (DIRECT) {
        const int orow = r0 + (int)(pending_at[i] / HP_W), ocol = c0 + (int)(pending_at[i] % HP_W);
        if (orow < oh && ocol < ow) oplane[(size_t)orow * opitch + ocol] = cell;
      } else {
        shade[pending_at[i]] = cell;
      }
    }
  }
  if constexpr (DIRECT) return;
  __syncthreads();
  // 16 consecutive pixels of a row per lane and step
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int chunk = threadIdx.x; chunk < HP_H * (HP_W / 16); chunk += 256) {
    const int row = chunk / (HP_W / 16), part = chunk % (HP_W / 16);
    const int orow = r0 + row, ocol = c0 + 16 * part;
    if (orow < oh && ocol < ow) {
      unsigned char* dst = oplane + (size_t)orow * opitch + ocol;
      const unsigned char* srcb = shade + row * HP_W + 16 * part;
      if (ocol + 16 <= ow && (((uintptr_t)dst) & 15) == 0) {
        __builtin_nontemporal_store(*(const u32x4*)srcb, (u32x4*)dst);
      } else if (ocol + 16 <= ow) {
        // (rows of w - 2 bytes start on 16 bytes one time in eight: the same store, unaligned -- the hardware splits it --
        //  instead of sixteen 1-byte stores)
        typedef unsigned u32x4u __attribute__((ext_vector_type(4), aligned(1)));
        __builtin_nontemporal_store(*(const u32x4*)srcb, (u32x4u*)dst);
      } else {
        for (int k = 0; k < 16 && ocol + k < ow; ++k) dst[k] = srcb[k];
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Gaussian smoothing, one axis (Smooth.process, spatial.py:276-290 = scipy.ndimage.gaussian_filter)
// ---------------------------------------------------------------------------------
// SciPy's correlate1d for a symmetric kernel, literally: the line extended with the constant, in
// float64, out = x[0] w[0] + sum over j = r..1 of (x[-j] + x[+j]) w[j] (farthest pair first), cast to
// the tile's dtype.  The first axis also performs values[values == ndv] = fill (spatial.py:284-285).
// grid: the output is the window [y0, y0+oh) x [x0, x0+ow) of the filtered tile (the last axis of an "exact" Smooth
// crops its margins); z over bands.
constexpr int GAUSS_ROWS = 8;  // outputs per lane

// The taps from LDS, through a sliding register window (round 4).  A lane owns 8 consecutive outputs ALONG the axis
// (8 rows of a column for y, 8 columns of a row for x).  At distance j it needs lo[r] = x[c + r - j] and
// hi[r] = x[c + r + j], r = 0..7; at distance j - 1 seven of each are the neighbours' values of the step before
// (lo[r] <- lo[r + 1], hi[r] <- hi[r - 1]) -- so a step costs TWO LDS reads per lane, not sixteen, and the window can
// hold the float64 values SciPy's line buffer holds (no data -> fill, outside the line -> cval, converted once when
// staged) without the 8-byte reads becoming the bound.  The rotation is free: the loop over j is unrolled by 8 and
// the registers are renamed (a ring of 8).  Round 3 served every tap as its own LDS read while the float64 window
// fitted 64 KiB (radius 12 along y) and as its own global load after that: 895 us for the two axes of a 4096^2 tile at
// radius 52, 2 x 104 loads per output.
//   along y: a workgroup makes 8 G x COLS outputs (G = 256 / COLS row groups) from a (8 G + 2 r) x COLS window;
//   along x: 4 rows x 512 columns (a wave per row) from 4 x (512 + 2 r) cells, column c of a row at
//            (c & 7) S + (c >> 3): the lanes of a wave read cells 8 apart, which land on consecutive words.
// Weights from device memory (scalar loads), so LDS alone bounds the radius.
template <typename T, bool ALONG_X, int COLS>
__global__ __launch_bounds__(256) void gauss_axis_slide_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w, int pitch,
                                                               int y0, int x0, int oh, int ow, int replace_ndv, T ndv,
                                                               T fill, const double* __restrict__ wts, int radius,
                                                               double cval) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* const tile = reinterpret_cast<double*>(smem);
  const int tid = threadIdx.x, b = blockIdx.z;
  const T* plane = in + (size_t)b * h * pitch;
  auto line_value = [&](int sy, int sx) -> double {  // SciPy's extended line at (sy, sx); sy / sx may lie outside along the axis
    const int cy = min(max(sy, 0), h - 1), cx = min(max(sx, 0), w - 1);
    const T v = plane[(size_t)cy * pitch + cx];
    const double inside = (double)((replace_ndv && v == ndv) ? fill : v);
    const int along = ALONG_X ? sx : sy;
    return (along < 0 || along >= (ALONG_X ? w : h)) ? cval : inside;
  };
  constexpr int G = 256 / COLS;  // row groups of a workgroup (along y)
  const int lane = tid & 63, wave = tid >> 6;
  const int tx = tid % COLS, group = tid / COLS;
  const int S = (((512 + 2 * radius + 7) >> 3) | 1);  // (odd: the 8 residues of a row land on different banks)
  int ox0, oy0;
  if (ALONG_X) {
    ox0 = blockIdx.x * 512;
    oy0 = blockIdx.y * 4;
    const int win_cols = 512 + 2 * radius;
    for (int c = tid; c < win_cols; c += 256) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = line_value(y0 + oy0 + u, x0 + ox0 + c - radius);
#pragma unroll
      for (int u = 0; u < 4; ++u) tile[u * 8 * S + (c & 7) * S + (c >> 3)] = v[u];
    }
  } else {
    ox0 = blockIdx.x * COLS;
    oy0 = blockIdx.y * (GAUSS_ROWS * G);
    const int win_h = GAUSS_ROWS * G + 2 * radius;
    constexpr int SB = 8;  // window rows fetched together per lane
    for (int wy0 = group; wy0 < win_h; wy0 += G * SB) {
      double v[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) v[u] = line_value(y0 + oy0 + min(wy0 + G * u, win_h - 1) - radius, x0 + ox0 + tx);
#pragma unroll
      for (int u = 0; u < SB; ++u)
        if (wy0 + G * u < win_h) tile[(wy0 + G * u) * COLS + tx] = v[u];
    }
  }
  __syncthreads();
  // W(k): cell k of this lane's line through the window; output r of the lane has its centre at W(c0 + r)
  const double* const mine = ALONG_X ? tile + wave * 8 * S : tile + tx;
  auto W = [&](int k) -> double { return ALONG_X ? mine[(k & 7) * S + (k >> 3)] : mine[k * COLS]; };
  const int c0 = (ALONG_X ? 8 * lane : GAUSS_ROWS * group) + radius;
  if (ALONG_X ? (ox0 + 8 * lane >= ow || oy0 + wave >= oh) : (ox0 + tx >= ow)) return;
  double acc[GAUSS_ROWS], lo[GAUSS_ROWS], hi[GAUSS_ROWS];
  {
    const double w0 = wts[0];
#pragma unroll
    for (int r = 0; r < GAUSS_ROWS; ++r) acc[r] = W(c0 + r) * w0;
  }
  int j = radius;
  for (int odd = radius & 7; odd > 0; --odd, --j) {  // the steps that do not fill a round of 8: every tap its own read
    const double wj = wts[j];
#pragma unroll
    for (int r = 0; r < GAUSS_ROWS; ++r) acc[r] += (W(c0 + r - j) + W(c0 + r + j)) * wj;
  }
  if (j >= 8) {
#pragma unroll
    for (int r = 0; r < GAUSS_ROWS; ++r) {
      lo[r] = W(c0 + r - j);
      hi[r] = W(c0 + r + j);
    }
  }
  for (; j >= 8; j -= 8) {  // farthest pair first, as SciPy adds them
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int jj = j - s;
      const double wj = wts[jj];
#pragma unroll
      for (int r = 0; r < GAUSS_ROWS; ++r) acc[r] += (lo[(r + s) & 7] + hi[(r - s) & 7]) * wj;
      lo[s & 7] = W(c0 + 8 - jj);        // the new last `lo` of distance jj - 1 takes the place of the first of distance jj
      hi[(7 - s) & 7] = W(c0 + jj - 1);  // the new first `hi` takes the place of the last
    }
  }
  if (ALONG_X) {
    T* const row = out + ((size_t)b * oh + oy0 + wave) * ow + ox0 + 8 * lane;
    const int count = min(GAUSS_ROWS, ow - (ox0 + 8 * lane));
    constexpr int PER16 = 16 / (int)sizeof(T);
    if (count == GAUSS_ROWS && ((uintptr_t)row & 15) == 0 && PER16 <= GAUSS_ROWS) {
      T packed[GAUSS_ROWS];
#pragma unroll
      for (int r = 0; r < GAUSS_ROWS; ++r) packed[r] = (T)acc[r];
#pragma unroll
      for (int q = 0; q < GAUSS_ROWS / PER16; ++q)
        *reinterpret_cast<uint4*>(row + q * PER16) = *reinterpret_cast<const uint4*>(packed + q * PER16);
    } else {
#pragma unroll
      for (int r = 0; r < GAUSS_ROWS; ++r)
        if (r < count) row[r] = (T)acc[r];
    }
  } else {
    const int row0 = oy0 + GAUSS_ROWS * group;
#pragma unroll
    for (int r = 0; r < GAUSS_ROWS; ++r)
      if (row0 + r < oh) out[((size_t)b * oh + row0 + r) * ow + ox0 + tx] = (T)acc[r];
  }
}

// From memory, any radius (a window that does not fit LDS): the kernel of round 3 with its weights in device memory.
template <typename T, bool ALONG_X>
__global__ __launch_bounds__(256) void gauss_axis_far_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w, int pitch,
                                                             int y0, int x0, int oh, int ow, int replace_ndv, T ndv, T fill,
                                                             const double* __restrict__ wts, int radius, double cval) {
  const int ox = blockIdx.x * 256 + threadIdx.x, oy0 = blockIdx.y * GAUSS_ROWS, b = blockIdx.z;
  if (ox >= ow) return;
  const int x = x0 + ox;
  const T* plane = in + (size_t)b * h * pitch;
  const int n = ALONG_X ? w : h;
  auto fetch = [&](int r, int i) -> T {
    const int y = min(y0 + oy0 + r, h - 1);
    const int ic = min(max(i, 0), n - 1);
    return ALONG_X ? plane[(size_t)y * pitch + ic] : plane[(size_t)ic * pitch + x];
  };
  auto value = [&](T v, int i) -> double {
    const double inside = (double)((replace_ndv && v == ndv) ? fill : v);
    return (i < 0 || i >= n) ? cval : inside;
  };
  auto centre = [&](int r) { return ALONG_X ? x : y0 + oy0 + r; };
  double acc[GAUSS_ROWS];
  {
    T raw[GAUSS_ROWS];
#pragma unroll
    for (int r = 0; r < GAUSS_ROWS; ++r) raw[r] = fetch(r, centre(r));
    const double w0 = wts[0];
#pragma unroll
    for (int r = 0; r < GAUSS_ROWS; ++r) acc[r] = value(raw[r], centre(r)) * w0;
  }
  for (int j = radius; j >= 1; --j) {
    const double wj = wts[j];
    T raw_lo[GAUSS_ROWS], raw_hi[GAUSS_ROWS];
#pragma unroll
    for (int r = 0; r < GAUSS_ROWS; ++r) {
      raw_lo[r] = fetch(r, centre(r) - j);
      raw_hi[r] = fetch(r, centre(r) + j);
    }
#pragma unroll
    for (int r = 0; r < GAUSS_ROWS; ++r) acc[r] += (value(raw_lo[r], centre(r) - j) + value(raw_hi[r], centre(r) + j)) * wj;
  }
#pragma unroll
  for (int r = 0; r < GAUSS_ROWS; ++r)
    if (oy0 + r < oh) out[((size_t)b * oh + oy0 + r) * ow + ox] = (T)acc[r];
}

// Both axes of a float32 tile in ONE launch (round 2): the two passes above move 16 bytes per pixel
// through HBM and serve every tap from LDS as an 8-byte read (LDS-bound from radius ~6 on).  Here a
// workgroup makes G2_TH x TW outputs, TW = 256 - 2 RCAP:
//   y axis: a lane owns one of the TW + 2 R columns the x axis will read and walks down the rows with a
//     register window of 8 + 2 R float64 values (8 new rows loaded per step, every row read once per
//     workgroup): 8 outputs per step, rounded to float32 -- what the first pass stores -- into LDS;
//   x axis: a lane owns 8 consecutive outputs of a row: 8 + 2 R floats from LDS (16-byte reads) into a
//     register window, the same arithmetic, two 16-byte stores.
// The operations and their order are those of the one-axis kernels (SciPy's: centre tap, then the pairs
// from the farthest in); columns outside the tile are the x line's constant extension, not filtered values.
#ifndef GM_G2_TH
#define GM_G2_TH 32  // (tuning builds; 48 / 56 rows read less of the source twice and are slower: MEASUREMENTS 11.11, 11.17)
#endif
constexpr int G2_TH = GM_G2_TH, G2_MAX_RADIUS = 16;
struct Gauss2Args {
  double wy[G2_MAX_RADIUS + 1], wx[G2_MAX_RADIUS + 1];  // weight at distance j along y / x
  double cval;
};
template <int R, int RCAP>
__global__ __launch_bounds__(256) void gauss2d_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w,
                                                      int pitch, int y0, int x0, int oh, int ow, int opitch, int replace_ndv, float ndv,
                                                      float fill, Gauss2Args A) {
  constexpr int TW = 256 - 2 * RCAP, WIN = 8 + 2 * R, WIN4 = (WIN + 3) / 4;
  __shared__ __align__(16) float mid[G2_TH * 256 + 4];
  const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * G2_TH, b = blockIdx.z;
  const float* plane = in + (size_t)b * h * w;
  const int rows = min(G2_TH, oh - oy0);  // output rows of this workgroup
  {
    const int e = threadIdx.x;            // column of the extended tile
    const int sx = x0 + ox0 - R + e;
    if (e < TW + 2 * R && sx >= 0 && sx < w) {
      const float* column = plane + sx;
      // loads first, conversions after: written as one expression per row the compiler waits for every
      // load before it issues the next (rows are uniform, the address arithmetic is scalar code)
      auto load = [&](int ty) -> float { return column[(size_t)min(max(ty, 0), h - 1) * pitch]; };
      auto value = [&](float v, int ty) -> double {  // what SciPy's line buffer holds for tile row ty
        const double inside = (double)((replace_ndv && v == ndv) ? fill : v);
        return (ty < 0 || ty >= h) ? A.cval : inside;
      };
      const int top = y0 + oy0 - R;  // tile row of win[0]
      // Every row of the column this lane will need is requested before the first is used (round 5: G2_TH + 2 R loads in
      // flight per lane; with 8 rows requested a slab ahead a workgroup waited for memory five times in a row, and what the
      // compute unit's waves had in flight did not cover the latency at HBM's rate)
      constexpr int TOTAL = G2_TH + 2 * R;
      float raw[TOTAL];
#pragma unroll
      for (int i = 0; i < TOTAL; ++i) raw[i] = load(top + i);
      __builtin_amdgcn_sched_barrier(0);
      double win[WIN];
#pragma unroll
      for (int i = 0; i < 2 * R; ++i) win[i] = value(raw[i], top + i);
#pragma unroll
      for (int slab = 0; slab < G2_TH / 8; ++slab) {
        const int first = top + slab * 8 + 2 * R;
#pragma unroll
        for (int i = 0; i < 8; ++i) win[2 * R + i] = value(raw[2 * R + slab * 8 + i], first + i);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          double acc = win[r + R] * A.wy[0];
#pragma unroll
          for (int j = R; j >= 1; --j) acc += (win[r + R - j] + win[r + R + j]) * A.wy[j];
          mid[(slab * 8 + r) * 256 + e] = (float)acc;
        }
#pragma unroll
        for (int i = 0; i < 2 * R; ++i) win[i] = win[i + 8];
      }
    }
  }
  __syncthreads();
  const bool at_edge = x0 + ox0 - R < 0 || x0 + ox0 + TW + R > w;  // (uniform) some columns are the constant extension
  constexpr int SEGS = TW / 8;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  for (int item = threadIdx.x; item < rows * SEGS; item += 256) {
    const int r = item / SEGS, seg = item - r * SEGS;
    const int ox = ox0 + seg * 8;
    if (ox >= ow) continue;
    f32x4 raw[WIN4];
    const f32x4* src = reinterpret_cast<const f32x4*>(mid + r * 256 + seg * 8);
#pragma unroll
    for (int i = 0; i < WIN4; ++i) raw[i] = src[i];
    double win[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i) win[i] = (double)raw[i / 4][i % 4];
    if (at_edge) {
      const int sx = x0 + ox - R;  // tile column of win[0]
#pragma unroll
      for (int i = 0; i < WIN; ++i)
        if (sx + i < 0 || sx + i >= w) win[i] = A.cval;
    }
    float res[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double acc = win[k + R] * A.wx[0];
#pragma unroll
      for (int j = R; j >= 1; --j) acc += (win[k + R - j] + win[k + R + j]) * A.wx[j];
      res[k] = (float)acc;
    }
    float* o = out + (size_t)b * oh * ow + (size_t)(oy0 + r) * opitch + ox;
    if (ox + 8 <= ow) {
      // (the result is not read again: past the caches, like every other kernel's; rows of any alignment)
      typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
      __builtin_nontemporal_store(f32x4{res[0], res[1], res[2], res[3]}, (f32x4u*)o);
      __builtin_nontemporal_store(f32x4{res[4], res[5], res[6], res[7]}, (f32x4u*)(o + 4));
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)  // (static indices: a loop over `res` would move the array into LDS)
        if (ox + k < ow) o[k] = res[k];
    }
  }
}

template <typename T>
T from_bits(uint64_t bits) {
  gm_scalar s;
  s.u64 = bits;
  if constexpr (std::is_floating_point<T>::value) return (T)s.f64;
  else if constexpr (std::is_same<T, unsigned long long>::value) return (T)s.u64;
  else return (T)s.i64;
}

template <typename T>
T lowest() {
  if constexpr (std::is_floating_point<T>::value) return -std::numeric_limits<T>::max();
  else return std::numeric_limits<T>::min();
}

}  // namespace

// The *_into entry points: the result written with a row pitch of its own (a window of a mosaic).  They call the
// *_window launchers with this thread's output pitch set; a route whose kernel writes compact rows only declines.
static thread_local int t_out_pitch = 0;
static inline int out_pitch_or(int ow) { return t_out_pitch ? t_out_pitch : ow; }
#define GM_COMPACT_ONLY(what)                                                                                    \
  do {                                                                                                           \
    if (t_out_pitch) return gm::fail(GM_ERR_UNSUPPORTED, what ": this route writes compact tiles only");         \
  } while (0)

namespace {
// Gaussian weights in device memory, keyed by content (a Smooth block asks for the same one or two kernels all day)
std::mutex g_weights_mu;
std::map<std::pair<int, std::vector<double>>, void*> g_weights;

int cached_weights(const double* host, int count, const double** dev_out) {
  int dev = 0;
  GM_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_weights_mu);
  auto key = std::make_pair(dev, std::vector<double>(host, host + count));
  auto it = g_weights.find(key);
  if (it != g_weights.end()) {
    *dev_out = (const double*)it->second;
    return GM_OK;
  }
  if (g_weights.size() > 256) {  // bounded: drop everything once nothing in flight reads it
    hipStream_t s = gm::current_stream();
    if (s) (void)hipStreamSynchronize(s);
    for (auto& kv : g_weights) gm_free(kv.second);
    g_weights.clear();
  }
  void* buf = nullptr;
  int rc = gm_malloc(&buf, sizeof(double) * (size_t)count);
  if (rc) return rc;
  rc = gm_h2d(buf, key.second.data(), sizeof(double) * (size_t)count);  // (the key's copy outlives the asynchronous upload)
  if (rc) {
    gm_free(buf);
    return rc;
  }
  g_weights.emplace(std::move(key), buf);
  *dev_out = (const double*)buf;
  return GM_OK;
}

constexpr size_t kGaussLdsDefault = 64 * 1024, kGaussLdsMax = 160 * 1024;

template <typename T, bool ALONG_X, int COLS>
int launch_gauss_slide(size_t lds, const void* in, int pitch, void* out, int bands, int h, int w, int y0, int x0, int oh, int ow,
                       int replace_ndv, uint64_t ndv_bits, uint64_t fill_bits, const double* wts, int radius, double cval,
                       hipStream_t s) {
  auto kernel = gauss_axis_slide_kernel<T, ALONG_X, COLS>;
  if (lds > kGaussLdsDefault)  // (a table lookup in the runtime after the first time)
    GM_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGaussLdsMax));
  constexpr int rows_wg = ALONG_X ? 4 : GAUSS_ROWS * (256 / COLS), cols_wg = ALONG_X ? 512 : COLS;
  dim3 grid((unsigned)((ow + cols_wg - 1) / cols_wg), (unsigned)((oh + rows_wg - 1) / rows_wg), (unsigned)bands);
  hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, (const T*)in, (T*)out, h, w, pitch, y0, x0, oh, ow, replace_ndv,
                     from_bits<T>(ndv_bits), from_bits<T>(fill_bits), wts, radius, cval);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

template <typename T, bool ALONG_X>
int launch_gauss_axis(const void* in, int pitch, void* out, int bands, int h, int w, int y0, int x0, int oh, int ow, int replace_ndv,
                      uint64_t ndv_bits, uint64_t fill_bits, const double* wts, int radius, double cval, hipStream_t s) {
  static const bool from_memory = getenv("GM_GAUSS_FAR") && atoi(getenv("GM_GAUSS_FAR"));
  static const int force_cols = getenv("GM_GAUSS_COLS") ? atoi(getenv("GM_GAUSS_COLS")) : 0;
#define GM_SLIDE(COLS) \
  return launch_gauss_slide<T, ALONG_X, COLS>(lds, in, pitch, out, bands, h, w, y0, x0, oh, ow, replace_ndv, ndv_bits, fill_bits, wts, radius, cval, s)
  if (!from_memory) {
    if (ALONG_X) {
      const size_t lds = (size_t)4 * 8 * ((((size_t)512 + 2 * (size_t)radius + 7) >> 3) | 1) * sizeof(double);
      if (lds <= kGaussLdsMax) GM_SLIDE(64);
    } else {
      // 16 columns x 128 rows per workgroup from radius 8 on: the window is read (128 + 2 r) / 128 times over and
      // several workgroups share a CU (y pass of a 4096^2 float32 tile at radius 52: 64 columns 217 us, 32: 158, 16: 135;
      // radius 104: 562 / 305 / 244; below radius 8 the three are equal and 64 columns read whole 256-byte rows)
      const int want = force_cols ? force_cols : (radius >= 8 ? 16 : 64);
      const size_t lds = ((size_t)GAUSS_ROWS * (256 / want) + 2 * (size_t)radius) * want * sizeof(double);
      if (lds <= kGaussLdsMax) {
        if (want == 64) GM_SLIDE(64);
        if (want == 32) GM_SLIDE(32);
        GM_SLIDE(16);
      }
    }
  }
#undef GM_SLIDE
  dim3 grid((unsigned)((ow + 255) / 256), (unsigned)((oh + GAUSS_ROWS - 1) / GAUSS_ROWS), (unsigned)bands);
  hipLaunchKernelGGL((gauss_axis_far_kernel<T, ALONG_X>), grid, dim3(256), 0, s, (const T*)in, (T*)out, h, w, pitch, y0, x0, oh,
                     ow, replace_ndv, from_bits<T>(ndv_bits), from_bits<T>(fill_bits), wts, radius, cval);
  GM_HIP(hipGetLastError());
  return GM_OK;
}
}  // namespace

extern "C" {

int gm_dilate(const void* in, int dtype, int bands, int h, int w, const uint64_t* values_bits, int n_values, void* out) {
  return gm_dilate_window(in, dtype, bands, h, w, w, values_bits, n_values, out);
}

int gm_dilate_window(const void* in, int dtype, int bands, int h, int w, int in_pitch, const uint64_t* values_bits, int n_values,
                     void* out) {
  if (!in || !out || (!values_bits && n_values)) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (in_pitch < w) return gm::fail(GM_ERR_INVALID, "gm_dilate: pitch %d of a %d-wide tile", in_pitch, w);
  const bool window = in_pitch != w;
  if (window && bands != 1) return gm::fail(GM_ERR_UNSUPPORTED, "gm_dilate: a window of %d bands", bands);
  if (bands <= 0 || h < 3 || w < 3) return gm::fail(GM_ERR_INVALID, "gm_dilate: tile %dx%dx%d has no interior", bands, h, w);
  if (n_values < 0) return gm::fail(GM_ERR_INVALID, "gm_dilate: %d values", n_values);
  if (bands > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "gm_dilate: too many bands");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  if (gm::dtype_size(dtype) == 1 && n_values >= 1 && n_values <= DIL8_MAX_VALUES) {
    Dilate8Args A8;
    A8.n_values = n_values;
    for (int i = 0; i < DIL8_MAX_VALUES; ++i) A8.splat[i] = i < n_values ? 0x01010101u * (unsigned)(values_bits[i] & 0xffu) : 0u;
    // (one band: with several, the cells of the neighbouring bands count too -- the row kernel below)
    static int d8h = 0;  // output rows per workgroup (GM_D8W_H: tuning runs)
    if (!d8h) {
      const char* e = getenv("GM_D8W_H");
      d8h = e ? atoi(e) : 8;  // (8192^2, 2 values: 16 rows 41 us, 8: 38, 4: 36; with a margin 31.6 / 29.4 / 29.9)
      if (d8h != 4 && d8h != 8 && d8h != 32) d8h = 16;
    }
    if (bands == 1 && w >= 64 && (h - 2 + d8h - 1) / d8h <= 65535) {
      dim3 wgrid((unsigned)((w - 2 + D8W_COLS - 1) / D8W_COLS), (unsigned)((h - 2 + d8h - 1) / d8h));
    gm::route_hit(GM_ROUTE_STENCILS, GM_RS_DILATE8_WIDE);
#define GM_D8W(H_) hipLaunchKernelGGL(dilate8_wide_kernel<H_>, wgrid, dim3(256), 0, s, (const unsigned char*)in, (unsigned char*)out, h, w, in_pitch, out_pitch_or(w - 2), A8)
      if (d8h == 4) GM_D8W(4);
      else if (d8h == 8) GM_D8W(8);
      else if (d8h == 32) GM_D8W(32);
      else GM_D8W(16);
#undef GM_D8W
      GM_HIP(hipGetLastError());
      return GM_OK;
    }
    if (window) return gm::fail(GM_ERR_UNSUPPORTED, "gm_dilate: this tile shape wants a compact tile");
    GM_COMPACT_ONLY("gm_dilate");
    dim3 grid8((unsigned)((w - 2 + 1023) / 1024), (unsigned)((h - 2 + D8_ROWS - 1) / D8_ROWS), (unsigned)bands);
    if (grid8.y > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "gm_dilate: tile taller than %d rows", 65535 * D8_ROWS);
    gm::route_hit(GM_ROUTE_STENCILS, GM_RS_DILATE_OTHER);
    hipLaunchKernelGGL(dilate8_kernel, grid8, dim3(256), 0, s, (const unsigned char*)in, (unsigned char*)out, bands, h, w, A8);
    GM_HIP(hipGetLastError());
    return GM_OK;
  }
  dim3 grid((unsigned)((w - 2 + 255) / 256), (unsigned)((h - 2 + ST_ROWS - 1) / ST_ROWS), (unsigned)bands);
  const int cell = gm::dtype_size(dtype);
  if (n_values > DIL_MAX_VALUES) {
    // the reference takes any number of values (spatial.py:147-154): groups of DIL_MAX_VALUES, every group testing
    // the original tile and writing on top of what the groups before it left in `out`
    GM_COMPACT_ONLY("gm_dilate");
#define GM_DILC(T)                                                                                                     \
  do {                                                                                                                 \
    for (int v0 = 0; v0 < n_values; v0 += DIL_MAX_VALUES) {                                                            \
      DilateArgs<T> A;                                                                                                 \
      A.n_values = n_values - v0 < DIL_MAX_VALUES ? n_values - v0 : DIL_MAX_VALUES;                                    \
      for (int i = 0; i < A.n_values; ++i) A.values[i] = from_bits<T>(values_bits[v0 + i]);                            \
      if (v0 == 0) hipLaunchKernelGGL((dilate_kernel<T, false>), grid, dim3(256), 0, s, (const T*)in, (T*)out, bands, h, w, in_pitch, A); \
      else hipLaunchKernelGGL((dilate_kernel<T, true>), grid, dim3(256), 0, s, (const T*)in, (T*)out, bands, h, w, in_pitch, A); \
    }                                                                                                                  \
  } while (0)
    switch (cell) {  // (values are compared by bit pattern for integers, by == for floats)
      case 1: GM_DILC(unsigned char); break;
      case 2: GM_DILC(unsigned short); break;
      case 4: if (dtype == GM_F32) GM_DILC(float); else GM_DILC(unsigned int); break;
      case 8: if (dtype == GM_F64) GM_DILC(double); else GM_DILC(unsigned long long); break;
      default: return gm::fail(GM_ERR_INVALID, "gm_dilate: dtype %d", dtype);
    }
#undef GM_DILC
    GM_HIP(hipGetLastError());
    return GM_OK;
  }
  if (bands == 1 && (cell == 2 || cell == 4) && (h - 2 + DT_H - 1) / DT_H <= 65535) {
    dim3 tgrid((unsigned)((w - 2 + DT_W - 1) / DT_W), (unsigned)((h - 2 + DT_H - 1) / DT_H));
#define GM_DILT(T)                                                                                  \
  do {                                                                                              \
    DilateArgs<T> A;                                                                                \
    A.n_values = n_values;                                                                          \
    for (int i = 0; i < n_values; ++i) A.values[i] = from_bits<T>(values_bits[i]);                  \
    hipLaunchKernelGGL(dilate_tile_kernel<T>, tgrid, dim3(256), 0, s, (const T*)in, (T*)out, h, w, in_pitch, out_pitch_or(w - 2), A); \
  } while (0)
    switch (dtype) {
      case GM_U16: GM_DILT(unsigned short); break;
      case GM_I16: GM_DILT(short); break;
      case GM_U32: GM_DILT(unsigned int); break;
      case GM_I32: GM_DILT(int); break;
      case GM_F32: GM_DILT(float); break;
      default: return gm::fail(GM_ERR_INVALID, "gm_dilate: dtype %d", dtype);
    }
#undef GM_DILT
    GM_HIP(hipGetLastError());
    return GM_OK;
  }
  GM_COMPACT_ONLY("gm_dilate");
#define GM_DIL(T)                                                                                    \
  do {                                                                                               \
    DilateArgs<T> A;                                                                                 \
    A.n_values = n_values;                                                                           \
    for (int i = 0; i < n_values; ++i) A.values[i] = from_bits<T>(values_bits[i]);                   \
    hipLaunchKernelGGL(dilate_kernel<T>, grid, dim3(256), 0, s, (const T*)in, (T*)out, bands, h, w, in_pitch, A); \
  } while (0)
  switch (dtype) {
    case GM_BOOL: case GM_U8: GM_DIL(unsigned char); break;
    case GM_I8: GM_DIL(signed char); break;
    case GM_U16: GM_DIL(unsigned short); break;
    case GM_I16: GM_DIL(short); break;
    case GM_U32: GM_DIL(unsigned int); break;
    case GM_I32: GM_DIL(int); break;
    case GM_U64: GM_DIL(unsigned long long); break;
    case GM_I64: GM_DIL(long long); break;
    case GM_F32: GM_DIL(float); break;
    case GM_F64: GM_DIL(double); break;
    default: return gm::fail(GM_ERR_INVALID, "gm_dilate: dtype %d", dtype);
  }
#undef GM_DIL
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_moving_max(const void* in, int dtype, int bands, int h, int w, int size, int has_ndv, uint64_t ndv_bits, void* out) {
  return gm_moving_max_window(in, dtype, bands, h, w, w, size, has_ndv, ndv_bits, out);
}

int gm_moving_max_window(const void* in, int dtype, int bands, int h, int w, int in_pitch, int size, int has_ndv,
                         uint64_t ndv_bits, void* out) {
  if (!in || !out) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (in_pitch < w) return gm::fail(GM_ERR_INVALID, "gm_moving_max: pitch %d of a %d-wide tile", in_pitch, w);
  const bool window = in_pitch != w;
  if (window && bands != 1) return gm::fail(GM_ERR_UNSUPPORTED, "gm_moving_max: a window of %d bands", bands);
  if (size < 3 || (size & 1) == 0 || size > 63) return gm::fail(GM_ERR_UNSUPPORTED, "gm_moving_max: size %d (odd, 3..63)", size);
  const int radius = size / 2;
  const int oh = h - 2 * radius, ow = w - 2 * radius;
  if (bands <= 0 || oh <= 0 || ow <= 0) return gm::fail(GM_ERR_INVALID, "gm_moving_max: tile smaller than the footprint");
  if (bands > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "gm_moving_max: too many bands");
  if ((size_t)(MM_TILE_H + 2 * radius) * (MM_TILE_W + 2 * radius) * (size_t)gm::dtype_size(dtype) > 65536)
    return gm::fail(GM_ERR_UNSUPPORTED, "gm_moving_max: the window of a size-%d footprint does not fit 64 KiB of LDS for this dtype", size);
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  if (size == 3 && gm::dtype_size(dtype) <= 4 && (oh + M3_H - 1) / M3_H <= 65535) {
    dim3 grid3((unsigned)((ow + M3_W - 1) / M3_W), (unsigned)((oh + M3_H - 1) / M3_H), (unsigned)bands);
    gm::route_hit(GM_ROUTE_STENCILS, GM_RS_MOVINGMAX_3X3);
#define GM_M3(T) hipLaunchKernelGGL(moving_max3_kernel<T>, grid3, dim3(256), 0, s, (const T*)in, (T*)out, h, w, in_pitch, out_pitch_or(ow), has_ndv, from_bits<T>(ndv_bits), lowest<T>())
    switch (dtype) {
      case GM_U8: GM_M3(unsigned char); break;
      case GM_I8: GM_M3(signed char); break;
      case GM_U16: GM_M3(unsigned short); break;
      case GM_I16: GM_M3(short); break;
      case GM_U32: GM_M3(unsigned int); break;
      case GM_I32: GM_M3(int); break;
      case GM_F32: GM_M3(float); break;
      default: return gm::fail(GM_ERR_INVALID, "gm_moving_max: dtype %d", dtype);
    }
#undef GM_M3
    GM_HIP(hipGetLastError());
    return GM_OK;
  }
  if (size >= 5 && size <= 15 && gm::dtype_size(dtype) <= 4 && (oh + 15) / 16 <= 65535) {
    const int th = disk_tile_rows(radius);
    gm::route_hit(GM_ROUTE_STENCILS, GM_RS_MOVINGMAX_STACK);
    dim3 gridd((unsigned)((ow + 255) / 256), (unsigned)((oh + th - 1) / th), (unsigned)bands);
#define GM_MD(T)                                                                                                  \
  do {                                                                                                            \
    if (size == 5)                                                                                                \
      hipLaunchKernelGGL((moving_max_disk_kernel<T, 2>), gridd, dim3(256), 0, s, (const T*)in, (T*)out, h, w,     \
                         in_pitch, out_pitch_or(ow), has_ndv, from_bits<T>(ndv_bits), lowest<T>());                        \
    else if (size == 7)                                                                                           \
      hipLaunchKernelGGL((moving_max_disk_kernel<T, 3>), gridd, dim3(256), 0, s, (const T*)in, (T*)out, h, w,     \
                         in_pitch, out_pitch_or(ow), has_ndv, from_bits<T>(ndv_bits), lowest<T>());                        \
    else if (size == 9)                                                                                           \
      hipLaunchKernelGGL((moving_max_disk_kernel<T, 4>), gridd, dim3(256), 0, s, (const T*)in, (T*)out, h, w,     \
                         in_pitch, out_pitch_or(ow), has_ndv, from_bits<T>(ndv_bits), lowest<T>());                        \
    else if (size == 11)                                                                                          \
      hipLaunchKernelGGL((moving_max_disk_kernel<T, 5>), gridd, dim3(256), 0, s, (const T*)in, (T*)out, h, w,     \
                         in_pitch, out_pitch_or(ow), has_ndv, from_bits<T>(ndv_bits), lowest<T>());                        \
    else if (size == 13)                                                                                          \
      hipLaunchKernelGGL((moving_max_disk_kernel<T, 6>), gridd, dim3(256), 0, s, (const T*)in, (T*)out, h, w,     \
                         in_pitch, out_pitch_or(ow), has_ndv, from_bits<T>(ndv_bits), lowest<T>());                        \
    else                                                                                                          \
      hipLaunchKernelGGL((moving_max_disk_kernel<T, 7>), gridd, dim3(256), 0, s, (const T*)in, (T*)out, h, w,     \
                         in_pitch, out_pitch_or(ow), has_ndv, from_bits<T>(ndv_bits), lowest<T>());                        \
  } while (0)
    switch (dtype) {
      case GM_U8: GM_MD(unsigned char); break;
      case GM_I8: GM_MD(signed char); break;
      case GM_U16: GM_MD(unsigned short); break;
      case GM_I16: GM_MD(short); break;
      case GM_U32: GM_MD(unsigned int); break;
      case GM_I32: GM_MD(int); break;
      case GM_F32: GM_MD(float); break;
      default: return gm::fail(GM_ERR_INVALID, "gm_moving_max: dtype %d", dtype);
    }
#undef GM_MD
    GM_HIP(hipGetLastError());
    return GM_OK;
  }
  if (window) return gm::fail(GM_ERR_UNSUPPORTED, "gm_moving_max: this footprint / dtype wants a compact tile");
  GM_COMPACT_ONLY("gm_moving_max");
  gm::route_hit(GM_ROUTE_STENCILS, GM_RS_MOVINGMAX_COLUMNS);
  dim3 grid((unsigned)((ow + MM_TILE_W - 1) / MM_TILE_W), (unsigned)((oh + MM_TILE_H - 1) / MM_TILE_H), (unsigned)bands);
  if (grid.y > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "gm_moving_max: tile too tall");
#define GM_MM_R(T, RMAX_)                                                                                      \
  hipLaunchKernelGGL((moving_max_kernel<T, RMAX_>), grid, dim3(256),                                           \
                     (size_t)(MM_TILE_H + 2 * radius) * (MM_TILE_W + 2 * radius) * sizeof(T), s, (const T*)in,   \
                     (T*)out, h, w, size, has_ndv, from_bits<T>(ndv_bits), lowest<T>())
#define GM_MM(T)                           \
  do {                                     \
    if (radius <= 3) GM_MM_R(T, 3);        \
    else if (radius <= 7) GM_MM_R(T, 7);   \
    else if (radius <= 15) GM_MM_R(T, 15); \
    else GM_MM_R(T, 31);                   \
  } while (0)
  switch (dtype) {
    case GM_U8: GM_MM(unsigned char); break;
    case GM_I8: GM_MM(signed char); break;
    case GM_U16: GM_MM(unsigned short); break;
    case GM_I16: GM_MM(short); break;
    case GM_U32: GM_MM(unsigned int); break;
    case GM_I32: GM_MM(int); break;
    case GM_U64: GM_MM(unsigned long long); break;
    case GM_I64: GM_MM(long long); break;
    case GM_F32: GM_MM(float); break;
    case GM_F64: GM_MM(double); break;
    default: return gm::fail(GM_ERR_INVALID, "gm_moving_max: dtype %d", dtype);
  }
#undef GM_MM
#undef GM_MM_R
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_gauss_axis_window(const void* in, int dtype, int bands, int h, int w, int in_pitch, int axis, const double* weights_host, int radius,
                  double cval, int replace_ndv, uint64_t ndv_bits, uint64_t fill_bits, int y0, int x0, int oh, int ow,
                  void* out) {
  if (!in || !out || !weights_host) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (gm::dtype_size(dtype) == 0) return gm::fail(GM_ERR_INVALID, "gm_gauss_axis: dtype %d", dtype);
  if (in_pitch < w) return gm::fail(GM_ERR_INVALID, "gm_gauss_axis: pitch %d of a %d-wide tile", in_pitch, w);
  if (in_pitch != w && bands != 1) return gm::fail(GM_ERR_UNSUPPORTED, "gm_gauss_axis: a window of %d bands", bands);
  if (axis != 1 && axis != 2) return gm::fail(GM_ERR_INVALID, "gm_gauss_axis: axis %d (1 = y, 2 = x)", axis);
  if (radius < 0 || radius > (1 << 24)) return gm::fail(GM_ERR_INVALID, "gm_gauss_axis: radius %d", radius);
  if (bands <= 0 || oh <= 0 || ow <= 0) return GM_OK;
  if (y0 < 0 || x0 < 0 || y0 + oh > h || x0 + ow > w) return gm::fail(GM_ERR_INVALID, "gm_gauss_axis: output window outside the tile");
  if (bands > 65535 || oh > 65535 * 4) return gm::fail(GM_ERR_UNSUPPORTED, "gm_gauss_axis: tile too tall / too many bands");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  const double* wts = nullptr;
  int rc = cached_weights(weights_host, radius + 1, &wts);
  if (rc) return rc;
#define GM_GAUSS(T)                                                                                                       \
  return axis == 2 ? launch_gauss_axis<T, true>(in, in_pitch, out, bands, h, w, y0, x0, oh, ow, replace_ndv, ndv_bits, fill_bits, wts, \
                                                radius, cval, s)                                                          \
                   : launch_gauss_axis<T, false>(in, in_pitch, out, bands, h, w, y0, x0, oh, ow, replace_ndv, ndv_bits, fill_bits, wts, \
                                                 radius, cval, s)
  // integer tiles: SciPy filters in float64 and stores with a C cast (truncation), measured against
  // scipy.ndimage.gaussian_filter(output=<integer array>)
  switch (dtype) {
    case GM_F32: GM_GAUSS(float);
    case GM_F64: GM_GAUSS(double);
    case GM_BOOL:  // (SciPy stores into a bool array with the C cast to npy_bool = unsigned char: truncation)
    case GM_U8: GM_GAUSS(unsigned char);
    case GM_I8: GM_GAUSS(signed char);
    case GM_U16: GM_GAUSS(unsigned short);
    case GM_I16: GM_GAUSS(short);
    case GM_U32: GM_GAUSS(unsigned int);
    case GM_I32: GM_GAUSS(int);
    case GM_U64: GM_GAUSS(unsigned long long);
    case GM_I64: GM_GAUSS(long long);
    default: return gm::fail(GM_ERR_INVALID, "gm_gauss_axis: dtype %d", dtype);
  }
#undef GM_GAUSS
}

int gm_gauss_axis(const void* in, int dtype, int bands, int h, int w, int axis, const double* weights_host, int radius,
                  double cval, int replace_ndv, uint64_t ndv_bits, uint64_t fill_bits, int y0, int x0, int oh, int ow,
                  void* out) {
  return gm_gauss_axis_window(in, dtype, bands, h, w, w, axis, weights_host, radius, cval, replace_ndv, ndv_bits, fill_bits, y0,
                              x0, oh, ow, out);
}

int gm_gauss_2d(const void* in, int dtype, int bands, int h, int w, const double* weights_y, const double* weights_x,
                int radius, double cval, int replace_ndv, uint64_t ndv_bits, uint64_t fill_bits, int y0, int x0, int oh,
                int ow, void* out) {
  return gm_gauss_2d_window(in, dtype, bands, h, w, w, weights_y, weights_x, radius, cval, replace_ndv, ndv_bits, fill_bits, y0, x0,
                            oh, ow, out);
}

int gm_gauss_2d_window(const void* in, int dtype, int bands, int h, int w, int in_pitch, const double* weights_y,
                       const double* weights_x, int radius, double cval, int replace_ndv, uint64_t ndv_bits, uint64_t fill_bits,
                       int y0, int x0, int oh, int ow, void* out) {
  if (!in || !out || !weights_y || !weights_x) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (in_pitch < w) return gm::fail(GM_ERR_INVALID, "gm_gauss_2d: pitch %d of a %d-wide tile", in_pitch, w);
  if (in_pitch != w && bands != 1) return gm::fail(GM_ERR_UNSUPPORTED, "gm_gauss_2d: a window of %d bands", bands);
  if (dtype != GM_F32) return gm::fail(GM_ERR_UNSUPPORTED, "gm_gauss_2d: dtype %d (float32 tiles)", dtype);
  if (radius < 1 || radius > G2_MAX_RADIUS) return gm::fail(GM_ERR_UNSUPPORTED, "gm_gauss_2d: radius %d (1..%d)", radius, G2_MAX_RADIUS);
  if (bands <= 0 || oh <= 0 || ow <= 0) return GM_OK;
  if (y0 < 0 || x0 < 0 || y0 + oh > h || x0 + ow > w) return gm::fail(GM_ERR_INVALID, "gm_gauss_2d: output window outside the tile");
  if (bands > 65535 || (oh + G2_TH - 1) / G2_TH > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "gm_gauss_2d: tile too tall / too many bands");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  Gauss2Args A;
  memset(&A, 0, sizeof(A));
  for (int j = 0; j <= radius; ++j) {
    A.wy[j] = weights_y[j];
    A.wx[j] = weights_x[j];
  }
  A.cval = cval;
  const float ndv = from_bits<float>(ndv_bits), fill = from_bits<float>(fill_bits);
  const int tw = 256 - 2 * (radius <= 8 ? 8 : 16);
  dim3 grid((unsigned)((ow + tw - 1) / tw), (unsigned)((oh + G2_TH - 1) / G2_TH), (unsigned)bands);
  gm::route_hit(GM_ROUTE_STENCILS, GM_RS_GAUSS_ONE_LAUNCH);
#define GM_G2(R, RCAP)                                                                                              \
  case R:                                                                                                           \
    hipLaunchKernelGGL((gauss2d_kernel<R, RCAP>), grid, dim3(256), 0, s, (const float*)in, (float*)out, h, w, in_pitch,   \
                       y0, x0, oh, ow, out_pitch_or(ow), replace_ndv, ndv, fill, A);                                \
    break
  switch (radius) {
    GM_G2(1, 8); GM_G2(2, 8); GM_G2(3, 8); GM_G2(4, 8); GM_G2(5, 8); GM_G2(6, 8); GM_G2(7, 8); GM_G2(8, 8);
    GM_G2(9, 16); GM_G2(10, 16); GM_G2(11, 16); GM_G2(12, 16); GM_G2(13, 16); GM_G2(14, 16); GM_G2(15, 16); GM_G2(16, 16);
  }
#undef GM_G2
  GM_HIP(hipGetLastError());
  return GM_OK;
}

// the scalars of spatial.py:370-383 as the kernels take them
static HillArgs hill_args(double xres, double yres, double altitude_deg, double azimuth_deg) {
  const double pi = 3.141592653589793;
  const double alt = altitude_deg * (pi / 180.0), az = azimuth_deg * (pi / 180.0);  // math.radians
  const double zsf = 1.0 / 8.0;
  HillArgs A;
  A.xres = xres;
  A.yres = yres;
  A.xres32 = (float)xres;
  A.yres32 = (float)yres;
  auto exact_inverse = [](double r) -> double {  // 1 / r when r is a power of two (normal, as its inverse), else 0
    int e;
    const double inv = 1.0 / r;
    return (std::isnormal(r) && std::isnormal(inv) && frexp(fabs(r), &e) == 0.5) ? inv : 0.0;
  };
  A.xinv = exact_inverse(xres);
  A.yinv = exact_inverse(yres);
  A.xinv32 = (float)exact_inverse((double)A.xres32);  // (a power of two in float32 is one in float64 and vice versa
  A.yinv32 = (float)exact_inverse((double)A.yres32);  //  within float32's exponent range; outside, the cast gives 0 / inf)
  if (!std::isnormal(A.xinv32)) A.xinv32 = 0.0f;
  if (!std::isnormal(A.yinv32)) A.yinv32 = 0.0f;
  A.sin_alt = (float)sin(alt);              // python floats meet float32 arrays as weak scalars:
  A.cos_alt_zsf = (float)(cos(alt) * zsf);  // the product of python floats is rounded to float32 once
  A.square_zsf = (float)(zsf * zsf);
  A.cos_az = (float)cos(az);
  A.sin_az = (float)sin(az);
  A.az32 = (float)az;
  return A;
}

int gm_hillshade(const void* in, int dtype, int bands, int h, int w, int has_ndv, uint64_t ndv_bits, uint64_t fill_bits,
                 double xres, double yres, double altitude_deg, double azimuth_deg, void* out) {
  return gm_hillshade_window(in, dtype, bands, h, w, w, has_ndv, ndv_bits, fill_bits, xres, yres, altitude_deg, azimuth_deg, out);
}

int gm_hillshade_window(const void* in, int dtype, int bands, int h, int w, int in_pitch, int has_ndv, uint64_t ndv_bits,
                        uint64_t fill_bits, double xres, double yres, double altitude_deg, double azimuth_deg, void* out) {
  if (!in || !out) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (in_pitch < w) return gm::fail(GM_ERR_INVALID, "gm_hillshade: pitch %d of a %d-wide tile", in_pitch, w);
  const bool window = in_pitch != w;
  if (window && bands != 1) return gm::fail(GM_ERR_UNSUPPORTED, "gm_hillshade: a window of %d bands", bands);
  if (bands <= 0 || h < 3 || w < 3) return gm::fail(GM_ERR_INVALID, "gm_hillshade: tile %dx%dx%d has no interior", bands, h, w);
  if (bands > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "gm_hillshade: too many bands");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  const HillArgs A = hill_args(xres, yres, altitude_deg, azimuth_deg);
  dim3 grid((unsigned)((w - 2 + 255) / 256), (unsigned)((h - 2 + ST_ROWS - 1) / ST_ROWS), (unsigned)bands);
  dim3 tgrid((unsigned)((w - 2 + HT_W - 1) / HT_W), (unsigned)((h - 2 + HT_H - 1) / HT_H), (unsigned)bands);
  const bool tiled = gm::dtype_size(dtype) <= 4 && tgrid.y <= 65535;  // (8-byte tiles: the window would not fit LDS twice)
  const bool pairs_fit = !getenv("GM_HS_SINGLE") && (h - 2 + HP_H - 1) / HP_H <= 65535;
  dim3 pgrid((unsigned)((w - 2 + HP_W - 1) / HP_W), (unsigned)((h - 2 + HP_H - 1) / HP_H), (unsigned)bands);
  gm::route_hit(GM_ROUTE_STENCILS, GM_RS_CALLS);
  if (dtype == GM_F32 && pairs_fit) {
    gm::route_hit(GM_ROUTE_STENCILS, GM_RS_HILLSHADE_PAIRS);
    const float ndv = from_bits<float>(ndv_bits), fill = from_bits<float>(fill_bits);
    static const bool direct_stores = getenv("GM_HS_DIRECT") && atoi(getenv("GM_HS_DIRECT")) != 0;
    if (direct_stores && A.xinv32 != 0.0f && A.yinv32 != 0.0f)
      hipLaunchKernelGGL((hillshade_pair_kernel<float, true, false, true>), pgrid, dim3(256), 0, s, (const float*)in, (unsigned char*)out, h, w, in_pitch, has_ndv, ndv, fill, A);
    else if (A.xinv32 != 0.0f && A.yinv32 != 0.0f)
      hipLaunchKernelGGL((hillshade_pair_kernel<float, true>), pgrid, dim3(256), 0, s, (const float*)in, (unsigned char*)out, h, w, in_pitch, has_ndv, ndv, fill, A);
    else
      hipLaunchKernelGGL((hillshade_pair_kernel<float, false>), pgrid, dim3(256), 0, s, (const float*)in, (unsigned char*)out, h, w, in_pitch, has_ndv, ndv, fill, A);
    GM_HIP(hipGetLastError());
    return GM_OK;
  }
  // integer tiles of up to 4 bytes at power-of-two resolutions (well inside float32's exponent range): the same kernel
  const bool int_pairs = pairs_fit && A.xinv32 != 0.0f && A.yinv32 != 0.0f && fabsf(A.xinv32) < 0x1p64f &&
                         fabsf(A.xinv32) > 0x1p-64f && fabsf(A.yinv32) < 0x1p64f && fabsf(A.yinv32) > 0x1p-64f;
  if (int_pairs && gm::dtype_size(dtype) <= 4 && !gm::dtype_is_float(dtype) && dtype != GM_BOOL) {
    gm::route_hit(GM_ROUTE_STENCILS, GM_RS_HILLSHADE_PAIRS);
#define GM_HP(T)                                                                                                          \
  hipLaunchKernelGGL((hillshade_pair_kernel<T, true>), pgrid, dim3(256), 0, s, (const T*)in, (unsigned char*)out, h, w, in_pitch, \
                     has_ndv, from_bits<T>(ndv_bits), from_bits<T>(fill_bits), A)
    switch (dtype) {
      case GM_U8: GM_HP(unsigned char); break;
      case GM_I8: GM_HP(signed char); break;
      case GM_U16: GM_HP(unsigned short); break;
      case GM_I16: GM_HP(short); break;
      case GM_U32: GM_HP(unsigned int); break;
      default: GM_HP(int); break;
    }
#undef GM_HP
    GM_HIP(hipGetLastError());
    return GM_OK;
  }
  gm::route_hit(GM_ROUTE_STENCILS, GM_RS_HILLSHADE_SINGLE);
#define GM_HS(T)                                                                                                     \
  do {                                                                                                               \
    if constexpr (sizeof(T) <= 4) {                                                                                  \
      if (tiled) {                                                                                                   \
        hipLaunchKernelGGL(hillshade_tile_kernel<T>, tgrid, dim3(256), 0, s, (const T*)in, (unsigned char*)out, h, w, \
                           in_pitch, has_ndv, from_bits<T>(ndv_bits), from_bits<T>(fill_bits), A);                  \
        break;                                                                                                       \
      }                                                                                                              \
    }                                                                                                                \
    if (window) return gm::fail(GM_ERR_UNSUPPORTED, "gm_hillshade: this dtype / tile shape wants a compact tile");  \
    hipLaunchKernelGGL(hillshade_kernel<T>, grid, dim3(256), 0, s, (const T*)in, (unsigned char*)out, h, w,         \
                       has_ndv, from_bits<T>(ndv_bits), from_bits<T>(fill_bits), A);                                \
  } while (0)
  switch (dtype) {
    case GM_U8: GM_HS(unsigned char); break;
    case GM_I8: GM_HS(signed char); break;
    case GM_U16: GM_HS(unsigned short); break;
    case GM_I16: GM_HS(short); break;
    case GM_U32: GM_HS(unsigned int); break;
    case GM_I32: GM_HS(int); break;
    case GM_I64: GM_HS(long long); break;
    case GM_U64: GM_HS(unsigned long long); break;  // (sums wrap modulo 2^64 like NumPy's; the quotient is float64)
    case GM_F32: GM_HS(float); break;
    case GM_F64: GM_HS(double); break;
    default: return gm::fail(GM_ERR_INVALID, "gm_hillshade: dtype %d", dtype);
  }
#undef GM_HS
  GM_HIP(hipGetLastError());
  return GM_OK;
}

// ---- results written with a row pitch of their own (one band; the tile's window of a mosaic) ----
#define GM_INTO(CALL)                                                                                         \
  do {                                                                                                        \
    if (!out) return gm::fail(GM_ERR_INVALID, "null pointer");                                                \
    if (bands != 1) return gm::fail(GM_ERR_UNSUPPORTED, "an output pitch needs a single band");               \
    if (out_pitch < min_pitch) return gm::fail(GM_ERR_INVALID, "output pitch %d of a %d-wide result", out_pitch, min_pitch); \
    t_out_pitch = out_pitch;                                                                                  \
    const int rc_ = (CALL);                                                                                   \
    t_out_pitch = 0;                                                                                          \
    return rc_;                                                                                               \
  } while (0)

int gm_dilate_into(const void* in, int dtype, int bands, int h, int w, int in_pitch, const uint64_t* values_bits, int n_values,
                   void* out, int out_pitch) {
  const int min_pitch = w - 2;
  GM_INTO(gm_dilate_window(in, dtype, bands, h, w, in_pitch, values_bits, n_values, out));
}

int gm_moving_max_into(const void* in, int dtype, int bands, int h, int w, int in_pitch, int size, int has_ndv,
                       uint64_t ndv_bits, void* out, int out_pitch) {
  const int min_pitch = w - 2 * (size / 2);
  GM_INTO(gm_moving_max_window(in, dtype, bands, h, w, in_pitch, size, has_ndv, ndv_bits, out));
}

int gm_gauss_2d_into(const void* in, int dtype, int bands, int h, int w, int in_pitch, const double* weights_y,
                     const double* weights_x, int radius, double cval, int replace_ndv, uint64_t ndv_bits, uint64_t fill_bits,
                     int y0, int x0, int oh, int ow, void* out, int out_pitch) {
  const int min_pitch = ow;
  GM_INTO(gm_gauss_2d_window(in, dtype, bands, h, w, in_pitch, weights_y, weights_x, radius, cval, replace_ndv, ndv_bits,
                             fill_bits, y0, x0, oh, ow, out));
}
#undef GM_INTO

int gm_hillshade_batch(int n_tiles, const void* const* in, int dtype, int h, int w, const int* in_pitch, int has_ndv,
                       uint64_t ndv_bits, uint64_t fill_bits, double xres, double yres, double altitude_deg, double azimuth_deg,
                       void* const* out, const int* out_pitch) {
  if (n_tiles < 0 || (n_tiles && (!in || !in_pitch || !out || !out_pitch))) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (n_tiles == 0) return GM_OK;
  if (h < 3 || w < 3) return gm::fail(GM_ERR_INVALID, "gm_hillshade_batch: tiles of %dx%d have no interior", h, w);
  for (int t = 0; t < n_tiles; ++t) {
    if (!in[t] || !out[t]) return gm::fail(GM_ERR_INVALID, "gm_hillshade_batch: tile %d is null", t);
    if (in_pitch[t] < w || out_pitch[t] < w - 2) return gm::fail(GM_ERR_INVALID, "gm_hillshade_batch: pitch of tile %d", t);
  }
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  const HillArgs A = hill_args(xres, yres, altitude_deg, azimuth_deg);
  // the pair kernel's routes only (float32 tiles; integer tiles of up to 4 bytes at power-of-two resolutions): anything else
  // is for the caller to run tile by tile
  const bool pow2 = A.xinv32 != 0.0f && A.yinv32 != 0.0f;
  const bool ints = pow2 && gm::dtype_size(dtype) <= 4 && !gm::dtype_is_float(dtype) && dtype != GM_BOOL &&
                    fabsf(A.xinv32) < 0x1p64f && fabsf(A.xinv32) > 0x1p-64f && fabsf(A.yinv32) < 0x1p64f && fabsf(A.yinv32) > 0x1p-64f;
  if ((dtype != GM_F32 && !ints) || getenv("GM_HS_SINGLE") || (h - 2 + HP_H - 1) / HP_H > 65535)
    return gm::fail(GM_ERR_UNSUPPORTED, "gm_hillshade_batch: dtype %d at this resolution runs tile by tile", dtype);
  for (int t0 = 0; t0 < n_tiles; t0 += HS_BATCH) {
    const int nb = n_tiles - t0 < HS_BATCH ? n_tiles - t0 : HS_BATCH;
    HillBatch B;
    memset(&B, 0, sizeof(B));
    for (int i = 0; i < nb; ++i) {
      B.in[i] = in[t0 + i];
      B.out[i] = (unsigned char*)out[t0 + i];
      B.in_pitch[i] = in_pitch[t0 + i];
      B.out_pitch[i] = out_pitch[t0 + i];
    }
    dim3 pgrid((unsigned)((w - 2 + HP_W - 1) / HP_W), (unsigned)((h - 2 + HP_H - 1) / HP_H), (unsigned)nb);
#define GM_HPB(T, INV)                                                                                               \
  hipLaunchKernelGGL((hillshade_pair_kernel<T, INV, true>), pgrid, dim3(256), 0, s, (const T*)nullptr,              \
                     (unsigned char*)nullptr, h, w, w, has_ndv, from_bits<T>(ndv_bits), from_bits<T>(fill_bits), A, B)
    switch (dtype) {
      case GM_F32: if (pow2) GM_HPB(float, true); else GM_HPB(float, false); break;
      case GM_U8: GM_HPB(unsigned char, true); break;
      case GM_I8: GM_HPB(signed char, true); break;
      case GM_U16: GM_HPB(unsigned short, true); break;
      case GM_I16: GM_HPB(short, true); break;
      case GM_U32: GM_HPB(unsigned int, true); break;
      default: GM_HPB(int, true); break;
    }
#undef GM_HPB
    GM_HIP(hipGetLastError());
  }
  return GM_OK;
}

}  // extern "C"
