// K6/K7: resampling kernels.
//
// K6 nearest: the warp every source read performs (raster/sources.py:119-149 hands it
//   to GDAL's GRA_NearestNeighbour) for axis-aligned same-CRS requests, and SciPy's
//   affine_transform(order=0) of Smooth's zoom branch (raster/spatial.py:296-305).
//   The separable destination->source index maps are computed once per launch in
//   float64 (same operation order as the oracle, FMA contraction off) into small
//   int32 tables; the gather kernel is output-stationary: a lane owns 4 consecutive
//   destination columns (one 16-byte store) and walks a strip of rows.
//
// K7 bilinear (our spec -- the reference has none): pixel-centre aligned,
//   edge-clamped, evaluated in float64 and rounded once.  A workgroup owns a
//   TILE_H x TILE_W block of destination pixels; the source rows/cols it touches
//   (block/zoom + 1 px apron) are staged in LDS once, so every source pixel is read
//   from HBM about once (5 B per output px for a 2x upsample of float32).
#include "gm_common.h"

namespace {

// ---------------------------------------------------------------------------------
// index tables
// ---------------------------------------------------------------------------------
// mode 0 (warp):   idx = floor(((o_dst + (k + 0.5) * s_dst) - o_src) / s_src + eps)
// mode 1 (affine): idx = floor(k * z + off + 0.5)            params = {z, off, -, -}
__global__ __launch_bounds__(256) void nearest_index_kernel(int* __restrict__ idx, int n_dst, int n_src, int mode,
                                                            double p0, double p1, double p2, double p3, double eps) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_dst) return;
  double pix;
  if (mode == 0) {
    const double world = p0 + ((double)k + 0.5) * p1;
    pix = (world - p2) / p3 + eps;
  } else {
    pix = ((double)k * p0 + p1) + 0.5;
  }
  const double f = floor(pix);
  int i = -1;
  if (f >= 0.0 && f < (double)n_src) i = (int)f;
  idx[k] = i;
}

template <typename T>
__device__ __forceinline__ bool is_finite_t(T v) {
  if constexpr (std::is_floating_point<T>::value) return isfinite(v);
  else return true;
}

// grid: x = groups of 4 destination columns (256 lanes -> 1024 columns), y = row strips, z = band
template <typename T, int ROWS>
__global__ __launch_bounds__(256) void nearest_gather_kernel(const T* __restrict__ src, T* __restrict__ dst, int hs,
                                                             int ws, int hd, int wd, const int* __restrict__ row_idx,
                                                             const int* __restrict__ col_idx, T fill,
                                                             int finite_to_fill) {
  const int band = blockIdx.z;
  const int c0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c0 >= wd) return;
  const T* sband = src + (size_t)band * hs * ws;
  T* dband = dst + (size_t)band * hd * wd;
  int ci[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ci[k] = (c0 + k < wd) ? col_idx[c0 + k] : -1;
  const int r_begin = blockIdx.y * ROWS;
  const bool vec_ok = (c0 + 3 < wd) && ((wd & 3) == 0) && ((((uintptr_t)dband) % (4 * sizeof(T))) == 0);
  for (int r = r_begin; r < r_begin + ROWS && r < hd; ++r) {
    const int ri = row_idx[r];  // wave-uniform
    T v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      T x = fill;
      if (ri >= 0 && ci[k] >= 0) {
        x = sband[(size_t)ri * ws + ci[k]];
        if (finite_to_fill && !is_finite_t<T>(x)) x = fill;
      }
      v[k] = x;
    }
    T* out = dband + (size_t)r * wd + c0;
    if (vec_ok) {
      typedef T VT __attribute__((ext_vector_type(4)));
      VT pack;
      pack[0] = v[0]; pack[1] = v[1]; pack[2] = v[2]; pack[3] = v[3];
      *(VT*)out = pack;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c0 + k < wd) out[k] = v[k];
    }
  }
}

// ---------------------------------------------------------------------------------
// bilinear
// ---------------------------------------------------------------------------------
constexpr int BL_TILE_H = 32;   // destination rows per workgroup
constexpr int BL_TILE_W = 128;  // destination columns per workgroup
constexpr int BL_LDS_BYTES = 64 * 1024;

__device__ __forceinline__ double src_coord(int k, double scale) { return ((double)k + 0.5) * scale - 0.5; }

// T: float or double.  When the source window of a block fits LDS it is staged there,
// otherwise (strong down-sampling) taps are read from global memory directly.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_kernel(const T* __restrict__ src, T* __restrict__ dst, int hs, int ws,
                                                       int hd, int wd, double sy, double sx, int has_ndv, T ndv,
                                                       int win_h_max, int win_w_max) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* tile = (T*)smem;
  const int band = blockIdx.z;
  const T* sband = src + (size_t)band * hs * ws;
  T* dband = dst + (size_t)band * hd * wd;
  const int r0 = blockIdx.y * BL_TILE_H, c0 = blockIdx.x * BL_TILE_W;
  const int r1 = min(r0 + BL_TILE_H, hd), c1 = min(c0 + BL_TILE_W, wd);

  // source window of this block (clamped taps): rows [wy0, wy1], cols [wx0, wx1]
  int wy0 = (int)floor(src_coord(r0, sy)), wy1 = (int)floor(src_coord(r1 - 1, sy)) + 1;
  int wx0 = (int)floor(src_coord(c0, sx)), wx1 = (int)floor(src_coord(c1 - 1, sx)) + 1;
  wy0 = max(wy0, 0); wx0 = max(wx0, 0);
  wy1 = min(wy1, hs - 1); wx1 = min(wx1, ws - 1);
  const int win_h = wy1 - wy0 + 1, win_w = wx1 - wx0 + 1;
  const bool staged = (win_h <= win_h_max) && (win_w <= win_w_max) && (win_h_max > 0);
  const int pitch = win_w_max;
  if (staged) {
    // coalesced staging: consecutive lanes read consecutive source columns
    for (int i = threadIdx.x; i < win_h * win_w; i += blockDim.x) {
      const int y = i / win_w, x = i - y * win_w;
      tile[y * pitch + x] = sband[(size_t)(wy0 + y) * ws + (wx0 + x)];
    }
    __syncthreads();
  }
  auto tap = [&](int y, int x) -> T {
    return staged ? tile[(y - wy0) * pitch + (x - wx0)] : sband[(size_t)y * ws + x];
  };

  // a lane owns one destination column per 256-column sweep and walks the rows
  for (int c = c0 + (int)(threadIdx.x % BL_TILE_W); c < c1; c += BL_TILE_W) {
    const double u = src_coord(c, sx);
    const double j0f = floor(u);
    const double fu = u - j0f;
    const int j0 = (int)j0f;
    const int ja = min(max(j0, 0), ws - 1), jb = min(max(j0 + 1, 0), ws - 1);
    for (int r = r0 + (int)(threadIdx.x / BL_TILE_W); r < r1; r += 256 / BL_TILE_W) {
      const double v = src_coord(r, sy);
      const double i0f = floor(v);
      const double fv = v - i0f;
      const int i0 = (int)i0f;
      const int ia = min(max(i0, 0), hs - 1), ib = min(max(i0 + 1, 0), hs - 1);
      const T A = tap(ia, ja), B = tap(ia, jb), C = tap(ib, ja), D = tap(ib, jb);
      const double top = (1.0 - fu) * (double)A + fu * (double)B;
      const double bot = (1.0 - fu) * (double)C + fu * (double)D;
      T out = (T)((1.0 - fv) * top + fv * bot);
      bool bad = !(isfinite(A) & isfinite(B) & isfinite(C) & isfinite(D));
      if (has_ndv) bad = bad | (A == ndv) | (B == ndv) | (C == ndv) | (D == ndv);
      if (bad && has_ndv) out = ndv;
      dband[(size_t)r * wd + c] = out;
    }
  }
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
  int* tables = nullptr;
  int rc = gm_malloc((void**)&tables, sizeof(int) * ((size_t)hd + (size_t)wd));
  if (rc) return rc;
  int* row_idx = tables;
  int* col_idx = tables + hd;
  hipLaunchKernelGGL(nearest_index_kernel, dim3((hd + 255) / 256), dim3(256), 0, s, row_idx, hd, hs, mode,
                     row_params[0], row_params[1], row_params[2], row_params[3], eps);
  hipLaunchKernelGGL(nearest_index_kernel, dim3((wd + 255) / 256), dim3(256), 0, s, col_idx, wd, ws, mode,
                     col_params[0], col_params[1], col_params[2], col_params[3], eps);
  constexpr int ROWS = 8;
  dim3 grid((unsigned)((wd + 1023) / 1024), (unsigned)((hd + ROWS - 1) / ROWS), (unsigned)bands);
  if (grid.y > 65535) {
    gm_free(tables);
    return gm::fail(GM_ERR_UNSUPPORTED, "destination taller than %d rows", 65535 * ROWS);
  }
#define GM_NEAREST(T, FILL)                                                                                    \
  hipLaunchKernelGGL((nearest_gather_kernel<T, ROWS>), grid, dim3(256), 0, s, (const T*)src, (T*)dst, hs, ws, \
                     hd, wd, row_idx, col_idx, (T)(FILL), finite_to_fill)
  switch (dtype) {
    case GM_BOOL: case GM_U8: case GM_I8: GM_NEAREST(unsigned char, fill.i64); break;
    case GM_U16: case GM_I16: GM_NEAREST(unsigned short, fill.i64); break;
    case GM_U32: case GM_I32: GM_NEAREST(unsigned int, fill.i64); break;
    case GM_U64: case GM_I64: GM_NEAREST(unsigned long long, fill.u64); break;
    case GM_F32: GM_NEAREST(float, fill.f64); break;
    case GM_F64: GM_NEAREST(double, fill.f64); break;
    default: gm_free(tables); return gm::fail(GM_ERR_INVALID, "gm_resample_nearest: dtype %d", dtype);
  }
#undef GM_NEAREST
  hipError_t e = hipGetLastError();
  gm_free(tables);  // stream-ordered reuse: later work on this stream runs after the gather
  if (e != hipSuccess) return gm::hip_fail(e, "gm_resample_nearest");
  return GM_OK;
}

int gm_resample_bilinear(const void* src, int dtype, int bands, int hs, int ws, void* dst, int hd, int wd,
                         int has_ndv, uint64_t ndv_bits) {
  if (!src || !dst) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (dtype != GM_F32 && dtype != GM_F64) return gm::fail(GM_ERR_UNSUPPORTED, "bilinear needs a float tile");
  if (bands <= 0 || hd <= 0 || wd <= 0) return GM_OK;
  if (hs <= 0 || ws <= 0) return gm::fail(GM_ERR_INVALID, "empty source");
  if (bands > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "more than 65535 bands");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  gm_scalar ndv;
  ndv.u64 = ndv_bits;
  const double sy = (double)hs / (double)hd, sx = (double)ws / (double)wd;
  // largest source window a block can need, + apron
  const int itemsize = dtype == GM_F32 ? 4 : 8;
  long long win_h = (long long)ceil(BL_TILE_H * sy) + 3, win_w = (long long)ceil(BL_TILE_W * sx) + 3;
  win_w = (win_w + 3) & ~3ll;
  int win_h_max = 0, win_w_max = 0;
  size_t lds = 0;
  if (win_h * win_w * itemsize <= BL_LDS_BYTES) {
    win_h_max = (int)win_h;
    win_w_max = (int)win_w;
    lds = (size_t)win_h * win_w * itemsize;
  }
  dim3 grid((unsigned)((wd + BL_TILE_W - 1) / BL_TILE_W), (unsigned)((hd + BL_TILE_H - 1) / BL_TILE_H), (unsigned)bands);
  if (grid.y > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "destination taller than %d rows", 65535 * BL_TILE_H);
  if (dtype == GM_F32)
    hipLaunchKernelGGL(bilinear_kernel<float>, grid, dim3(256), lds, s, (const float*)src, (float*)dst, hs, ws, hd,
                       wd, sy, sx, has_ndv, (float)ndv.f64, win_h_max, win_w_max);
  else
    hipLaunchKernelGGL(bilinear_kernel<double>, grid, dim3(256), lds, s, (const double*)src, (double*)dst, hs, ws,
                       hd, wd, sy, sx, has_ndv, ndv.f64, win_h_max, win_w_max);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

}  // extern "C"
