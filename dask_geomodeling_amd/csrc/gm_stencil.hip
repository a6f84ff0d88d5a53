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
template <typename T>
__global__ __launch_bounds__(256) void dilate_kernel(const T* __restrict__ in, T* __restrict__ out, int bands, int h,
                                                     int w, DilateArgs<T> A) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  const int ow = w - 2, oh = h - 2;
  if (c >= ow) return;
  const size_t plane = (size_t)h * w;
  const T* p = in + (size_t)b * plane + (size_t)(r + 1) * w + (c + 1);  // the centre in the padded tile
  const T centre = p[0], north = p[-w], south = p[w], west = p[-1], east = p[1];
  const bool has_up = b > 0, has_down = b + 1 < bands;
  const T up = has_up ? p[-(ptrdiff_t)plane] : centre, down = has_down ? p[plane] : centre;
  T result = centre;
  for (int i = 0; i < A.n_values; ++i) {  // in order: later values overwrite earlier ones
    const T v = A.values[i];
    const bool hit = (centre == v) | (north == v) | (south == v) | (west == v) | (east == v) |
                     (has_up & (up == v)) | (has_down & (down == v));
    result = hit ? v : result;
  }
  __builtin_nontemporal_store(result, out + ((size_t)b * oh + r) * ow + c);
}

// ---------------------------------------------------------------------------------
// MovingMax
// ---------------------------------------------------------------------------------
constexpr int MM_TILE_H = 32, MM_TILE_W = 64;

template <typename T>
__global__ __launch_bounds__(256) void moving_max_kernel(const T* __restrict__ in, T* __restrict__ out, int h, int w,
                                                         int size, int has_ndv, T ndv, T minimum) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* const tile = (T*)smem;
  const int radius = size / 2;
  const int oh = h - 2 * radius, ow = w - 2 * radius;
  const int r0 = blockIdx.y * MM_TILE_H, c0 = blockIdx.x * MM_TILE_W, b = blockIdx.z;
  const T* sband = in + (size_t)b * h * w;
  const int win_h = min(MM_TILE_H, oh - r0) + 2 * radius, win_w = min(MM_TILE_W, ow - c0) + 2 * radius;
  const int pitch = MM_TILE_W + 2 * radius;
  // stage the window, no-data lowered to the dtype's minimum (spatial.py:199-203)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int y = wave; y < win_h; y += 4) {
    const T* row = sband + (size_t)(r0 + y) * w + c0;
    for (int x = lane; x < win_w; x += 64) {
      const T v = row[x];
      tile[y * pitch + x] = (has_ndv && v == ndv) ? minimum : v;
    }
  }
  __syncthreads();
  const int c = threadIdx.x & 63;
  if (c0 + c >= ow) return;
  const int s2 = size * size;
  for (int rr = threadIdx.x >> 6; rr < MM_TILE_H && r0 + rr < oh; rr += 4) {
    T best = minimum;
    for (int dy = 0; dy < size; ++dy) {
      for (int dx = 0; dx < size; ++dx) {  // wave-uniform footprint test: (x^2 + y^2) < (size/2)^2 (utils.py:536-547)
        const int yy = dy - radius, xx = dx - radius;
        if (4 * (xx * xx + yy * yy) < s2) {
          const T v = tile[(rr + dy) * pitch + (c + dx)];
          best = v > best ? v : best;
        }
      }
    }
    // still the minimum on a no-data centre: no-data again (spatial.py:208-209)
    const T centre = sband[(size_t)(r0 + rr + radius) * w + (c0 + c + radius)];
    if (has_ndv && best == minimum && centre == ndv) best = ndv;
    __builtin_nontemporal_store(best, out + ((size_t)b * oh + (r0 + rr)) * ow + (c0 + c));
  }
}

// ---------------------------------------------------------------------------------
// HillShade
// ---------------------------------------------------------------------------------
struct HillArgs {
  double xres, yres;       // division of integer / float64 tiles happens in float64
  float xres32, yres32;    // float32 tiles divide by the float32-rounded resolution (NumPy weak scalar)
  float sin_alt, cos_alt_zsf, square_zsf, az32;
};

// T arithmetic exactly as NumPy carries it out on a T array: integers wrap modulo 2^bits,
// float32 / float64 round after every operation (-ffp-contract=off)
template <typename T>
__device__ __forceinline__ float gradient(T a, T b2, T c, T d, T e2, T f, double res, float res32) {
  if constexpr (std::is_integral<T>::value) {
    typedef unsigned long long U;  // modular arithmetic: the truncation to T equals NumPy's wrap-around
    const U wide = (U)(long long)a + 2ull * (U)(long long)b2 + (U)(long long)c - (U)(long long)d - 2ull * (U)(long long)e2 - (U)(long long)f;
    return (float)((double)(T)wide / res);
  } else if constexpr (std::is_same<T, float>::value) {
    const float s = ((((a + 2.0f * b2) + c) - d) - 2.0f * e2) - f;
    return s / res32;
  } else {
    const double s = ((((a + 2.0 * b2) + c) - d) - 2.0 * e2) - f;
    return (float)(s / res);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void hillshade_kernel(const T* __restrict__ in, unsigned char* __restrict__ out, int h,
                                                        int w, int has_ndv, T ndv, T fill, HillArgs A) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  const int ow = w - 2, oh = h - 2;
  if (c >= ow) return;
  const T* p = in + (size_t)b * h * w + (size_t)r * w + c;  // top-left of the 3x3 window
  T v[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const T x = p[(size_t)i * w + j];
      v[i][j] = (has_ndv && x == ndv) ? fill : x;  // spatial.py:363-364
    }
  // spatial.py:384-404 (s0..s8 = rows top/middle/bottom x columns left/middle/right)
  const float y = gradient<T>(v[0][0], v[0][1], v[0][2], v[2][0], v[2][1], v[2][2], A.yres, A.yres32);
  const float x = gradient<T>(v[0][0], v[1][0], v[2][0], v[0][2], v[1][2], v[2][2], A.xres, A.xres32);
  const float xx_plus_yy = x * x + y * y;
  // float32 arctan2 / sin of NumPy are reproduced as the float32 rounding of the float64 functions
  const float aspect = (float)atan2((double)y, (double)x);
  const float shade = (float)sin((double)(aspect - A.az32));
  const float cang = (A.sin_alt - ((A.cos_alt_zsf * sqrtf(xx_plus_yy)) * shade)) / sqrtf(1.0f + A.square_zsf * xx_plus_yy);
  const float scaled = 255.0f * cang;
  const unsigned char result = (cang <= 0.0f) ? (unsigned char)0 : (unsigned char)(int)scaled;
  out[((size_t)b * oh + r) * ow + c] = result;
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

extern "C" {

int gm_dilate(const void* in, int dtype, int bands, int h, int w, const uint64_t* values_bits, int n_values, void* out) {
  if (!in || !out || (!values_bits && n_values)) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (bands <= 0 || h < 3 || w < 3) return gm::fail(GM_ERR_INVALID, "gm_dilate: tile %dx%dx%d has no interior", bands, h, w);
  if (n_values < 0 || n_values > DIL_MAX_VALUES) return gm::fail(GM_ERR_UNSUPPORTED, "gm_dilate: %d values (max %d)", n_values, DIL_MAX_VALUES);
  if (bands > 65535 || h - 2 > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "gm_dilate: tile too tall / too many bands");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  dim3 grid((unsigned)((w - 2 + 255) / 256), (unsigned)(h - 2), (unsigned)bands);
#define GM_DIL(T)                                                                                    \
  do {                                                                                               \
    DilateArgs<T> A;                                                                                 \
    A.n_values = n_values;                                                                           \
    for (int i = 0; i < n_values; ++i) A.values[i] = from_bits<T>(values_bits[i]);                   \
    hipLaunchKernelGGL(dilate_kernel<T>, grid, dim3(256), 0, s, (const T*)in, (T*)out, bands, h, w, A); \
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
  if (!in || !out) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (size < 3 || (size & 1) == 0 || size > 63) return gm::fail(GM_ERR_UNSUPPORTED, "gm_moving_max: size %d (odd, 3..63)", size);
  const int radius = size / 2;
  const int oh = h - 2 * radius, ow = w - 2 * radius;
  if (bands <= 0 || oh <= 0 || ow <= 0) return gm::fail(GM_ERR_INVALID, "gm_moving_max: tile smaller than the footprint");
  if (bands > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "gm_moving_max: too many bands");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  dim3 grid((unsigned)((ow + MM_TILE_W - 1) / MM_TILE_W), (unsigned)((oh + MM_TILE_H - 1) / MM_TILE_H), (unsigned)bands);
  if (grid.y > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "gm_moving_max: tile too tall");
#define GM_MM(T)                                                                                              \
  hipLaunchKernelGGL(moving_max_kernel<T>, grid, dim3(256),                                                   \
                     (size_t)(MM_TILE_H + 2 * radius) * (MM_TILE_W + 2 * radius) * sizeof(T), s, (const T*)in, \
                     (T*)out, h, w, size, has_ndv, from_bits<T>(ndv_bits), lowest<T>())
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
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_hillshade(const void* in, int dtype, int bands, int h, int w, int has_ndv, uint64_t ndv_bits, uint64_t fill_bits,
                 double xres, double yres, double altitude_deg, double azimuth_deg, void* out) {
  if (!in || !out) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (bands <= 0 || h < 3 || w < 3) return gm::fail(GM_ERR_INVALID, "gm_hillshade: tile %dx%dx%d has no interior", bands, h, w);
  if (bands > 65535 || h - 2 > 65535) return gm::fail(GM_ERR_UNSUPPORTED, "gm_hillshade: tile too tall / too many bands");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  const double pi = 3.141592653589793;
  const double alt = altitude_deg * (pi / 180.0), az = azimuth_deg * (pi / 180.0);  // math.radians
  const double zsf = 1.0 / 8.0;
  HillArgs A;
  A.xres = xres;
  A.yres = yres;
  A.xres32 = (float)xres;
  A.yres32 = (float)yres;
  A.sin_alt = (float)sin(alt);              // python floats meet float32 arrays as weak scalars:
  A.cos_alt_zsf = (float)(cos(alt) * zsf);  // the product of python floats is rounded to float32 once
  A.square_zsf = (float)(zsf * zsf);
  A.az32 = (float)az;
  dim3 grid((unsigned)((w - 2 + 255) / 256), (unsigned)(h - 2), (unsigned)bands);
#define GM_HS(T)                                                                                          \
  hipLaunchKernelGGL(hillshade_kernel<T>, grid, dim3(256), 0, s, (const T*)in, (unsigned char*)out, h, w, \
                     has_ndv, from_bits<T>(ndv_bits), from_bits<T>(fill_bits), A)
  switch (dtype) {
    case GM_U8: GM_HS(unsigned char); break;
    case GM_I8: GM_HS(signed char); break;
    case GM_U16: GM_HS(unsigned short); break;
    case GM_I16: GM_HS(short); break;
    case GM_U32: GM_HS(unsigned int); break;
    case GM_I32: GM_HS(int); break;
    case GM_I64: GM_HS(long long); break;
    case GM_F32: GM_HS(float); break;
    case GM_F64: GM_HS(double); break;
    default: return gm::fail(GM_ERR_INVALID, "gm_hillshade: dtype %d", dtype);
  }
#undef GM_HS
  GM_HIP(hipGetLastError());
  return GM_OK;
}

}  // extern "C"
