// K8: helper kernels -- constant fill, synthetic tile generator, all-equal reduction
// (Clip's early-out, raster/misc.py:108), checksum (null sink), mosaic paste
// (device side of RasterTiler.process, raster/parallelize.py:118-124).
#include <vector>

#include "gm_common.h"

namespace {

__device__ __host__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(T* out, T value, unsigned long long n) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = value;
}

template <typename T>
__device__ __forceinline__ T cast_uniform(double x) { return (T)x; }

// window [row0, row0+h) x [col0, col0+w) of a procedural H x W mosaic: the value of a pixel is
// a function of (seed, global row * W + global col) only, so tiles of any size agree
template <typename T>
__global__ __launch_bounds__(256) void generate_kernel(T* out, unsigned long long seed, long long row0, long long col0,
                                                       int h, int w, long long H, long long W, double lo, double hi,
                                                       unsigned ly, unsigned lx, T ndv) {
  const unsigned long long n = (unsigned long long)h * (unsigned long long)w;
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const unsigned long long key = splitmix64(seed);
  const double span = hi - lo;
  for (; i < n; i += stride) {
    const long long gr = row0 + (long long)(i / (unsigned long long)w);
    const long long gc = col0 + (long long)(i % (unsigned long long)w);
    T v = ndv;
    if (gr >= 0 && gr < H && gc >= 0 && gc < W) {
      const unsigned long long bits = splitmix64(key ^ ((unsigned long long)gr * (unsigned long long)W + (unsigned long long)gc));
      const double u = (double)(bits >> 11) * (1.0 / 9007199254740992.0);  // [0, 1)
      v = cast_uniform<T>(lo + u * span);
      if (lx != 0 && (gr % ly) == 0 && (gc % lx) == 0) v = ndv;
    }
    out[i] = v;
  }
}

// np.all(tile == value).  16 bytes per lane and load (a uint8 mask read cell by cell moved 64 bytes per wave and load:
// 1 TB/s); a wave that met a mismatch says so and stops, the others look at the flag every eighth load -- the answer
// "no" (the usual one where the question is "is there any data at all?") costs the first sweep of the grid, not the tile.
template <typename T>
__global__ __launch_bounds__(256) void all_equal_kernel(const T* in, T value, unsigned long long n, int* flag) {
  constexpr int PER = 16 / (int)sizeof(T);
  typedef T Vec __attribute__((ext_vector_type(PER)));
  const unsigned long long misaligned = (16 - ((unsigned long long)in & 15)) & 15;  // bytes up to the first 16-byte boundary
  const unsigned long long head = misaligned / sizeof(T) < n ? misaligned / sizeof(T) : n;
  const Vec* body = (const Vec*)(in + head);
  const unsigned long long groups = (n - head) / PER;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  bool ok = true;
  int round = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += stride) {
    const Vec v = body[i];
#pragma unroll
    for (int k = 0; k < PER; ++k) ok = ok & (v[k] == value);
    if (__ballot(!ok) != 0ull) break;
    if ((++round & 7) == 0 && *(volatile int*)flag) return;
  }
  if (blockIdx.x == 0) {  // the cells before the first and after the last whole group
    const unsigned long long done = head + groups * PER;
    for (unsigned long long i = threadIdx.x; i < head; i += blockDim.x) ok = ok & (in[i] == value);
    for (unsigned long long i = done + threadIdx.x; i < n; i += blockDim.x) ok = ok & (in[i] == value);
  }
  // one store per wave that saw a mismatch (flag only ever goes 0 -> 1)
  if (__ballot(!ok) != 0ull && (threadIdx.x & 63) == 0) *flag = 1;
}

template <typename T>
__global__ __launch_bounds__(256) void all_finite_kernel(const T* in, unsigned long long n, int* flag) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  bool ok = true;
  for (; i < n; i += stride) ok = ok & isfinite(in[i]);
  if (__ballot(!ok) != 0ull && (threadIdx.x & 63) == 0) *flag = 1;
}

__global__ __launch_bounds__(256) void checksum_kernel(const unsigned int* words, unsigned long long n_words,
                                                       const unsigned char* tail, unsigned n_tail,
                                                       unsigned long long* result) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (; i < n_words; i += stride) acc += splitmix64(((i + 1) << 32) ^ (unsigned long long)words[i]);
  if (blockIdx.x == 0 && threadIdx.x == 0 && n_tail) {
    unsigned int w = 0;
    for (unsigned k = 0; k < n_tail; ++k) w |= (unsigned int)tail[k] << (8 * k);
    acc += splitmix64(((n_words + 1) << 32) ^ (unsigned long long)w);
  }
  // wave reduction, then one atomic per wave
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(result, acc);
}

typedef unsigned Word16 __attribute__((ext_vector_type(4)));  // 16 bytes moved as one word
template <typename T>
__global__ __launch_bounds__(256) void paste_kernel(T* mosaic, int H, int W, const T* tile, int bands, int th, int tw,
                                                    int row0, int col0, int row_base) {
  // grid.y walks (band, tile row); x walks columns (coalesced)
  const int br = row_base + (int)blockIdx.y;
  const int band = br / th, r = br % th;
  const T* src = tile + ((size_t)band * th + r) * tw;
  T* dst = mosaic + ((size_t)band * H + (row0 + r)) * W + col0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < tw; c += gridDim.x * blockDim.x) dst[c] = src[c];
}

// All tiles of a mosaic in one launch (grid.z = tile): a RasterTiler over a store with a kernel of its own pastes
// 16 ... 256 tiles per request, and a launch per tile was a quarter of the time of a 16-tile HillShade mosaic.
constexpr int PASTE_BATCH = 64;
struct PasteBatch {
  const void* tile[PASTE_BATCH];
  int th[PASTE_BATCH], tw[PASTE_BATCH], row0[PASTE_BATCH], col0[PASTE_BATCH];
};
template <typename T>
__global__ __launch_bounds__(256) void paste_batch_kernel(T* mosaic, int H, int W, PasteBatch B, int bands) {
  const int t = blockIdx.z;
  const int th = B.th[t], tw = B.tw[t];
  const int br = (int)blockIdx.y;
  if (br >= bands * th) return;
  const int band = br / th, r = br % th;
  const T* src = (const T*)B.tile[t] + ((size_t)band * th + r) * tw;
  T* dst = mosaic + ((size_t)band * H + (B.row0[t] + r)) * W + B.col0[t];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < tw; c += gridDim.x * blockDim.x) dst[c] = src[c];
}

unsigned grid_for(size_t n, int cus) {
  size_t blocks = (n + 255) / 256;
  size_t cap = (size_t)(cus > 0 ? cus : 256) * 8;
  if (blocks > cap) blocks = cap;
  if (blocks == 0) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace

extern "C" {

int gm_fill(void* out, int dtype, uint64_t value_bits, size_t n) {
  gm_scalar value;
  value.u64 = value_bits;
  if (!out) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (n == 0) return GM_OK;
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  dim3 grid(grid_for(n, gm::current_cu_count()));
  switch (dtype) {
    case GM_BOOL: case GM_U8: case GM_I8:
      hipLaunchKernelGGL(fill_kernel<unsigned char>, grid, dim3(256), 0, s, (unsigned char*)out, (unsigned char)value.i64, n);
      break;
    case GM_U16: case GM_I16:
      hipLaunchKernelGGL(fill_kernel<unsigned short>, grid, dim3(256), 0, s, (unsigned short*)out, (unsigned short)value.i64, n);
      break;
    case GM_U32: case GM_I32:
      hipLaunchKernelGGL(fill_kernel<unsigned int>, grid, dim3(256), 0, s, (unsigned int*)out, (unsigned int)value.i64, n);
      break;
    case GM_U64: case GM_I64:
      hipLaunchKernelGGL(fill_kernel<unsigned long long>, grid, dim3(256), 0, s, (unsigned long long*)out, (unsigned long long)value.u64, n);
      break;
    case GM_F32:
      hipLaunchKernelGGL(fill_kernel<float>, grid, dim3(256), 0, s, (float*)out, (float)value.f64, n);
      break;
    case GM_F64:
      hipLaunchKernelGGL(fill_kernel<double>, grid, dim3(256), 0, s, (double*)out, value.f64, n);
      break;
    default:
      return gm::fail(GM_ERR_INVALID, "gm_fill: dtype %d", dtype);
  }
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_generate(void* out, int dtype, uint64_t seed, int64_t row0, int64_t col0, int h, int w, int64_t H, int64_t W,
                double lo, double hi, uint32_t lattice_y, uint32_t lattice_x, uint64_t ndv_bits) {
  gm_scalar ndv;
  ndv.u64 = ndv_bits;
  if (!out) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (h < 0 || w < 0 || H < 0 || W < 0) return gm::fail(GM_ERR_INVALID, "negative extent");
  const size_t n = (size_t)h * (size_t)w;
  if (n == 0) return GM_OK;
  if (lattice_x != 0 && lattice_y == 0) lattice_y = 1;
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  dim3 grid(grid_for(n, gm::current_cu_count()));
#define GM_GEN(T, NDV)                                                                                    \
  hipLaunchKernelGGL(generate_kernel<T>, grid, dim3(256), 0, s, (T*)out, (unsigned long long)seed,        \
                     (long long)row0, (long long)col0, h, w, (long long)H, (long long)W, lo, hi, lattice_y, \
                     lattice_x, (T)(NDV))
  switch (dtype) {
    case GM_U8: GM_GEN(unsigned char, ndv.i64); break;
    case GM_I8: GM_GEN(signed char, ndv.i64); break;
    case GM_U16: GM_GEN(unsigned short, ndv.i64); break;
    case GM_I16: GM_GEN(short, ndv.i64); break;
    case GM_U32: GM_GEN(unsigned int, ndv.i64); break;
    case GM_I32: GM_GEN(int, ndv.i64); break;
    case GM_I64: GM_GEN(long long, ndv.i64); break;
    case GM_F32: GM_GEN(float, ndv.f64); break;
    case GM_F64: GM_GEN(double, ndv.f64); break;
    default: return gm::fail(GM_ERR_INVALID, "gm_generate: dtype %d", dtype);
  }
#undef GM_GEN
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_all_equal(const void* in, int dtype, uint64_t value_bits, size_t n, int* result) {
  gm_scalar value;
  value.u64 = value_bits;
  if (!result) return gm::fail(GM_ERR_INVALID, "null pointer");
  *result = 1;
  if (n == 0) return GM_OK;
  if (!in) return gm::fail(GM_ERR_INVALID, "null pointer");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  int* flag = nullptr;
  int rc = gm_malloc((void**)&flag, sizeof(int));
  if (rc) return rc;
  GM_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
  dim3 grid(grid_for(n, gm::current_cu_count()));
#define GM_AE(T, V) hipLaunchKernelGGL(all_equal_kernel<T>, grid, dim3(256), 0, s, (const T*)in, (T)(V), (unsigned long long)n, flag)
  switch (dtype) {
    case GM_BOOL: case GM_U8: GM_AE(unsigned char, value.i64); break;
    case GM_I8: GM_AE(signed char, value.i64); break;
    case GM_U16: GM_AE(unsigned short, value.i64); break;
    case GM_I16: GM_AE(short, value.i64); break;
    case GM_U32: GM_AE(unsigned int, value.i64); break;
    case GM_I32: GM_AE(int, value.i64); break;
    case GM_U64: GM_AE(unsigned long long, value.u64); break;
    case GM_I64: GM_AE(long long, value.i64); break;
    case GM_F32: GM_AE(float, value.f64); break;
    case GM_F64: GM_AE(double, value.f64); break;
    default: gm_free(flag); return gm::fail(GM_ERR_INVALID, "gm_all_equal: dtype %d", dtype);
  }
#undef GM_AE
  int mismatch = 0;
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&mismatch, flag, sizeof(int), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  gm_free(flag);
  if (e != hipSuccess) return gm::hip_fail(e, "gm_all_equal");
  *result = mismatch ? 0 : 1;
  return GM_OK;
}

int gm_all_finite(const void* in, int dtype, size_t n, int* result) {
  if (!result) return gm::fail(GM_ERR_INVALID, "null pointer");
  *result = 1;
  if (n == 0 || (dtype != GM_F32 && dtype != GM_F64)) return GM_OK;  // integers are always finite
  if (!in) return gm::fail(GM_ERR_INVALID, "null pointer");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  int* flag = nullptr;
  int rc = gm_malloc((void**)&flag, sizeof(int));
  if (rc) return rc;
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
  dim3 grid(grid_for(n, gm::current_cu_count()));
  if (e == hipSuccess) {
    if (dtype == GM_F32)
      hipLaunchKernelGGL(all_finite_kernel<float>, grid, dim3(256), 0, s, (const float*)in, (unsigned long long)n, flag);
    else
      hipLaunchKernelGGL(all_finite_kernel<double>, grid, dim3(256), 0, s, (const double*)in, (unsigned long long)n, flag);
    e = hipGetLastError();
  }
  int bad = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  gm_free(flag);
  if (e != hipSuccess) return gm::hip_fail(e, "gm_all_finite");
  *result = bad ? 0 : 1;
  return GM_OK;
}

int gm_checksum(const void* in, size_t bytes, uint64_t* result) {
  if (!result) return gm::fail(GM_ERR_INVALID, "null pointer");
  *result = 0;
  if (bytes == 0) return GM_OK;
  if (!in) return gm::fail(GM_ERR_INVALID, "null pointer");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  unsigned long long* acc = nullptr;
  int rc = gm_malloc((void**)&acc, sizeof(unsigned long long));
  if (rc) return rc;
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(unsigned long long), s);
  const unsigned long long n_words = bytes / 4;
  const unsigned n_tail = (unsigned)(bytes % 4);
  dim3 grid(grid_for(n_words ? n_words : 1, gm::current_cu_count()));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(checksum_kernel, grid, dim3(256), 0, s, (const unsigned int*)in, n_words,
                       (const unsigned char*)in + n_words * 4, n_tail, acc);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(result, acc, sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  gm_free(acc);
  if (e != hipSuccess) return gm::hip_fail(e, "gm_checksum");
  return GM_OK;
}

int gm_paste_tile(void* mosaic, int H, int W, const void* tile, int bands, int th, int tw, int row0, int col0,
                  int itemsize) {
  if (!mosaic || !tile) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (bands <= 0 || th <= 0 || tw <= 0) return GM_OK;
  if (row0 < 0 || col0 < 0 || row0 + th > H || col0 + tw > W)
    return gm::fail(GM_ERR_INVALID, "gm_paste_tile: tile (%d,%d)+(%d,%d) outside mosaic (%d,%d)", row0, col0, th, tw, H, W);
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  // rows that are whole 16-byte groups on both sides (4064-wide uint8 tiles of a 16256-wide mosaic ...) move as 16-byte words
  // through the widest instantiation: one byte per lane made the paste of a HillShade tile slower than its shading
  const long long row_bytes = (long long)tw * itemsize, mosaic_row_bytes = (long long)W * itemsize;
  if (itemsize < 16 && row_bytes % 16 == 0 && mosaic_row_bytes % 16 == 0 && ((long long)col0 * itemsize) % 16 == 0 &&
      ((uintptr_t)mosaic % 16) == 0 && ((uintptr_t)tile % 16) == 0)
    return gm_paste_tile(mosaic, H, (int)(mosaic_row_bytes / 16), tile, bands, th, (int)(row_bytes / 16), row0,
                         (int)((long long)col0 * itemsize / 16), 16);
  unsigned gx = (unsigned)((tw + 255) / 256);
  if (gx > 64) gx = 64;
  // grid.y is limited to 65535: walk the (band, row) space in chunks
  const long long rows_total = (long long)bands * th;
  if (rows_total > 2147483647ll) return gm::fail(GM_ERR_UNSUPPORTED, "gm_paste_tile: too many rows");
  for (long long base = 0; base < rows_total; base += 65535) {
    long long nrows = rows_total - base;
    if (nrows > 65535) nrows = 65535;
    dim3 grid(gx, (unsigned)nrows);
#define GM_PASTE(T)                                                                                   \
  hipLaunchKernelGGL(paste_kernel<T>, grid, dim3(256), 0, s, (T*)mosaic, H, W, (const T*)tile, bands, \
                     th, tw, row0, col0, (int)base)
    switch (itemsize) {
      case 1: GM_PASTE(unsigned char); break;
      case 2: GM_PASTE(unsigned short); break;
      case 4: GM_PASTE(unsigned int); break;
      case 8: GM_PASTE(unsigned long long); break;
      case 16: GM_PASTE(Word16); break;
      default: return gm::fail(GM_ERR_INVALID, "gm_paste_tile: itemsize %d", itemsize);
    }
#undef GM_PASTE
    GM_HIP(hipGetLastError());
  }
  return GM_OK;
}

int gm_paste_tiles(void* mosaic, int H, int W, int n_tiles, const void* const* tiles, const int* th, const int* tw,
                   const int* row0, const int* col0, int bands, int itemsize) {
  if (!mosaic || (n_tiles && (!tiles || !th || !tw || !row0 || !col0))) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (n_tiles <= 0 || bands <= 0) return GM_OK;
  if (itemsize != 1 && itemsize != 2 && itemsize != 4 && itemsize != 8) return gm::fail(GM_ERR_INVALID, "gm_paste_tiles: itemsize %d", itemsize);
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  // as in gm_paste_tile: rows that are whole 16-byte groups on both sides move as 16-byte words
  bool words = ((uintptr_t)mosaic % 16) == 0 && ((long long)W * itemsize) % 16 == 0;
  long long most_rows = 0;
  // a tile without cells pastes nothing, as in gm_paste_tile (the reference's int() of a tile size can leave a tile of no
  // columns at the edge of a request -- parallelize.py:85-86; found by the temporal fuzzer on 17 x 31 rasters): it is left out
  std::vector<const void*> kept_tile;
  std::vector<int> kept_th, kept_tw, kept_row0, kept_col0;
  for (int t = 0; t < n_tiles; ++t) {
    if (!tiles[t]) return gm::fail(GM_ERR_INVALID, "gm_paste_tiles: tile %d is null", t);
    if (th[t] <= 0 || tw[t] <= 0) continue;
    if (row0[t] < 0 || col0[t] < 0 || row0[t] + th[t] > H || col0[t] + tw[t] > W)
      return gm::fail(GM_ERR_INVALID, "gm_paste_tiles: tile %d (%d,%d)+(%d,%d) outside mosaic (%d,%d)", t, row0[t], col0[t], th[t], tw[t], H, W);
    words = words && ((uintptr_t)tiles[t] % 16) == 0 && ((long long)tw[t] * itemsize) % 16 == 0 && ((long long)col0[t] * itemsize) % 16 == 0;
    if ((long long)bands * th[t] > most_rows) most_rows = (long long)bands * th[t];
    kept_tile.push_back(tiles[t]);
    kept_th.push_back(th[t]);
    kept_tw.push_back(tw[t]);
    kept_row0.push_back(row0[t]);
    kept_col0.push_back(col0[t]);
  }
  n_tiles = (int)kept_tile.size();
  if (n_tiles == 0) return GM_OK;
  tiles = kept_tile.data();
  th = kept_th.data();
  tw = kept_tw.data();
  row0 = kept_row0.data();
  col0 = kept_col0.data();
  if (most_rows > 65535) {  // (grid.y): tile by tile
    for (int t = 0; t < n_tiles; ++t) {
      int rc = gm_paste_tile(mosaic, H, W, tiles[t], bands, th[t], tw[t], row0[t], col0[t], itemsize);
      if (rc) return rc;
    }
    return GM_OK;
  }
  const int scale = words ? 16 / itemsize : 1;  // elements per moved word
  for (int t0 = 0; t0 < n_tiles; t0 += PASTE_BATCH) {
    const int nb = n_tiles - t0 < PASTE_BATCH ? n_tiles - t0 : PASTE_BATCH;
    PasteBatch B;
    memset(&B, 0, sizeof(B));
    int widest = 1;
    for (int i = 0; i < nb; ++i) {
      B.tile[i] = tiles[t0 + i];
      B.th[i] = th[t0 + i];
      B.tw[i] = tw[t0 + i] / scale;
      B.row0[i] = row0[t0 + i];
      B.col0[i] = col0[t0 + i] / scale;
      if (B.tw[i] > widest) widest = B.tw[i];
    }
    unsigned gx = (unsigned)((widest + 255) / 256);
    if (gx > 64) gx = 64;
    dim3 grid(gx, (unsigned)most_rows, (unsigned)nb);
#define GM_PASTEB(T) hipLaunchKernelGGL(paste_batch_kernel<T>, grid, dim3(256), 0, s, (T*)mosaic, H, W / scale, B, bands)
    if (words) GM_PASTEB(Word16);
    else if (itemsize == 1) GM_PASTEB(unsigned char);
    else if (itemsize == 2) GM_PASTEB(unsigned short);
    else if (itemsize == 4) GM_PASTEB(unsigned int);
    else GM_PASTEB(unsigned long long);
#undef GM_PASTEB
    GM_HIP(hipGetLastError());
  }
  return GM_OK;
}

}  // extern "C"

// Device-to-device copy of a 2-D / 3-D window (compaction of a strided view), asynchronous on the compute stream.  Round 5: a
// kernel of this library instead of hipMemcpy2DAsync / hipMemcpy3DAsync -- the runtime's copy kernels take a window that does
// not start on 16 bytes, or whose rows are no multiple of 16 bytes wide, several times slower (the 8 frames of a 2048^2 window
// of a float32 series: 45 us on a 16-byte boundary, 150 us one cell further), and every Block that wants a compact tile pays
// for it.  A lane moves 16 bytes of a row, element-aligned on both sides (the hardware takes unaligned accesses); a
// workgroup 4 KiB of each of COPY_ROWS consecutive rows, every load issued before the first store; the last bytes of a row
// go one by one.
namespace {
constexpr int COPY_ROWS = 4;
__global__ __launch_bounds__(256) void copy_window_kernel(unsigned char* __restrict__ dst, size_t dst_pitch, size_t dst_plane,
                                                          const unsigned char* __restrict__ src, size_t src_pitch, size_t src_plane,
                                                          size_t width, size_t rows, unsigned blocks_per_row) {
  typedef unsigned u32x4u __attribute__((ext_vector_type(4), aligned(1)));
  const size_t row0 = (size_t)(blockIdx.x / blocks_per_row) * COPY_ROWS;
  const size_t col = ((size_t)(blockIdx.x % blocks_per_row) * 256 + threadIdx.x) * 16;
  if (col >= width) return;
  const unsigned char* s = src + (size_t)blockIdx.y * src_plane + row0 * src_pitch + col;
  unsigned char* d = dst + (size_t)blockIdx.y * dst_plane + row0 * dst_pitch + col;
  if (col + 16 <= width) {
    u32x4u v[COPY_ROWS];
#pragma unroll
    for (int r = 0; r < COPY_ROWS; ++r)  // (rows past the window re-read its last row: unconditional loads, nothing stored)
      v[r] = __builtin_nontemporal_load((const u32x4u*)(s + (row0 + r < rows ? (size_t)r : rows - 1 - row0) * src_pitch));
#pragma unroll
    for (int r = 0; r < COPY_ROWS; ++r)
      if (row0 + r < rows) *(u32x4u*)(d + (size_t)r * dst_pitch) = v[r];
  } else {
    for (int r = 0; r < COPY_ROWS && row0 + r < rows; ++r)
      for (size_t k = 0; col + k < width; ++k) d[(size_t)r * dst_pitch + k] = s[(size_t)r * src_pitch + k];
  }
}

int copy_window(void* dst, size_t dst_pitch, size_t dst_plane, const void* src, size_t src_pitch, size_t src_plane, size_t width,
                size_t rows, size_t planes, hipStream_t stream) {
  const size_t blocks_per_row = (width + 4095) / 4096, row_groups = (rows + COPY_ROWS - 1) / COPY_ROWS;
  if (blocks_per_row * row_groups > 0x7fffffffull || blocks_per_row > 0xffffffffull || planes > 65535)
    return gm::fail(GM_ERR_UNSUPPORTED, "copy of a window of %zu x %zu bytes x %zu planes", rows, width, planes);
  hipLaunchKernelGGL(copy_window_kernel, dim3((unsigned)(blocks_per_row * row_groups), (unsigned)planes), dim3(256), 0, stream,
                     (unsigned char*)dst, dst_pitch, dst_plane, (const unsigned char*)src, src_pitch, src_plane, width, rows,
                     (unsigned)blocks_per_row);
  GM_HIP(hipGetLastError());
  return GM_OK;
}
}  // namespace

extern "C" int gm_copy_2d(void* dst, size_t dst_pitch_bytes, const void* src, size_t src_pitch_bytes,
                          size_t width_bytes, size_t rows) {
  if (!dst || !src) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (width_bytes == 0 || rows == 0) return GM_OK;
  if (width_bytes > dst_pitch_bytes || width_bytes > src_pitch_bytes)
    return gm::fail(GM_ERR_INVALID, "gm_copy_2d: rows of %zu bytes in pitches of %zu / %zu", width_bytes, dst_pitch_bytes, src_pitch_bytes);
  hipStream_t stream = gm::current_stream();
  if (!stream) return GM_ERR_NO_DEVICE;
  return copy_window(dst, dst_pitch_bytes, 0, src, src_pitch_bytes, 0, width_bytes, rows, 1, stream);
}

// ... of a window of several planes (the frames of a series inside a tile of its raster): `planes` windows of `rows` rows,
// the source's `src_plane_rows` rows of src_pitch_bytes apart, the destination's dst_plane_rows.  One launch.
extern "C" int gm_copy_3d(void* dst, size_t dst_pitch_bytes, size_t dst_plane_rows, const void* src, size_t src_pitch_bytes,
                          size_t src_plane_rows, size_t width_bytes, size_t rows, size_t planes) {
  if (!dst || !src) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (width_bytes == 0 || rows == 0 || planes == 0) return GM_OK;
  if (rows > dst_plane_rows || rows > src_plane_rows || width_bytes > dst_pitch_bytes || width_bytes > src_pitch_bytes)
    return gm::fail(GM_ERR_INVALID, "gm_copy_3d: window larger than its plane");
  hipStream_t stream = gm::current_stream();
  if (!stream) return GM_ERR_NO_DEVICE;
  for (size_t p0 = 0; p0 < planes; p0 += 65535) {  // (grid.y)
    const size_t np = planes - p0 < 65535 ? planes - p0 : 65535;
    if (int rc = copy_window((char*)dst + p0 * dst_plane_rows * dst_pitch_bytes, dst_pitch_bytes, dst_plane_rows * dst_pitch_bytes,
                             (const char*)src + p0 * src_plane_rows * src_pitch_bytes, src_pitch_bytes,
                             src_plane_rows * src_pitch_bytes, width_bytes, rows, np, stream))
      return rc;
  }
  return GM_OK;
}
