// Micro-benchmark (not product code): what limits a 2-read + 1-write float4 stream on MI355X?
// Sweeps workgroup size, groups per lane, grid shape, non-temporal hints, array stagger and
// tiles per launch.  The VM kernel is judged against the best line printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Tiles { const f32x4* a[16]; const f32x4* b[16]; f32x4* o[16]; };

template <int NT>
__device__ __forceinline__ f32x4 ldv(const f32x4* p) {
  if (NT & 1) return __builtin_nontemporal_load(p);
  return *p;
}
template <int NT>
__device__ __forceinline__ void stv(f32x4* p, f32x4 v) {
  if (NT & 2) __builtin_nontemporal_store(v, p);
  else *p = v;
}

__device__ __forceinline__ f32x4 chain(f32x4 x, f32x4 y, float ndv) {
  f32x4 r;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float av = x[q], bv = y[q];
    float t = av + bv; t = (av == ndv || bv == ndv || !isfinite(t)) ? ndv : t;
    float t2 = t * 2.5f; t2 = (t == ndv || !isfinite(t2)) ? ndv : t2;
    float t3 = t2 - bv; t3 = (t2 == ndv || bv == ndv || !isfinite(t3)) ? ndv : t3;
    float t4 = t3 + 1.5f; t4 = (t3 == ndv || !isfinite(t4)) ? ndv : t4;
    r[q] = (av == ndv) ? ndv : t4;
  }
  return r;
}

// blockIdx.y = tile; blockIdx.x strides over block-iterations of WG*G groups
template <int WG, int G, int NT>
__global__ __launch_bounds__(WG) void k(Tiles T, size_t ngroups, float ndv) {
  const f32x4* __restrict__ a = T.a[blockIdx.y];
  const f32x4* __restrict__ b = T.b[blockIdx.y];
  f32x4* __restrict__ o = T.o[blockIdx.y];
  for (size_t it = blockIdx.x; it * WG * G < ngroups; it += gridDim.x) {
    f32x4 x[G], y[G];
#pragma unroll
    for (int u = 0; u < G; ++u) { size_t g = it * WG * G + u * WG + threadIdx.x; if (g < ngroups) { x[u] = ldv<NT>(a + g); y[u] = ldv<NT>(b + g); } }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      size_t g = it * WG * G + u * WG + threadIdx.x;
      if (g < ngroups) stv<NT>(o + g, chain(x[u], y[u], ndv));
    }
  }
}

// chunked: every workgroup owns one contiguous chunk of the tile
template <int WG, int G, int NT>
__global__ __launch_bounds__(WG) void kc(Tiles T, size_t ngroups, float ndv) {
  const f32x4* __restrict__ a = T.a[blockIdx.y];
  const f32x4* __restrict__ b = T.b[blockIdx.y];
  f32x4* __restrict__ o = T.o[blockIdx.y];
  const size_t iters = (ngroups + WG * G - 1) / (WG * G);
  const size_t per = (iters + gridDim.x - 1) / gridDim.x;
  size_t it = blockIdx.x * per, end = it + per; if (end > iters) end = iters;
  for (; it < end; ++it) {
    f32x4 x[G], y[G];
#pragma unroll
    for (int u = 0; u < G; ++u) { size_t g = it * WG * G + u * WG + threadIdx.x; if (g < ngroups) { x[u] = ldv<NT>(a + g); y[u] = ldv<NT>(b + g); } }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      size_t g = it * WG * G + u * WG + threadIdx.x;
      if (g < ngroups) stv<NT>(o + g, chain(x[u], y[u], ndv));
    }
  }
}

struct Setup { std::vector<Tiles> batches; int tiles_per_launch; };

template <typename K>
float timeit(K launch, int nlaunch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < nlaunch; ++i) launch(i);
  hipEventRecord(e0);
  const int reps = 4;
  for (int r = 0; r < reps; ++r) for (int i = 0; i < nlaunch; ++i) launch(i);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps * 1e3f;  // us for all launches
}

int main(int argc, char** argv) {
  const size_t n = 4096ull * 4096ull, ngroups = n / 4; const int T = 16;
  const size_t tile_bytes = n * 4;
  // stagger sweep: arrays carved from one slab with `stagger` extra bytes between consecutive arrays
  for (size_t stagger : {(size_t)0, (size_t)4096, (size_t)(64 * 1024 + 4096), (size_t)(1 << 20) + 8192}) {
    char* slab; size_t stride = tile_bytes + stagger;
    CK(hipMalloc(&slab, stride * 3 * T + (1 << 21)));
    CK(hipMemset(slab, 0x3f, stride * 3 * T));
    std::vector<const f32x4*> A(T), B(T); std::vector<f32x4*> O(T);
    for (int i = 0; i < T; ++i) { A[i] = (const f32x4*)(slab + stride * (3 * i)); B[i] = (const f32x4*)(slab + stride * (3 * i + 1)); O[i] = (f32x4*)(slab + stride * (3 * i + 2)); }
    printf("stagger=%zu\n", stagger);
    auto run = [&](const char* name, auto kern, int wg, int g, int tpl, int gridx) {
      int nl = T / tpl;
      std::vector<Tiles> bt(nl);
      for (int l = 0; l < nl; ++l) for (int t = 0; t < tpl; ++t) { bt[l].a[t] = A[l * tpl + t]; bt[l].b[t] = B[l * tpl + t]; bt[l].o[t] = O[l * tpl + t]; }
      size_t iters = (ngroups + (size_t)wg * g - 1) / ((size_t)wg * g);
      int gx = gridx > 0 ? (int)std::min<size_t>(iters, gridx) : (int)iters;
      float us = timeit([&](int i) { hipLaunchKernelGGL(kern, dim3(gx, tpl), dim3(wg), 0, 0, bt[i], ngroups, 3.4028235e38f); }, nl);
      printf("  %-28s wg=%4d G=%d tiles/launch=%2d grid=%6d : %.1f us/tile %.0f GB/s\n", name, wg, g, tpl, gx, us / T, 12.0 * n * T / us / 1e3);
    };
#define RUN(KERN, WG, G, NT, TPL, GRID) run(#KERN "<nt" #NT ">", KERN<WG, G, NT>, WG, G, TPL, GRID)
    RUN(k, 256, 1, 0, 1, 0);
    RUN(k, 256, 1, 3, 1, 0);
    RUN(k, 256, 1, 2, 1, 0);
    RUN(k, 256, 1, 1, 1, 0);
    RUN(k, 512, 1, 0, 1, 0);
    RUN(k, 1024, 1, 0, 1, 0);
    RUN(k, 256, 2, 0, 1, 0);
    RUN(k, 512, 2, 0, 1, 0);
    RUN(k, 256, 1, 0, 1, 2048);
    RUN(k, 256, 1, 0, 1, 4096);
    RUN(k, 256, 2, 0, 1, 2048);
    RUN(k, 256, 1, 0, 4, 0);
    RUN(k, 256, 1, 0, 16, 0);
    RUN(k, 256, 1, 3, 16, 0);
    RUN(k, 256, 2, 0, 16, 0);
    RUN(k, 256, 1, 0, 16, 512);
    RUN(kc, 256, 1, 0, 1, 2048);
    RUN(kc, 256, 1, 0, 1, 4096);
    RUN(kc, 256, 2, 0, 1, 2048);
    RUN(kc, 256, 1, 0, 16, 256);
    CK(hipFree(slab));
  }
  return 0;
}
