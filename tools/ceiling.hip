// Micro-benchmark (not product code): how fast can a hand-specialised kernel stream the canonical
// chain (2 reads + 1 write of float4) on this GPU?  Gives the ceiling the VM kernel is judged against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int G, bool CHAIN>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ o,
                                         size_t ngroups, float ndv) {
  size_t it = blockIdx.x;
  for (; it * 256 * G < ngroups; it += gridDim.x) {
    f32x4 x[G], y[G];
#pragma unroll
    for (int u = 0; u < G; ++u) { size_t g = it * 256 * G + u * 256 + threadIdx.x; if (g < ngroups) { x[u] = a[g]; y[u] = b[g]; } }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      size_t g = it * 256 * G + u * 256 + threadIdx.x;
      if (g >= ngroups) continue;
      f32x4 r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float av = x[u][q], bv = y[u][q];
        float t = av + bv; t = (av == ndv || bv == ndv || !isfinite(t)) ? ndv : t;
        if (CHAIN) {
          float t2 = t * 2.5f; t2 = (t == ndv || !isfinite(t2)) ? ndv : t2;
          float t3 = t2 - bv; t3 = (t2 == ndv || bv == ndv || !isfinite(t3)) ? ndv : t3;
          float t4 = t3 + 1.5f; t4 = (t3 == ndv || !isfinite(t4)) ? ndv : t4;
          t = (av == ndv) ? ndv : t4;
        }
        r[q] = t;
      }
      o[g] = r;
    }
  }
}

template <int G, bool CHAIN>
float run(const std::vector<f32x4*>& A, const std::vector<f32x4*>& B, const std::vector<f32x4*>& O, size_t ngroups, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  size_t iters = (ngroups + 256 * G - 1) / (256 * G);
  int grid = blocks > 0 ? (int)std::min<size_t>(iters, blocks) : (int)iters;
  for (size_t i = 0; i < A.size(); ++i) hipLaunchKernelGGL((k<G, CHAIN>), dim3(grid), dim3(256), 0, 0, A[i], B[i], O[i], ngroups, 3.4028235e38f);
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    for (size_t i = 0; i < A.size(); ++i) hipLaunchKernelGGL((k<G, CHAIN>), dim3(grid), dim3(256), 0, 0, A[i], B[i], O[i], ngroups, 3.4028235e38f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / (reps * A.size()) * 1e3f;
}

int main() {
  const size_t n = 4096ull * 4096ull, ngroups = n / 4; const int T = 8;
  std::vector<f32x4*> A(T), B(T), O(T);
  for (int i = 0; i < T; ++i) { CK(hipMalloc(&A[i], n * 4)); CK(hipMalloc(&B[i], n * 4)); CK(hipMalloc(&O[i], n * 4)); CK(hipMemset(A[i], 0x3f, n * 4)); CK(hipMemset(B[i], 0x40, n * 4)); }
  for (int blocks : {0, 2048, 4096, 8192}) {
    printf("blocks=%d\n", blocks);
#define R(G, C) { float us = run<G, C>(A, B, O, ngroups, blocks); printf("  G=%d %s: %.1f us  %.0f GB/s\n", G, C ? "chain5" : "add   ", us, 12.0 * n / us / 1e3); }
    R(1, false) R(2, false) R(4, false) R(1, true) R(2, true) R(4, true)
  }
  return 0;
}
