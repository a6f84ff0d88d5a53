// Micro-benchmark (not product code): streaming ceilings of read:write mixes other than the 2:1 of the
// element-wise chain -- 4 reads : 1 write of 16 bytes per lane (int32 -> uint8 lookup, configs[2]), 1 read :
// 4 writes (2x bilinear upsample, configs[3]), write only, read only.  Full grid, one group per lane,
// non-temporal 16-byte accesses; every wave instruction touches one contiguous 1 KiB span.
//   hipcc -O3 --offload-arch=gfx950 tools/ceiling5.hip -o tools/ceiling5.bin && tools/ceiling5.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// lane reads R groups (stride = one wave-span apart) and writes W groups
template <int R, int W, int WG>
__global__ __launch_bounds__(WG) void mix(const f4* __restrict__ in, f4* __restrict__ out, size_t spans) {
  const size_t s = (size_t)blockIdx.x * (WG / 64) + (threadIdx.x >> 6);  // wave-span index
  if (s >= spans) return;
  const unsigned lane = threadIdx.x & 63;
  f4 v[R > 0 ? R : 1];
  f4 acc = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = __builtin_nontemporal_load(in + (s * R + r) * 64 + lane);
#pragma unroll
  for (int r = 0; r < R; ++r) acc += v[r];
#pragma unroll
  for (int w = 0; w < W; ++w) __builtin_nontemporal_store(acc + (float)w, out + (s * W + w) * 64 + lane);
  if (W == 0 && acc[0] == 12345.678f) out[0] = acc;
}

template <int R, int W, int WG>
void run(const char* label, const f4* in, f4* out, size_t in_bytes, size_t out_bytes) {
  // spans so that neither buffer is exceeded
  size_t spans = (size_t)1 << 60;
  if (R) spans = in_bytes / (1024 * (size_t)R);
  if (W && out_bytes / (1024 * (size_t)W) < spans) spans = out_bytes / (1024 * (size_t)W);
  const size_t blocks = (spans + WG / 64 - 1) / (WG / 64);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix<R, W, WG>), dim3((unsigned)blocks), dim3(WG), 0, 0, in, out, spans);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mix<R, W, WG>), dim3((unsigned)blocks), dim3(WG), 0, 0, in, out, spans);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)spans * 1024.0 * (R + W);
  printf("%-28s wg=%4d  %8.1f us  %7.1f GB/s  (%4.1f %% of 8 TB/s)\n", label, WG, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9,
         bytes / (ms / reps * 1e-3) / 1e9 / 80.0);
}

int main() {
  const size_t in_bytes = (size_t)4 << 30, out_bytes = (size_t)4 << 30;
  f4 *in, *out;
  CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, out_bytes));
  CK(hipMemset(in, 1, in_bytes)); CK(hipMemset(out, 0, out_bytes));
  run<1, 1, 256>("copy 1R:1W", in, out, in_bytes, out_bytes);
  run<2, 1, 256>("2R:1W (chain)", in, out, in_bytes, out_bytes);
  run<4, 1, 256>("4R:1W (lookup32)", in, out, (size_t)1 << 30, out_bytes);
  run<4, 1, 512>("4R:1W (lookup32)", in, out, (size_t)1 << 30, out_bytes);
  run<8, 2, 256>("8R:2W (lookup32 x2)", in, out, (size_t)1 << 30, out_bytes);
  run<1, 4, 256>("1R:4W (bilinear 2x)", in, out, (size_t)1 << 30, out_bytes);
  run<1, 4, 512>("1R:4W (bilinear 2x)", in, out, (size_t)1 << 30, out_bytes);
  run<2, 8, 256>("2R:8W (bilinear 2x, x2)", in, out, (size_t)1 << 30, out_bytes);
  run<0, 1, 256>("write only", in, out, in_bytes, out_bytes);
  run<0, 4, 256>("write only x4", in, out, in_bytes, out_bytes);
  run<1, 0, 256>("read only", in, out, in_bytes, out_bytes);
  run<4, 0, 256>("read only x4", in, out, in_bytes, out_bytes);
  return 0;
}
