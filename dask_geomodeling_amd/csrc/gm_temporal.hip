// N4: statistics over the band (time) axis -- TemporalAggregate.process (raster/temporal.py:676-768)
// and Cumulative.process (raster/temporal.py:912-1003).
//
// The host maps every output label to a run of consecutive source frames (pandas resampler bins of
// an ascending time axis).  A lane owns V consecutive pixels and walks the frames of its bin in
// order, so every wave-instruction reads one contiguous span of a frame; a workgroup row (blockIdx.y)
// is one bin.  Every frame is read once (twice for var / std: mean first, deviations second), the
// result is written once: an HBM stream of frames * itemsize + labels * out_itemsize bytes per pixel.
//
// Arithmetic follows NumPy's reductions over axis 0 of a C-contiguous stack, which accumulate frame
// by frame in the working dtype C = result_type(float32, out dtype): no data (and NaN) is skipped
// (nansum adds +0, nanmin / nanmax are fmin / fmax), the mean is float64(sum) / float64(count)
// rounded to C (np.true_divide of a float32 array by an intp count runs the float64 loop), the
// variance is the mean of squared float-C deviations from that mean.  No FMA contraction
// (-ffp-contract=off) so every step rounds as NumPy's does.
#include <algorithm>
#include <cmath>
#include <limits>
#include <type_traits>

#include "gm_common.h"

namespace {

constexpr int TB_MAX_BINS = 96;  // bins per launch (kernarg resident)
struct Bins {
  int first[TB_MAX_BINS];  // first source frame of the bin
  int count[TB_MAX_BINS];  // frames in the bin (0: the label has no frame)
  int out[TB_MAX_BINS];    // aggregate: output frame of the bin
};

// gm_scalar payload -> T (float tiles carry float64 bits, integer tiles int64 bits)
template <typename T>
T from_bits(uint64_t bits) {
  gm_scalar s;
  s.u64 = bits;
  if constexpr (std::is_floating_point<T>::value) return (T)s.f64;
  else return (T)s.i64;
}

// working dtype: result_type(float32, out dtype)
template <typename Tout>
struct Work {
  using type = typename std::conditional<(sizeof(Tout) <= 2) || std::is_same<Tout, float>::value, float, double>::type;
};
// output dtype of a statistic (utils.py:826-845)
template <typename Tin, int STAT>
struct OutOf {
  using sum_t = typename std::conditional<
      std::is_floating_point<Tin>::value, Tin,
      typename std::conditional<(sizeof(Tin) < 4), int,
                                typename std::conditional<std::is_same<Tin, unsigned int>::value, long long, Tin>::type>::type>::type;
  using mean_t = typename std::conditional<(sizeof(Tin) <= 2) || std::is_same<Tin, float>::value, float, double>::type;
  using type = typename std::conditional<
      STAT == GM_STAT_MIN || STAT == GM_STAT_MAX, Tin,
      typename std::conditional<STAT == GM_STAT_SUM, sum_t,
                                typename std::conditional<STAT == GM_STAT_COUNT, int, mean_t>::type>::type>::type;
};

template <typename T, int V>
struct Pack {
  T v[V];
};

template <int ALIGN>
struct Unaligned;  // the access types of load_pack / store_pack, aligned like the cells they hold
#define GM_UNALIGNED(A, A2, A4, A8, A16)                                           \
  template <>                                                                      \
  struct Unaligned<A> {                                                            \
    typedef unsigned short u16 __attribute__((aligned(A2)));                       \
    typedef unsigned u32 __attribute__((aligned(A4)));                             \
    typedef unsigned u2 __attribute__((ext_vector_type(2), aligned(A8)));          \
    typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(A16)));         \
  }
GM_UNALIGNED(1, 1, 1, 1, 1);
GM_UNALIGNED(2, 2, 2, 2, 2);
GM_UNALIGNED(4, 2, 4, 4, 4);
GM_UNALIGNED(8, 2, 4, 8, 8);
#undef GM_UNALIGNED
// V cells as one access.  The vector types promise element alignment only (round 5): the instructions are the same, and a frame
// that has no multiple of V cells -- its successors then start off 16 bytes -- keeps the vector kernels (the last group of a
// frame is moved back to end with the frame: see the kernels).
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> load_pack(const T* p) {
  Pack<T, V> r;
  if constexpr (V == 1) {
    r.v[0] = __builtin_nontemporal_load(p);
  } else if constexpr (sizeof(T) * V == 2) {
    typedef typename Unaligned<(int)alignof(T)>::u16 u16u;
    const unsigned short w = __builtin_nontemporal_load((const u16u*)p);
    memcpy(&r, &w, 2);
  } else if constexpr (sizeof(T) * V == 4) {
    typedef typename Unaligned<(int)alignof(T)>::u32 u32u;
    const unsigned w = __builtin_nontemporal_load((const u32u*)p);
    memcpy(&r, &w, 4);
  } else if constexpr (sizeof(T) * V == 8) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef typename Unaligned<(int)alignof(T)>::u2 u2u;
    const u2 w = __builtin_nontemporal_load((const u2u*)p);
    memcpy(&r, &w, 8);
  } else {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef typename Unaligned<(int)alignof(T)>::u4 u4u;
    constexpr int N = (int)(sizeof(T) * V / 16);
    u4 w[N];
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = __builtin_nontemporal_load((const u4u*)p + i);
    memcpy(&r, w, sizeof(r));
  }
  return r;
}

template <typename T, int V>
__device__ __forceinline__ void store_pack(T* p, const Pack<T, V>& r) {
  if constexpr (V == 1) {
    __builtin_nontemporal_store(r.v[0], p);
  } else if constexpr (sizeof(T) * V == 2) {
    typedef typename Unaligned<(int)alignof(T)>::u16 u16u;
    unsigned short w;
    memcpy(&w, &r, 2);
    __builtin_nontemporal_store(w, (u16u*)p);
  } else if constexpr (sizeof(T) * V == 4) {
    typedef typename Unaligned<(int)alignof(T)>::u32 u32u;
    unsigned w;
    memcpy(&w, &r, 4);
    __builtin_nontemporal_store(w, (u32u*)p);
  } else if constexpr (sizeof(T) * V == 8) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef typename Unaligned<(int)alignof(T)>::u2 u2u;
    u2 w;
    memcpy(&w, &r, 8);
    __builtin_nontemporal_store(w, (u2u*)p);
  } else {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef typename Unaligned<(int)alignof(T)>::u4 u4u;
    constexpr int N = (int)(sizeof(T) * V / 16);
    u4 w[N];
    memcpy(w, &r, sizeof(r));
#pragma unroll
    for (int i = 0; i < N; ++i) __builtin_nontemporal_store(w[i], (u4u*)p + i);
  }
}

template <typename C>
__device__ __forceinline__ bool finite_(C x) {
  return fabs(x) <= std::numeric_limits<C>::max();  // false for NaN and +-inf
}

// One frame value -> (working value, skipped?).  values == no_data_value on the source dtype, NaN
// sources are NaN in the working dtype already (temporal.py:738-741).
template <typename Tin, typename C>
__device__ __forceinline__ bool skipped(Tin v, Tin ndv, bool has_ndv, C& x) {
  x = (C)v;
  bool nd = has_ndv & (v == ndv);
  if constexpr (std::is_floating_point<Tin>::value) nd |= (v != v);
  return nd;
}


constexpr int TB_UNROLL = 4;  // frames in flight per lane (the median kernels' passes)

// grid: x over groups of 256 * V pixels, y over bins
template <typename Tin, int STAT, int V>
__global__ __launch_bounds__(256) void aggregate_kernel(const Tin* __restrict__ frames, Tin ndv, bool has_ndv,
                                                        long long pixels, Bins B,
                                                        typename OutOf<Tin, STAT>::type* __restrict__ out,
                                                        typename OutOf<Tin, STAT>::type fill) {
  using Tout = typename OutOf<Tin, STAT>::type;
  using C = typename Work<Tout>::type;
  long long p = ((long long)blockIdx.x * 256 + threadIdx.x) * V;
  if (p >= pixels) return;
  if (p + V > pixels) p = pixels - V;  // (the frame's last group ends with the frame: a few cells are worked out twice)
  const int first = B.first[blockIdx.y], count = B.count[blockIdx.y];
  const Tin* src = frames + (long long)first * pixels + p;
  C acc[V];    // sum / min / max
  int cnt[V];  // frames that are data
#pragma unroll
  for (int e = 0; e < V; ++e) {
    acc[e] = (STAT == GM_STAT_MIN || STAT == GM_STAT_MAX) ? std::numeric_limits<C>::quiet_NaN() : (C)0;
    cnt[e] = 0;
  }
  auto take = [&](const Pack<Tin, V>& f, bool is_first) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      C x;
      const bool nd = skipped<Tin, C>(f.v[e], ndv, has_ndv, x);
      cnt[e] += nd ? 0 : 1;
      if constexpr (STAT == GM_STAT_MIN) {
        // np.fmin.reduce over the frame axis, the accumulator being the first operand.  A tie goes to the NEW frame: only
        // -0.0 against +0.0 can tell, and NumPy's SIMD loops (x86 min / max return their second operand on a tie) say so
        // on the hosts the golden vectors and the fuzzing oracle ran on; the scalar loop would keep the accumulator.
        const C y = nd ? std::numeric_limits<C>::quiet_NaN() : x;
        acc[e] = (acc[e] < y || y != y) ? acc[e] : y;
      } else if constexpr (STAT == GM_STAT_MAX) {
        const C y = nd ? std::numeric_limits<C>::quiet_NaN() : x;
        acc[e] = (acc[e] > y || y != y) ? acc[e] : y;
      } else if constexpr (STAT != GM_STAT_COUNT) {
        const C y = nd ? (C)0 : x;
        // np.nansum = np.sum of the frames with NaN -> 0, and np.sum over the frame axis starts from add's identity:
        // 0.0 + f0 + f1 ... (a cell whose every frame is -0.0 sums to +0.0; measured, NumPy 2.2: round-4 fuzzing)
        acc[e] = acc[e] + y;
      }
    }
  };
  // Rounds of 8, 4, 2, 1 frames (`count` is the same for the whole workgroup), every round's loads requested before the
  // first of them is looked at: a bin of any length takes its frames in at most count / 8 + 3 dependent memory round
  // trips and reads nothing twice (round 3: the first frame used to go alone and the last count mod 4 one by one).
  int j = 0;
  auto round = [&](auto n_) {
    constexpr int N = decltype(n_)::value;
    Pack<Tin, V> f[N];
#pragma unroll
    for (int u = 0; u < N; ++u) f[u] = load_pack<Tin, V>(src + (long long)(j + u) * pixels);
#pragma unroll
    for (int u = 0; u < N; ++u) take(f[u], j + u == 0);
    j += N;
  };
  while (count - j >= 8) round(std::integral_constant<int, 8>());
  if (count - j >= 4) round(std::integral_constant<int, 4>());
  if (count - j >= 2) round(std::integral_constant<int, 2>());
  if (count - j >= 1) round(std::integral_constant<int, 1>());

  Pack<Tout, V> res;
  if constexpr (STAT == GM_STAT_COUNT) {
#pragma unroll
    for (int e = 0; e < V; ++e) res.v[e] = cnt[e];
  } else {
    C r[V];
#pragma unroll
    for (int e = 0; e < V; ++e) r[e] = acc[e];
    if constexpr (STAT == GM_STAT_MEAN || STAT == GM_STAT_VAR || STAT == GM_STAT_STD) {
#pragma unroll
      for (int e = 0; e < V; ++e) r[e] = (C)((double)acc[e] / (double)cnt[e]);  // 0 / 0 = NaN: no data
    }
    if constexpr (STAT == GM_STAT_VAR || STAT == GM_STAT_STD) {
      // second pass: squared deviations from the (rounded) mean, summed frame by frame
      C sq[V];
#pragma unroll
      for (int e = 0; e < V; ++e) sq[e] = (C)0;
      auto dev = [&](const Pack<Tin, V>& f, bool is_first) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          C x;
          const bool nd = skipped<Tin, C>(f.v[e], ndv, has_ndv, x);
          const C d = nd ? (C)0 : x - r[e];
          const C y = d * d;
          sq[e] = is_first ? y : sq[e] + y;
        }
      };
      int k = 0;
      auto round2 = [&](auto n_) {
        constexpr int N = decltype(n_)::value;
        Pack<Tin, V> f[N];
#pragma unroll
        for (int u = 0; u < N; ++u) f[u] = load_pack<Tin, V>(src + (long long)(k + u) * pixels);
#pragma unroll
        for (int u = 0; u < N; ++u) dev(f[u], k + u == 0);
        k += N;
      };
      while (count - k >= 8) round2(std::integral_constant<int, 8>());
      if (count - k >= 4) round2(std::integral_constant<int, 4>());
      if (count - k >= 2) round2(std::integral_constant<int, 2>());
      if (count - k >= 1) round2(std::integral_constant<int, 1>());
#pragma unroll
      for (int e = 0; e < V; ++e) {
        C var = cnt[e] > 0 ? (C)((double)sq[e] / (double)cnt[e]) : std::numeric_limits<C>::quiet_NaN();
        if constexpr (STAT == GM_STAT_STD) var = sqrt(var);
        r[e] = var;
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) res.v[e] = (count > 0 && finite_<C>(r[e])) ? (Tout)r[e] : fill;
  }
  if constexpr (STAT == GM_STAT_COUNT) {
    if (count == 0) {
#pragma unroll
      for (int e = 0; e < V; ++e) res.v[e] = fill;
    }
  }
  store_pack<Tout, V>(out + (long long)B.out[blockIdx.y] * pixels + p, res);
}

// A raster of ONE cell (a point request: the time series of a location) is another reduction in NumPy: with nothing left
// beside the frame axis, np.sum(work, axis=0) runs its contiguous inner loop -- PAIRWISE summation over pieces of at most
// 8192 frames (loops_utils.h; the same rule as TemporalSum's frames_sum_cell_kernel) -- instead of adding frame by frame.
// A lane per bin walks its frames that way: the sum, the mean from it, and for var / std the squared deviations summed
// the same way (np.nanvar, nanfunctions.py).  Found by the round-4 fuzzing (a float64 mean of 24 frames, last bit).
template <typename C, typename F>
__device__ C pairwise_terms(const F& term, int start, int n) {
  if (n < 8) {
    C r = (C)-0.0;
    for (int i = 0; i < n; ++i) r += term(start + i);
    return r;
  }
  if (n <= 128) {
    C r[8];
    for (int u = 0; u < 8; ++u) r[u] = term(start + u);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int u = 0; u < 8; ++u) r[u] += term(start + i + u);
    C res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += term(start + i);
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise_terms<C, F>(term, start, n2) + pairwise_terms<C, F>(term, start + n2, n - n2);
}
template <typename C, typename F>
__device__ C chunked_pairwise(const F& term, int n) {
  C total = (C)0.0;  // add's identity
  for (int piece = 0; piece < n; piece += 8192) total += pairwise_terms<C, F>(term, piece, n - piece < 8192 ? n - piece : 8192);
  return total;
}
template <typename Tin, int STAT>
__global__ void aggregate_cell_kernel(const Tin* __restrict__ frames, Tin ndv, bool has_ndv, Bins B, int n_bins,
                                      typename OutOf<Tin, STAT>::type* __restrict__ out,
                                      typename OutOf<Tin, STAT>::type fill) {
  using Tout = typename OutOf<Tin, STAT>::type;
  using C = typename Work<Tout>::type;
  const int bin = blockIdx.x * blockDim.x + threadIdx.x;
  if (bin >= n_bins) return;
  const int count = B.count[bin];
  const Tin* src = frames + B.first[bin];
  int cnt = 0;
  for (int i = 0; i < count; ++i) {
    C x;
    cnt += skipped<Tin, C>(src[i], ndv, has_ndv, x) ? 0 : 1;
  }
  auto value = [&](int i) -> C {
    C x;
    return skipped<Tin, C>(src[i], ndv, has_ndv, x) ? (C)0 : x;
  };
  C r = chunked_pairwise<C>(value, count);
  if constexpr (STAT == GM_STAT_MEAN || STAT == GM_STAT_VAR || STAT == GM_STAT_STD) r = (C)((double)r / (double)cnt);  // 0 / 0 = NaN
  if constexpr (STAT == GM_STAT_VAR || STAT == GM_STAT_STD) {
    const C mean = r;
    auto squared = [&](int i) -> C {
      C x;
      const C d = skipped<Tin, C>(src[i], ndv, has_ndv, x) ? (C)0 : x - mean;
      return d * d;
    };
    const C sq = chunked_pairwise<C>(squared, count);
    r = cnt > 0 ? (C)((double)sq / (double)cnt) : std::numeric_limits<C>::quiet_NaN();
    if constexpr (STAT == GM_STAT_STD) r = sqrt(r);
  }
  out[B.out[bin]] = (count > 0 && finite_<C>(r)) ? (Tout)r : fill;
}

// Cumulative: running nancumsum / count of data within every bin; frame j is written to output frame
// j - emit_first when emit_first <= j < emit_stop (the requested window, temporal.py:979-1001).
template <typename Tin, int STAT, int V>
__global__ __launch_bounds__(256) void cumulative_kernel(const Tin* __restrict__ frames, Tin ndv, bool has_ndv,
                                                         long long pixels, Bins B, int emit_first, int emit_stop,
                                                         typename OutOf<Tin, STAT>::type* __restrict__ out,
                                                         typename OutOf<Tin, STAT>::type fill) {
  using Tout = typename OutOf<Tin, STAT>::type;
  using C = typename Work<Tout>::type;
  long long p = ((long long)blockIdx.x * 256 + threadIdx.x) * V;
  if (p >= pixels) return;
  if (p + V > pixels) p = pixels - V;  // (the frame's last group ends with the frame: a few cells are worked out twice)
  const int first = B.first[blockIdx.y], count = B.count[blockIdx.y];
  const Tin* src = frames + (long long)first * pixels + p;
  C acc[V];
  int cnt[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    acc[e] = (C)0;
    cnt[e] = 0;
  }
  auto step = [&](const Pack<Tin, V>& f, int j) {
    Pack<Tout, V> res;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      C x;
      const bool nd = skipped<Tin, C>(f.v[e], ndv, has_ndv, x);
      cnt[e] += nd ? 0 : 1;
      const C y = nd ? (C)0 : x;
      acc[e] = j == 0 ? y : acc[e] + y;
      if constexpr (STAT == GM_STAT_COUNT) res.v[e] = cnt[e];
      else res.v[e] = finite_<C>(acc[e]) ? (Tout)acc[e] : fill;
    }
    const int frame = first + j;
    if (frame >= emit_first && frame < emit_stop) store_pack<Tout, V>(out + (long long)(frame - emit_first) * pixels + p, res);
  };
  int j = 0;  // (rounds of 8, 4, 2, 1 frames as in aggregate_kernel)
  auto round = [&](auto n_) {
    constexpr int N = decltype(n_)::value;
    Pack<Tin, V> f[N];
#pragma unroll
    for (int u = 0; u < N; ++u) f[u] = load_pack<Tin, V>(src + (long long)(j + u) * pixels);
#pragma unroll
    for (int u = 0; u < N; ++u) step(f[u], j + u);
    j += N;
  };
  while (count - j >= 8) round(std::integral_constant<int, 8>());
  if (count - j >= 4) round(std::integral_constant<int, 4>());
  if (count - j >= 2) round(std::integral_constant<int, 2>());
  if (count - j >= 1) round(std::integral_constant<int, 1>());
}

// Median: the data frames of a cell are insertion-sorted into the lane's column of an LDS array
// (slot-major: lanes hit distinct banks), then (low + high) / 2 like np.ma.median, which is what
// np.nanmedian runs for fewer than 600 frames.  Bins of up to GM_MAX_MEDIAN_FRAMES frames: 256-lane
// workgroups, the percentile plan in the kernarg segment; longer bins (up to GM_MAX_MEDIAN_FRAMES_LONG): 64-lane
// workgroups -- a quarter of the columns, four times the frames in the same LDS, up to the 160 KiB of a CU --
// and the plan in device memory.
struct Plan {  // gm_percentile_step per number of data frames (1 .. GM_MAX_MEDIAN_FRAMES)
  short lower[GM_MAX_MEDIAN_FRAMES], upper[GM_MAX_MEDIAN_FRAMES];
  double gamma[GM_MAX_MEDIAN_FRAMES], one_minus_gamma[GM_MAX_MEDIAN_FRAMES];
};

template <typename Tin, bool PERCENTILE, int WG, bool LONG>
__global__ __launch_bounds__(WG) void median_kernel(const Tin* __restrict__ frames, Tin ndv, int flags,
                                                    long long pixels, Bins B, Plan P,
                                                    const gm_percentile_step* __restrict__ long_plan,
                                                    typename OutOf<Tin, GM_STAT_MEDIAN>::type* __restrict__ out,
                                                    typename OutOf<Tin, GM_STAT_MEDIAN>::type fill) {
  using C = typename OutOf<Tin, GM_STAT_MEDIAN>::type;  // = result_type(float32, dtype), the working dtype too
  const bool has_ndv = (flags & 1) != 0, keep = (flags & GM_TEMPORAL_KEEP_NONFINITE) != 0;
  extern __shared__ unsigned char lds_raw[];
  C* sorted = reinterpret_cast<C*>(lds_raw);  // [slot][lane]
  const long long p = (long long)blockIdx.x * WG + threadIdx.x;
  if (p >= pixels) return;
  const int first = B.first[blockIdx.y], count = B.count[blockIdx.y];
  const Tin* src = frames + (long long)first * pixels + p;
  C* column = sorted + threadIdx.x;
  int n = 0;
  auto insert = [&](Tin v) {
    C x;
    if (skipped<Tin, C>(v, ndv, has_ndv, x)) return;
    int i = n;
    while (i > 0 && column[(i - 1) * WG] > x) {
      column[i * WG] = column[(i - 1) * WG];
      --i;
    }
    column[i * WG] = x;
    ++n;
  };
  int j = 0;
  for (; j + TB_UNROLL <= count; j += TB_UNROLL) {
    Tin f[TB_UNROLL];
#pragma unroll
    for (int u = 0; u < TB_UNROLL; ++u) f[u] = __builtin_nontemporal_load(src + (long long)(j + u) * pixels);
#pragma unroll
    for (int u = 0; u < TB_UNROLL; ++u) insert(f[u]);
  }
  for (; j < count; ++j) insert(__builtin_nontemporal_load(src + (long long)j * pixels));
  C r = std::numeric_limits<C>::quiet_NaN();
  if (n > 0) {
    if constexpr (PERCENTILE) {
      // numpy's _lerp between the two neighbours of the virtual index (n varies per lane: the plan is
      // read with a lane-dependent index)
      int lower, upper;
      C t, u;
      if constexpr (LONG) {
        const gm_percentile_step step = long_plan[n - 1];
        lower = step.lower; upper = step.upper; t = (C)step.gamma; u = (C)step.one_minus_gamma;
      } else {
        lower = P.lower[n - 1]; upper = P.upper[n - 1]; t = (C)P.gamma[n - 1]; u = (C)P.one_minus_gamma[n - 1];
      }
      const C a = column[lower * WG], b = column[upper * WG];
      const C d = b - a;
      r = (t >= (C)0.5) ? b - d * u : a + d * t;
    } else {
      const int h = n / 2;
      const C high = column[h * WG];
      const C low = (n & 1) ? high : column[(h - 1) * WG];
      r = ((C)0 + low + high) / (C)2;  // (np.mean of the middle pair starts from add's identity: a -0.0 median is +0.0)
    }
  }
  // (TemporalAggregate fills every non-finite result, temporal.py:760-766; reduce_rasters only the cells without
  //  a data raster, reduction.py:113-115: GM_TEMPORAL_KEEP_NONFINITE)
  __builtin_nontemporal_store((keep ? n > 0 : finite_<C>(r)) ? r : fill, out + (long long)B.out[blockIdx.y] * pixels + p);
}

// Bins of up to 16 frames (hourly -> daily of 6-hourly data, 5-minute -> hourly ...): the data frames of a cell are
// sorted in REGISTERS by Batcher's odd-even merge network -- frames that are no data enter as +inf and end up behind
// the data, every index a compile-time constant -- instead of the LDS insertion sort above (one dependent LDS
// read-modify-write chain per shifted element; the kernel sat at 38 % of the HBM roofline with 6-frame bins).
// The element a lane needs (its rank depends on how many of its frames are data) is picked with a select chain.
template <typename C, int N>
__device__ __forceinline__ void sort_network(C (&v)[N]) {
#pragma unroll
  for (int p = 1; p < N; p *= 2) {
#pragma unroll
    for (int k = p; k >= 1; k /= 2) {
#pragma unroll
      for (int j = k % p; j <= N - 1 - k; j += 2 * k) {
#pragma unroll
        for (int i = 0; i <= k - 1; ++i) {
          if (i + j + k <= N - 1 && (i + j) / (2 * p) == (i + j + k) / (2 * p)) {
            const C lo = v[i + j] < v[i + j + k] ? v[i + j] : v[i + j + k];
            const C hi = v[i + j] < v[i + j + k] ? v[i + j + k] : v[i + j];
            v[i + j] = lo;
            v[i + j + k] = hi;
          }
        }
      }
    }
  }
}
// (round 6: the chain's indices as template constants -- as a loop left to the unroller, the 32-element chain (and the 16-element
//  one of the 8-byte dtypes) kept a run-time index, which put the whole sorted array into scratch memory: 16 + 4 N bytes per lane)
template <int I, int N, typename C>
__device__ __forceinline__ void pick_step(const C (&v)[N], int index, C& r) {
  if constexpr (I < N) {
    r = index == I ? v[I] : r;
    pick_step<I + 1, N, C>(v, index, r);
  }
}
template <typename C, int N>
__device__ __forceinline__ C pick(const C (&v)[N], int index) {
  C r = v[0];
  pick_step<1, N, C>(v, index, r);
  return r;
}

template <typename Tin, bool PERCENTILE, int N>
__global__ __launch_bounds__(256) void median_small_kernel(const Tin* __restrict__ frames, Tin ndv, int flags,
                                                           long long pixels, Bins B, Plan P,
                                                           typename OutOf<Tin, GM_STAT_MEDIAN>::type* __restrict__ out,
                                                           typename OutOf<Tin, GM_STAT_MEDIAN>::type fill) {
  using C = typename OutOf<Tin, GM_STAT_MEDIAN>::type;
  const bool has_ndv = (flags & 1) != 0, keep = (flags & GM_TEMPORAL_KEEP_NONFINITE) != 0;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= pixels) return;
  const int first = B.first[blockIdx.y], count = B.count[blockIdx.y];
  const Tin* src = frames + (long long)first * pixels + p;
  Tin raw[N];
#pragma unroll
  for (int j = 0; j < N; ++j) raw[j] = __builtin_nontemporal_load(src + (long long)(j < count ? j : 0) * pixels);
  C v[N];
  int n = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    C x;
    const bool nd = j >= count || skipped<Tin, C>(raw[j], ndv, has_ndv, x);
    v[j] = nd ? std::numeric_limits<C>::infinity() : x;  // (a data value of +inf sorts among the sentinels: same place)
    n += nd ? 0 : 1;
  }
  sort_network<C, N>(v);
  C r = std::numeric_limits<C>::quiet_NaN();
  if (n > 0) {
    if constexpr (PERCENTILE) {
      const C a = pick<C, N>(v, P.lower[n - 1]), b = pick<C, N>(v, P.upper[n - 1]);
      const C t = (C)P.gamma[n - 1], u = (C)P.one_minus_gamma[n - 1];
      const C d = b - a;
      r = (t >= (C)0.5) ? b - d * u : a + d * t;
    } else {
      const int h = n / 2;
      const C high = pick<C, N>(v, h);
      const C low = (n & 1) ? high : pick<C, N>(v, h - 1);
      r = ((C)0 + low + high) / (C)2;  // (np.mean of the middle pair starts from add's identity: a -0.0 median is +0.0)
    }
  }
  // (TemporalAggregate fills every non-finite result, temporal.py:760-766; reduce_rasters only the cells without
  //  a data raster, reduction.py:113-115: GM_TEMPORAL_KEEP_NONFINITE)
  __builtin_nontemporal_store((keep ? n > 0 : finite_<C>(r)) ? r : fill, out + (long long)B.out[blockIdx.y] * pixels + p);
}

// Bins too long for the LDS sort (a year of hourly frames ...): the order statistics are SELECTED instead of sorted.
// A lane counts the data frames of its cell, takes the two ranks it needs (the middle pair, or the percentile plan's
// neighbours), and finds the value of the lower rank by a radix select over the order-preserving integer image of the
// working dtype -- bit by bit from the top, each bit one pass over the bin counting the frames below the candidate --
// and the upper rank's value with one more pass (the same value when it occurs again, else the smallest larger one).
// 34 (float32) / 66 (float64) passes over the bin, every pass a coalesced stream (lane = pixel) that mostly stays
// in the Infinity Cache; no limit on the length of a bin.  np.nanmedian's >= 600-frame route (np.median of the data
// frames: the mean of the middle pair) and np.nanpercentile give the very same two values and combine them as below.
template <typename C>
struct SelectKey;
template <>
struct SelectKey<float> {
  typedef unsigned K;
  static constexpr int BITS = 32;
  static __device__ __forceinline__ K encode(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  }
  static __device__ __forceinline__ float decode(K k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
};
template <>
struct SelectKey<double> {
  typedef unsigned long long K;
  static constexpr int BITS = 64;
  static __device__ __forceinline__ K encode(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
  }
  static __device__ __forceinline__ double decode(K k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
  }
};

template <typename Tin, bool PERCENTILE>
__global__ __launch_bounds__(256) void median_select_kernel(const Tin* __restrict__ frames, Tin ndv, int flags,
                                                            long long pixels, Bins B,
                                                            const gm_percentile_step* __restrict__ long_plan,
                                                            typename OutOf<Tin, GM_STAT_MEDIAN>::type* __restrict__ out,
                                                            typename OutOf<Tin, GM_STAT_MEDIAN>::type fill) {
  using C = typename OutOf<Tin, GM_STAT_MEDIAN>::type;
  const bool has_ndv = (flags & 1) != 0, keep = (flags & GM_TEMPORAL_KEEP_NONFINITE) != 0;
  using SK = SelectKey<C>;
  using K = typename SK::K;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= pixels) return;
  const int first = B.first[blockIdx.y], count = B.count[blockIdx.y];
  const Tin* src = frames + (long long)first * pixels + p;
  // one pass over the bin: visit(key) for every data frame (TB_UNROLL loads in flight)
  auto sweep = [&](auto visit) {
    int j = 0;
    for (; j + TB_UNROLL <= count; j += TB_UNROLL) {
      Tin f[TB_UNROLL];
#pragma unroll
      for (int u = 0; u < TB_UNROLL; ++u) f[u] = src[(long long)(j + u) * pixels];
#pragma unroll
      for (int u = 0; u < TB_UNROLL; ++u) {
        C x;
        if (!skipped<Tin, C>(f[u], ndv, has_ndv, x)) visit(SK::encode(x));
      }
    }
    for (; j < count; ++j) {
      C x;
      if (!skipped<Tin, C>(src[(long long)j * pixels], ndv, has_ndv, x)) visit(SK::encode(x));
    }
  };
  int n = 0;
  sweep([&](K) { ++n; });
  C r = std::numeric_limits<C>::quiet_NaN();
  if (n > 0) {
    int lower, upper;
    C t = (C)0, u = (C)0;
    if constexpr (PERCENTILE) {
      const gm_percentile_step step = long_plan[n - 1];
      lower = step.lower; upper = step.upper; t = (C)step.gamma; u = (C)step.one_minus_gamma;
    } else {
      upper = n / 2;
      lower = (n & 1) ? upper : upper - 1;
    }
    // the largest candidate with at most `lower` frames below it is the frame of that rank
    K prefix = 0;
    for (int bit = SK::BITS - 1; bit >= 0; --bit) {
      const K candidate = prefix | ((K)1 << bit);
      int below = 0;
      sweep([&](K key) { below += key < candidate ? 1 : 0; });
      if (below <= lower) prefix = candidate;
    }
    const C a = SK::decode(prefix);
    C b = a;
    if (upper != lower) {  // (the next rank: the same value when it occurs again, else the smallest larger one)
      int not_above = 0;
      K next = ~(K)0;
      sweep([&](K key) {
        not_above += key <= prefix ? 1 : 0;
        next = (key > prefix && key < next) ? key : next;
      });
      if (not_above <= lower + 1) b = SK::decode(next);
    }
    if constexpr (PERCENTILE) {
      const C d = b - a;
      r = (t >= (C)0.5) ? b - d * u : a + d * t;
    } else {
      r = ((C)0 + a + b) / (C)2;
    }
  }
  // (TemporalAggregate fills every non-finite result, temporal.py:760-766; reduce_rasters only the cells without
  //  a data raster, reduction.py:113-115: GM_TEMPORAL_KEEP_NONFINITE)
  __builtin_nontemporal_store((keep ? n > 0 : finite_<C>(r)) ? r : fill, out + (long long)B.out[blockIdx.y] * pixels + p);
}

template <typename Tin>
int launch_median(const void* frames, uint64_t ndv_bits, int has_ndv, long long pixels, const gm_frame_bin* bins,
                  int n_bins, void* out, uint64_t fill_bits, hipStream_t s, const gm_percentile_step* plan = nullptr,
                  int n_plan = 0) {
  using C = typename OutOf<Tin, GM_STAT_MEDIAN>::type;
  int longest_all = 1;
  for (int i = 0; i < n_bins; ++i)
    if (bins[i].count > longest_all) longest_all = bins[i].count;
  if (plan && longest_all > n_plan) return gm::fail(GM_ERR_INVALID, "gm_temporal_percentile: plan covers %d frames, a bin has %d", n_plan, longest_all);
  for (int i = 0; plan && i < n_plan; ++i)
    if (plan[i].lower < 0 || plan[i].lower > i || plan[i].upper < 0 || plan[i].upper > i)
      return gm::fail(GM_ERR_INVALID, "gm_temporal_percentile: plan[%d] points outside %d values", i, i + 1);
  const bool long_bins = longest_all > GM_MAX_MEDIAN_FRAMES || (size_t)longest_all * 256 * sizeof(C) > 64 * 1024;
  Plan P;
  memset(&P, 0, sizeof(P));
  gm_percentile_step* plan_dev = nullptr;
  // bins beyond what the 160 KiB of a CU sort: selection in global memory (median_select_kernel)
  // (round 5: every long bin.  The 64-lane insertion sort that used to take bins of up to GM_MAX_MEDIAN_FRAMES_LONG frames keeps
  //  one wave per compute unit busy with ~n^2 / 4 LDS moves per cell: the median of 365 daily frames of a 1024^2 raster took
  //  337 ms; selected -- 34 passes over the bin -- 12 ms.  GM_MEDIAN_LONG_SORT=1 brings the sort back: tuning runs, tests.)
  static const bool long_sort = getenv("GM_MEDIAN_LONG_SORT") && atoi(getenv("GM_MEDIAN_LONG_SORT"));
  const bool select = long_bins && (!long_sort || longest_all > GM_MAX_MEDIAN_FRAMES_LONG || (size_t)longest_all * 64 * sizeof(C) > 160 * 1024);
  if (long_bins) {
    if (plan) {
      int rc = gm_malloc((void**)&plan_dev, sizeof(gm_percentile_step) * (size_t)n_plan);
      if (rc) return rc;
      hipError_t e = hipMemcpyAsync(plan_dev, plan, sizeof(gm_percentile_step) * (size_t)n_plan, hipMemcpyHostToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);  // (the caller's plan may go away after this call)
      if (e != hipSuccess) {
        (void)gm_free(plan_dev);
        return gm::hip_fail(e, "gm_temporal_percentile: plan upload");
      }
    }
  } else {
    for (int i = 0; i < n_plan && i < GM_MAX_MEDIAN_FRAMES; ++i) {
      P.lower[i] = (short)plan[i].lower;
      P.upper[i] = (short)plan[i].upper;
      P.gamma[i] = plan[i].gamma;
      P.one_minus_gamma[i] = plan[i].one_minus_gamma;
    }
  }
  gm::route_hit(GM_ROUTE_MEDIANS, GM_RM_CALLS);
  gm::route_hit(GM_ROUTE_MEDIANS, longest_all <= 32 ? GM_RM_NETWORK : (select ? GM_RM_SELECT : GM_RM_LDS_SORT));
  if (longest_all <= 32) {  // short bins: the register sorting network (up to 32 frames: a month of daily frames; round 5)
    const unsigned gxs = (unsigned)((pixels + 255) / 256);
    for (int b0 = 0; b0 < n_bins; b0 += TB_MAX_BINS) {
      Bins B;
      const int nb = n_bins - b0 < TB_MAX_BINS ? n_bins - b0 : TB_MAX_BINS;
      for (int i = 0; i < nb; ++i) {
        B.first[i] = bins[b0 + i].first;
        B.count[i] = bins[b0 + i].count;
        B.out[i] = bins[b0 + i].out;
      }
#define GM_MEDIAN_SMALL(PCT, N_)                                                                                     \
  hipLaunchKernelGGL((median_small_kernel<Tin, PCT, N_>), dim3(gxs, nb), dim3(256), 0, s, (const Tin*)frames,        \
                     from_bits<Tin>(ndv_bits), has_ndv, pixels, B, P, (C*)out, from_bits<C>(fill_bits))
      if (longest_all <= 4) { if (plan) GM_MEDIAN_SMALL(true, 4); else GM_MEDIAN_SMALL(false, 4); }
      else if (longest_all <= 8) { if (plan) GM_MEDIAN_SMALL(true, 8); else GM_MEDIAN_SMALL(false, 8); }
      else if (longest_all <= 16) { if (plan) GM_MEDIAN_SMALL(true, 16); else GM_MEDIAN_SMALL(false, 16); }
      else { if (plan) GM_MEDIAN_SMALL(true, 32); else GM_MEDIAN_SMALL(false, 32); }
#undef GM_MEDIAN_SMALL
    }
    return GM_OK;
  }
  if (select) {
    const unsigned gxs = (unsigned)((pixels + 255) / 256);
    for (int b0 = 0; b0 < n_bins; b0 += TB_MAX_BINS) {
      Bins B;
      const int nb = n_bins - b0 < TB_MAX_BINS ? n_bins - b0 : TB_MAX_BINS;
      for (int i = 0; i < nb; ++i) {
        B.first[i] = bins[b0 + i].first;
        B.count[i] = bins[b0 + i].count;
        B.out[i] = bins[b0 + i].out;
      }
      if (plan)
        hipLaunchKernelGGL((median_select_kernel<Tin, true>), dim3(gxs, nb), dim3(256), 0, s, (const Tin*)frames, from_bits<Tin>(ndv_bits),
                           has_ndv, pixels, B, (const gm_percentile_step*)plan_dev, (C*)out, from_bits<C>(fill_bits));
      else
        hipLaunchKernelGGL((median_select_kernel<Tin, false>), dim3(gxs, nb), dim3(256), 0, s, (const Tin*)frames, from_bits<Tin>(ndv_bits),
                           has_ndv, pixels, B, (const gm_percentile_step*)nullptr, (C*)out, from_bits<C>(fill_bits));
    }
    hipError_t e = hipGetLastError();
    if (plan_dev) (void)gm_free(plan_dev);
    return e == hipSuccess ? GM_OK : gm::hip_fail(e, "gm_temporal_aggregate: median selection");
  }
  const int WG = long_bins ? 64 : 256;
  const unsigned gx = (unsigned)((pixels + WG - 1) / WG);
  int rc = GM_OK;
  for (int b0 = 0; b0 < n_bins && rc == GM_OK; b0 += TB_MAX_BINS) {
    Bins B;
    const int nb = n_bins - b0 < TB_MAX_BINS ? n_bins - b0 : TB_MAX_BINS;
    int longest = 1;
    for (int i = 0; i < nb; ++i) {
      B.first[i] = bins[b0 + i].first;
      B.count[i] = bins[b0 + i].count;
      B.out[i] = bins[b0 + i].out;
      if (bins[b0 + i].count > longest) longest = bins[b0 + i].count;
    }
    const size_t lds = (size_t)longest * WG * sizeof(C);
#define GM_MEDIAN_LAUNCH(PCT, WG_, LONG_)                                                                              \
  do {                                                                                                                 \
    auto kernel = median_kernel<Tin, PCT, WG_, LONG_>;                                                                 \
    if (lds > 64 * 1024 &&                                                                                             \
        hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
      (void)hipGetLastError();                                                                                         \
      rc = gm::fail(GM_ERR_UNSUPPORTED, "gm_temporal_aggregate: %zu bytes of LDS per workgroup refused", lds);         \
      break;                                                                                                           \
    }                                                                                                                  \
    hipLaunchKernelGGL(kernel, dim3(gx, nb), dim3(WG_), lds, s, (const Tin*)frames, from_bits<Tin>(ndv_bits), has_ndv, \
                       pixels, B, P, (const gm_percentile_step*)plan_dev, (C*)out, from_bits<C>(fill_bits));           \
  } while (0)
    if (long_bins) {
      if (plan) GM_MEDIAN_LAUNCH(true, 64, true);
      else GM_MEDIAN_LAUNCH(false, 64, true);
    } else {
      if (plan) GM_MEDIAN_LAUNCH(true, 256, false);
      else GM_MEDIAN_LAUNCH(false, 256, false);
    }
#undef GM_MEDIAN_LAUNCH
  }
  if (plan_dev) (void)gm_free(plan_dev);  // (stream-ordered pool: reuse happens behind the kernels above)
  return rc;
}

bool ascending_bins(const gm_frame_bin* bins, int n_bins, long long n_frames, int n_out) {
  for (int i = 0; i < n_bins; ++i) {
    if (bins[i].first < 0 || bins[i].count < 0 || (long long)bins[i].first + bins[i].count > n_frames) return false;
    if (n_out >= 0 && (bins[i].out < 0 || bins[i].out >= n_out)) return false;
  }
  return true;
}

template <typename Tin, int STAT>
int launch_aggregate(const void* frames, uint64_t ndv_bits, int has_ndv, long long pixels, const gm_frame_bin* bins,
                     int n_bins, void* out, uint64_t fill_bits, hipStream_t s) {
  using Tout = typename OutOf<Tin, STAT>::type;
  const bool vec = pixels >= 4;  // (any alignment: load_pack)
  const int V = vec ? 4 : 1;
  const unsigned gx = (unsigned)((pixels + 256LL * V - 1) / (256LL * V));
  for (int b0 = 0; b0 < n_bins; b0 += TB_MAX_BINS) {
    Bins B;
    const int nb = n_bins - b0 < TB_MAX_BINS ? n_bins - b0 : TB_MAX_BINS;
    for (int i = 0; i < nb; ++i) {
      B.first[i] = bins[b0 + i].first;
      B.count[i] = bins[b0 + i].count;
      B.out[i] = bins[b0 + i].out;
    }
    constexpr bool additive = STAT == GM_STAT_SUM || STAT == GM_STAT_MEAN || STAT == GM_STAT_VAR || STAT == GM_STAT_STD;
    if constexpr (additive) {  // (integer rasters too: their means and deviations are summed in the float work dtype)
      if (pixels == 1) {  // the time series of one cell: NumPy sums it pairwise (aggregate_cell_kernel)
        hipLaunchKernelGGL((aggregate_cell_kernel<Tin, STAT>), dim3(1), dim3(128), 0, s, (const Tin*)frames, from_bits<Tin>(ndv_bits),
                           (has_ndv & 1) != 0, B, nb, (Tout*)out, from_bits<Tout>(fill_bits));
        continue;
      }
    }
    if (vec)
      hipLaunchKernelGGL((aggregate_kernel<Tin, STAT, 4>), dim3(gx, nb), dim3(256), 0, s, (const Tin*)frames,
                         from_bits<Tin>(ndv_bits), (has_ndv & 1) != 0, pixels, B, (Tout*)out, from_bits<Tout>(fill_bits));
    else
      hipLaunchKernelGGL((aggregate_kernel<Tin, STAT, 1>), dim3(gx, nb), dim3(256), 0, s, (const Tin*)frames,
                         from_bits<Tin>(ndv_bits), (has_ndv & 1) != 0, pixels, B, (Tout*)out, from_bits<Tout>(fill_bits));
  }
  return GM_OK;
}

template <typename Tin, int STAT>
int launch_cumulative(const void* frames, uint64_t ndv_bits, int has_ndv, long long pixels, const gm_frame_bin* bins,
                      int n_bins, int emit_first, int emit_stop, void* out, uint64_t fill_bits, hipStream_t s) {
  using Tout = typename OutOf<Tin, STAT>::type;
  const bool vec = pixels >= 4;  // (any alignment: load_pack)
  const int V = vec ? 4 : 1;
  const unsigned gx = (unsigned)((pixels + 256LL * V - 1) / (256LL * V));
  for (int b0 = 0; b0 < n_bins; b0 += TB_MAX_BINS) {
    Bins B;
    const int nb = n_bins - b0 < TB_MAX_BINS ? n_bins - b0 : TB_MAX_BINS;
    for (int i = 0; i < nb; ++i) {
      B.first[i] = bins[b0 + i].first;
      B.count[i] = bins[b0 + i].count;
      B.out[i] = 0;
    }
    if (vec)
      hipLaunchKernelGGL((cumulative_kernel<Tin, STAT, 4>), dim3(gx, nb), dim3(256), 0, s, (const Tin*)frames,
                         from_bits<Tin>(ndv_bits), (has_ndv & 1) != 0, pixels, B, emit_first, emit_stop, (Tout*)out,
                         from_bits<Tout>(fill_bits));
    else
      hipLaunchKernelGGL((cumulative_kernel<Tin, STAT, 1>), dim3(gx, nb), dim3(256), 0, s, (const Tin*)frames,
                         from_bits<Tin>(ndv_bits), (has_ndv & 1) != 0, pixels, B, emit_first, emit_stop, (Tout*)out,
                         from_bits<Tout>(fill_bits));
  }
  return GM_OK;
}

template <typename Tin>
int aggregate_of(int statistic, const void* frames, uint64_t ndv_bits, int has_ndv, long long pixels,
                 const gm_frame_bin* bins, int n_bins, void* out, uint64_t fill_bits, hipStream_t s) {
  switch (statistic) {
    case GM_STAT_SUM: return launch_aggregate<Tin, GM_STAT_SUM>(frames, ndv_bits, has_ndv, pixels, bins, n_bins, out, fill_bits, s);
    case GM_STAT_COUNT: return launch_aggregate<Tin, GM_STAT_COUNT>(frames, ndv_bits, has_ndv, pixels, bins, n_bins, out, fill_bits, s);
    case GM_STAT_MIN: return launch_aggregate<Tin, GM_STAT_MIN>(frames, ndv_bits, has_ndv, pixels, bins, n_bins, out, fill_bits, s);
    case GM_STAT_MAX: return launch_aggregate<Tin, GM_STAT_MAX>(frames, ndv_bits, has_ndv, pixels, bins, n_bins, out, fill_bits, s);
    case GM_STAT_MEAN: return launch_aggregate<Tin, GM_STAT_MEAN>(frames, ndv_bits, has_ndv, pixels, bins, n_bins, out, fill_bits, s);
    case GM_STAT_VAR: return launch_aggregate<Tin, GM_STAT_VAR>(frames, ndv_bits, has_ndv, pixels, bins, n_bins, out, fill_bits, s);
    case GM_STAT_STD: return launch_aggregate<Tin, GM_STAT_STD>(frames, ndv_bits, has_ndv, pixels, bins, n_bins, out, fill_bits, s);
    case GM_STAT_MEDIAN: return launch_median<Tin>(frames, ndv_bits, has_ndv, pixels, bins, n_bins, out, fill_bits, s);
    default: return gm::fail(GM_ERR_UNSUPPORTED, "gm_temporal_aggregate: statistic %d", statistic);
  }
}

template <typename Tin>
int cumulative_of(int statistic, const void* frames, uint64_t ndv_bits, int has_ndv, long long pixels,
                  const gm_frame_bin* bins, int n_bins, int emit_first, int emit_stop, void* out, uint64_t fill_bits,
                  hipStream_t s) {
  switch (statistic) {
    case GM_STAT_SUM:
      return launch_cumulative<Tin, GM_STAT_SUM>(frames, ndv_bits, has_ndv, pixels, bins, n_bins, emit_first, emit_stop, out, fill_bits, s);
    case GM_STAT_COUNT:
      return launch_cumulative<Tin, GM_STAT_COUNT>(frames, ndv_bits, has_ndv, pixels, bins, n_bins, emit_first, emit_stop, out, fill_bits, s);
    default: return gm::fail(GM_ERR_UNSUPPORTED, "gm_temporal_cumulative: statistic %d", statistic);
  }
}

}  // namespace

// ---- Snap / Resample (raster/temporal.py:165-182, 1196-1214): values[nearest] --------------------------------
// out frame j = source frame index[j]: a gather of whole frames, 16 bytes per lane and load, four loads in flight.
// grid: x over pieces of a frame, y over output frames.
template <typename W>
__global__ __launch_bounds__(256) void frames_take_kernel(const W* __restrict__ frames, long long frame_words,
                                                          const int* __restrict__ index, W* __restrict__ out) {
  const W* src = frames + (long long)index[blockIdx.y] * frame_words;
  W* dst = out + (long long)blockIdx.y * frame_words;
  const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
  W w[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = base + u * 256;
    w[u] = __builtin_nontemporal_load(src + (i < frame_words ? i : frame_words - 1));
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = base + u * 256;
    if (i < frame_words) __builtin_nontemporal_store(w[u], dst + i);
  }
}

// ---- TemporalSum.process (raster/temporal.py:239-257): values.sum(axis=0) ------------------------------------
// A plain NumPy sum over the frame axis: no-data values are added like any other.  np.sum accumulates integers
// narrower than the platform integer in int64 (signed, bool) / uint64 (unsigned) and floats in their own type;
// over axis 0 of a C-contiguous (frames, h, w) stack it adds frame by frame (the reduction runs over the outer
// axis).  A raster of ONE cell is a contiguous 1-D reduction instead: pairwise (frames_sum_cell_kernel).
template <typename Tin>
struct SumOf {
  using type = typename std::conditional<
      std::is_floating_point<Tin>::value, Tin,
      typename std::conditional<std::is_signed<Tin>::value, long long, unsigned long long>::type>::type;
};
template <typename Tin, bool IS_BOOL, int V>
__global__ __launch_bounds__(256) void frames_sum_kernel(const Tin* __restrict__ frames, int n, long long pixels,
                                                         typename std::conditional<IS_BOOL, long long, typename SumOf<Tin>::type>::type* __restrict__ out) {
  using A = typename std::conditional<IS_BOOL, long long, typename SumOf<Tin>::type>::type;
  long long p = ((long long)blockIdx.x * 256 + threadIdx.x) * V;
  if (p >= pixels) return;
  if (p + V > pixels) p = pixels - V;  // (the frame's last group ends with the frame: a few cells are worked out twice)
  const Tin* src = frames + p;
  A acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = (A)0;
  int j = 0;
  for (; j + 8 <= n; j += 8) {
    Pack<Tin, V> f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) f[u] = load_pack<Tin, V>(src + (long long)(j + u) * pixels);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = acc[e] + (A)f[u].v[e];  // (from add's identity: -0.0 frames sum to +0.0)
  }
  for (; j < n; ++j) {
    const Pack<Tin, V> f = load_pack<Tin, V>(src + (long long)j * pixels);
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = acc[e] + (A)f.v[e];
  }
  Pack<A, V> res;
#pragma unroll
  for (int e = 0; e < V; ++e) res.v[e] = acc[e];
  store_pack<A, V>(out + p, res);
}

// One cell, float: NumPy's pairwise sum (loops_utils.h) over pieces of at most 8192 elements -- what the ufunc's
// buffered iterator hands the loop at a time -- added to add's identity.  One lane does it (a raster of one cell).
template <typename T>
__device__ T cell_pairwise(const T* a, int n) {
  if (n < 8) {
    T r = (T)-0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
    T r[8];
    for (int u = 0; u < 8; ++u) r[u] = a[u];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int u = 0; u < 8; ++u) r[u] += a[i + u];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return cell_pairwise(a, n2) + cell_pairwise(a + n2, n - n2);  // (six levels deep at most: a piece is 8192 long)
}
template <typename T>
__global__ void frames_sum_cell_kernel(const T* __restrict__ frames, int n, T* __restrict__ out) {
  T total = (T)0.0;  // (add's identity; the pieces themselves chain from -0.0, loops_utils.h)
  for (int piece = 0; piece < n; piece += 8192) total += cell_pairwise(frames + piece, n - piece < 8192 ? n - piece : 8192);
  *out = total;
}

template <typename Tin, bool IS_BOOL>
int launch_frames_sum(const void* frames, int n, long long pixels, void* out, hipStream_t s) {
  using A = typename std::conditional<IS_BOOL, long long, typename SumOf<Tin>::type>::type;
  if constexpr (std::is_floating_point<Tin>::value) {
    if (pixels == 1) {
      hipLaunchKernelGGL((frames_sum_cell_kernel<Tin>), dim3(1), dim3(1), 0, s, (const Tin*)frames, n, (Tin*)out);
      return GM_OK;
    }
  }
  constexpr int VV = 16 / (int)sizeof(A);  // the wider side (the accumulator) sets the vector width
  const bool vec = VV > 1 && pixels >= VV;  // (any alignment: load_pack)
  if (vec) {
    const unsigned gx = (unsigned)(((pixels + VV - 1) / VV + 255) / 256);
    hipLaunchKernelGGL((frames_sum_kernel<Tin, IS_BOOL, VV>), dim3(gx), dim3(256), 0, s, (const Tin*)frames, n, pixels, (A*)out);
  } else {
    const unsigned gx = (unsigned)((pixels + 255) / 256);
    hipLaunchKernelGGL((frames_sum_kernel<Tin, IS_BOOL, 1>), dim3(gx), dim3(256), 0, s, (const Tin*)frames, n, pixels, (A*)out);
  }
  return GM_OK;
}

#define GM_TEMPORAL_DISPATCH(CALL)                                   \
  switch (dtype) {                                                   \
    case GM_BOOL: case GM_U8: rc = CALL(unsigned char); break;       \
    case GM_I8: rc = CALL(signed char); break;                       \
    case GM_U16: rc = CALL(unsigned short); break;                   \
    case GM_I16: rc = CALL(short); break;                            \
    case GM_U32: rc = CALL(unsigned int); break;                     \
    case GM_I32: rc = CALL(int); break;                              \
    case GM_I64: rc = CALL(long long); break;                        \
    case GM_U64: rc = CALL(unsigned long long); break;               \
    case GM_F32: rc = CALL(float); break;                            \
    case GM_F64: rc = CALL(double); break;                           \
    default: return gm::fail(GM_ERR_UNSUPPORTED, "temporal statistics: dtype %d", dtype); \
  }

extern "C" {

int gm_temporal_out_dtype(int dtype, int statistic) {
  const bool is_float = gm::dtype_is_float(dtype);
  switch (statistic) {
    case GM_STAT_MIN: case GM_STAT_MAX: return dtype == GM_BOOL ? GM_U8 : dtype;
    case GM_STAT_COUNT: return GM_I32;
    case GM_STAT_SUM:
      if (is_float) return dtype;
      if (gm::dtype_size(dtype) < 4) return GM_I32;
      return dtype == GM_U32 ? GM_I64 : dtype;
    case GM_STAT_MEAN: case GM_STAT_VAR: case GM_STAT_STD: case GM_STAT_MEDIAN:
      return (gm::dtype_size(dtype) <= 2 || dtype == GM_F32) ? GM_F32 : GM_F64;
    default: return -1;
  }
}

int gm_temporal_aggregate(const void* frames, int dtype, int has_ndv, uint64_t ndv_bits, int64_t n_frames, int64_t pixels,
                          const gm_frame_bin* bins, int n_bins, int n_out, int statistic, void* out, uint64_t fill_bits) {
  if (!frames || !out || (!bins && n_bins)) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (n_frames < 0 || pixels <= 0 || n_bins < 0 || n_out < 0) return gm::fail(GM_ERR_INVALID, "gm_temporal_aggregate: bad shape");
  if (!ascending_bins(bins, n_bins, n_frames, n_out)) return gm::fail(GM_ERR_INVALID, "gm_temporal_aggregate: bin outside the stack");
  if (pixels > (1LL << 40)) return gm::fail(GM_ERR_UNSUPPORTED, "gm_temporal_aggregate: too many pixels");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  int rc = GM_OK;
#define GM_AGG(T) aggregate_of<T>(statistic, frames, ndv_bits, has_ndv, (long long)pixels, bins, n_bins, out, fill_bits, s)
  GM_TEMPORAL_DISPATCH(GM_AGG)
#undef GM_AGG
  if (rc != GM_OK) return rc;
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_temporal_percentile(const void* frames, int dtype, int has_ndv, uint64_t ndv_bits, int64_t n_frames, int64_t pixels,
                           const gm_frame_bin* bins, int n_bins, int n_out, const gm_percentile_step* plan, int n_plan,
                           void* out, uint64_t fill_bits) {
  if (!frames || !out || !plan || (!bins && n_bins)) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (n_frames < 0 || pixels <= 0 || n_bins < 0 || n_out < 0) return gm::fail(GM_ERR_INVALID, "gm_temporal_percentile: bad shape");
  if (n_plan < 1) return gm::fail(GM_ERR_INVALID, "gm_temporal_percentile: plan of %d steps", n_plan);
  if (!ascending_bins(bins, n_bins, n_frames, n_out)) return gm::fail(GM_ERR_INVALID, "gm_temporal_percentile: bin outside the stack");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  int rc = GM_OK;
#define GM_PCT(T) launch_median<T>(frames, ndv_bits, has_ndv, (long long)pixels, bins, n_bins, out, fill_bits, s, plan, n_plan)
  GM_TEMPORAL_DISPATCH(GM_PCT)
#undef GM_PCT
  if (rc != GM_OK) return rc;
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_temporal_cumulative(const void* frames, int dtype, int has_ndv, uint64_t ndv_bits, int64_t n_frames, int64_t pixels,
                           const gm_frame_bin* bins, int n_bins, int emit_first, int emit_stop, int statistic, void* out,
                           uint64_t fill_bits) {
  if (!frames || !out || (!bins && n_bins)) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (n_frames < 0 || pixels <= 0 || n_bins < 0) return gm::fail(GM_ERR_INVALID, "gm_temporal_cumulative: bad shape");
  if (emit_first < 0 || emit_stop < emit_first || emit_stop > n_frames) return gm::fail(GM_ERR_INVALID, "gm_temporal_cumulative: bad window");
  if (!ascending_bins(bins, n_bins, n_frames, -1)) return gm::fail(GM_ERR_INVALID, "gm_temporal_cumulative: bin outside the stack");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  int rc = GM_OK;
#define GM_CUM(T) cumulative_of<T>(statistic, frames, ndv_bits, has_ndv, (long long)pixels, bins, n_bins, emit_first, emit_stop, out, fill_bits, s)
  GM_TEMPORAL_DISPATCH(GM_CUM)
#undef GM_CUM
  if (rc != GM_OK) return rc;
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_frames_take(const void* frames, size_t frame_bytes, int64_t n_frames, const int32_t* index, int n_out, void* out) {
  if (!frames || !out || (!index && n_out)) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (n_frames < 0 || n_out < 0 || n_out > 65535) return gm::fail(GM_ERR_INVALID, "gm_frames_take: %d frames out of %lld", n_out, (long long)n_frames);
  for (int j = 0; j < n_out; ++j)
    if (index[j] < 0 || index[j] >= n_frames) return gm::fail(GM_ERR_INVALID, "gm_frames_take: frame %d of %lld", index[j], (long long)n_frames);
  if (n_out == 0 || frame_bytes == 0) return GM_OK;
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  int* index_dev = nullptr;
  int rc = gm_malloc((void**)&index_dev, sizeof(int) * (size_t)n_out);
  if (rc) return rc;
  hipError_t e = hipMemcpyAsync(index_dev, index, sizeof(int) * (size_t)n_out, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // (the caller's index may go away after this call)
  if (e != hipSuccess) {
    (void)gm_free(index_dev);
    return gm::hip_fail(e, "gm_frames_take: index upload");
  }
  // the widest word that divides the frame size and both base addresses
  const uintptr_t both = (uintptr_t)frames | (uintptr_t)out | (uintptr_t)frame_bytes;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define GM_TAKE(W)                                                                                                     \
  do {                                                                                                                 \
    const long long words = (long long)(frame_bytes / sizeof(W));                                                      \
    hipLaunchKernelGGL((frames_take_kernel<W>), dim3((unsigned)((words + 1023) / 1024), (unsigned)n_out), dim3(256), 0, s, \
                       (const W*)frames, words, (const int*)index_dev, (W*)out);                                      \
  } while (0)
  if (both % 16 == 0) GM_TAKE(u4);
  else if (both % 4 == 0) GM_TAKE(unsigned);
  else GM_TAKE(unsigned char);
#undef GM_TAKE
  e = hipGetLastError();
  (void)gm_free(index_dev);  // (stream-ordered pool: reuse happens behind the kernel)
  return e == hipSuccess ? GM_OK : gm::hip_fail(e, "gm_frames_take");
}

int gm_frames_sum_dtype(int dtype) {
  switch (dtype) {
    case GM_BOOL: case GM_I8: case GM_I16: case GM_I32: case GM_I64: return GM_I64;
    case GM_U8: case GM_U16: case GM_U32: case GM_U64: return GM_U64;
    case GM_F32: return GM_F32;
    case GM_F64: return GM_F64;
    default: return -1;
  }
}

int gm_frames_sum(const void* frames, int dtype, int64_t n_frames, int64_t pixels, void* out) {
  if (!frames || !out) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (n_frames <= 0 || n_frames > 0x7fffffff || pixels < 0) return gm::fail(GM_ERR_INVALID, "gm_frames_sum: %lld frames of %lld cells", (long long)n_frames, (long long)pixels);
  if (pixels == 0) return GM_OK;
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  int rc = GM_OK;
  const int n = (int)n_frames;
  switch (dtype) {
    case GM_BOOL: rc = launch_frames_sum<unsigned char, true>(frames, n, pixels, out, s); break;
    case GM_U8: rc = launch_frames_sum<unsigned char, false>(frames, n, pixels, out, s); break;
    case GM_I8: rc = launch_frames_sum<signed char, false>(frames, n, pixels, out, s); break;
    case GM_U16: rc = launch_frames_sum<unsigned short, false>(frames, n, pixels, out, s); break;
    case GM_I16: rc = launch_frames_sum<short, false>(frames, n, pixels, out, s); break;
    case GM_U32: rc = launch_frames_sum<unsigned int, false>(frames, n, pixels, out, s); break;
    case GM_I32: rc = launch_frames_sum<int, false>(frames, n, pixels, out, s); break;
    case GM_U64: rc = launch_frames_sum<unsigned long long, false>(frames, n, pixels, out, s); break;
    case GM_I64: rc = launch_frames_sum<long long, false>(frames, n, pixels, out, s); break;
    case GM_F32: rc = launch_frames_sum<float, false>(frames, n, pixels, out, s); break;
    case GM_F64: rc = launch_frames_sum<double, false>(frames, n, pixels, out, s); break;
    default: return gm::fail(GM_ERR_UNSUPPORTED, "gm_frames_sum: dtype %d", dtype);
  }
  if (rc != GM_OK) return rc;
  GM_HIP(hipGetLastError());
  return GM_OK;
}

}  // extern "C"
