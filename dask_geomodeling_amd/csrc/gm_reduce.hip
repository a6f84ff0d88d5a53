// N4: reduce_rasters (raster/reduction.py:38-119) -- a statistic over a stack of rasters, skipping no data.
//
// The host has cast every raster of the stack to the reference's work dtype
// W = result_type(dtype, float16) with NaN for no data (program op NANIFNODATA, one launch per raster),
// exactly as reduction.py:99-107 fills `stack_array`; this file is NumPy's nan-function along axis 0 of
// that (n, pixels) stack:
//   * a lane owns V consecutive pixels and walks the frames in order.  The array the reference reduces is
//     `stack_array[:, not_all_nan]` (reduction.py:113-115): boolean-mask indexing hands NumPy an (n, K) array
//     whose FAST axis is n (strides (itemsize, n * itemsize)), so `func(..., axis=0)` reduces along the
//     contiguous axis: sums are NumPy's PAIRWISE summation over the n frames (loops_utils.h: fewer than 8
//     terms one after the other from -0.0; up to 128 in eight interleaved accumulators combined as
//     ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) plus the remainder; beyond that halves, the first a
//     multiple of 8), products run frame by frame; a float16 work dtype accumulates both in float32 and
//     rounds ONCE at the end (HALF_add / HALF_multiply with a reduced inner axis), while the element-wise
//     steps of var / std (deviation, square) round per element.  Round 3: until then the frames were added
//     one after the other, which is the same thing below 8 rasters in float32 / float64 -- all the vectors
//     of round 2 -- and is not from 8 rasters on, nor for float16 sums that actually round;
//   * nansum adds +0 for NaN, nanprod multiplies by 1, nanmin / nanmax are fmin / fmax with the
//     accumulator first, nanargmin / nanargmax replace NaN by +-inf and take the first extremum;
//   * nanmean = float64(sum) / float64(count) rounded once to W (np.true_divide of a W array by an intp
//     count runs the float64 loop); nanvar = mean of squared W-deviations from that rounded mean,
//     nanstd its square root in W;
//   * nanmedian (< 600 frames: np.ma.median) sorts the data frames of a cell and returns (low + high) / 2;
//     nanpercentile interpolates between the two neighbours of the virtual index with NumPy's _lerp, the
//     host having derived index and weight per number of data frames with NumPy's own arithmetic.
// W = float16 (1-byte output dtypes) is float32 storage here; NumPy's element-wise float16 loops compute in
// float32 and round the result to float16 after every operation, and so does HALF = true (division by the
// count: float64 quotient rounded straight to float16).  Cells that are NaN in every raster get `fill`; every other
// result is left as it is (incl. NaN / inf of overflowing data): the cast into the output dtype is the
// caller's, like the reference's `out[not_all_nan] = func(...)`.
// HBM traffic: n * itemsize + itemsize bytes per pixel (twice the reads for var / std).
#include <cmath>
#include <limits>
#include <type_traits>

#include "gm_common.h"

namespace {

template <typename C, int V>
struct Pack {
  C v[V];
};
template <typename C, int V>
__device__ __forceinline__ Pack<C, V> load_pack(const C* p) {
  Pack<C, V> r;
  if constexpr (V == 1) {
    r.v[0] = __builtin_nontemporal_load(p);
  } else {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 w = __builtin_nontemporal_load((const u4*)p);
    memcpy(&r, &w, 16);
  }
  return r;
}
template <typename C, int V>
__device__ __forceinline__ void store_pack(C* p, const Pack<C, V>& r) {
  if constexpr (V == 1) {
    __builtin_nontemporal_store(r.v[0], p);
  } else {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 w;
    memcpy(&w, &r, 16);
    __builtin_nontemporal_store(w, (u4*)p);
  }
}

// Where the frames of a cell group come from.  FramesOfStack: frame j of the (n, pixels) stack the reference builds (NaN for no
// data).  FramesOfRasters (round 6): n SEPARATE rasters of the work dtype read in place -- `stack_array[i, index] = values[index]`
// (reduction.py:102-107) happens in the load: a cell that utils.get_index calls no data (~np.isclose(values, ndv), utils.py:61-64,
// with the parameters the host works out per raster) arrives as NaN.  Until then every raster was coded into a stack by a launch
// of its own and the stack read again: three passes over the bytes and ~20 us of host time per raster.
struct RasterRef {  // = gm_raster_ref
  const void* values;
  double y, tol;
  unsigned flags, reserved;
};
template <typename C>
struct FramesOfStack {
  const C* src;
  long long pixels;
  template <int V>
  __device__ __forceinline__ Pack<C, V> load(int j) const { return load_pack<C, V>(src + (long long)j * pixels); }
};
template <typename C>
struct FramesOfRasters {
  const RasterRef* refs;  // (wave-uniform reads: scalar loads)
  long long p;
  template <int V>
  __device__ __forceinline__ Pack<C, V> load(int j) const {
    const RasterRef r = refs[j];
    Pack<C, V> f = load_pack<C, V>((const C*)r.values + p);
    const bool in_f64 = (r.flags & 1u) != 0, y_finite = (r.flags & 2u) != 0;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      bool close;
      if (in_f64) {
        const double xd = (double)f.v[e];
        close = ((fabs(xd - r.y) <= r.tol) & y_finite) | (xd == r.y);
      } else {
        const float xf = (float)f.v[e], yf = (float)r.y;
        close = ((fabsf(xf - yf) <= (float)r.tol) & y_finite) | (xf == yf);
      }
      f.v[e] = close ? std::numeric_limits<C>::quiet_NaN() : f.v[e];
    }
    return f;
  }
};
// first / last / count (reduction.py:78-91) tell data from no data by utils.get_index alone: a cell that IS NaN is data there (it
// is picked, it is counted), so the NaN coding above cannot serve them -- the rasters' own cells and the index side by side
template <typename C, int V>
struct MaskedPack {
  Pack<C, V> f;
  bool data[V];
};
template <typename C, int V>
__device__ __forceinline__ MaskedPack<C, V> load_masked(const RasterRef* refs, long long p, int j) {
  const RasterRef r = refs[j];
  MaskedPack<C, V> m;
  m.f = load_pack<C, V>((const C*)r.values + p);
  const bool in_f64 = (r.flags & 1u) != 0, y_finite = (r.flags & 2u) != 0;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    bool close;
    if (in_f64) {
      const double xd = (double)m.f.v[e];
      close = ((fabs(xd - r.y) <= r.tol) & y_finite) | (xd == r.y);
    } else {
      const float xf = (float)m.f.v[e], yf = (float)r.y;
      close = ((fabsf(xf - yf) <= (float)r.tol) & y_finite) | (xf == yf);
    }
    m.data[e] = !close;
  }
  return m;
}
constexpr int RD_PICK_UNROLL = 4;
// STAT 12 / 13 / 1: first / last / count over n separate rasters
template <typename C, int STAT, int V>
__global__ __launch_bounds__(256) void pick_rasters_kernel(const RasterRef* __restrict__ refs, int n, long long pixels, C* __restrict__ out, C fill) {
  const long long p = ((long long)blockIdx.x * 256 + threadIdx.x) * V;
  if (p >= pixels) return;
  C acc[V];
  int cnt[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    acc[e] = fill;
    cnt[e] = 0;
  }
  auto take = [&](const MaskedPack<C, V>& m) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      if (m.data[e]) {
        acc[e] = m.f.v[e];  // (first: the rasters are walked from the last to the first, like reduction.py:80-81)
        cnt[e] += 1;
      }
    }
  };
  int j = 0;
  for (; j + RD_PICK_UNROLL <= n; j += RD_PICK_UNROLL) {
    MaskedPack<C, V> m[RD_PICK_UNROLL];
#pragma unroll
    for (int u = 0; u < RD_PICK_UNROLL; ++u) m[u] = load_masked<C, V>(refs, p, STAT == 12 ? n - 1 - (j + u) : j + u);
#pragma unroll
    for (int u = 0; u < RD_PICK_UNROLL; ++u) take(m[u]);
  }
  for (; j < n; ++j) take(load_masked<C, V>(refs, p, STAT == 12 ? n - 1 - j : j));
  Pack<C, V> res;
#pragma unroll
  for (int e = 0; e < V; ++e) res.v[e] = STAT == 1 ? (C)cnt[e] : acc[e];
  store_pack<C, V>(out + p, res);
}

template <typename C>
struct StackSource {
  const C* stack;
  long long pixels;
  __device__ __forceinline__ FramesOfStack<C> at(long long p) const { return FramesOfStack<C>{stack + p, pixels}; }
};
template <typename C>
struct RasterSource {
  const RasterRef* refs;
  __device__ __forceinline__ FramesOfRasters<C> at(long long p) const { return FramesOfRasters<C>{refs, p}; }
};

// rounding of a float32 result to float16 (NumPy's HALF loops: float32 operation, then npy_float_to_half)
template <bool HALF, typename C>
__device__ __forceinline__ C rnd(C x) {
  if constexpr (HALF) return (C)(float)(_Float16)(float)x;
  else return x;
}
// float64 -> W: for float16 one rounding straight from the float64 value (npy_double_to_half): float64 ->
// float32 rounded to odd keeps the sticky information the final float32 -> float16 rounding needs
template <bool HALF, typename C>
__device__ __forceinline__ C from_f64(double d) {
  if constexpr (!HALF) {
    return (C)d;
  } else {
    float f = __double2float_rz(d);
    if ((double)f != d && d == d && fabs(d) <= 3.0e38) f = __uint_as_float(__float_as_uint(f) | 1u);
    return (C)(float)(_Float16)f;
  }
}

constexpr int RD_UNROLL = 4;
constexpr int RD_MAX_SORT = 64;
constexpr int RD_PW_BLOCK = 128, RD_PW_DEPTH = 5;  // NumPy's PW_BLOCKSIZE; halvings this file unrolls: n <= 128 * 2^5

// What a frame contributes to a sum.  MODE 0: the value, NaN (no data) as +0 (np.nansum's _replace_nan), counting the
// data frames; MODE 1: the squared deviation from `mean` (np.nanvar: arr - avg where data, 0 elsewhere, then
// np.multiply(arr, arr): two element-wise operations, each rounded to W).
template <typename C, typename A, bool HALF, int V, int MODE>
__device__ __forceinline__ void pw_term(const Pack<C, V>& f, const C (&mean)[V], A (&term)[V], int (&cnt)[V]) {
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const C x = f.v[e];
    const bool nd = x != x;
    if constexpr (MODE == 0) {
      cnt[e] += nd ? 0 : 1;
      term[e] = nd ? (A)0 : (A)x;
    } else {
      const C d = nd ? (C)0 : rnd<HALF, C>(x - mean[e]);
      term[e] = (A)rnd<HALF, C>(d * d);
    }
  }
}

// NumPy's pairwise_sum over frames [start, start + len), len <= 128 (loops_utils.h, @TYPE@_pairwise_sum)
template <typename C, typename A, bool HALF, int V, int MODE, typename F>
__device__ __forceinline__ void pw_leaf(const F& fr, int start, int len, const C (&mean)[V], A (&res)[V],
                                        int (&cnt)[V]) {
  A term[V];
  if (len < 8) {
#pragma unroll
    for (int e = 0; e < V; ++e) res[e] = (A)-0.0;  // (the loop's own start; the reduction adds it to add's identity: reduce_kernel)
    for (int i = 0; i < len; ++i) {
      pw_term<C, A, HALF, V, MODE>(fr.template load<V>(start + i), mean, term, cnt);
#pragma unroll
      for (int e = 0; e < V; ++e) res[e] += term[e];
    }
    return;
  }
  A r[8][V];
  {
    Pack<C, V> f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) f[u] = fr.template load<V>(start + u);
#pragma unroll
    for (int u = 0; u < 8; ++u) pw_term<C, A, HALF, V, MODE>(f[u], mean, r[u], cnt);
  }
  int i = 8;
  for (; i < len - (len % 8); i += 8) {
    Pack<C, V> f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) f[u] = fr.template load<V>(start + i + u);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      pw_term<C, A, HALF, V, MODE>(f[u], mean, term, cnt);
#pragma unroll
      for (int e = 0; e < V; ++e) r[u][e] += term[e];
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) res[e] = ((r[0][e] + r[1][e]) + (r[2][e] + r[3][e])) + ((r[4][e] + r[5][e]) + (r[6][e] + r[7][e]));
  for (; i < len; ++i) {
    pw_term<C, A, HALF, V, MODE>(fr.template load<V>(start + i), mean, term, cnt);
#pragma unroll
    for (int e = 0; e < V; ++e) res[e] += term[e];
  }
}

// DEPTH >= 0: the recursion unrolled (everything stays in registers; n <= 128 * 2^DEPTH).  DEPTH < 0: stacks of any
// length -- the recursion walked with an explicit stack of the left halves' sums (a lane-private array indexed at run
// time lives in scratch memory, which is why the short stacks do not take this route), piece by piece.
template <typename C, typename A, bool HALF, int V, int MODE, int DEPTH, typename F>
__device__ void pw_sum(const F& fr, int start, int n, const C (&mean)[V], A (&res)[V], int (&cnt)[V]) {
  if constexpr (DEPTH < 0) {
    // NumPy hands the reduction loop at most `bufsize` = 8192 elements at a time (the ufunc's buffered iterator),
    // so a longer stack is the chain  0.0 + pairwise(first 8192) + pairwise(next 8192) + ...  (measured against
    // np.nansum on the (n, cells) array the reference reduces; 8192 = 128 * 2^6: six halvings inside a piece)
    constexpr int PIECE = 8192, LEVELS = 8;
    int at[LEVELS], len[LEVELS], phase[LEVELS];
    A left[LEVELS][V], part[V];
#pragma unroll
    for (int e = 0; e < V; ++e) res[e] = (A)0.0;
    for (int piece = 0; piece < n; piece += PIECE) {
      int sp = 0;
      at[0] = start + piece; len[0] = n - piece < PIECE ? n - piece : PIECE; phase[0] = 0;
      while (sp >= 0) {
        if (phase[sp] == 0) {
          if (len[sp] <= RD_PW_BLOCK || sp == LEVELS - 1) {
            pw_leaf<C, A, HALF, V, MODE>(fr, at[sp], len[sp], mean, part, cnt);
            --sp;
            continue;
          }
          int n2 = len[sp] / 2;
          n2 -= n2 % 8;
          phase[sp] = 1;
          at[sp + 1] = at[sp]; len[sp + 1] = n2; phase[sp + 1] = 0;
          ++sp;
        } else if (phase[sp] == 1) {  // the left half is in `part`: keep it, walk the right half
          int n2 = len[sp] / 2;
          n2 -= n2 % 8;
#pragma unroll
          for (int e = 0; e < V; ++e) left[sp][e] = part[e];
          phase[sp] = 2;
          at[sp + 1] = at[sp] + n2; len[sp + 1] = len[sp] - n2; phase[sp + 1] = 0;
          ++sp;
        } else {
#pragma unroll
          for (int e = 0; e < V; ++e) part[e] = left[sp][e] + part[e];
          --sp;
        }
      }
#pragma unroll
      for (int e = 0; e < V; ++e) res[e] += part[e];
    }
  } else if constexpr (DEPTH == 0) {
    pw_leaf<C, A, HALF, V, MODE>(fr, start, n, mean, res, cnt);  // (longer stacks take the DEPTH < 0 form)
  } else {
    if (n <= RD_PW_BLOCK) {
      pw_leaf<C, A, HALF, V, MODE>(fr, start, n, mean, res, cnt);
      return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    A right[V];
    pw_sum<C, A, HALF, V, MODE, DEPTH - 1>(fr, start, n2, mean, res, cnt);
    pw_sum<C, A, HALF, V, MODE, DEPTH - 1>(fr, start + n2, n - n2, mean, right, cnt);
#pragma unroll
    for (int e = 0; e < V; ++e) res[e] += right[e];
  }
}

// sum / product / min / max / mean / var / std / argmin / argmax
template <typename C, bool HALF, int STAT, int V, int PW_DEPTH = RD_PW_DEPTH, typename SRC = StackSource<C>>
__global__ __launch_bounds__(256) void reduce_kernel(const SRC source, int n, long long pixels,
                                                     C* __restrict__ out, C fill) {
  const long long p = ((long long)blockIdx.x * 256 + threadIdx.x) * V;
  if (p >= pixels) return;
  const auto fr = source.at(p);
  const C nan = std::numeric_limits<C>::quiet_NaN(), inf = std::numeric_limits<C>::infinity();
  C acc[V];
  int cnt[V], arg[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    acc[e] = (STAT == GM_RED_MIN || STAT == GM_RED_MAX) ? nan : (C)0;
    cnt[e] = 0;
    arg[e] = 0;
  }
  auto take = [&](const Pack<C, V>& f, int j) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const C x = f.v[e];
      const bool nd = x != x;
      cnt[e] += nd ? 0 : 1;
      if constexpr (STAT == GM_RED_MIN) {
        // np.fmin.reduce over the rasters.  A tie -- +0.0 against -0.0 -- goes to the LATER raster, as in the temporal kernels
        // (gm_temporal.hip aggregate_kernel): NumPy's loops on x86 hosts end in minps / maxps, which hand back their second
        // operand when the two compare equal (np.nanmin([[0.], [-0.]], axis=0) is -0.0, the other order +0.0)
        acc[e] = (acc[e] < x || nd) ? acc[e] : x;
      } else if constexpr (STAT == GM_RED_MAX) {
        acc[e] = (acc[e] > x || nd) ? acc[e] : x;
      } else if constexpr (STAT == GM_RED_PRODUCT) {
        const C y = nd ? (C)1 : x;
        acc[e] = j == 0 ? y : acc[e] * y;  // (float16: the product runs in float32 and is rounded once, below)
      } else if constexpr (STAT == GM_RED_ARGMIN) {
        const C y = nd ? inf : x;  // _replace_nan(a, +inf), np.argmin: the first minimum
        if (j == 0 || y < acc[e]) { acc[e] = y; arg[e] = j; }
      } else if constexpr (STAT == GM_RED_ARGMAX) {
        const C y = nd ? -inf : x;
        if (j == 0 || y > acc[e]) { acc[e] = y; arg[e] = j; }
      }
    }
  };
  constexpr bool ADDITIVE = STAT == GM_RED_SUM || STAT == GM_RED_MEAN || STAT == GM_RED_VAR || STAT == GM_RED_STD;
  typedef typename std::conditional<std::is_same<C, double>::value, double, float>::type A;  // what the sums run in
  C r[V];
  if constexpr (ADDITIVE) {
    A total[V];
    C none[V];
#pragma unroll
    for (int e = 0; e < V; ++e) none[e] = (C)0;
    pw_sum<C, A, HALF, V, 0, PW_DEPTH>(fr, 0, n, none, total, cnt);
    // (np.sum puts the loop's result on top of add's identity: cells whose every raster is -0.0 sum to +0.0 -- measured,
    //  NumPy 2.2, round-4 fuzzing; until then such a cell came out as -0.0)
#pragma unroll
    for (int e = 0; e < V; ++e) r[e] = rnd<HALF, C>((C)((A)0.0 + total[e]));
    if constexpr (STAT != GM_RED_SUM) {
#pragma unroll
      for (int e = 0; e < V; ++e) r[e] = from_f64<HALF, C>((double)r[e] / (double)cnt[e]);
    }
    if constexpr (STAT == GM_RED_VAR || STAT == GM_RED_STD) {
      A squares[V];
      int ignored[V];
      pw_sum<C, A, HALF, V, 1, PW_DEPTH>(fr, 0, n, r, squares, ignored);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        C var = from_f64<HALF, C>((double)rnd<HALF, C>((C)squares[e]) / (double)cnt[e]);
        if constexpr (STAT == GM_RED_STD) var = rnd<HALF, C>(sqrt(var));
        r[e] = var;
      }
    }
  } else {
    int j = 0;
    for (; j + RD_UNROLL <= n; j += RD_UNROLL) {
      Pack<C, V> f[RD_UNROLL];
#pragma unroll
      for (int u = 0; u < RD_UNROLL; ++u) f[u] = fr.template load<V>(j + u);
#pragma unroll
      for (int u = 0; u < RD_UNROLL; ++u) take(f[u], j + u);
    }
    for (; j < n; ++j) take(fr.template load<V>(j), j);
#pragma unroll
    for (int e = 0; e < V; ++e)
      r[e] = (STAT == GM_RED_ARGMIN || STAT == GM_RED_ARGMAX) ? (C)arg[e] : (STAT == GM_RED_PRODUCT ? rnd<HALF, C>(acc[e]) : acc[e]);
  }
  Pack<C, V> res;
#pragma unroll
  for (int e = 0; e < V; ++e) res.v[e] = cnt[e] > 0 ? r[e] : fill;
  store_pack<C, V>(out + p, res);
}

struct SortPlan {  // gm_percentile_step per number of data frames (1 .. RD_MAX_SORT)
  short lower[RD_MAX_SORT], upper[RD_MAX_SORT];
  double gamma[RD_MAX_SORT], one_minus_gamma[RD_MAX_SORT];
};

// median / percentile: the data frames of a cell insertion-sorted into the lane's column of an LDS array.
// Up to RD_MAX_SORT rasters: 256-lane workgroups, the plan in the kernarg segment; more (LONG): 64-lane
// workgroups -- four times the rasters in the same LDS, up to the CU's 160 KiB -- and the plan in memory.
template <typename C, bool HALF, bool PERCENTILE, int WG, bool LONG>
__global__ __launch_bounds__(WG) void reduce_sorted_kernel(const C* __restrict__ stack, int n, long long pixels,
                                                           SortPlan P, const gm_percentile_step* __restrict__ long_plan,
                                                           C* __restrict__ out, C fill) {
  extern __shared__ unsigned char lds_raw[];
  C* column = reinterpret_cast<C*>(lds_raw) + threadIdx.x;  // [slot][lane]
  const long long p = (long long)blockIdx.x * WG + threadIdx.x;
  if (p >= pixels) return;
  const C* src = stack + p;
  int m = 0;
  for (int j = 0; j < n; ++j) {
    const C x = __builtin_nontemporal_load(src + (long long)j * pixels);
    if (x != x) continue;
    int i = m;
    while (i > 0 && column[(i - 1) * WG] > x) {
      column[i * WG] = column[(i - 1) * WG];
      --i;
    }
    column[i * WG] = x;
    ++m;
  }
  C r = fill;
  if (m > 0) {
    if constexpr (PERCENTILE) {
      int lower, upper;
      C t, u;
      if constexpr (LONG) {
        const gm_percentile_step step = long_plan[m - 1];
        lower = step.lower; upper = step.upper; t = (C)step.gamma; u = (C)step.one_minus_gamma;
      } else {
        lower = P.lower[m - 1]; upper = P.upper[m - 1]; t = (C)P.gamma[m - 1]; u = (C)P.one_minus_gamma[m - 1];
      }
      const C a = column[lower * WG], b = column[upper * WG];
      const C d = rnd<HALF, C>(b - a);
      r = (t >= (C)0.5) ? rnd<HALF, C>(b - rnd<HALF, C>(d * u)) : rnd<HALF, C>(a + rnd<HALF, C>(d * t));
    } else {
      const int h = m / 2;
      const C high = column[h * WG];
      const C low = (m & 1) ? high : column[(h - 1) * WG];
      r = rnd<HALF, C>(rnd<HALF, C>((low + high) + (C)0) / (C)2);  // np.ma.median: sum of the two, then / 2.
    }
  }
  __builtin_nontemporal_store(r, out + p);
}

// Up to 16 rasters -- the usual stack -- sort in registers (Batcher's odd-even merge network, NaN frames entering as
// +inf behind the data) instead of LDS; the element of the rank a cell needs is picked with a select chain.
// Batcher's odd-even merge sort as a network of compare-exchanges on a lane's registers.  The indices are template constants
// (round 6): written as nested loops for the unroller, the 32- and 64-element networks -- and the 16-element one of float64 -- were
// left with run-time indices, so the values lived in scratch memory (16 + 4 N bytes per lane) and every comparator went there.
template <int Begin, int End, int Step, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (Begin < End) {
    f(std::integral_constant<int, Begin>{});
    static_for<Begin + Step, End, Step>(f);
  }
}
template <typename C, int N, int P, int K>
__device__ __forceinline__ void sort_network_step(C (&v)[N]) {
  static_for<K % P, N - K, 2 * K>([&](auto j) {
    static_for<0, K, 1>([&](auto i) {
      constexpr int a = decltype(i)::value + decltype(j)::value, b = a + K;
      if constexpr (b <= N - 1 && a / (2 * P) == b / (2 * P)) {
        const C lo = v[a] < v[b] ? v[a] : v[b];
        const C hi = v[a] < v[b] ? v[b] : v[a];
        v[a] = lo;
        v[b] = hi;
      }
    });
  });
  if constexpr (K / 2 >= 1) sort_network_step<C, N, P, K / 2>(v);
}
template <typename C, int N, int P = 1>
__device__ __forceinline__ void sort_network(C (&v)[N]) {
  sort_network_step<C, N, P, P>(v);
  if constexpr (2 * P < N) sort_network<C, N, 2 * P>(v);
}
template <typename C, int N>
__device__ __forceinline__ C pick(const C (&v)[N], int index) {
  C r = v[0];
  static_for<1, N, 1>([&](auto i) { r = index == decltype(i)::value ? v[decltype(i)::value] : r; });  // (see sort_network)
  return r;
}

template <typename C, bool HALF, bool PERCENTILE, int N, typename SRC = StackSource<C>>
__global__ __launch_bounds__(256) void reduce_sorted_small_kernel(const SRC source, int n, long long pixels,
                                                                  SortPlan P, C* __restrict__ out, C fill) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= pixels) return;
  const auto fr = source.at(p);
  C v[N];
  int m = 0;
  // (all N loads first: with each value examined right behind its load the compiler waited for every load before it
  // issued the next -- N memory round trips per pixel)
  // (... sixteen at a time: all 64 of the largest network in flight at once doubled its registers and spilled)
  constexpr int CH = N < 16 ? N : 16;
#pragma unroll
  for (int j0 = 0; j0 < N; j0 += CH) {
    C raw[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) raw[k] = fr.template load<1>(j0 + k < n ? j0 + k : 0).v[0];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const C x = raw[k];
      const bool nd = j0 + k >= n || x != x;
      v[j0 + k] = nd ? std::numeric_limits<C>::infinity() : x;
      m += nd ? 0 : 1;
    }
  }
  sort_network<C, N>(v);
  C r = fill;
  if (m > 0) {
    if constexpr (PERCENTILE) {
      const C a = pick<C, N>(v, P.lower[m - 1]), b = pick<C, N>(v, P.upper[m - 1]);
      const C t = (C)P.gamma[m - 1], u = (C)P.one_minus_gamma[m - 1];
      const C d = rnd<HALF, C>(b - a);
      r = (t >= (C)0.5) ? rnd<HALF, C>(b - rnd<HALF, C>(d * u)) : rnd<HALF, C>(a + rnd<HALF, C>(d * t));
    } else {
      const int h = m / 2;
      const C high = pick<C, N>(v, h);
      const C low = (m & 1) ? high : pick<C, N>(v, h - 1);
      r = rnd<HALF, C>(rnd<HALF, C>((low + high) + (C)0) / (C)2);
    }
  }
  __builtin_nontemporal_store(r, out + p);
}

template <typename C>
C fill_of(uint64_t bits) {
  gm_scalar s;
  s.u64 = bits;
  return (C)s.f64;
}

template <typename C, bool HALF>
int launch_reduce(const void* stack, int n, long long pixels, int statistic, const gm_percentile_step* plan,
                  void* out, uint64_t fill_bits, hipStream_t s) {
  const C fill = fill_of<C>(fill_bits);
  if (statistic == GM_RED_MEDIAN || statistic == GM_RED_PERCENTILE) {
    const bool percentile = statistic == GM_RED_PERCENTILE;
    if (percentile && !plan) return gm::fail(GM_ERR_INVALID, "gm_reduce_stack: percentile without a plan");
    for (int i = 0; percentile && i < n; ++i)
      if (plan[i].lower < 0 || plan[i].lower > i || plan[i].upper < 0 || plan[i].upper > i)
        return gm::fail(GM_ERR_INVALID, "gm_reduce_stack: plan[%d] points outside %d values", i, i + 1);
    const bool long_stack = n > RD_MAX_SORT || (size_t)n * 256 * sizeof(C) > 64 * 1024;
    if (long_stack && (n > GM_MAX_MEDIAN_FRAMES_LONG || (size_t)n * 64 * sizeof(C) > 160 * 1024))
      return gm::fail(GM_ERR_UNSUPPORTED, "gm_reduce_stack: median / percentile over %d rasters", n);
    SortPlan P;
    memset(&P, 0, sizeof(P));
    gm_percentile_step* plan_dev = nullptr;
    if (percentile && long_stack) {
      int rc = gm_malloc((void**)&plan_dev, sizeof(gm_percentile_step) * (size_t)n);
      if (rc) return rc;
      hipError_t e = hipMemcpyAsync(plan_dev, plan, sizeof(gm_percentile_step) * (size_t)n, hipMemcpyHostToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) {
        (void)gm_free(plan_dev);  // (not leaked on the error path)
        return gm::hip_fail(e, "gm_reduce_stack: percentile plan upload");
      }
    } else if (percentile) {
      for (int i = 0; i < n; ++i) {
        P.lower[i] = (short)plan[i].lower;
        P.upper[i] = (short)plan[i].upper;
        P.gamma[i] = plan[i].gamma;
        P.one_minus_gamma[i] = plan[i].one_minus_gamma;
      }
    }
    // the usual stack: sorted in registers.  Round 6: up to 32 rasters (64 of a 4-byte work dtype) -- the LDS insertion sort that
    // took over from 17 rasters on ran at 0.2 TB/s (60 rasters of 2048^2: 6.6 ms; their maximum 0.23 ms)
    if (n <= 32 || (n <= 64 && sizeof(C) == 4)) {
      const unsigned gxs = (unsigned)((pixels + 255) / 256);
#define GM_SORTED_SMALL(PCT, N_)                                                                                   \
  hipLaunchKernelGGL((reduce_sorted_small_kernel<C, HALF, PCT, N_>), dim3(gxs), dim3(256), 0, s, StackSource<C>{(const C*)stack, pixels}, n, \
                     pixels, P, (C*)out, fill)
      if (n <= 4) { if (percentile) GM_SORTED_SMALL(true, 4); else GM_SORTED_SMALL(false, 4); }
      else if (n <= 8) { if (percentile) GM_SORTED_SMALL(true, 8); else GM_SORTED_SMALL(false, 8); }
      else if (n <= 16) { if (percentile) GM_SORTED_SMALL(true, 16); else GM_SORTED_SMALL(false, 16); }
      else if (n <= 32) { if (percentile) GM_SORTED_SMALL(true, 32); else GM_SORTED_SMALL(false, 32); }
      else if constexpr (sizeof(C) == 4) { if (percentile) GM_SORTED_SMALL(true, 64); else GM_SORTED_SMALL(false, 64); }
#undef GM_SORTED_SMALL
      return GM_OK;
    }
    const int WG = long_stack ? 64 : 256;
    const unsigned gx = (unsigned)((pixels + WG - 1) / WG);
    const size_t lds = (size_t)n * WG * sizeof(C);
    int rc = GM_OK;
#define GM_SORTED_LAUNCH(PCT, WG_, LONG_)                                                                              \
  do {                                                                                                                 \
    auto kernel = reduce_sorted_kernel<C, HALF, PCT, WG_, LONG_>;                                                      \
    if (lds > 64 * 1024 &&                                                                                             \
        hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
      (void)hipGetLastError();                                                                                         \
      rc = gm::fail(GM_ERR_UNSUPPORTED, "gm_reduce_stack: %zu bytes of LDS per workgroup refused", lds);               \
      break;                                                                                                           \
    }                                                                                                                  \
    hipLaunchKernelGGL(kernel, dim3(gx), dim3(WG_), lds, s, (const C*)stack, n, pixels, P,                            \
                       (const gm_percentile_step*)plan_dev, (C*)out, fill);                                            \
  } while (0)
    if (long_stack) {
      if (percentile) GM_SORTED_LAUNCH(true, 64, true);
      else GM_SORTED_LAUNCH(false, 64, true);
    } else {
      if (percentile) GM_SORTED_LAUNCH(true, 256, false);
      else GM_SORTED_LAUNCH(false, 256, false);
    }
#undef GM_SORTED_LAUNCH
    if (plan_dev) (void)gm_free(plan_dev);
    return rc;
  }
  if (n > (RD_PW_BLOCK << RD_PW_DEPTH) &&
      (statistic == GM_RED_SUM || statistic == GM_RED_MEAN || statistic == GM_RED_VAR || statistic == GM_RED_STD)) {
    // sums over more rasters than the unrolled recursion covers: the explicit-stack form, one cell per lane
    const unsigned g1 = (unsigned)((pixels + 255) / 256);
#define GM_RD_DEEP(STAT)                                                                                               \
  case STAT:                                                                                                           \
    hipLaunchKernelGGL((reduce_kernel<C, HALF, STAT, 1, -1>), dim3(g1), dim3(256), 0, s, StackSource<C>{(const C*)stack, pixels}, n, pixels, (C*)out, fill); \
    break;
    switch (statistic) {
      GM_RD_DEEP(GM_RED_SUM) GM_RD_DEEP(GM_RED_MEAN) GM_RD_DEEP(GM_RED_VAR) GM_RD_DEEP(GM_RED_STD)
    }
#undef GM_RD_DEEP
    return GM_OK;
  }
  constexpr int VV = 16 / (int)sizeof(C);
  const bool vec = (pixels % VV) == 0 && ((uintptr_t)stack % 16) == 0 && ((uintptr_t)out % 16) == 0;
  const int V = vec ? VV : 1;
  const unsigned gx = (unsigned)((pixels + 256LL * V - 1) / (256LL * V));
#define GM_RD(STAT)                                                                                                    \
  case STAT:                                                                                                           \
    if (vec) hipLaunchKernelGGL((reduce_kernel<C, HALF, STAT, VV>), dim3(gx), dim3(256), 0, s, StackSource<C>{(const C*)stack, pixels}, n, pixels, (C*)out, fill); \
    else hipLaunchKernelGGL((reduce_kernel<C, HALF, STAT, 1>), dim3(gx), dim3(256), 0, s, StackSource<C>{(const C*)stack, pixels}, n, pixels, (C*)out, fill);      \
    break;
  // The sums of up to 128 rasters are ONE leaf of NumPy's pairwise recursion: the kernel without the recursion's five unrolled
  // halvings (round 6: with them the 16-byte float32 kernel took 252 VGPRs and spilled 128 bytes -- whatever n was --, which is
  // why sums over a stack ran at 1 TB/s next to minima at 4)
#define GM_RD_ADD(STAT)                                                                                                \
  case STAT:                                                                                                           \
    if (n <= RD_PW_BLOCK) {                                                                                            \
      if (vec) hipLaunchKernelGGL((reduce_kernel<C, HALF, STAT, VV, 0>), dim3(gx), dim3(256), 0, s, StackSource<C>{(const C*)stack, pixels}, n, pixels, (C*)out, fill); \
      else hipLaunchKernelGGL((reduce_kernel<C, HALF, STAT, 1, 0>), dim3(gx), dim3(256), 0, s, StackSource<C>{(const C*)stack, pixels}, n, pixels, (C*)out, fill);      \
    } else { /* (129 .. 4096 rasters: one cell per lane -- the 16-byte form of the unrolled recursion spills) */          \
      hipLaunchKernelGGL((reduce_kernel<C, HALF, STAT, 1>), dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, s, StackSource<C>{(const C*)stack, pixels}, n, pixels, (C*)out, fill); \
    }                                                                                                                  \
    break;
  switch (statistic) {
    GM_RD_ADD(GM_RED_SUM) GM_RD(GM_RED_MIN) GM_RD(GM_RED_MAX) GM_RD_ADD(GM_RED_MEAN) GM_RD_ADD(GM_RED_VAR) GM_RD_ADD(GM_RED_STD)
    GM_RD(GM_RED_PRODUCT) GM_RD(GM_RED_ARGMIN) GM_RD(GM_RED_ARGMAX)
    default: return gm::fail(GM_ERR_INVALID, "gm_reduce_stack: statistic %d", statistic);
  }
#undef GM_RD
#undef GM_RD_ADD
  return GM_OK;
}

// n separate rasters of the work dtype, read in place
template <typename C>
int launch_reduce_rasters(const gm_raster_ref* rasters, int n, long long pixels, int statistic, const gm_percentile_step* plan, void* out,
                          uint64_t fill_bits, hipStream_t s) {
  const C fill = fill_of<C>(fill_bits);
  const bool sorted = statistic == GM_RED_MEDIAN || statistic == GM_RED_PERCENTILE;
  if (sorted) {
    if (!(n <= 32 || (n <= 64 && sizeof(C) == 4)))
      return gm::fail(GM_ERR_UNSUPPORTED, "gm_reduce_rasters: median / percentile over %d rasters (up to 32, 64 of a 4-byte dtype: longer stacks take gm_reduce_stack)", n);
    if (statistic == GM_RED_PERCENTILE && !plan) return gm::fail(GM_ERR_INVALID, "gm_reduce_rasters: percentile without a plan");
    for (int i = 0; statistic == GM_RED_PERCENTILE && i < n; ++i)
      if (plan[i].lower < 0 || plan[i].lower > i || plan[i].upper < 0 || plan[i].upper > i)
        return gm::fail(GM_ERR_INVALID, "gm_reduce_rasters: plan[%d] points outside %d values", i, i + 1);
  }
  constexpr int VV = 16 / (int)sizeof(C);
  bool vec = (pixels % VV) == 0 && ((uintptr_t)out % 16) == 0;
  for (int i = 0; i < n; ++i) vec = vec && ((uintptr_t)rasters[i].values % 16) == 0;
  RasterRef* refs = nullptr;
  int rc = gm_malloc((void**)&refs, sizeof(RasterRef) * (size_t)n);
  if (rc) return rc;
  static_assert(sizeof(RasterRef) == sizeof(gm_raster_ref), "gm_raster_ref layout");
  rc = gm_h2d(refs, rasters, sizeof(RasterRef) * (size_t)n);  // (through the calling thread's pinned bounce buffer)
  if (rc) {
    (void)gm_free(refs);
    return rc;
  }
  if (statistic == 1 || statistic == 12 || statistic == 13) {  // count / first / last: the rasters' cells and the index side by side
    const int Vp = vec ? VV : 1;
    const unsigned gp = (unsigned)((pixels + 256LL * Vp - 1) / (256LL * Vp));
#define GM_PICK(STAT)                                                                                                          \
  case STAT:                                                                                                                   \
    if (vec) hipLaunchKernelGGL((pick_rasters_kernel<C, STAT, VV>), dim3(gp), dim3(256), 0, s, refs, n, pixels, (C*)out, fill); \
    else hipLaunchKernelGGL((pick_rasters_kernel<C, STAT, 1>), dim3(gp), dim3(256), 0, s, refs, n, pixels, (C*)out, fill);     \
    break;
    switch (statistic) { GM_PICK(1) GM_PICK(12) GM_PICK(13) }
#undef GM_PICK
    (void)gm_free(refs);
    return GM_OK;
  }
  const RasterSource<C> source{refs};
  if (sorted) {
    const bool percentile = statistic == GM_RED_PERCENTILE;
    SortPlan P;
    memset(&P, 0, sizeof(P));
    for (int i = 0; percentile && i < n; ++i) {
      P.lower[i] = (short)plan[i].lower;
      P.upper[i] = (short)plan[i].upper;
      P.gamma[i] = plan[i].gamma;
      P.one_minus_gamma[i] = plan[i].one_minus_gamma;
    }
    const unsigned gxs = (unsigned)((pixels + 255) / 256);
#define GM_SORTED_RASTERS(PCT, N_)                                                                                              \
  hipLaunchKernelGGL((reduce_sorted_small_kernel<C, false, PCT, N_, RasterSource<C>>), dim3(gxs), dim3(256), 0, s, source, n, pixels, P, \
                     (C*)out, fill)
    if (n <= 4) { if (percentile) GM_SORTED_RASTERS(true, 4); else GM_SORTED_RASTERS(false, 4); }
    else if (n <= 8) { if (percentile) GM_SORTED_RASTERS(true, 8); else GM_SORTED_RASTERS(false, 8); }
    else if (n <= 16) { if (percentile) GM_SORTED_RASTERS(true, 16); else GM_SORTED_RASTERS(false, 16); }
    else if (n <= 32) { if (percentile) GM_SORTED_RASTERS(true, 32); else GM_SORTED_RASTERS(false, 32); }
    else if constexpr (sizeof(C) == 4) { if (percentile) GM_SORTED_RASTERS(true, 64); else GM_SORTED_RASTERS(false, 64); }
#undef GM_SORTED_RASTERS
    (void)gm_free(refs);
    return GM_OK;
  }
  // (sums over more rasters than one leaf of the pairwise recursion: the explicit-stack form, one cell per lane -- the unrolled
  //  recursion spills once the no-data test is part of every load)
  const bool deep = n > RD_PW_BLOCK &&
                    (statistic == GM_RED_SUM || statistic == GM_RED_MEAN || statistic == GM_RED_VAR || statistic == GM_RED_STD);
  const int V = (vec && !deep) ? VV : 1;
  const unsigned gx = (unsigned)((pixels + 256LL * V - 1) / (256LL * V));
#define GM_RR(STAT)                                                                                                                  \
  case STAT:                                                                                                                         \
    if (deep) hipLaunchKernelGGL((reduce_kernel<C, false, STAT, 1, -1, RasterSource<C>>), dim3(gx), dim3(256), 0, s, source, n, pixels, (C*)out, fill); \
    else if (V > 1) hipLaunchKernelGGL((reduce_kernel<C, false, STAT, VV, 0, RasterSource<C>>), dim3(gx), dim3(256), 0, s, source, n, pixels, (C*)out, fill); \
    else hipLaunchKernelGGL((reduce_kernel<C, false, STAT, 1, 0, RasterSource<C>>), dim3(gx), dim3(256), 0, s, source, n, pixels, (C*)out, fill); \
    break;
  switch (statistic) {
    GM_RR(GM_RED_SUM) GM_RR(GM_RED_MIN) GM_RR(GM_RED_MAX) GM_RR(GM_RED_MEAN) GM_RR(GM_RED_VAR) GM_RR(GM_RED_STD)
    GM_RR(GM_RED_PRODUCT) GM_RR(GM_RED_ARGMIN) GM_RR(GM_RED_ARGMAX)
    default: rc = gm::fail(GM_ERR_UNSUPPORTED, "gm_reduce_rasters: statistic %d (medians and percentiles take gm_reduce_stack)", statistic);
  }
#undef GM_RR
  (void)gm_free(refs);  // (the pool is stream-ordered: the launch above reads it before anybody can be handed the block)
  return rc;
}

}  // namespace

extern "C" int gm_reduce_rasters(const gm_raster_ref* rasters, int work_dtype, int64_t n, int64_t pixels, int statistic,
                                 const gm_percentile_step* plan, void* out, uint64_t fill_bits) {
  if (!rasters || !out) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (n <= 0 || n > 0x7fffffff || pixels < 0) return gm::fail(GM_ERR_INVALID, "gm_reduce_rasters: %lld rasters of %lld pixels", (long long)n, (long long)pixels);
  for (int64_t i = 0; i < n; ++i)
    if (!rasters[i].values) return gm::fail(GM_ERR_INVALID, "gm_reduce_rasters: raster %lld is null", (long long)i);
  if (pixels == 0) return GM_OK;
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  int rc;
  if (work_dtype == GM_F32) rc = launch_reduce_rasters<float>(rasters, (int)n, pixels, statistic, plan, out, fill_bits, s);
  else if (work_dtype == GM_F64) rc = launch_reduce_rasters<double>(rasters, (int)n, pixels, statistic, plan, out, fill_bits, s);
  else return gm::fail(GM_ERR_UNSUPPORTED, "gm_reduce_rasters: rasters of dtype %d (float32 / float64 rasters in their own work dtype)", work_dtype);
  if (rc) return rc;
  GM_HIP(hipGetLastError());
  return GM_OK;
}

extern "C" int gm_reduce_stack(const void* stack, int work_dtype, int half, int64_t n, int64_t pixels, int statistic,
                               const gm_percentile_step* plan, void* out, uint64_t fill_bits) {
  if (!stack || !out) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (n <= 0 || n > 0x7fffffff || pixels < 0) return gm::fail(GM_ERR_INVALID, "gm_reduce_stack: %lld rasters of %lld pixels", (long long)n, (long long)pixels);
  if (pixels == 0) return GM_OK;
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  int rc;
  if (work_dtype == GM_F32) rc = half ? launch_reduce<float, true>(stack, (int)n, pixels, statistic, plan, out, fill_bits, s)
                                      : launch_reduce<float, false>(stack, (int)n, pixels, statistic, plan, out, fill_bits, s);
  else if (work_dtype == GM_F64 && !half) rc = launch_reduce<double, false>(stack, (int)n, pixels, statistic, plan, out, fill_bits, s);
  else return gm::fail(GM_ERR_INVALID, "gm_reduce_stack: work dtype %d (half %d)", work_dtype, half);
  if (rc) return rc;
  GM_HIP(hipGetLastError());
  return GM_OK;
}
