// N4: labelled (zonal) statistics -- the numeric core of geometry.AggregateRaster
// (geometry/aggregate.py:160-203): for one frame of a raster and a label raster (cell -> index of the
// geometry that covers it, as utils.rasterize_geoseries paints it), one statistic per label over the
// label's ACTIVE cells (data, and not below the label's threshold when thresholds are given).
//
// The reference computes these with scipy.ndimage.{sum, mean, minimum, maximum, median} and its own
// measurements.percentile, whose results depend on the ORDER of the cells:
//   * sum / mean are np.bincount(labels, weights=values): float64 accumulation cell after cell in raster
//     order -- float64 addition is not associative, so only the same order gives the same bits;
//   * minimum / maximum / median / percentile pick from the cells sorted by (label, value), NaN last.
// Where the order matters the cells are sorted, not reduced in a tree: a STABLE sort by label -- one counting pass for up to
// 2047 labels, this file's own least-significant-digit radix sort (passes of that same counting pass over 11-bit digits)
// beyond -- leaves every label's active cells contiguous and in raster order; the statistics that pick sort by value first.  Then
//   * sum / mean (zonal_sum_kernel, zonal_short_sum_kernel): a wave (a lane for labels of a few cells) adds the label's
//     cells in that order in float64 -- as a tree for the steps that are certified exact, as a chain of dependent
//     additions for the others: bincount's bits either way;
//   * median / percentile (zonal_pick_kernel): a lane per label picks from its sorted segment and combines exactly as
//     SciPy / measurements.percentile do (median of integers in float64, of float32 in float32; the percentile's
//     difference in the raster's own dtype, wrap-around included).
// Where it does not, nothing is sorted: count, min and max take one pass of atomics (zonal_direct_kernel).  Sums and means
// first take the chunk records (round 5: zonal_chunk_kernel / zonal_chunk_combine_kernel, further down): one pass over the
// raster leaves, per stretch of cells and label, a tree sum and what certifies it; a wave per label adds the certified chunks
// in order -- one addition each -- and reads the cells of the others again.  Label rasters that are not made of runs, and
// labels that have to be replayed too often, fall back: sums whose every partial sum is exactly representable -- integer
// rasters, dyadic data, small polygons -- take two passes of atomics (zonal_exact_kernel; when only a few labels fail that
// test only their cells are gathered and sorted), everything else the sort.
// Round 6: what the caller keeps of a sum is its rounding into the float32 `agg` array, and bincount's sequential float64 sum
// lies within a computable bound of ANY float64 sum of the same cells (derived at zonal_chunk_combine_kernel): a label whose sum
// +/- that bound rounds to one float32 is answered without the order, every other label is added in order as above.  The
// test runs first of all -- over the chunk records (labels made of runs), over per-workgroup LDS tables (zonal_tables_lazy_kernel:
// up to 85 labels, labels scattered cell by cell, stripes), over the parallel sums of rasters of thousands of small labels
// (zonal_exact_kernel PHASE 3 + zonal_few_*: the handful that fails added in order from lists gathered off the raster) -- and
// answers nearly every label of nearly every raster in one pass.  GM_ZONAL_LAZY=0: never (the tests run both).
// gm_zonal_statistic_host hands the results back in host arrays with one wait per call.
// Results are float32 like the reference's `agg` array; labels without an active cell stay NaN.
// HBM traffic: 8 algorithmic bytes per cell (value + label); the chunk records move 1.2 x that
// (profiles/r05_zonal_pmc_1024.json), the full sort ~6-7 x.
#include <algorithm>
#include <climits>
#include <cmath>
#include <limits>
#include <type_traits>

#include <mutex>

#include "gm_common.h"

namespace {

enum { Z_SUM = 0, Z_COUNT = 1, Z_MIN = 2, Z_MAX = 3, Z_MEAN = 4, Z_MEDIAN = 5, Z_PERCENTILE = 6 };

// Order-preserving unsigned image of a value in NumPy's sort order: ascending, NaN after everything else.
template <typename T>
struct Ordered {
  typedef typename std::conditional<sizeof(T) == 8, unsigned long long, unsigned>::type Key;
  static __device__ __forceinline__ Key encode(T v) {
    if constexpr (std::is_same<T, float>::value) {
      if (v != v) return 0xFFFFFFFFu;
      const unsigned u = __float_as_uint(v);
      return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    } else if constexpr (std::is_same<T, double>::value) {
      if (v != v) return ~0ull;
      const unsigned long long u = (unsigned long long)__double_as_longlong(v);
      return (u >> 63) ? ~u : (u | (1ull << 63));
    } else if constexpr (std::is_signed<T>::value) {
      typedef typename std::make_unsigned<T>::type U;
      return (Key)(U)((U)v ^ (U)((U)1 << (8 * sizeof(T) - 1)));
    } else {
      return (Key)v;
    }
  }
  static __device__ __forceinline__ T decode(Key k) {
    if constexpr (std::is_same<T, float>::value) {
      if (k == 0xFFFFFFFFu) return std::numeric_limits<float>::quiet_NaN();
      return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
    } else if constexpr (std::is_same<T, double>::value) {
      if (k == ~0ull) return std::numeric_limits<double>::quiet_NaN();
      return __longlong_as_double((long long)((k >> 63) ? (k & ~(1ull << 63)) : ~k));
    } else if constexpr (std::is_signed<T>::value) {
      typedef typename std::make_unsigned<T>::type U;
      return (T)(U)((U)k ^ (U)((U)1 << (8 * sizeof(T) - 1)));
    } else {
      return (T)k;
    }
  }
};

// `present[label] = 1` for every labelled cell, without one store instruction per step: a step whose 64 cells carry the label
// of the step before it (the rule inside a polygon) has nothing new to say.  `*last` is the wave's label of the step before
// (INT_MIN: none / mixed); wave-uniform control flow.
__device__ __forceinline__ void mark_present(unsigned char* __restrict__ present, int label, bool labelled, int* last) {
  if (!present) return;
  const int lead = __builtin_amdgcn_readfirstlane(label);
  const bool uniform = __all(label == lead);
  if (uniform && lead == *last) return;
  if (labelled) present[label] = 1;
  *last = uniform ? lead : INT_MIN;
}

// aggregate.py:177-186: active = frame != no_data_value, and with thresholds only cells whose label has one
// (not NaN) and whose value reaches it; the comparison promotes like NumPy (value vs float32 threshold:
// float32 for float32 and narrow integers, float64 otherwise -- all exact images in float64).
// key = label of an active cell, n_labels for every other cell (sorts behind all labels).
template <typename T>
__global__ __launch_bounds__(256) void zonal_keys_kernel(const T* __restrict__ values, const int* __restrict__ labels, size_t cells,
                                                         int n_labels, int has_ndv, T ndv, const float* __restrict__ thresholds,
                                                         unsigned* __restrict__ keys, typename Ordered<T>::Key* __restrict__ payload,
                                                         unsigned char* __restrict__ present) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += stride) {
    const T v = values[i];
    const int label = labels[i];
    const bool labelled = (unsigned)label < (unsigned)n_labels;
    if (present && labelled) present[label] = 1;
    bool active = labelled && !(has_ndv && v == ndv);
    if (thresholds && active) {
      const float t = thresholds[label];
      active = !(t != t) && (double)v >= (double)t;
    }
    keys[i] = active ? (unsigned)label : (unsigned)n_labels;
    payload[i] = Ordered<T>::encode(v);
  }
}

// first[l] = index of the first sorted cell whose key is >= l, for l = 0 .. n_labels (first[n_labels] = the
// number of active cells): label l owns the sorted cells [first[l], first[l + 1])
__global__ __launch_bounds__(256) void zonal_bounds_kernel(const unsigned* __restrict__ keys, unsigned cells, int n_labels,
                                                           unsigned* __restrict__ first) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l > n_labels) return;
  unsigned lo = 0, hi = cells;
  while (lo < hi) {
    const unsigned mid = lo + (hi - lo) / 2;
    if (keys[mid] < (unsigned)l) lo = mid + 1;
    else hi = mid;
  }
  first[l] = lo;
}

__device__ __forceinline__ void exponent_span(double x, int* top, int* low) {  // x finite, not zero
  const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
  const int e = (int)((bits >> 52) & 0x7ff);
  unsigned long long mant = bits & ((1ull << 52) - 1);
  if (e != 0) mant |= 1ull << 52;
  const int base = (e != 0 ? e : 1) - 1075;  // value = mant * 2^base
  *top = base + 63 - __clzll((long long)mant);
  *low = base + __ffsll((long long)mant) - 1;
}
__device__ __forceinline__ double shuffle_xor(double x, int offset) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = __shfl_xor((unsigned)u, offset), hi = __shfl_xor((unsigned)(u >> 32), offset);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// Wave-wide reductions through the data-parallel-primitive paths of the vector unit (a few cycles per step;
// __shfl goes through the LDS crossbar at ~100): butterflies inside each row of 16 lanes, then the row totals are
// passed on with row_bcast15 / row_bcast31; lane 63 ends up with the total, which is read back for everybody.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_move(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int wave_max(int v) {
  v = max(v, dpp_move<0xB1>(v, v));    // quad_perm [1,0,3,2]
  v = max(v, dpp_move<0x4E>(v, v));    // quad_perm [2,3,0,1]
  v = max(v, dpp_move<0x141>(v, v));   // row_half_mirror
  v = max(v, dpp_move<0x140>(v, v));   // row_mirror
  v = max(v, dpp_move<0x142, 0xA>(INT_MIN, v));  // row_bcast15 into rows 1 and 3
  v = max(v, dpp_move<0x143, 0xC>(INT_MIN, v));  // row_bcast31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min(int v) { return -wave_max(v == INT_MIN ? INT_MAX : -v); }
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_move_f64(double identity, double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v), w = (unsigned long long)__double_as_longlong(identity);
  const unsigned lo = (unsigned)dpp_move<CTRL, ROW_MASK>((int)(unsigned)w, (int)(unsigned)u);
  const unsigned hi = (unsigned)dpp_move<CTRL, ROW_MASK>((int)(unsigned)(w >> 32), (int)(unsigned)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_exact(double v) {  // (for sums known to be exact: the tree's shape is immaterial)
  v += dpp_move_f64<0xB1>(v, v);
  v += dpp_move_f64<0x4E>(v, v);
  v += dpp_move_f64<0x141>(v, v);
  v += dpp_move_f64<0x140>(v, v);
  v += dpp_move_f64<0x142, 0xA>(0.0, v);
  v += dpp_move_f64<0x143, 0xC>(0.0, v);
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, 63), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Inclusive scans over the lanes in lane order through the same paths (shifts inside the rows of 16 lanes -- a lane without a
// source takes the identity --, then the row totals are passed on).
__device__ __forceinline__ double lane_scan_sum_f64(double v) {
  v += dpp_move_f64<0x111>(0.0, v);        // row_shr:1
  v += dpp_move_f64<0x112>(0.0, v);        // row_shr:2
  v += dpp_move_f64<0x114>(0.0, v);        // row_shr:4
  v += dpp_move_f64<0x118>(0.0, v);        // row_shr:8
  v += dpp_move_f64<0x142, 0xA>(0.0, v);   // row_bcast15 into rows 1 and 3
  v += dpp_move_f64<0x143, 0xC>(0.0, v);   // row_bcast31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ double lane_scan_min_f64(double v) {
  const double inf = __builtin_huge_val();
  v = fmin(v, dpp_move_f64<0x111>(inf, v));
  v = fmin(v, dpp_move_f64<0x112>(inf, v));
  v = fmin(v, dpp_move_f64<0x114>(inf, v));
  v = fmin(v, dpp_move_f64<0x118>(inf, v));
  v = fmin(v, dpp_move_f64<0x142, 0xA>(inf, v));
  v = fmin(v, dpp_move_f64<0x143, 0xC>(inf, v));
  return v;
}
__device__ __forceinline__ double lane_scan_max_f64(double v) { return -lane_scan_min_f64(-v); }
__device__ __forceinline__ int lane_scan_min_i32(int v) {
  v = min(v, dpp_move<0x111>(INT_MAX, v));
  v = min(v, dpp_move<0x112>(INT_MAX, v));
  v = min(v, dpp_move<0x114>(INT_MAX, v));
  v = min(v, dpp_move<0x118>(INT_MAX, v));
  v = min(v, dpp_move<0x142, 0xA>(INT_MAX, v));
  v = min(v, dpp_move<0x143, 0xC>(INT_MAX, v));
  return v;
}

// ... and for the counters of a run (zonal_direct_kernel / zonal_exact_kernel flush a run whenever the wave's label changes -- every
// other step inside a 128-cell polygon row: the __shfl_xor butterflies there cost 6 LDS-crossbar round trips per quantity)
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
  v += (unsigned)dpp_move<0xB1>(0, (int)v);
  v += (unsigned)dpp_move<0x4E>(0, (int)v);
  v += (unsigned)dpp_move<0x141>(0, (int)v);
  v += (unsigned)dpp_move<0x140>(0, (int)v);
  v += (unsigned)dpp_move<0x142, 0xA>(0, (int)v);
  v += (unsigned)dpp_move<0x143, 0xC>(0, (int)v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
  v |= (unsigned)dpp_move<0xB1>(0, (int)v);
  v |= (unsigned)dpp_move<0x4E>(0, (int)v);
  v |= (unsigned)dpp_move<0x141>(0, (int)v);
  v |= (unsigned)dpp_move<0x140>(0, (int)v);
  v |= (unsigned)dpp_move<0x142, 0xA>(0, (int)v);
  v |= (unsigned)dpp_move<0x143, 0xC>(0, (int)v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) { return (unsigned)wave_max((int)(v ^ 0x80000000u)) ^ 0x80000000u; }
__device__ __forceinline__ unsigned wave_umin(unsigned v) { return ~wave_umax(~v); }

// top exponent / exponent of the lowest set bit of a cell as bincount sees it ((double)v), from the cell's own
// representation where that is cheaper; zero, NaN and infinities are the caller's business (finite, not zero here)
template <typename T>
__device__ __forceinline__ void cell_span(T v, int* top, int* low) {
  if constexpr (std::is_same<T, float>::value) {
    const unsigned bits = __float_as_uint(v);
    const int e = (int)((bits >> 23) & 0xff);
    const unsigned mant = (bits & 0x7fffffu) | (e != 0 ? 0x800000u : 0u);
    const int base = (e != 0 ? e : 1) - 150;  // value = mant * 2^base
    *top = base + 31 - __clz((int)mant);
    *low = base + __ffs((int)mant) - 1;
  } else if constexpr (std::is_integral<T>::value && sizeof(T) <= 4) {
    const long long wide = (long long)v;
    const unsigned mag = (unsigned)(wide < 0 ? -wide : wide);
    *top = 31 - __clz((int)mag);
    *low = __ffs((int)mag) - 1;
  } else {
    exponent_span((double)v, top, low);
  }
}

// np.bincount(labels, weights=values)[l]: sums[l] += (double)value, cell after cell in raster order.
// One wave per label, 256 cells per step (a lane loads 4 consecutive cells with one 16-byte load).
//   * The step is EXACT when none of its additions can round: with lo = the lowest set bit among the 256 cells and the
//     running sum, every intermediate value is a multiple of 2^lo, and it stays below 2^(lo + 53) when
//     max(top of the sum, top of the cells + 8) + 2 - lo <= 53 -- then the order inside the step is immaterial: every
//     lane adds its 4 cells, the wave adds the 64 partial sums as a tree and puts the total onto the running sum; the
//     bits are those of bincount's 256 additions.  Sums of ordinary data over thousands of cells run on this path.
//   * Otherwise the step is replayed literally: the values parked in LDS, every lane runs the same chain of dependent
//     additions on broadcast reads (the wave as one sequential adder, ~20 cycles per cell).
// Cells past the end of the label add +0.0, which changes nothing (the sum starts from +0.0 like bincount's zeros).
#ifndef GM_ZP_TILE
#define GM_ZP_TILE 16384
#endif
constexpr int ZP_TILE = GM_ZP_TILE;  // cells per tile of the label partition (further down)
constexpr int ZS_PER_LANE = 4, ZS_STEP = 64 * ZS_PER_LANE;
constexpr unsigned ZS_SHORT = 192;  // labels of up to this many active cells: a lane each (zonal_short_sum_kernel)
template <typename T>
__global__ __launch_bounds__(256) void zonal_sum_kernel(const typename Ordered<T>::Key* __restrict__ payload,
                                                        const unsigned* __restrict__ first, int n_labels, int statistic,
                                                        const unsigned char* __restrict__ only, float* __restrict__ out,
                                                        long long* __restrict__ counts, unsigned long_from = 0) {
  typedef typename Ordered<T>::Key Key;
  __shared__ __align__(16) double parked[4][ZS_STEP];
  const int l = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (l >= n_labels || (only && !only[l])) return;  // (only: the labels this launch is about)
  const unsigned a = first[l], b = first[l + 1];
  if (b - a <= ZS_SHORT) return;  // (zonal_short_sum_kernel's)
  if (long_from && b - a >= long_from) return;  // (the block pipeline's: zonal_compose_kernel)
  double* mine = parked[wave];
  double acc = 0.0;
  struct Four { Key k[ZS_PER_LANE]; };
  const Key nothing = Ordered<T>::encode(T(0));
  // The lane's 4 consecutive cells of a step.  The loads of the full steps are unconditional (addresses clamped into
  // the label, nothing selected): with a branch or a select around them the compiler drains all outstanding loads at
  // every step, and a wave that is alone on its SIMD then pays the full memory latency 40 times per 10 k cells.  Only
  // the last, partial step masks the cells past the label (they become zeros).
  const unsigned last = b - 1;  // (b > a here)
  auto fetch = [&](unsigned step) -> Four {
    Four f;
    const unsigned at = a + step * ZS_STEP + ZS_PER_LANE * lane;  // (fewer than 2^31 cells: no wrap for the steps asked)
#pragma unroll
    for (int k = 0; k < ZS_PER_LANE; ++k) f.k[k] = payload[min(at + k, last)];
    return f;
  };
  const unsigned full_steps = (b - a) / ZS_STEP, steps = (b - a + ZS_STEP - 1) / ZS_STEP;
  // Steps are taken UNROLL at a time: what a step needs from its own cells -- their span, the tree sum -- does not depend
  // on the running sum, so the instruction streams of UNROLL steps interleave (a wave that is alone on its SIMD has no
  // other wave to fill the latency of its dependent instructions); only the verdict and the addition go step by step.
  constexpr int UNROLL = 4, AHEAD = 2;  // AHEAD groups of UNROLL steps of cells in flight
  struct Step {
    T cell[ZS_PER_LANE];
    int top, low;
    bool finite;
    double total;
  };
  auto analyse = [&](Four now, unsigned step) -> Step {
    if (step >= full_steps) {  // (wave-uniform; the one partial step, or a step past the label: all zeros)
      const unsigned at = a + step * ZS_STEP + ZS_PER_LANE * lane;
#pragma unroll
      for (int k = 0; k < ZS_PER_LANE; ++k)
        if (at + k >= b || step >= steps) now.k[k] = nothing;
    }
    Step s;
    s.top = INT_MIN;
    s.low = INT_MAX;
    bool finite = true;
    double partial = 0.0;
#pragma unroll
    for (int k = 0; k < ZS_PER_LANE; ++k) {
      s.cell[k] = Ordered<T>::decode(now.k[k]);
      const double x = (double)s.cell[k];
      if constexpr (std::is_floating_point<T>::value) finite = finite && fabs(x) <= 1.7976931348623157e308;
      if (x != 0.0 && fabs(x) <= 1.7976931348623157e308) {
        int t, w;
        cell_span<T>(s.cell[k], &t, &w);
        s.top = max(s.top, t);
        s.low = min(s.low, w);
      }
      partial += x;
    }
    s.top = wave_max(s.top);
    s.low = wave_min(s.low);
    s.finite = __all(finite);
    s.total = wave_sum_exact(partial);  // (only used when the step is exact)
    return s;
  };
  auto apply = [&](const Step& s) {
    int acc_top = INT_MIN, acc_low = INT_MAX;
    const bool acc_finite = fabs(acc) <= 1.7976931348623157e308;
    if (acc_finite && acc != 0.0) exponent_span(acc, &acc_top, &acc_low);
    const int lo = min(s.low, acc_low);
    const long long hi = max((long long)acc_top, (long long)s.top + 8) + 2;  // 256 cells: 8 more bits
    const bool exact = s.finite && acc_finite && (s.top == INT_MIN || (hi - lo <= 53 && hi < 1023));
    if (exact) {
      acc += s.total;
      return;
    }
    // replay: park the 256 values in order (a wave's LDS accesses complete in order: no barrier between its own write
    // and reads), then 4 rounds of: all 32 LDS reads first (128 registers), then the chain of 64 additions
#pragma unroll
    for (int k = 0; k < ZS_PER_LANE; ++k) mine[ZS_PER_LANE * lane + k] = (double)s.cell[k];
    // 16 values (8 sixteen-byte reads) per round, the next round's reads issued before this round's 16 additions: the
    // chain never waits for LDS (8 reads outstanding fit the 4-bit counter the wait instruction can express)
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    constexpr int ROUND = 16, ROUNDS = ZS_STEP / ROUND;
    f64x2 buffer[2][ROUND / 2];
#pragma unroll
    for (int k = 0; k < ROUND / 2; ++k) buffer[0][k] = *reinterpret_cast<const f64x2*>(mine + 2 * k);
#pragma unroll
    for (int round = 0; round < ROUNDS; ++round) {
      if (round + 1 < ROUNDS) {
#pragma unroll
        for (int k = 0; k < ROUND / 2; ++k)
          buffer[(round + 1) & 1][k] = *reinterpret_cast<const f64x2*>(mine + ROUND * (round + 1) + 2 * k);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < ROUND / 2; ++k) {
        acc += buffer[round & 1][k].x;
        acc += buffer[round & 1][k].y;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  Four ahead[AHEAD][UNROLL];
#pragma unroll
  for (int g = 0; g < AHEAD; ++g)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) ahead[g][u] = fetch((unsigned)(g * UNROLL + u));
  for (unsigned step = 0; step < steps; step += UNROLL) {
    Step group[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) group[u] = analyse(ahead[0][u], step + u);
#pragma unroll
    for (int g = 0; g + 1 < AHEAD; ++g)
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) ahead[g][u] = ahead[g + 1][u];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) ahead[AHEAD - 1][u] = fetch(step + AHEAD * UNROLL + u);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (step + u < steps) apply(group[u]);  // (wave-uniform)
  }
  if (lane == 0) {
    const unsigned n = b - a;
    counts[l] = n;
    const float nan = std::numeric_limits<float>::quiet_NaN();
    // ndimage.mean: sum / count in float64; the assignment into the float32 `agg` rounds once
    out[l] = n == 0 ? nan : (statistic == Z_MEAN ? (float)(acc / (double)n) : (float)acc);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Long labels (round 3): bincount's sequential float64 sum, block-parallel.
//
// One wave adding one label's cells in order -- the kernel above -- is bound by the chain of dependent additions
// whenever a step cannot be certified exact (12.6 M cells of wide-range float32 data under ONE label: 41.6 ms).  What
// makes the sum parallel after all: while the running sum stays inside one binade [2^E, 2^(E+1)) its unit in the last
// place u = 2^(E-52) does not move, and adding a cell x is adding round(x / u) units -- the direction of the rounding
// depends on the cell alone, EXCEPT for an exact tie (x mod u == u / 2), which rounds to even, i.e. depends on the
// parity of the sum before it ... and leaves an even sum behind, whatever came before.  So a block of cells acts on
// the sum s (in units of u) as s -> s + D[s mod 2] with two precomputable integers D[0], D[1], valid while every
// prefix stays inside the binade, which two more numbers per parity (the lowest and the highest prefix) decide.
//   1. zonal_block_sum_kernel: a plain tree sum per block of 256 sorted cells (blocks aligned to the sorted array);
//   2. zonal_block_prefix_kernel: per long label the running total BEFORE each of its blocks -- approximate (tree
//      order), only used to PREDICT the binade the exact sum will be in there;
//   3. zonal_block_info_kernel: per block, for the predicted binade, D / lowest / highest prefix for both parities (the
//      256 cells in order: a lane composes its 4 cells, the wave scans the 64 parity -> (sum, parity) functions);
//   4. zonal_compose_kernel: a wave per label walks its blocks: when the exact running sum is in the predicted binade and
//      the block's prefixes stay inside it, the block is ONE integer addition; otherwise -- the sum crosses a power of
//      two inside the block, the sum is still small against the cells, a cell is not finite, the prediction was off
//      -- the block's 256 cells are added literally, like above.  Correctness never rests on the prediction.
// The restatement was checked against literal float64 accumulation on 7 000 random and edge-case blocks before it
// was written down here (sums next to powers of two, exact half-unit cells, both signs).
constexpr unsigned ZB = 256;            // cells per block
// labels of at least this many cells take the pipeline (1024 labels of 11 k cells of zero-mean data: 0.52 ms on the
// wave-per-label kernel, 0.61 through the pipeline -- their sums stay small against the cells, most blocks are replayed);
// GM_ZONAL_LONG overrides it (the tests run small labels through the pipeline too)
constexpr unsigned ZL_LONG = 65536;
struct BlockInfo {  // 32 bytes
  // Units added for an even incoming sum; for an odd one D0 + delta (the two trajectories take the same steps until the first
  // exact tie, differ by one unit in that step, and are both even -- hence identical -- from there on: delta is -1, 0 or 1).
  long long D0;
  // lowest / highest prefix relative to the incoming sum, over BOTH parities (they differ by at most that one unit, so the
  // test is conservative by at most one unit -- a block it turns away needlessly is added literally, which is always right)
  long long lo, hi;
  int E;      // the binade the block was analysed for; INT_MIN: not analysed (add it literally)
  int flags;  // bit 0: the sum it was analysed for is negative (the block is mirrored); bit 1: expected to be added literally;
              // bits 2-3: delta + 1
  __device__ __forceinline__ long long D(int parity) const { return D0 + (parity ? ((flags >> 2) & 3) - 1 : 0); }
};

// exponent of the lowest set bit of a finite non-zero double: x is an odd multiple of 2^lowbit_exp(x)
__device__ __forceinline__ int lowbit_exp(double x) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(x) & 0x7fffffffffffffffull;
  const int field = (int)(bits >> 52);
  const unsigned long long mantissa = bits & 0xfffffffffffffull;
  if (field == 0) return -1074 + __builtin_ctzll(mantissa);  // subnormal
  return field - 1075 + __builtin_ctzll(mantissa | (1ull << 52));
}

// Besides the tree sum, what CERTIFIES it (round 4): the sum of the |cells| and the finest bit any cell carries.  A block
// whose cells are all multiples of 2^g, entered by a running sum s that is one too, with |s| + sum|x| < 2^(53 + g), is
// added without a single rounding -- every prefix is a multiple of 2^g below 2^(53 + g), i.e. a float64 -- so the order does
// not matter, the tree sum is that exact sum, and s + tree sum is what bincount's chain leaves, whatever powers of two the
// prefixes cross on the way.  The composing wave asks this of the blocks the binade analysis turns away (zero-mean data:
// sums that are small against the cells, sums that wander around a power of two) before it adds them literally.
template <typename T>
__global__ __launch_bounds__(256) void zonal_block_sum_kernel(const typename Ordered<T>::Key* __restrict__ payload,
                                                              const unsigned* __restrict__ keys, unsigned cells, int n_labels,
                                                              double* __restrict__ blk_sum, int* __restrict__ blk_label,
                                                              double* __restrict__ blk_abs, int* __restrict__ blk_gexp) {
  const unsigned j = (unsigned)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if ((size_t)j * ZB >= cells) return;
  const unsigned base = j * ZB, last = cells - 1;
  double partial = 0.0, magnitude = 0.0;
  int finest = INT_MAX;  // INT_MAX: nothing but zeros; INT_MIN: a cell that is not finite
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned at = base + 4 * lane + k;
    const double x = at <= last ? (double)Ordered<T>::decode(payload[at]) : 0.0;
    partial += x;
    magnitude += fabs(x);
    if (!(fabs(x) <= 1.7976931348623157e308)) finest = INT_MIN;
    else if (x != 0.0 && finest != INT_MIN) finest = min(finest, lowbit_exp(x));
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    partial += shuffle_xor(partial, off);
    magnitude += shuffle_xor(magnitude, off);
    finest = min(finest, __shfl_xor(finest, off));
  }
  if (lane == 0) {
    blk_abs[j] = magnitude;
    blk_gexp[j] = finest;
    blk_sum[j] = partial;
    const unsigned ka = keys[base], kb = keys[min(base + ZB - 1, last)];
    blk_label[j] = (base + ZB <= cells && ka == kb && ka < (unsigned)n_labels) ? (int)ka : -1;
  }
}

// wave-wide exclusive prefix sum of one double per lane (Hillis-Steele over shuffles); total in *total
__device__ __forceinline__ double shuffle_up_f64(double x, int delta) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = __shfl_up((unsigned)u, delta), hi = __shfl_up((unsigned)(u >> 32), delta);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// A workgroup of eight waves per long label.  A round is 64 x PER consecutive block sums (a lane takes PER of them; any order
// of addition will do: the result only predicts); wave w takes the rounds w, w + 8, ...: first their totals (into LDS), then
// -- once one wave has turned the totals into the sums before every round -- the prefixes inside them.  (One label covering
// the default raster limit, 49 152 blocks: 362 us with one wave, one block per lane and round and the loads not ahead; 59 us
// with 16 blocks per lane, the loads ahead and four waves; round 5: eight waves, the loads of the first pass ahead as well and
// the sums over the lanes through the data-parallel primitives instead of the LDS crossbar -- 63 -> 14 us.)
constexpr int ZPRE_WAVES = 8;
template <typename T>
__global__ __launch_bounds__(64 * ZPRE_WAVES) void zonal_block_prefix_kernel(const typename Ordered<T>::Key* __restrict__ payload,
                                                                             const unsigned* __restrict__ first, int n_labels,
                                                                             const double* __restrict__ blk_sum,
                                                                             double* __restrict__ blk_pred, unsigned long_from) {
  constexpr int PER = 16, ROUND = 64 * PER, RMAX = 2048;  // (RMAX rounds = 2 M blocks = 537 M cells; more: one wave, in order)
  __shared__ double before_round[RMAX];
  const int l = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (l >= n_labels) return;
  const unsigned a = first[l], b = first[l + 1];
  if (b - a < long_from) return;  // (workgroup-uniform)
  const unsigned j0 = (a + ZB - 1) / ZB, j1 = b / ZB;  // the label's whole blocks [j0, j1)
  if (j0 >= j1) return;
  const unsigned rounds = (j1 - j0 + ROUND - 1) / ROUND;
  const bool team = rounds <= (unsigned)RMAX;
  if (!team && wave != 0) return;
  double head = 0.0;  // the cells before the first whole block
  {
    double partial = 0.0;
    for (unsigned at = a + lane; at < j0 * ZB; at += 64) partial += (double)Ordered<T>::decode(payload[at]);
    head = wave_sum_exact(partial);
  }
  // (a lane's PER blocks of a round are 64 apart: every load and store of the wave is one run of 512 bytes -- with PER consecutive
  //  blocks per lane each of them touched 64 cache lines, 7 000 cycles per round and pass)
  auto fetch = [&](unsigned j, double* v) {
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = blk_sum[min(j + 64u * k + lane, j1 - 1)];
  };
  const unsigned stride = team ? (unsigned)ZPRE_WAVES : 1u, mine_first = team ? (unsigned)wave : 0u;
  double ahead[PER];
  if (team) {
    // pass A: the totals of this wave's rounds (the next round's sums requested before this round's are added up)
    if (mine_first < rounds) fetch(j0 + mine_first * ROUND, ahead);
    for (unsigned r = mine_first; r < rounds; r += stride) {
      const unsigned j = j0 + r * ROUND;
      double total = 0.0;
#pragma unroll
      for (int k = 0; k < PER; ++k) total += j + 64u * k + lane < j1 ? ahead[k] : 0.0;
      if (r + stride < rounds) fetch(j0 + (r + stride) * ROUND, ahead);
      total = wave_sum_exact(total);
      if (lane == 0) before_round[r] = total;
    }
    __syncthreads();
    if (wave == 0) {  // the sums before every round: a lane takes RMAX / 64 consecutive rounds, the wave scans the lane totals
      constexpr int CHUNK = RMAX / 64;
      double mine[CHUNK], total = 0.0;
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) {
        const unsigned r = (unsigned)(CHUNK * lane + k);
        mine[k] = r < rounds ? before_round[r] : 0.0;
        total += mine[k];
      }
      const double scan = lane_scan_sum_f64(total);
      double before = head + (scan - total);
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) {
        const unsigned r = (unsigned)(CHUNK * lane + k);
        if (r < rounds) before_round[r] = before;
        before += mine[k];
      }
    }
    __syncthreads();
  }
  // pass B: the prefixes inside the rounds (team: each round starts from its sum before; one wave: a running sum); a round is PER
  // slabs of 64 consecutive blocks, a lane each: a scan over the lanes per slab, the slabs' totals carried along
  double running = head;
  if (mine_first < rounds) fetch(j0 + mine_first * ROUND, ahead);
  for (unsigned r = mine_first; r < rounds; r += stride) {
    const unsigned j = j0 + r * ROUND;
    double mine[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) mine[k] = j + 64u * k + lane < j1 ? ahead[k] : 0.0;
    if (r + stride < rounds) fetch(j0 + (r + stride) * ROUND, ahead);
    if (team) running = before_round[r];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const double scan = lane_scan_sum_f64(mine[k]);
      if (j + 64u * k + lane < j1) blk_pred[j + 64u * k + lane] = running + (scan - mine[k]);
      const unsigned long long u = (unsigned long long)__double_as_longlong(scan);
      running += __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(u >> 32), 63) << 32) |
                                                  (unsigned)__builtin_amdgcn_readlane((int)u, 63)));
    }
  }
}

// what a run of cells does to (sum in units of u, parity): for either parity coming in
struct ParityFn {
  double S[2];
  int q[2];
};
__device__ __forceinline__ ParityFn compose(const ParityFn& first_fn, const ParityFn& then) {
  ParityFn r;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int mid = first_fn.q[p];
    r.S[p] = first_fn.S[p] + (mid ? then.S[1] : then.S[0]);
    r.q[p] = mid ? then.q[1] : then.q[0];
  }
  return r;
}
__device__ __forceinline__ ParityFn shuffle_up_fn(const ParityFn& f, int delta) {
  ParityFn r;
  r.S[0] = shuffle_up_f64(f.S[0], delta);
  r.S[1] = shuffle_up_f64(f.S[1], delta);
  const int packed = __shfl_up(f.q[0] | (f.q[1] << 1), delta);
  r.q[0] = packed & 1;
  r.q[1] = (packed >> 1) & 1;
  return r;
}

// what a lane's 4 cells do to (sum in units of u, parity) for the binade E and sign of the sum, and the lowest / highest prefix
// inside them, for either parity coming in
struct LaneFn {
  ParityFn f;
  double lo[2], hi[2];
};
__device__ __forceinline__ LaneFn lane_function(const double* x, int E, bool negative) {
  const double scale = E == INT_MIN ? 1.0 : ldexp(negative ? -1.0 : 1.0, 52 - E);  // cell -> signed units of u
  double k[4], frac[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double t = x[c] * scale;  // exact: a power of two (|E| < 900 keeps it far from overflow and underflow)
    k[c] = floor(t);
    frac[c] = t - k[c];             // exact, in [0, 1)
  }
  LaneFn r;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    int parity = p;
    double sum = 0.0, low = 1.0e300, high = -1.0e300;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k_odd = (int)(k[c] - 2.0 * floor(k[c] * 0.5));
      // the rounding: up for more than half a unit, to even on exactly half (which leaves an even sum)
      const int up = frac[c] == 0.5 ? (parity ^ k_odd) : (frac[c] > 0.5 ? 1 : 0);
      sum += k[c] + (double)up;
      parity ^= k_odd ^ up;
      low = fmin(low, sum);
      high = fmax(high, sum);
    }
    r.f.S[p] = sum;
    r.f.q[p] = parity;
    r.lo[p] = low;
    r.hi[p] = high;
  }
  return r;
}
// the functions of WIDTH consecutive lanes composed in lane order (the group's last lane ends up with the total) and the
// lowest / highest prefix anywhere in the group (every lane of the group ends up with them)
template <int WIDTH>
__device__ __forceinline__ void group_function(const LaneFn& m, int lane, ParityFn* total, double* lowest, double* highest) {
  const int at = lane & (WIDTH - 1);
  ParityFn scan = m.f;
#pragma unroll
  for (int off = 1; off < WIDTH; off <<= 1) {
    const ParityFn before = shuffle_up_fn(scan, off);
    if (at >= off) scan = compose(before, scan);
  }
  ParityFn excl = shuffle_up_fn(scan, 1);
  if (at == 0) {
    excl.S[0] = excl.S[1] = 0.0;
    excl.q[0] = 0;
    excl.q[1] = 1;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    lowest[p] = excl.S[p] + (excl.q[p] ? m.lo[1] : m.lo[0]);
    highest[p] = excl.S[p] + (excl.q[p] ? m.hi[1] : m.hi[0]);
#pragma unroll
    for (int off = 1; off < WIDTH; off <<= 1) {
      lowest[p] = fmin(lowest[p], shuffle_xor(lowest[p], off));
      highest[p] = fmax(highest[p], shuffle_xor(highest[p], off));
    }
  }
  *total = scan;
}
// the description of a group of cells from its composed function (call on the group's last lane); *risky: the sum predicted
// to enter it would take a prefix out of the binade (or the group is not analysable at all)
__device__ __forceinline__ BlockInfo describe(const ParityFn& total, const double* lowest, const double* highest, int E, bool negative,
                                              double pred, bool all_finite) {
  BlockInfo out;
  // (offsets that could take a sum out of its binade anyway -- beyond 2^60 units -- are not worth an integer)
  bool small = true;
#pragma unroll
  for (int p = 0; p < 2; ++p) small = small && fabs(total.S[p]) < 0x1p60 && fabs(lowest[p]) < 0x1p60 && fabs(highest[p]) < 0x1p60;
  const long long d0 = small ? (long long)total.S[0] : 0, d1 = small ? (long long)total.S[1] : 0;
  const long long delta = d1 - d0;  // (-1, 0 or 1: see BlockInfo; anything else would be a flaw in the analysis)
  out.E = (E != INT_MIN && all_finite && small && delta >= -1 && delta <= 1) ? E : INT_MIN;
  out.D0 = d0;
  out.lo = small ? (long long)fmin(lowest[0], lowest[1]) : 0;
  out.hi = small ? (long long)fmax(highest[0], highest[1]) : 0;
  // Will the group be added literally?  Known ahead with near certainty: the predicted sum entering it is off the exact one
  // by parts in 10^9 of its binade, its prefixes range over parts in 10^3 and more.
  bool risky = out.E == INT_MIN;
  if (!risky) {
    const double entering = fabs(pred) * ldexp(1.0, 52 - E), slack = 0x1p30;  // units of u, in [2^52, 2^53)
#pragma unroll
    for (int p = 0; p < 2; ++p) risky = risky || entering + lowest[p] < 0x1p52 + slack || entering + highest[p] > 0x1p53 - slack;
  }
  out.flags = (negative ? 1 : 0) | (risky ? 2 : 0) | ((int)((delta >= -1 && delta <= 1 ? delta : 0) + 1) << 2);
  return out;
}
__device__ __forceinline__ int binade_of(double pred) {
  if (pred != 0.0 && fabs(pred) <= 1.7976931348623157e308) {
    const int e = ilogb(fabs(pred));
    if (e > -900 && e < 900) return e;
  }
  return INT_MIN;
}

// Per block: its description for the binade the predicted sum is in (flags bit 1: the block will be added literally -- the
// composing wave requests the cells of those blocks ahead).
template <typename T>
__global__ __launch_bounds__(256) void zonal_block_info_kernel(const typename Ordered<T>::Key* __restrict__ payload,
                                                               const unsigned* __restrict__ first, unsigned cells,
                                                               const int* __restrict__ blk_label, const double* __restrict__ blk_pred,
                                                               BlockInfo* __restrict__ info, unsigned long_from) {
  const unsigned j = (unsigned)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if ((size_t)j * ZB >= cells) return;
  const int l = blk_label[j];
  if (l < 0) return;
  if (first[l + 1] - first[l] < long_from) return;
  const double pred = blk_pred[j];
  const int E = binade_of(pred);
  const bool negative = pred < 0.0;
  double x[4];
  bool finite = true;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    x[c] = (double)Ordered<T>::decode(payload[j * ZB + 4 * lane + c]);
    finite = finite && fabs(x[c]) <= 1.7976931348623157e308;
  }
  const unsigned long long finite_lanes = __ballot(finite);
  const LaneFn mine = lane_function(x, E, negative);
  ParityFn total;
  double lowest[2], highest[2];
  group_function<64>(mine, (int)lane, &total, lowest, highest);
  if (lane == 63) info[j] = describe(total, lowest, highest, E, negative, pred, finite_lanes == ~0ull);
}

// the same composition over whole blocks, in integer units
struct UnitsFn {
  long long S[2];
  int q[2];
};
__device__ __forceinline__ UnitsFn compose_units(const UnitsFn& first_fn, const UnitsFn& then) {
  UnitsFn r;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int mid = first_fn.q[p];
    r.S[p] = first_fn.S[p] + (mid ? then.S[1] : then.S[0]);
    r.q[p] = mid ? then.q[1] : then.q[0];
  }
  return r;
}
__device__ __forceinline__ long long shuffle_up_i64(long long x, int delta) {
  const unsigned lo = __shfl_up((unsigned)x, delta), hi = __shfl_up((unsigned)((unsigned long long)x >> 32), delta);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ UnitsFn shuffle_up_units(const UnitsFn& f, int delta) {
  UnitsFn r;
  r.S[0] = shuffle_up_i64(f.S[0], delta);
  r.S[1] = shuffle_up_i64(f.S[1], delta);
  const int packed = __shfl_up(f.q[0] | (f.q[1] << 1), delta);
  r.q[0] = packed & 1;
  r.q[1] = (packed >> 1) & 1;
  return r;
}

// The inclusive composition over the lanes in lane order, through the data-parallel primitives (shifts inside the rows of 16
// lanes, then the row totals passed on: see wave_scan_f64); a lane without a source composes with the identity.  Until round 5
// the scan went through __shfl_up: five LDS-crossbar round trips per step and function, six steps -- most of what the composing
// wave of a long label spent per group of blocks it had to walk.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ UnitsFn dpp_units(const UnitsFn& f) {
  UnitsFn r;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const unsigned lo = (unsigned)dpp_move<CTRL, ROW_MASK>(0, (int)(unsigned)f.S[p]);
    const unsigned hi = (unsigned)dpp_move<CTRL, ROW_MASK>(0, (int)(unsigned)((unsigned long long)f.S[p] >> 32));
    r.S[p] = (long long)(((unsigned long long)hi << 32) | lo);
  }
  const int packed = dpp_move<CTRL, ROW_MASK>(2, f.q[0] | (f.q[1] << 1));  // (identity: parity p stays p)
  r.q[0] = packed & 1;
  r.q[1] = (packed >> 1) & 1;
  return r;
}
__device__ __forceinline__ UnitsFn scan_units(UnitsFn scan) {
  scan = compose_units(dpp_units<0x111>(scan), scan);       // row_shr:1
  scan = compose_units(dpp_units<0x112>(scan), scan);       // row_shr:2
  scan = compose_units(dpp_units<0x114>(scan), scan);       // row_shr:4
  scan = compose_units(dpp_units<0x118>(scan), scan);       // row_shr:8
  scan = compose_units(dpp_units<0x142, 0xA>(scan), scan);  // row_bcast15 into rows 1 and 3
  scan = compose_units(dpp_units<0x143, 0xC>(scan), scan);  // row_bcast31 into rows 2 and 3
  return scan;
}

// The same once more over groups of 256 blocks (aligned to the block index): what the group adds for either parity of the sum
// entering it and the extreme prefixes anywhere inside it -- usable when its 256 blocks are whole blocks of ONE long label,
// analysed for one binade and sign, none of them expected to be added literally.  The composing wave then takes such a
// group in one step (and still checks it against the exact sum entering it); groups that hold a flagged block -- or any
// other irregularity -- are walked block by block as before.
struct GroupInfo {
  long long D[2], lo[2], hi[2];
  int E;      // INT_MIN: not usable
  int flags;  // bit 0: negative
};
__device__ __forceinline__ long long shuffle_xor_i64(long long x, int offset) {
  const unsigned lo = __shfl_xor((unsigned)x, offset), hi = __shfl_xor((unsigned)((unsigned long long)x >> 32), offset);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
constexpr int ZG = 256;  // blocks per group

__global__ __launch_bounds__(256) void zonal_group_info_kernel(const int* __restrict__ blk_label, const BlockInfo* __restrict__ info,
                                                               const unsigned* __restrict__ first, unsigned whole_blocks,
                                                               unsigned long_from, GroupInfo* __restrict__ ginfo, unsigned n_groups) {
  const unsigned G = (unsigned)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (G >= n_groups) return;
  GroupInfo out;
  out.D[0] = out.D[1] = out.lo[0] = out.lo[1] = out.hi[0] = out.hi[1] = 0;
  out.E = INT_MIN;
  out.flags = 0;
  const unsigned base = G * ZG;
  if (base + ZG > whole_blocks) {  // (wave-uniform)
    if (lane == 0) ginfo[G] = out;
    return;
  }
  BlockInfo B[4];
  bool regular = true;
  const int l = __builtin_amdgcn_readfirstlane(blk_label[base]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    B[k] = info[base + 4 * lane + k];
    regular = regular && blk_label[base + 4 * lane + k] == l;
  }
  const int E0 = __builtin_amdgcn_readfirstlane(B[0].E), neg0 = __builtin_amdgcn_readfirstlane(B[0].flags & 1);
#pragma unroll
  for (int k = 0; k < 4; ++k) regular = regular && B[k].E == E0 && (B[k].flags & 3) == neg0;  // (same sign, not flagged)
  const bool long_label = l >= 0 && first[max(l, 0) + 1] - first[max(l, 0)] >= long_from;
  if (!(__all(regular) && long_label && E0 != INT_MIN)) {
    if (lane == 0) ginfo[G] = out;
    return;
  }
  // the lane's four blocks in order, for either parity coming in
  UnitsFn mine;
  long long llo[2], lhi[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    long long run = 0, low = LLONG_MAX, high = LLONG_MIN;
    int parity = p;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      low = min(low, run + B[k].lo);
      high = max(high, run + B[k].hi);
      const long long d = B[k].D(parity);
      run += d;
      parity = (int)((parity + d) & 1);
    }
    mine.S[p] = run;
    mine.q[p] = parity;
    llo[p] = low;
    lhi[p] = high;
  }
  const UnitsFn scan = scan_units(mine);
  UnitsFn excl = shuffle_up_units(scan, 1);
  if (lane == 0) {
    excl.S[0] = excl.S[1] = 0;
    excl.q[0] = 0;
    excl.q[1] = 1;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    long long lowest = excl.S[p] + (excl.q[p] ? llo[1] : llo[0]), highest = excl.S[p] + (excl.q[p] ? lhi[1] : lhi[0]);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      lowest = min(lowest, shuffle_xor_i64(lowest, off));
      highest = max(highest, shuffle_xor_i64(highest, off));
    }
    out.lo[p] = lowest;
    out.hi[p] = highest;
    out.D[p] = scan.S[p];  // (lane 63's is the group's)
  }
  out.E = E0;
  out.flags = neg0;
  if (lane == 63) ginfo[G] = out;
}

// The exactness certificate of zonal_block_sum_kernel for a whole group of 256 blocks (round 4).  With t_j the blocks' tree
// sums, o_j = t_0 + ... + t_(j-1) and a_j their sums of |cells|, every prefix inside the group lies, relative to the sum
// entering the group, in [lo, hi] = [min_j (o_j - a_j), max_j (o_j + a_j)].  The composing wave certifies the group for the
// exact sum s entering it when max(|s + lo|, |s + hi|) < 2^(53 + g), g = min(finest bit of the group's cells, lowest bit of s):
// no addition of the group rounds, and s + total is what the chain leaves.  (hi - lo < 2^(52 + g) makes sure of the o_j
// themselves: every range of blocks then sums without rounding.)  Usable under the conditions of GroupInfo: 256 whole blocks
// of ONE long label, every cell finite.
struct GroupExact {
  double total, lo, hi;
  int grain;  // INT_MIN: not usable; INT_MAX: nothing but zeros
  int pad;
};
__global__ __launch_bounds__(256) void zonal_group_exact_kernel(const int* __restrict__ blk_label, const double* __restrict__ blk_sum,
                                                                const double* __restrict__ blk_abs, const int* __restrict__ blk_gexp,
                                                                const unsigned* __restrict__ first, unsigned whole_blocks,
                                                                unsigned long_from, GroupExact* __restrict__ gexact, unsigned n_groups) {
  const unsigned G = (unsigned)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (G >= n_groups) return;
  GroupExact out;
  out.total = out.lo = out.hi = 0.0;
  out.grain = INT_MIN;
  out.pad = 0;
  const unsigned base = G * ZG;
  if (base + ZG > whole_blocks) {  // (wave-uniform)
    if (lane == 0) gexact[G] = out;
    return;
  }
  const int l = __builtin_amdgcn_readfirstlane(blk_label[base]);
  bool regular = true;
  int finest = INT_MAX;
  double t[4], a[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned j = base + 4 * lane + k;
    regular = regular && blk_label[j] == l && blk_gexp[j] != INT_MIN;
    finest = min(finest, blk_gexp[j]);
    t[k] = blk_sum[j];
    a[k] = blk_abs[j];
  }
  const bool long_label = l >= 0 && first[max(l, 0) + 1] - first[max(l, 0)] >= long_from;
  if (!(__all(regular) && long_label)) {
    if (lane == 0) gexact[G] = out;
    return;
  }
  double run = 0.0, low = 1.0e300, high = -1.0e300;  // the lane's four blocks, relative to the sum entering the lane
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    low = fmin(low, run - a[k]);
    high = fmax(high, run + a[k]);
    run += t[k];
  }
  double scan = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double up = shuffle_up_f64(scan, off);
    if (lane >= off) scan += up;
  }
  const double before = scan - run;  // (exact whenever the certificate will hold: see above)
  low += before;
  high += before;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    low = fmin(low, shuffle_xor(low, off));
    high = fmax(high, shuffle_xor(high, off));
    finest = min(finest, __shfl_xor(finest, off));
  }
  out.total = scan;  // (lane 63's is the group's)
  out.lo = low;
  out.hi = high;
  out.grain = finest;
  if (lane == 63) gexact[G] = out;
}

template <typename T>
__global__ __launch_bounds__(256) void zonal_compose_kernel(const typename Ordered<T>::Key* __restrict__ payload,
                                                            const unsigned* __restrict__ first, int n_labels, int statistic,
                                                            const int* __restrict__ blk_label, const BlockInfo* __restrict__ info,
                                                            const GroupInfo* __restrict__ ginfo, float* __restrict__ out,
                                                            long long* __restrict__ counts,
                                                            unsigned* __restrict__ replayed, unsigned long_from,
                                                            const double* __restrict__ blk_sum, const double* __restrict__ blk_abs,
                                                            const int* __restrict__ blk_gexp, const GroupExact* __restrict__ gexact) {
  __shared__ __align__(16) double parked[4][ZB];
  const int l = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (l >= n_labels) return;
  const unsigned a = first[l], b = first[l + 1];
  if (b - a < long_from) return;
  double* mine = parked[wave];
  double acc = 0.0;
  // cells [from, to), at most ZB of them, added one after the other (the chain of zonal_sum_kernel's replay)
  typedef typename Ordered<T>::Key Key;
  // the n parked values (n <= ZB; the rest of the last round is +0.0 padding) added one after the other: 16 values (8
  // sixteen-byte reads) per round, the next round's reads issued before this round's 16 additions -- the chain never waits for LDS
  auto chain = [&](unsigned n) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    constexpr int ROUND = 16;
    const int rounds = (int)((n + ROUND - 1) / ROUND);
    f64x2 buffer[2][ROUND / 2];
#pragma unroll
    for (int k = 0; k < ROUND / 2; ++k) buffer[0][k] = *reinterpret_cast<const f64x2*>(mine + 2 * k);
#pragma unroll 1  // (a real loop also where n is a constant: unrolled, the optimiser hoists every LDS read to the front -- 512 registers)
    for (int round = 0; round < rounds; round += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int next = min(round + half + 1, (int)(ZB / ROUND) - 1);
#pragma unroll
        for (int k = 0; k < ROUND / 2; ++k) buffer[(half + 1) & 1][k] = *reinterpret_cast<const f64x2*>(mine + ROUND * next + 2 * k);
        __builtin_amdgcn_sched_barrier(0);
        if (round + half < rounds) {
#pragma unroll
          for (int k = 0; k < ROUND / 2; ++k) {
            acc += buffer[half & 1][k].x;
            acc += buffer[half & 1][k].y;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto literally = [&](unsigned from, unsigned to) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned at = from + 4 * lane + k;
      mine[4 * lane + k] = at < to ? (double)Ordered<T>::decode(payload[at]) : 0.0;  // (+0.0 changes nothing)
    }
    chain(to - from);
  };
  struct Cells {
    Key k[4];  // the lane's four cells of the block
  };
  auto request = [&](unsigned block) -> Cells {  // (used later: the loads are only issued here)
    Cells c;
#pragma unroll
    for (int k = 0; k < 4; ++k) c.k[k] = payload[block * ZB + 4 * lane + k];
    return c;
  };
  const unsigned j0 = (a + ZB - 1) / ZB, j1 = b / ZB;
  unsigned n_replayed = 0, n_ahead = 0, n_exact = 0, n_exact_groups = 0;  // literal blocks; ... requested ahead; certified blocks / groups
  unsigned n_scans = 0, n_walked = 0, n_binade_groups = 0;               // compositions of a group's rest; groups walked / taken in one step
  if (j0 * ZB > a) literally(a, min(j0 * ZB, b));
  // Between literal stretches the sum lives as (binade E, sign, integer s = |sum| / u in [2^52, 2^53)): a block that is
  // valid for it is a handful of scalar integer operations.  The blocks' descriptions are fetched 64 at a time (a lane
  // each) and handed round with readlane: a load per block would cost the wave one memory round trip per block
  // (measured: 500 ns of the 520 per block).
  bool as_integer = false;
  long long units = 0;
  int E = 0, negative = 0;
  auto to_integer = [&]() {
    as_integer = false;
    if (acc != 0.0 && fabs(acc) <= 1.7976931348623157e308) {
      const int e = ilogb(fabs(acc));
      if (e > -900 && e < 900) {
        E = __builtin_amdgcn_readfirstlane(e);
        negative = __builtin_amdgcn_readfirstlane(acc < 0.0 ? 1 : 0);
        const long long v = (long long)ldexp(fabs(acc), 52 - e);  // exact: an integer in [2^52, 2^53)
        units = ((long long)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        as_integer = true;
      }
    }
  };
  auto to_double = [&]() {
    if (as_integer) {
      const double magnitude = ldexp((double)units, E - 52);  // exact
      acc = negative ? -magnitude : magnitude;
    }
  };
  auto pass = [&](long long mine64, int i) -> long long {  // lane i's value for everybody
    return ((long long)__builtin_amdgcn_readlane((int)(mine64 >> 32), i) << 32) | (unsigned)__builtin_amdgcn_readlane((int)mine64, i);
  };
  to_integer();
  // Groups of 64 x PER blocks, a lane holding PER consecutive ones (the descriptions of the next group that is sure to be
  // walked are requested while the groups before it go by).  Every block is a function parity -> (units added, parity left):
  // a lane composes its own blocks, the wave composes the lanes' functions in lane order (six steps of data-parallel
  // primitives), every lane then knows the sum ENTERING its first block and walks its blocks, testing each one's validity
  // for the sum entering it; up to the first block that fails all are one addition; that block -- and, in one lane-parallel
  // step, the run of blocks behind it that the exactness certificates cover -- is certified exact or added literally, and the
  // rest of the group is composed again from the sum they leave.  (One label over the default raster limit, 49 152 blocks, 16 of
  // them literal: 743 us at one block per lane and nothing requested ahead; round 5: 148 us for one label of zero-mean noise.)
  constexpr int PER = 4;
  // (with a block's description come the three numbers of its exactness certificate: asked for only when a block is turned
  //  away they were two dependent memory round trips per such block -- ~2.5 us, 164 times for one label of noise over the
  //  default raster limit: most of this kernel's 500 us)
  struct Certificate {
    double sum, magnitude;
    int finest;
  };
  auto fetch = [&](unsigned g, BlockInfo* B, Certificate* C) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const unsigned jj = min(g + (unsigned)(PER * lane + k), j1 - 1);
      B[k] = info[jj];
      if (blk_label[jj] != l) B[k].E = INT_MIN;
      C[k].finest = blk_gexp ? blk_gexp[jj] : INT_MIN;
      C[k].sum = blk_sum[jj];
      C[k].magnitude = blk_abs[jj];
    }
  };
  static_assert(64 * PER == ZG, "groups");
  // (the first group ends at the next multiple of ZG, so that the later ones coincide with the groups of ginfo[])
  // The groups' descriptions -- of both kinds -- are held 64 groups at a time, a lane each (one memory round trip per 64 groups =
  // 16 384 blocks), and read off the lanes where they are needed.  A group's 256 BLOCK descriptions are requested ahead only
  // where the held descriptions say that the group will have to be walked (neither kind covers it): a usable group is expected
  // to go in one step, and asking for its blocks anyway would put a memory round trip into every iteration.
  const unsigned last_group = (j1 - 1) / ZG;  // (j1 > j0 wherever this is used)
  unsigned held_from = ~0u;  // group index of lane 0's descriptions
  GroupInfo held;
  GroupExact certified;
  held.E = INT_MIN;
  certified.grain = INT_MIN;
  unsigned long long to_walk = 0;  // the held groups that no description covers
  auto hold = [&](unsigned gi) {  // make sure group gi's descriptions are in a lane; returns the lane
    if (held_from == ~0u || gi < held_from || gi >= held_from + 64u) {
      held_from = gi;
      const unsigned mine_g = min(gi + (unsigned)lane, last_group);
      held = ginfo[mine_g];
      if (gexact) certified = gexact[mine_g];
      to_walk = __ballot(held.E == INT_MIN && certified.grain == INT_MIN);
    }
    return (int)(gi - held_from);
  };
  auto pass_f64 = [&](double mine64, int i) -> double {
    const unsigned long long u = (unsigned long long)__double_as_longlong(mine64);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(u >> 32), i) << 32) |
                                            (unsigned)__builtin_amdgcn_readlane((int)u, i)));
  };
  // a whole group of ginfo[] / gexact[], all of it blocks of this label's?
  auto whole_group = [&](unsigned g) -> bool { return g < j1 && g % ZG == 0 && j1 - g >= (unsigned)ZG; };
  // the first group from block g on that will be walked whatever the sum is by then: a group that is not whole, or a held one
  // that no description covers (j1: none as far as the held descriptions reach)
  auto next_walked = [&](unsigned g) -> unsigned {
    if (g >= j1) return j1;
    if (!whole_group(g)) return g;
    const int at = hold(g / ZG);
    const unsigned long long later = to_walk >> at;
    if (later == 0) return j1;
    const unsigned found = g + (unsigned)(__ffsll((long long)later) - 1) * (unsigned)ZG;
    return found < j1 ? found : j1;
  };
  BlockInfo ahead[PER];
  Certificate ahead_cert[PER];
  unsigned ahead_for = j1;  // the group (by its first block) whose descriptions were requested into `ahead` (j1: none)
  if (j0 < j1) {
    ahead_for = next_walked(j0);
    if (ahead_for < j1) fetch(ahead_for, ahead, ahead_cert);
  }
  unsigned here = 0;
  for (unsigned g = j0; g < j1; g += here) {
    here = min((unsigned)ZG - g % ZG, j1 - g);
    const bool whole = whole_group(g);
    int at = 0;
    if (whole) at = hold(g / ZG);
    if (whole && as_integer && __builtin_amdgcn_readlane(held.E, at) == E && (__builtin_amdgcn_readlane(held.flags, at) & 1) == negative) {
      const int p0 = (int)(units & 1);
      const long long lo = pass(p0 ? held.lo[1] : held.lo[0], at), hi = pass(p0 ? held.hi[1] : held.hi[0], at);
      if (units + lo >= 4503599627370497ll && units + hi <= 9007199254740991ll) {
        units += pass(p0 ? held.D[1] : held.D[0], at);  // (256 blocks, 65 536 cells: one addition)
        ++n_binade_groups;
        continue;
      }
    }
    if (whole && gexact) {
      const int finest = __builtin_amdgcn_readlane(certified.grain, at);
      if (finest != INT_MIN) {
        to_double();
        if (fabs(acc) <= 1.7976931348623157e308) {
          const int held_bit = acc == 0.0 ? INT_MAX : lowbit_exp(acc);
          const int grain = min(finest, held_bit);
          const double lo = pass_f64(certified.lo, at), hi = pass_f64(certified.hi, at);
          const double reach = fmax(fabs(acc + lo), fabs(acc + hi)) * 1.001;  // (the bounds are float64 sums themselves)
          if (grain == INT_MAX || (grain > -1000 && grain < 900 && reach < ldexp(0.99, 53 + grain) && hi - lo < ldexp(0.99, 52 + grain))) {
            acc += pass_f64(certified.total, at);  // 256 blocks without a rounding: one addition
            to_integer();
            ++n_exact_groups;
            continue;
          }
        }
      }
    }
    // This group is walked.  Its blocks' descriptions: requested earlier, or now (a group that was expected to go in one step
    // and did not); then the same is requested for the next group that is sure to be walked.
    BlockInfo B[PER];
    Certificate C[PER];
    if (ahead_for == g) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        B[k] = ahead[k];
        C[k] = ahead_cert[k];
      }
    } else {
      fetch(g, B, C);
    }
    {
      const unsigned coming = next_walked(g + here);  // (may move the window of held descriptions on: nothing below needs them)
      if (coming != ahead_for) {
        if (coming < j1) fetch(coming, ahead, ahead_cert);
        ahead_for = coming;
      }
    }
    ++n_walked;
    // the first block from `from_idx` on that the info kernel expects to be added literally (64 * PER: none in this group)
    auto next_risky = [&](unsigned from_idx) -> unsigned {
      unsigned first_risky = 64u * PER;
#pragma unroll
      for (int k = PER - 1; k >= 0; --k) {
        const unsigned idx = (unsigned)(PER * lane + k);
        if (idx >= from_idx && idx < here && (B[k].flags & 2)) first_risky = idx;
      }
      const unsigned long long any = __ballot(first_risky < 64u * PER);
      if (any == 0) return 64u * PER;
      return (unsigned)__builtin_amdgcn_readlane((int)first_risky, __ffsll((long long)any) - 1);
    };
    unsigned start = 0;
    unsigned wanted = next_risky(0);  // its cells are requested now and used when the composition stops there
    Cells ahead_cells = request(g + min(wanted, here - 1));
    while (start < here) {
      ++n_scans;
      UnitsFn scan;
      scan.S[0] = scan.S[1] = 0;
      scan.q[0] = 0;
      scan.q[1] = 1;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const unsigned idx = (unsigned)(PER * lane + k);
        if (idx >= start && idx < here) {
          UnitsFn f;
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            f.S[p] = B[k].D(p);
            f.q[p] = (int)((p + B[k].D(p)) & 1);
          }
          scan = compose_units(scan, f);
        }
      }
      scan = scan_units(scan);
      UnitsFn excl = shuffle_up_units(scan, 1);
      if (lane == 0) {
        excl.S[0] = excl.S[1] = 0;
        excl.q[0] = 0;
        excl.q[1] = 1;
      }
      const int p0 = (int)(units & 1);
      long long entering = units + (p0 ? excl.S[1] : excl.S[0]);
      int parity = p0 ? excl.q[1] : excl.q[0];
      int failed = -1;           // the lane's first block that is not valid for the sum entering it
      long long failed_entering = 0;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const unsigned idx = (unsigned)(PER * lane + k);
        if (idx >= start && idx < here && failed < 0) {
          const long long lowest = entering + B[k].lo, highest = entering + B[k].hi;
          // every prefix strictly inside the binade (one unit of margin below: the exact value before a rounding may lie
          // half a unit under the rounded one)
          const bool valid = as_integer && B[k].E == E && (B[k].flags & 1) == negative && lowest >= 4503599627370497ll &&
                             highest <= 9007199254740991ll;
          if (!valid) {
            failed = (int)idx;
            failed_entering = entering;
          } else {
            const long long d = B[k].D(parity);
            entering += d;
            parity = (int)((parity + d) & 1);
          }
        }
      }
      const unsigned long long bad = __ballot(failed >= 0);
      if (bad == 0) {
        const long long total = p0 ? scan.S[1] : scan.S[0];
        units += pass(total, 63);
        start = here;
      } else {
        const int from = __ffsll((long long)bad) - 1;  // (lanes hold ascending blocks: the lowest lane has the first failure)
        unsigned k = (unsigned)__builtin_amdgcn_readlane(failed, from);
        units = pass(failed_entering, from);
        to_double();
        // the exactness certificate of zonal_block_sum_kernel: no rounding anywhere in this block for the sum entering it?
        bool exact = false;
        // the block's certificate from the lane that holds it (k = PER * lane + which)
        const int owner = (int)(k / PER), which = (int)(k % PER);
        Certificate mine_c = C[0];
#pragma unroll
        for (int c = 1; c < PER; ++c)
          if (which == c) mine_c = C[c];  // (wave-uniform)
        const int finest = __builtin_amdgcn_readlane(mine_c.finest, owner);
        const double block_sum = pass_f64(mine_c.sum, owner), block_abs = pass_f64(mine_c.magnitude, owner);
        if (blk_gexp) {
          if (finest != INT_MIN && fabs(acc) <= 1.7976931348623157e308) {
            const int held_bit = acc == 0.0 ? INT_MAX : lowbit_exp(acc);
            const int grain = min(finest, held_bit);
            // (0.99: the two magnitudes are float64 sums themselves; the bound has to hold for the exact ones)
            exact = grain == INT_MAX || (grain > -1000 && grain < 900 && fabs(acc) + block_abs < ldexp(0.99, 53 + grain));
          }
        }
        if (exact) {
          acc += block_sum;  // exact: the block in one addition
          ++n_exact;
          // Blocks turned away come in stretches (a sum that wanders about a power of two, a prediction that is off for a
          // while): the blocks that follow are asked for the same certificate straight away, one after the other, before the
          // functions of the rest of the group are composed again -- an addition certified exact is bincount's whichever way it
          // was arrived at.  (Three labels of 4 M cells of zero-mean noise: 2 397 blocks turned away, 2.55 ms; see MEASUREMENTS.)
          if (blk_gexp && k + 1 < here && fabs(acc) <= 1.7976931348623157e308) {
            // Every lane looks at its own blocks from k + 1 on: the partial sums of the run up to a block lie, relative to the sum
            // entering the run, inside [LO, HI] = [min (o_j - a_j), max (o_j + a_j)] over the blocks j up to it (o_j: the tree sums
            // before block j, a_j: its sum of |cells|), and everything is a multiple of 2^G, G = the finest bit of those blocks and
            // of the sum entering the run.  max(|acc + LO|, |acc + HI|) < 2^(53 + G) and HI - LO < 2^(52 + G): no addition up to
            // that block rounds, every range of its blocks sums without rounding (the o_j are what they stand for) and the run
            // goes in ONE addition.  The test can only turn false once along the run: a ballot finds where.
            const unsigned from = k + 1;
            const double inf = __builtin_huge_val();
            double lane_lo = inf, lane_hi = -inf, lane_run = 0.0;
            int lane_fin = INT_MAX;
            double o[PER], lo_c[PER], hi_c[PER];
            int fin_c[PER];
#pragma unroll
            for (int c = 0; c < PER; ++c) {
              const unsigned idx = (unsigned)(PER * lane + c);
              const bool in = idx >= from && idx < here;
              o[c] = lane_run;
              if (in) {
                lane_lo = fmin(lane_lo, lane_run - C[c].magnitude);
                lane_hi = fmax(lane_hi, lane_run + C[c].magnitude);
                lane_fin = min(lane_fin, C[c].finest);  // (INT_MIN: a block without a certificate ends the run)
                lane_run += C[c].sum;
              }
              lo_c[c] = lane_lo;
              hi_c[c] = lane_hi;
              fin_c[c] = lane_fin;
            }
            // (the sum before the lane's blocks from the lanes below alone: `inclusive - lane_run` would carry what a block past the
            //  end of the run -- of any size -- rounded away)
            const double inclusive = lane_scan_sum_f64(lane_run);
            double before = shuffle_up_f64(inclusive, 1);
            if (lane == 0) before = 0.0;
            const double scan_lo = lane_scan_min_f64(before + lane_lo), scan_hi = lane_scan_max_f64(before + lane_hi);
            const int scan_fin = lane_scan_min_i32(lane_fin);
            double ex_lo = shuffle_up_f64(scan_lo, 1), ex_hi = shuffle_up_f64(scan_hi, 1);
            int ex_fin = __shfl_up(scan_fin, 1);
            if (lane == 0) {
              ex_lo = inf;
              ex_hi = -inf;
              ex_fin = INT_MAX;
            }
            const int held_bit = acc == 0.0 ? INT_MAX : lowbit_exp(acc);
            int stop = PER;  // the lane's first block that the test turns away (PER: none)
#pragma unroll
            for (int c = PER - 1; c >= 0; --c) {
              const unsigned idx = (unsigned)(PER * lane + c);
              if (idx >= from && idx < here) {
                const int G = min(held_bit, min(ex_fin, fin_c[c]));
                const double LO = fmin(ex_lo, before + lo_c[c]), HI = fmax(ex_hi, before + hi_c[c]);
                const double reach = fmax(fabs(acc + LO), fabs(acc + HI)) * 1.001;  // (the bounds are float64 sums themselves)
                const bool ok = G == INT_MAX || (G > -1000 && G < 900 && reach < ldexp(0.99, 53 + G) && HI - LO < ldexp(0.99, 52 + G));
                if (!ok) stop = c;
              }
            }
            const unsigned long long stopped = __ballot(stop < PER);
            unsigned taken;   // blocks of the run that go in the one addition
            double run_total;
            if (stopped == 0) {
              taken = here - from;
              run_total = pass_f64(inclusive, 63);
            } else {
              const int at_lane = __ffsll((long long)stopped) - 1;
              const int at_c = __builtin_amdgcn_readlane(stop, at_lane);
              double upto = before + o[0];
#pragma unroll
              for (int c = 1; c < PER; ++c)
                if (at_c == c) upto = before + o[c];  // (wave-uniform)
              taken = (unsigned)(PER * at_lane + at_c) - from;
              run_total = pass_f64(upto, at_lane);
            }
            if (taken) {
              acc += run_total;
              n_exact += taken;
              k += taken;
            }
            // Where the run ends the blocks that follow are still asked one by one, a few of them: a block on its own needs no
            // bound on the run's range (|sum entering it| + its sum of |cells| < 2^(53 + grain) is all), which is what still
            // holds when the running sum is hardly larger than a block's cells and carries a bit at its last place -- there the
            // run test gives out after a block or two and every composition of the rest of the group costs ~2.5 us.  (Three
            // labels of 4 M cells of zero-mean noise: 1 483 compositions with the run test alone, 3.9 ms.)
            // (after a long run the rest of the group is better composed: a composition takes up to 256 valid blocks at the price of three)
            for (int more = 0; taken < 8 && more < 16 && k + 1 < here; ++more) {
              const unsigned nk = k + 1;
              const int n_owner = (int)(nk / PER), n_which = (int)(nk % PER);
              Certificate next_c = C[0];
#pragma unroll
              for (int c = 1; c < PER; ++c)
                if (n_which == c) next_c = C[c];  // (wave-uniform)
              const int n_finest = __builtin_amdgcn_readlane(next_c.finest, n_owner);
              if (n_finest == INT_MIN || !(fabs(acc) <= 1.7976931348623157e308)) break;
              const int n_grain = min(n_finest, acc == 0.0 ? INT_MAX : lowbit_exp(acc));
              const double n_abs = pass_f64(next_c.magnitude, n_owner);
              if (!(n_grain == INT_MAX || (n_grain > -1000 && n_grain < 900 && fabs(acc) + n_abs < ldexp(0.99, 53 + n_grain)))) break;
              acc += pass_f64(next_c.sum, n_owner);
              ++n_exact;
              k = nk;
            }
          }
          if (wanted < k + 1 || k == wanted) {
            wanted = next_risky(k + 1);
            ahead_cells = request(g + min(wanted, here - 1));
          }
        } else if (k == wanted) {
          // the cells asked for earlier; the next expected block's are requested before this block's chain starts.
          // (Tried and dropped: descriptions per quarter of such a block, so that only the quarter in which the sum crosses
          // the power of two is added literally -- around a power of two the sum of zero-mean data wanders back and forth,
          // 2.04 of the 4 quarters stay literal, and the extra conversions cost more than the shorter chains save: 3.9 -> 4.6 ms.)
          const Cells now = ahead_cells;
          wanted = next_risky(k + 1);
          ahead_cells = request(g + min(wanted, here - 1));
#pragma unroll
          for (int c = 0; c < 4; ++c) mine[4 * lane + c] = (double)Ordered<T>::decode(now.k[c]);
          chain(ZB);
          ++n_ahead;
        } else {
          literally((g + k) * ZB, (g + k) * ZB + ZB);
          if (wanted < k + 1) {  // (the expected block was passed without stopping: look further)
            wanted = next_risky(k + 1);
            ahead_cells = request(g + min(wanted, here - 1));
          }
        }
        to_integer();
        n_replayed += exact ? 0u : 1u;
        start = k + 1;
      }
    }
  }
  to_double();
  if (j1 >= j0 && j1 * ZB < b) literally(max(j1 * ZB, a), b);
  if (lane == 0) {
    const unsigned n = b - a;
    counts[l] = n;
    out[l] = statistic == Z_MEAN ? (float)(acc / (double)n) : (float)acc;
    if (replayed) {
      atomicAdd(replayed, n_replayed);
      atomicAdd(replayed + 1, n_ahead);
      atomicAdd(replayed + 2, n_exact);
      atomicAdd(replayed + 3, n_exact_groups);
      atomicAdd(replayed + 4, n_scans);
      atomicAdd(replayed + 5, n_walked);
      atomicAdd(replayed + 6, n_binade_groups);
    }
  }
}

// Labels of a few cells (a raster of small polygons has hundreds of thousands): a lane per label adds its cells one after
// the other straight from memory -- a wave per label would spend its life waiting for 40 cells.
template <typename T>
__global__ __launch_bounds__(256) void zonal_short_sum_kernel(const typename Ordered<T>::Key* __restrict__ payload,
                                                              const unsigned* __restrict__ first, int n_labels, int statistic,
                                                              const unsigned char* __restrict__ only, float* __restrict__ out,
                                                              long long* __restrict__ counts) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n_labels || (only && !only[l])) return;
  const unsigned a = first[l], b = first[l + 1], n = b - a;
  if (n > ZS_SHORT) return;  // (zonal_sum_kernel's)
  double acc = 0.0;
  for (unsigned i = a; i < b; ++i) acc += (double)Ordered<T>::decode(payload[i]);
  counts[l] = n;
  out[l] = n == 0 ? std::numeric_limits<float>::quiet_NaN() : (statistic == Z_MEAN ? (float)(acc / (double)n) : (float)acc);
}

// count (ndimage.sum of ones) and the statistics that pick from the cells sorted by (label, value)
template <typename T>
__global__ __launch_bounds__(256) void zonal_pick_kernel(const typename Ordered<T>::Key* __restrict__ payload,
                                                         const unsigned* __restrict__ first, int n_labels, int statistic,
                                                         double qval, float* __restrict__ out, long long* __restrict__ counts) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n_labels) return;
  const unsigned a = first[l], b = first[l + 1], n = b - a;
  counts[l] = n;
  if (n == 0) {
    out[l] = std::numeric_limits<float>::quiet_NaN();
    return;
  }
  auto cell = [&](unsigned i) -> T { return Ordered<T>::decode(payload[i]); };
  float result;
  if (statistic == Z_COUNT) {
    result = (float)(double)n;
  } else if (statistic == Z_MIN) {
    result = (float)cell(a);
  } else if (statistic == Z_MAX) {
    result = (float)cell(b - 1);  // (NaN sorts last: a label with a NaN cell has a NaN maximum, like SciPy's)
  } else if (statistic == Z_MEDIAN) {
    // scipy.ndimage._measurements._select: the two middle cells (the same cell for an odd count) averaged --
    // integers and booleans in float64, floats in their own precision
    const unsigned step = (n - 1) / 2, lo = a + step, hi = (b - 1) - step;
    const T x = cell(lo), y = cell(hi);
    if constexpr (std::is_same<T, float>::value) result = (x + y) / 2.0f;
    else if constexpr (std::is_same<T, double>::value) result = (float)((x + y) / 2.0);
    else result = (float)(((double)x + (double)y) / 2.0);
  } else {
    // measurements.percentile (measurements.py:126-134): np.percentile's linear interpolation between the
    // floor and the ceiling of the fractional index; the difference is taken in the raster's dtype
    const double frac = (double)(n - 1) * (qval / 100.0);
    const double below = floor(frac), above = ceil(frac);
    const T x = cell(a + (unsigned)below), y = cell(a + (unsigned)above);
    const T diff = (T)(y - x);
    result = (float)((double)x + (frac - below) * (double)diff);
  }
  out[l] = result;
}

// The walk of a wave over its stretch of cells [begin, end), 64 at a time: `body(value, label, inside, index)`.  The
// loads of the full steps are unconditional and DEPTH steps ahead (a select or a branch around a load makes the
// compiler drain every outstanding load at each step); only the last, partial step tests `inside`.
template <typename T, typename Body>
__device__ __forceinline__ void walk_cells(const T* __restrict__ values, const int* __restrict__ labels, size_t begin, size_t end,
                                           int lane, Body body) {
  if (begin >= end) return;
  constexpr int DEPTH = 4;
  const size_t full = (end - begin) / 64, last = end - 1;
  T ahead_v[DEPTH];
  int ahead_l[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const size_t at = min(begin + (size_t)d * 64 + lane, last);
    ahead_v[d] = values[at];
    ahead_l[d] = labels[at];
  }
  for (size_t step = 0; step < full; ++step) {
    const T v = ahead_v[0];
    const int label = ahead_l[0];
#pragma unroll
    for (int d = 0; d + 1 < DEPTH; ++d) {
      ahead_v[d] = ahead_v[d + 1];
      ahead_l[d] = ahead_l[d + 1];
    }
    const size_t at = min(begin + (step + DEPTH) * 64 + lane, last);
    ahead_v[DEPTH - 1] = values[at];
    ahead_l[DEPTH - 1] = labels[at];
    body(v, label, true, begin + step * 64 + lane);
  }
  if (begin + full * 64 < end) {  // (whole waves enter the partial step too: cross-lane operations in the body)
    const size_t i = begin + full * 64 + lane;
    const bool inside = i < end;
    body(inside ? values[i] : T(0), inside ? labels[i] : -1, inside, i);
  }
}

// A step whose 64 cells carry several labels: the lanes of every run of equal labels combine first (a segmented
// inclusive scan over the lanes, log2(64) exchange steps per quantity) and only the last lane of a run touches the
// label's counters -- a raster of small polygons is all such steps, and one atomic per cell is an order of magnitude
// slower than one per run, more when neighbouring rows hammer the same few counters.
struct LabelRuns {
  int distance;  // lanes between this lane and the first lane of its run
  bool last;     // the last lane of its run
};
__device__ __forceinline__ LabelRuns label_runs(int label, int lane) {
  const int before = __shfl_up(label, 1), after = __shfl_down(label, 1);
  int start = (lane == 0 || label != before) ? lane : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int other = __shfl_up(start, d);
    if (lane >= d) start = max(start, other);
  }
  return LabelRuns{lane - start, lane == 63 || label != after};
}
template <typename V>
__device__ __forceinline__ V lane_up(V v, int d) {
  if constexpr (sizeof(V) == 8) {
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    const unsigned lo = __shfl_up((unsigned)u, d), hi = __shfl_up((unsigned)(u >> 32), d);
    u = ((unsigned long long)hi << 32) | lo;
    V r;
    __builtin_memcpy(&r, &u, 8);
    return r;
  } else {
    return __shfl_up(v, d);
  }
}
template <typename V, typename Op>
__device__ __forceinline__ V run_total(V v, const LabelRuns& runs, Op op) {  // at the last lane of a run: the run's total
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const V other = lane_up(v, d);
    if (runs.distance >= d) v = op(v, other);
  }
  return v;
}

// count / min / max need no order at all: one pass of atomics on the per-label counters and on the ordered
// keys (the smallest key is the minimum with NaN last -- NaN only if every cell is; the largest key is the
// maximum -- NaN if any cell is: exactly the first / last cell of the sorted segment).  Label rasters are
// made of runs: a wave walks a contiguous stretch of cells and, as long as the 64 cells of a step carry one label
// (the usual case), only accumulates per lane; when the label changes it reduces across its lanes once and issues
// one atomic per quantity.  Steps that straddle a label boundary fall back to one atomic per cell.
template <typename T, int STAT>
__global__ __launch_bounds__(256) void zonal_direct_kernel(const T* __restrict__ values, const int* __restrict__ labels, size_t cells,
                                                           size_t cells_per_wave, int n_labels, int has_ndv, T ndv,
                                                           const float* __restrict__ thresholds,
                                                           unsigned long long* __restrict__ counts,
                                                           typename Ordered<T>::Key* __restrict__ extreme,
                                                           unsigned char* __restrict__ present) {
  typedef typename Ordered<T>::Key Key;
  constexpr Key NEUTRAL = STAT == Z_MIN ? ~(Key)0 : (Key)0;
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const size_t begin = wave * cells_per_wave, end = min(begin + cells_per_wave, cells);  // (cells_per_wave: a multiple of 64)
  // the run the wave is in: its label (wave-uniform), and per lane the cells counted / the extreme key so far
  int run_label = -1;
  int marked = INT_MIN;  // (mark_present)
  unsigned run_count = 0;
  Key run_key = NEUTRAL;
  auto better = [](Key a, Key b) -> Key { return STAT == Z_MIN ? (b < a ? b : a) : (b > a ? b : a); };
  auto wave_best = [&](Key key) -> Key {  // the wave's extreme key, in every lane
    if constexpr (sizeof(Key) == 4) {
      return STAT == Z_MIN ? wave_umin((unsigned)key) : wave_umax((unsigned)key);
    } else {
#pragma unroll
      for (int offset = 32; offset >= 1; offset >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)key, offset), hi = __shfl_xor((unsigned)(key >> 32), offset);
        key = better(key, ((Key)hi << 32) | lo);
      }
      return key;
    }
  };
  // What the wave has gathered for the labels it met, a lane per label (round 5): a finished run, and every label of a step that
  // straddles labels, is merged into its lane here and goes to the label's counters ONCE, when the wave is through (or the 64
  // lanes are taken: all are written out and start afresh).  Until then a run cost one atomic per quantity -- and a label raster
  // that is not made of runs one per CELL: 12.6 M cells under 2 labels at random 44 ms, under 40 labels 20 ms.
  int slot_label = -1, n_slots = 0;
  unsigned long long slot_count = 0;
  Key slot_key = NEUTRAL;
  auto write_slots = [&]() {
    if (lane < n_slots && slot_count != 0) {
      atomicAdd(&counts[slot_label], slot_count);
      if constexpr (STAT == Z_MIN) atomicMin(&extreme[slot_label], slot_key);
      if constexpr (STAT == Z_MAX) atomicMax(&extreme[slot_label], slot_key);
    }
    slot_label = -1;
    slot_count = 0;
    slot_key = NEUTRAL;
    n_slots = 0;
  };
  auto merge = [&](int label, unsigned long long n, Key key) {  // (wave-uniform arguments)
    if (n == 0) return;
    const unsigned long long hit = __ballot(slot_label == label);
    int slot;
    if (hit) {
      slot = __ffsll((long long)hit) - 1;
    } else {
      if (n_slots == 64) write_slots();
      slot = n_slots++;
    }
    if (lane == slot) {
      slot_label = label;
      slot_count += n;
      slot_key = better(slot_key, key);
    }
  };
  auto flush = [&]() {  // (wave-uniform call)
    if (run_label < 0) return;
    unsigned long long n = wave_sum_u32(run_count);  // (fewer than 2^32 cells)
    Key key = run_key;
    if constexpr (STAT != Z_COUNT) key = wave_best(key);
    merge(run_label, n, key);
    run_label = -1;
    run_count = 0;
    run_key = NEUTRAL;
  };
  walk_cells<T>(values, labels, begin, end, lane, [&](T v, int label, bool inside, size_t i) {
    (void)inside;
    (void)i;
    const bool labelled = (unsigned)label < (unsigned)n_labels;
    mark_present(present, label, labelled, &marked);
    bool active = labelled && !(has_ndv && v == ndv);
    if (thresholds && active) {
      const float t = thresholds[label];
      active = !(t != t) && (double)v >= (double)t;
    }
    const unsigned long long mask = __ballot(active);
    if (mask == 0) return;  // (wave-uniform)
    const int leader_label = __builtin_amdgcn_readlane(label, __ffsll((long long)mask) - 1);  // (a uniform lane index)
    const Key key = Ordered<T>::encode(v);
    if (__all(!active || label == leader_label)) {
      // one label among the active cells of this step: it joins the wave's run, no traffic between lanes
      if (leader_label != run_label) {
        flush();
        run_label = leader_label;
      }
      if (active) {
        run_count += 1;
        run_key = better(run_key, key);
      }
    } else {  // (a step across labels: its distinct labels one after the other, a ballot each)
      unsigned long long open = mask;
      while (open) {
        const int lead = __builtin_amdgcn_readlane(label, __ffsll((long long)open) - 1);
        const bool mine = active && label == lead;
        const unsigned long long same = __ballot(mine);
        Key best = NEUTRAL;
        if constexpr (STAT != Z_COUNT) best = wave_best(mine ? key : NEUTRAL);
        merge(lead, (unsigned long long)__popcll(same), best);
        open &= ~same;
      }
    }
  });
  write_slots();
  // The run a wave ENDS in meets the other three waves' in LDS, and waves that end in the same label issue their atomics
  // together: under a label that covers the raster every wave ends on the same counters, and atomics on one address
  // serialise (round 3: 4 x fewer of them; count under one label 111 -> see MEASUREMENTS K14).
  __shared__ int end_label[4];
  __shared__ unsigned long long end_count[4];
  __shared__ Key end_key[4];
  {
    unsigned long long n = wave_sum_u32(run_count);  // (fewer than 2^32 cells)
    Key key = run_key;
    if constexpr (STAT != Z_COUNT) key = wave_best(key);
    if (lane == 0) {
      const int w = threadIdx.x >> 6;
      end_label[w] = (run_label >= 0 && n != 0) ? run_label : -1;
      end_count[w] = n;
      end_key[w] = key;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 0; w < 4; ++w) {
      const int label = end_label[w];
      if (label < 0) continue;
      unsigned long long n = end_count[w];
      Key key = end_key[w];
      for (int v = w + 1; v < 4; ++v) {
        if (end_label[v] == label) {
          n += end_count[v];
          key = better(key, end_key[v]);
          end_label[v] = -1;
        }
      }
      atomicAdd(&counts[label], n);
      if constexpr (STAT == Z_MIN) atomicMin(&extreme[label], key);
      if constexpr (STAT == Z_MAX) atomicMax(&extreme[label], key);
    }
  }
}

// count / min / max through tables PRIVATE to a workgroup (round 6).  zonal_direct_kernel above merges a step's distinct labels
// one ballot at a time and keeps a wave's labels in 64 lanes: right for rasters made of runs, an order of magnitude off for label
// rasters without runs -- 1000 labels at random per cell: 64 ballots per step and a write-out of all 64 lanes every step, 1.6 ms
// (count) and 2.8 ms (max) for 12.6 M cells against 60 us for blocky labels.  Here every workgroup keeps a counter (and an extreme
// key) per label in LDS: a step whose active cells carry one label joins the wave's run in registers as before (one LDS atomic
// per quantity when the run ends), any other step is one LDS atomic per active lane -- no loop over its labels --, and the
// workgroup's table goes to the global counters once, entries that were touched only.  Up to ZT_LDS_BYTES of table per workgroup.
constexpr size_t ZT_LDS_BYTES = 64 * 1024;
constexpr int ZT_THREADS = 1024;
template <typename T, int STAT>
__global__ __launch_bounds__(ZT_THREADS) void zonal_tables_kernel(const T* __restrict__ values, const int* __restrict__ labels, size_t cells,
                                                                  size_t cells_per_wave, int n_labels, int has_ndv, T ndv,
                                                                  const float* __restrict__ thresholds,
                                                                  unsigned long long* __restrict__ counts,
                                                                  typename Ordered<T>::Key* __restrict__ extreme,
                                                                  unsigned char* __restrict__ present) {
  typedef typename Ordered<T>::Key Key;
  constexpr Key NEUTRAL = STAT == Z_MIN ? ~(Key)0 : (Key)0;
  extern __shared__ unsigned long long zt_lds[];  // [n_labels] keys (min / max only), then [n_labels] counters
  Key* const best_of = (Key*)zt_lds;
  unsigned* const count_of = (unsigned*)((unsigned char*)zt_lds + (STAT == Z_COUNT ? 0 : sizeof(Key) * (size_t)((n_labels + 1) & ~1)));
  for (int l = threadIdx.x; l < n_labels; l += ZT_THREADS) {
    count_of[l] = 0u;
    if constexpr (STAT != Z_COUNT) best_of[l] = NEUTRAL;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * ZT_THREADS + threadIdx.x) >> 6;
  const size_t begin = wave * cells_per_wave, end = min(begin + cells_per_wave, cells);  // (cells_per_wave: a multiple of 64)
  auto better = [](Key a, Key b) -> Key { return STAT == Z_MIN ? (b < a ? b : a) : (b > a ? b : a); };
  auto wave_best = [&](Key key) -> Key {
    if constexpr (sizeof(Key) == 4) {
      return STAT == Z_MIN ? wave_umin((unsigned)key) : wave_umax((unsigned)key);
    } else {
#pragma unroll
      for (int offset = 32; offset >= 1; offset >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)key, offset), hi = __shfl_xor((unsigned)(key >> 32), offset);
        key = better(key, ((Key)hi << 32) | lo);
      }
      return key;
    }
  };
  int run_label = -1, marked = INT_MIN;
  unsigned run_count = 0;
  Key run_key = NEUTRAL;
  auto flush = [&]() {  // (wave-uniform call)
    if (run_label < 0) return;
    const unsigned n = wave_sum_u32(run_count);
    Key key = run_key;
    if constexpr (STAT != Z_COUNT) key = wave_best(key);
    if (lane == 0 && n != 0) {
      atomicAdd(&count_of[run_label], n);
      if constexpr (STAT == Z_MIN) atomicMin(&best_of[run_label], key);
      if constexpr (STAT == Z_MAX) atomicMax(&best_of[run_label], key);
    }
    run_label = -1;
    run_count = 0;
    run_key = NEUTRAL;
  };
  walk_cells<T>(values, labels, begin, end, lane, [&](T v, int label, bool inside, size_t i) {
    (void)inside;
    (void)i;
    const bool labelled = (unsigned)label < (unsigned)n_labels;
    mark_present(present, label, labelled, &marked);
    bool active = labelled && !(has_ndv && v == ndv);
    if (thresholds && active) {
      const float t = thresholds[label];
      active = !(t != t) && (double)v >= (double)t;
    }
    const unsigned long long mask = __ballot(active);
    if (mask == 0) return;  // (wave-uniform)
    const int leader_label = __builtin_amdgcn_readlane(label, __ffsll((long long)mask) - 1);
    const Key key = Ordered<T>::encode(v);
    if (__all(!active || label == leader_label)) {
      if (leader_label != run_label) {
        flush();
        run_label = leader_label;
      }
      if (active) {
        run_count += 1;
        run_key = better(run_key, key);
      }
    } else if (active) {  // a step across labels: every active lane on its own counter of the workgroup's table
      atomicAdd(&count_of[label], 1u);
      if constexpr (STAT == Z_MIN) atomicMin(&best_of[label], key);
      if constexpr (STAT == Z_MAX) atomicMax(&best_of[label], key);
    }
  });
  flush();
  __syncthreads();
  for (int l = threadIdx.x; l < n_labels; l += ZT_THREADS) {
    const unsigned n = count_of[l];
    if (n == 0) continue;
    atomicAdd(&counts[l], (unsigned long long)n);
    if constexpr (STAT == Z_MIN) atomicMin(&extreme[l], best_of[l]);
    if constexpr (STAT == Z_MAX) atomicMax(&extreme[l], best_of[l]);
  }
}

// tally = 0, extreme = the neutral key, present = 0 (first frame) in ONE launch: three fills of a few KiB cost three launches,
// a fifth of what a count over 12.6 M cells takes
template <typename K>
__global__ __launch_bounds__(256) void zonal_direct_init_kernel(unsigned long long* __restrict__ tally, K* __restrict__ extreme, K neutral,
                                                                unsigned char* __restrict__ present, int n_labels) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n_labels) return;
  tally[l] = 0ull;
  extreme[l] = neutral;
  if (present) present[l] = 0;
}

template <typename T, int STAT>
__global__ __launch_bounds__(256) void zonal_direct_result_kernel(const unsigned long long* __restrict__ tally,
                                                                  const typename Ordered<T>::Key* __restrict__ extreme, int n_labels,
                                                                  float* __restrict__ out, long long* __restrict__ counts) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n_labels) return;
  const unsigned long long n = tally[l];
  counts[l] = (long long)n;
  float result = std::numeric_limits<float>::quiet_NaN();
  if (n != 0) {
    if constexpr (STAT == Z_COUNT) result = (float)(double)n;
    else result = (float)Ordered<T>::decode(extreme[l]);
  }
  out[l] = result;
}

// sum / mean without the sort, when the order cannot matter.  Every partial sum of a label's cells -- in ANY
// order -- is a multiple of 2^lo (lo = the lowest set bit among the cells) and smaller than count * 2^(hi + 1)
// (hi = the highest exponent): if hi + 1 + ceil(log2(count)) - lo <= 53 all of them are exactly representable in
// float64, bincount's sequential sum rounds nowhere, and a parallel sum gives the same bits.  Integer rasters and
// data on a dyadic grid qualify (class maps, masks, counts, centimetres as integers); data that spans many octaves
// does not and keeps the sorted, sequential path.  PHASE 0 gathers count / hi / lo per label, PHASE 1 adds.
struct ExactRange {
  unsigned long long count;
  int hi, lo;       // top exponent / exponent of the lowest set bit over the non-zero cells
  unsigned bad;     // a NaN or an infinity among the cells
};
struct ExactVerdict {  // over all labels: how many are not exact, how many active cells they hold; the longest label's cells
  unsigned labels, longest;
  unsigned long long cells;
};
template <typename T, int PHASE>
__global__ __launch_bounds__(256) void zonal_exact_kernel(const T* __restrict__ values, const int* __restrict__ labels, size_t cells,
                                                          size_t cells_per_wave, int n_labels, int has_ndv, T ndv,
                                                          const float* __restrict__ thresholds, ExactRange* __restrict__ range,
                                                          double* __restrict__ sums, unsigned char* __restrict__ present,
                                                          unsigned* __restrict__ keys, typename Ordered<T>::Key* __restrict__ payload,
                                                          unsigned* __restrict__ tile_counts, double* __restrict__ mags = nullptr) {
  // (PHASE 0 also writes the sort keys of zonal_keys_kernel: if the order turns out to matter the sort can start at once --
  // and, with tile_counts, what the label partition's counting pass would find: the cells of every key per tile of TILE
  // cells, one atomic per run and tile)
  constexpr size_t TILE = ZP_TILE;
  const int bins = n_labels + 1;
  size_t run_tile = 0;
  unsigned idle = 0;    // cells of the tile that carry the key n_labels (wave-uniform)
  unsigned tile_n = 0;  // per lane: cells of the run inside the tile
  int marked = INT_MIN;  // (mark_present)
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const size_t begin = wave * cells_per_wave, end = min(begin + cells_per_wave, cells);
  int run_label = -1;  // the run the wave is in (wave-uniform); per lane what it has gathered for it
  unsigned run_count = 0, run_bad = 0;
  int run_hi = INT_MIN, run_lo = INT_MAX;
  double run_sum = 0.0, run_mag = 0.0;  // (PHASE 2 = PHASE 1 + the sums of the magnitudes: the float32 certificate of the caller;
                                         //  PHASE 3 = PHASE 2 + the labels' counts and `present`: the one pass of a raster of small labels)
  auto tile_flush = [&]() {  // (wave-uniform call; a run that crosses a tile border keeps gathering for its label's range)
    if (run_label < 0) return;
    const unsigned n = wave_sum_u32(tile_n);
    if (lane == 0 && n != 0) atomicAdd(&tile_counts[run_tile * bins + run_label], n);
    tile_n = 0;
  };
  auto flush = [&]() {  // (wave-uniform call)
    if (run_label < 0) return;
    if constexpr (PHASE == 0) {
      if (tile_counts) tile_flush();
      const unsigned n = wave_sum_u32(run_count), bad = wave_or_u32(run_bad);
      const int hi = wave_max(run_hi), lo = wave_min(run_lo);
      if (lane == 0 && n != 0) {
        ExactRange* slot = range + run_label;
        atomicAdd(&slot->count, (unsigned long long)n);
        atomicMax(&slot->hi, hi);
        atomicMin(&slot->lo, lo);
        if (bad) atomicOr(&slot->bad, 1u);
      }
    } else {
      const double total = wave_sum_exact(run_sum);  // (exact: see above)
      if (lane == 0) atomicAdd(&sums[run_label], total);
      if constexpr (PHASE >= 2) {
        const double mag = wave_sum_exact(run_mag);
        if (lane == 0) atomicAdd(&mags[run_label], mag);
        run_mag = 0.0;
      }
      if constexpr (PHASE == 3) {  // (bits 40 and up of the count: how many additions the label's sum has taken -- its error bound)
        const unsigned n = wave_sum_u32(run_count);
        if (lane == 0 && n != 0) atomicAdd(&range[run_label].count, (1ull << 40) | (unsigned long long)n);
      }
    }
    run_label = -1;
    run_count = run_bad = 0;
    run_hi = INT_MIN;
    run_lo = INT_MAX;
    run_sum = 0.0;
  };
  walk_cells<T>(values, labels, begin, end, lane, [&](T v, int label, bool inside, size_t i) {
    const bool labelled = (unsigned)label < (unsigned)n_labels;
    if constexpr (PHASE == 0 || PHASE == 3) mark_present(present, label, labelled, &marked);
    bool active = labelled && !(has_ndv && v == ndv);
    if (thresholds && active) {
      const float t = thresholds[label];
      active = !(t != t) && (double)v >= (double)t;
    }
    if constexpr (PHASE == 0) {
      if (keys && inside) {
        keys[i] = active ? (unsigned)label : (unsigned)n_labels;
        payload[i] = Ordered<T>::encode(v);
      }
    }
    const unsigned long long mask = __ballot(active);
    if constexpr (PHASE == 0) {
      if (tile_counts) {
        const size_t tile = (i - lane) / TILE;  // (a step never straddles two tiles: stretches start on multiples of 64)
        if (tile != run_tile) {
          tile_flush();
          if (lane == 0 && idle) atomicAdd(&tile_counts[run_tile * bins + n_labels], idle);
          idle = 0;
          run_tile = tile;
        }
        idle += (unsigned)__popcll(__ballot(inside && !active));
      }
    }
    if (mask == 0) return;  // (wave-uniform)
    const int leader_label = __builtin_amdgcn_readlane(label, __ffsll((long long)mask) - 1);  // (a uniform lane index)
    const double x = (double)v;  // what bincount adds
    if (__all(!active || label == leader_label)) {
      if (leader_label != run_label) {
        flush();
        run_label = leader_label;
      }
      if (active) {
        if constexpr (PHASE == 0) {
          run_count += 1;
          tile_n += 1;
          if (!(fabs(x) <= 1.7976931348623157e308)) {
            run_bad = 1;
          } else if (x != 0.0) {
            int top, low;
            cell_span<T>(v, &top, &low);
            run_hi = max(run_hi, top);
            run_lo = min(run_lo, low);
          }
        } else {
          run_sum += x;
          if constexpr (PHASE >= 2) run_mag += fabs(x);
          if constexpr (PHASE == 3) run_count += 1;
        }
      }
    } else {  // (a step across label boundaries: run by run)
      const LabelRuns runs = label_runs(labelled ? label : -1, lane);
      if constexpr (PHASE == 0) {
        int top = INT_MIN, low = INT_MAX;
        unsigned bad = 0;
        if (active) {
          if (!(fabs(x) <= 1.7976931348623157e308)) bad = 1;
          else if (x != 0.0) cell_span<T>(v, &top, &low);
        }
        const unsigned n = run_total(active ? 1u : 0u, runs, [](unsigned a, unsigned b) { return a + b; });
        top = run_total(top, runs, [](int a, int b) { return max(a, b); });
        low = run_total(low, runs, [](int a, int b) { return min(a, b); });
        bad = run_total(bad, runs, [](unsigned a, unsigned b) { return a | b; });
        if (runs.last && labelled && n != 0) {
          ExactRange* slot = range + label;
          atomicAdd(&slot->count, (unsigned long long)n);
          if (top != INT_MIN) {
            atomicMax(&slot->hi, top);
            atomicMin(&slot->lo, low);
          }
          if (bad) atomicOr(&slot->bad, 1u);
          if (tile_counts) atomicAdd(&tile_counts[run_tile * bins + label], n);
        }
      } else {
        const double total = run_total(active ? x : 0.0, runs, [](double a, double b) { return a + b; });  // (exact: see above)
        if (runs.last && labelled && total != 0.0) atomicAdd(&sums[label], total);
        if constexpr (PHASE >= 2) {
          const double mag = run_total(active ? fabs(x) : 0.0, runs, [](double a, double b) { return a + b; });
          if (runs.last && labelled && mag != 0.0) atomicAdd(&mags[label], mag);
        }
        if constexpr (PHASE == 3) {
          const unsigned n = run_total(active ? 1u : 0u, runs, [](unsigned a, unsigned b) { return a + b; });
          if (runs.last && labelled && n != 0) atomicAdd(&range[label].count, (1ull << 40) | (unsigned long long)n);
        }
      }
    }
  });
  // the run a wave ENDS in: combined across the workgroup before it goes to the label's counters (see zonal_direct_kernel)
  __shared__ int end_label[4];
  __shared__ unsigned long long end_count[4];
  __shared__ int end_hi[4], end_lo[4];
  __shared__ unsigned end_bad[4];
  __shared__ double end_sum[4], end_mag[4];
  {
    const int w = threadIdx.x >> 6;
    if constexpr (PHASE == 0) {
      if (tile_counts && run_label >= 0) tile_flush();
      const unsigned n = wave_sum_u32(run_count), bad = wave_or_u32(run_bad);
      const int hi = wave_max(run_hi), lo = wave_min(run_lo);
      if (lane == 0) {
        end_label[w] = (run_label >= 0 && n != 0) ? run_label : -1;
        end_count[w] = n;
        end_hi[w] = hi;
        end_lo[w] = lo;
        end_bad[w] = bad;
      }
    } else {
      const double total = wave_sum_exact(run_sum);  // (exact: see above)
      double mag = 0.0;
      if constexpr (PHASE >= 2) mag = wave_sum_exact(run_mag);
      unsigned n = 0;
      if constexpr (PHASE == 3) n = wave_sum_u32(run_count);
      if (lane == 0) {
        end_label[w] = run_label;
        end_sum[w] = total;
        end_mag[w] = mag;
        end_count[w] = n;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 0; w < 4; ++w) {
      const int label = end_label[w];
      if (label < 0) continue;
      if constexpr (PHASE == 0) {
        unsigned long long n = end_count[w];
        int hi = end_hi[w], lo = end_lo[w];
        unsigned bad = end_bad[w];
        for (int v = w + 1; v < 4; ++v) {
          if (end_label[v] == label) {
            n += end_count[v];
            hi = max(hi, end_hi[v]);
            lo = min(lo, end_lo[v]);
            bad |= end_bad[v];
            end_label[v] = -1;
          }
        }
        ExactRange* slot = range + label;
        atomicAdd(&slot->count, n);
        atomicMax(&slot->hi, hi);
        atomicMin(&slot->lo, lo);
        if (bad) atomicOr(&slot->bad, 1u);
      } else {
        double total = end_sum[w], mag = end_mag[w];
        unsigned long long n = end_count[w];
        for (int v = w + 1; v < 4; ++v) {
          if (end_label[v] == label) {
            total += end_sum[v];  // (exact like every addition of this pass)
            mag += end_mag[v];
            n += end_count[v];
            end_label[v] = -1;
          }
        }
        atomicAdd(&sums[label], total);
        if constexpr (PHASE >= 2) atomicAdd(&mags[label], mag);
        if constexpr (PHASE == 3) {
          if (n != 0) atomicAdd(&range[label].count, (1ull << 40) | n);
        }
      }
    }
  }
  if constexpr (PHASE == 0) {
    if (tile_counts && lane == 0 && idle) atomicAdd(&tile_counts[run_tile * bins + n_labels], idle);
  }
}

// The two passes of zonal_exact_kernel for label rasters WITHOUT runs (round 6): the ranges (PHASE 0) and the sums (PHASE 1) of
// a workgroup's cells gather in LDS tables -- a step of one label in registers as above, any other step one LDS atomic per
// active lane and quantity -- and go to the labels' global records once per workgroup.  zonal_exact_kernel pays a global atomic
// per run, which for labels scattered cell by cell is one per CELL: such rasters used to skip the certificates altogether and
// take the partition and the block pipeline (int32 cells under 1000 labels at random: 560 us; here ~150).  The sums are exact
// whenever they are used (the verdict of PHASE 0 says so), so the order the LDS atomics arrive in is immaterial.
constexpr int ZX_THREADS = 1024;
template <typename T, int PHASE>
__global__ __launch_bounds__(ZX_THREADS) void zonal_tables_exact_kernel(const T* __restrict__ values, const int* __restrict__ labels, size_t cells,
                                                                        size_t cells_per_wave, int n_labels, int has_ndv, T ndv,
                                                                        const float* __restrict__ thresholds, ExactRange* __restrict__ range,
                                                                        double* __restrict__ sums, unsigned char* __restrict__ present) {
  extern __shared__ unsigned long long zx_lds[];
  // PHASE 0: [n] counts, [n] hi, [n] lo, [n] bad (4 bytes each); PHASE 1: [n] sums (8 bytes each)
  unsigned* const count_of = (unsigned*)zx_lds;
  int* const hi_of = (int*)(count_of + n_labels);
  int* const lo_of = hi_of + n_labels;
  unsigned* const bad_of = (unsigned*)(lo_of + n_labels);
  double* const sum_of = (double*)zx_lds;
  for (int l = threadIdx.x; l < n_labels; l += ZX_THREADS) {
    if constexpr (PHASE == 0) {
      count_of[l] = 0u;
      hi_of[l] = INT_MIN;
      lo_of[l] = INT_MAX;
      bad_of[l] = 0u;
    } else {
      sum_of[l] = 0.0;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * ZX_THREADS + threadIdx.x) >> 6;
  const size_t begin = wave * cells_per_wave, end = min(begin + cells_per_wave, cells);
  int run_label = -1, marked = INT_MIN;
  unsigned run_count = 0, run_bad = 0;
  int run_hi = INT_MIN, run_lo = INT_MAX;
  double run_sum = 0.0;
  auto flush = [&]() {  // (wave-uniform call)
    if (run_label < 0) return;
    if constexpr (PHASE == 0) {
      const unsigned n = wave_sum_u32(run_count), bad = wave_or_u32(run_bad);
      const int hi = wave_max(run_hi), lo = wave_min(run_lo);
      if (lane == 0 && n != 0) {
        atomicAdd(&count_of[run_label], n);
        atomicMax(&hi_of[run_label], hi);
        atomicMin(&lo_of[run_label], lo);
        if (bad) atomicOr(&bad_of[run_label], 1u);
      }
    } else {
      const double total = wave_sum_exact(run_sum);
      if (lane == 0 && total != 0.0) atomicAdd(&sum_of[run_label], total);
    }
    run_label = -1;
    run_count = run_bad = 0;
    run_hi = INT_MIN;
    run_lo = INT_MAX;
    run_sum = 0.0;
  };
  walk_cells<T>(values, labels, begin, end, lane, [&](T v, int label, bool inside, size_t i) {
    (void)inside;
    (void)i;
    const bool labelled = (unsigned)label < (unsigned)n_labels;
    if constexpr (PHASE == 0) mark_present(present, label, labelled, &marked);
    bool active = labelled && !(has_ndv && v == ndv);
    if (thresholds && active) {
      const float t = thresholds[label];
      active = !(t != t) && (double)v >= (double)t;
    }
    const unsigned long long mask = __ballot(active);
    if (mask == 0) return;  // (wave-uniform)
    const double x = (double)v;
    int top = INT_MIN, low = INT_MAX;
    unsigned bad = 0;
    if constexpr (PHASE == 0) {
      if (active) {
        if (!(fabs(x) <= 1.7976931348623157e308)) bad = 1;
        else if (x != 0.0) cell_span<T>(v, &top, &low);
      }
    }
    const int leader_label = __builtin_amdgcn_readlane(label, __ffsll((long long)mask) - 1);
    if (__all(!active || label == leader_label)) {
      if (leader_label != run_label) {
        flush();
        run_label = leader_label;
      }
      if (active) {
        if constexpr (PHASE == 0) {
          run_count += 1;
          run_bad |= bad;
          run_hi = max(run_hi, top);
          run_lo = min(run_lo, low);
        } else {
          run_sum += x;  // (exact: the verdict of PHASE 0)
        }
      }
    } else if (active) {
      if constexpr (PHASE == 0) {
        atomicAdd(&count_of[label], 1u);
        if (top != INT_MIN) {
          atomicMax(&hi_of[label], top);
          atomicMin(&lo_of[label], low);
        }
        if (bad) atomicOr(&bad_of[label], 1u);
      } else if (x != 0.0) {
        atomicAdd(&sum_of[label], x);
      }
    }
  });
  flush();
  __syncthreads();
  for (int l = threadIdx.x; l < n_labels; l += ZX_THREADS) {
    if constexpr (PHASE == 0) {
      const unsigned n = count_of[l];
      if (n == 0) continue;
      ExactRange* slot = range + l;
      atomicAdd(&slot->count, (unsigned long long)n);
      if (hi_of[l] != INT_MIN) {
        atomicMax(&slot->hi, hi_of[l]);
        atomicMin(&slot->lo, lo_of[l]);
      }
      if (bad_of[l]) atomicOr(&slot->bad, 1u);
    } else {
      const double total = sum_of[l];
      if (total != 0.0) atomicAdd(&sums[l], total);
    }
  }
}

// Sums of labels WITHOUT runs whose cells do round (round 6: zero-mean noise under 20 labels at random: 1 ms through the sort
// and the block pipeline): the float32 certificate of zonal_chunk_combine_kernel (derived there) over workgroups instead of
// chunks.  A workgroup's waves walk adjacent stretches, so a workgroup covers a contiguous range of the raster; per label it
// leaves a row {sum of its cells in whatever order the LDS atomics arrived, sum of their magnitudes, count, non-finite?}, and
// a wave per label then walks the workgroups' rows in raster order: the partial sums inside workgroup g lie within
// (the sum entering g) -/+ mag_g, which bounds sum_k |X_k| by W = sum_g count_g (|entering_g| + mag_g).  What the rows' sums and
// their combination can be off by: a lane's chain over a run of whole steps adds at most (steps per wave) results, each at most
// the lane's share of mag_g; the reduction across the wave 6 levels of at most mag_g together; every addition onto the table
// entry -- a flushed run, a scattered cell: counted in `adds` -- a result of at most mag_g; the combination here up to 16 + 6
// levels more: V = sum_g (adds_g + steps + 32) mag_g.  A label that fails -- a sum next to a float32 tie, a non-finite cell -- raises
// `failed` and the whole frame goes on to the routes that add in order.
struct LazyRow {  // 24 bytes per (workgroup, label)
  double sum, mag;
  unsigned count;
  unsigned adds;  // additions onto the workgroup's table entry (a flushed run, a scattered cell); bit 31: a non-finite cell
};
template <typename T, int THREADS>
__global__ __launch_bounds__(THREADS) void zonal_tables_lazy_kernel(const T* __restrict__ values, const int* __restrict__ labels, size_t cells,
                                                                    size_t cells_per_wave, int n_labels, int has_ndv, T ndv,
                                                                    const float* __restrict__ thresholds, LazyRow* __restrict__ rows,
                                                                    unsigned char* __restrict__ present) {
  extern __shared__ unsigned long long zx_lds[];
  double* const sum_of = (double*)zx_lds;
  double* const mag_of = sum_of + n_labels;
  unsigned* const count_of = (unsigned*)(mag_of + n_labels);
  unsigned* const bad_of = count_of + n_labels;
  for (int l = threadIdx.x; l < n_labels; l += THREADS) {
    sum_of[l] = 0.0;
    mag_of[l] = 0.0;
    count_of[l] = 0u;
    bad_of[l] = 0u;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * THREADS + threadIdx.x) >> 6;
  const size_t begin = min(wave * cells_per_wave, cells), end = min(begin + cells_per_wave, cells);
  int run_label = -1, marked = INT_MIN;
  unsigned run_count = 0, run_bad = 0;
  double run_sum = 0.0, run_mag = 0.0;
  auto flush = [&]() {  // (wave-uniform call)
    if (run_label < 0) return;
    const unsigned n = wave_sum_u32(run_count), bad = wave_or_u32(run_bad);
    const double total = wave_sum_exact(run_sum), mag = wave_sum_exact(run_mag);
    if (lane == 0 && n != 0) {
      atomicAdd(&count_of[run_label], n);
      if (mag != 0.0) {
        atomicAdd(&sum_of[run_label], total);
        atomicAdd(&mag_of[run_label], mag);
        atomicAdd(&bad_of[run_label], 1u);
      }
      if (bad) atomicOr(&bad_of[run_label], 0x80000000u);
    }
    run_label = -1;
    run_count = run_bad = 0;
    run_sum = run_mag = 0.0;
  };
  walk_cells<T>(values, labels, begin, end, lane, [&](T v, int label, bool inside, size_t i) {
    (void)inside;
    (void)i;
    const bool labelled = (unsigned)label < (unsigned)n_labels;
    mark_present(present, label, labelled, &marked);
    bool active = labelled && !(has_ndv && v == ndv);
    if (thresholds && active) {
      const float t = thresholds[label];
      active = !(t != t) && (double)v >= (double)t;
    }
    const unsigned long long mask = __ballot(active);
    if (mask == 0) return;  // (wave-uniform)
    const double x = (double)v;
    const bool bad = active && !(fabs(x) <= 1.7976931348623157e308);
    const int leader_label = __builtin_amdgcn_readlane(label, __ffsll((long long)mask) - 1);
    if (__all(!active || label == leader_label)) {
      if (leader_label != run_label) {
        flush();
        run_label = leader_label;
      }
      if (active) {
        run_count += 1;
        run_bad |= bad ? 1u : 0u;
        if (!bad) {
          run_sum += x;
          run_mag += fabs(x);
        }
      }
    } else {
      // a step of several labels: the lanes of the run's label (a label with strays in it, the larger part of a boundary step)
      // stay in registers like a step of one label, the others go to the table one by one
      if (run_label < 0) run_label = leader_label;  // (wave-uniform)
      if (active && label == run_label) {
        run_count += 1;
        run_bad |= bad ? 1u : 0u;
        if (!bad) {
          run_sum += x;
          run_mag += fabs(x);
        }
      } else if (active) {
        atomicAdd(&count_of[label], 1u);
        if (bad) {
          atomicOr(&bad_of[label], 0x80000000u);
        } else if (x != 0.0) {
          atomicAdd(&sum_of[label], x);
          atomicAdd(&mag_of[label], fabs(x));
          atomicAdd(&bad_of[label], 1u);
        }
      }
    }
  });
  flush();
  __syncthreads();
  LazyRow* const mine = rows + (size_t)blockIdx.x * n_labels;
  for (int l = threadIdx.x; l < n_labels; l += THREADS) mine[l] = LazyRow{sum_of[l], mag_of[l], count_of[l], bad_of[l]};
}

__global__ __launch_bounds__(256) void zonal_exact_init_kernel(ExactRange* __restrict__ range, double* __restrict__ sums, int n_labels,
                                                               ExactVerdict* __restrict__ verdict, unsigned* __restrict__ cursor,
                                                               double* __restrict__ mags = nullptr, unsigned char* __restrict__ inexact = nullptr,
                                                               unsigned* __restrict__ slots = nullptr, int n_slot_words = 0) {
  // (the one-pass route of the small labels: its other buffers in the same launch -- every launch of a few waves takes ~5 us)
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l == 0) {
    *verdict = ExactVerdict{0u, 0u, 0ull};
    *cursor = 0;
  }
  if (slots && l < n_slot_words) slots[l] = 0u;
  if (l >= n_labels) return;
  range[l] = ExactRange{0ull, INT_MIN, INT_MAX, 0u};
  sums[l] = 0.0;
  if (mags) mags[l] = 0.0;
  if (inexact) inexact[l] = 1;
}

// per label: can the order matter?  The verdict counts the labels for which it can and the cells they hold, and flags them.
__global__ __launch_bounds__(256) void zonal_exact_decide_kernel(const ExactRange* __restrict__ range, int n_labels,
                                                                 unsigned char* __restrict__ inexact, ExactVerdict* __restrict__ verdict) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n_labels) return;
  const ExactRange r = range[l];
  bool exact = !r.bad;
  if (exact && r.count != 0 && r.hi != INT_MIN) {
    const int growth = r.count <= 1 ? 0 : 64 - __clzll((long long)(r.count - 1));  // ceil(log2(count))
    const int top = r.hi + 1 + growth;
    exact = top - r.lo <= 53 && top < 1023;
  }
  inexact[l] = exact ? 0 : 1;
  if (r.count) atomicMax(&verdict->longest, (unsigned)min(r.count, 0xffffffffull));
  if (!exact) {
    atomicAdd(&verdict->labels, 1u);
    atomicAdd(&verdict->cells, r.count);
  }
}

// Round 6: the labels the exponents turn away, asked again with the float32 certificate of zonal_chunk_combine_kernel (derived
// there) in its crudest form.  PHASE 2 of zonal_exact_kernel has added a label's cells in some order -- every intermediate
// result a sum of some of them, at most mag = the sum of their magnitudes; at most 2 count + 16 results: a lane's chain, the
// reductions across a wave, one atomic per run -- so its sum lies within tau = 1.5 u (2 count + 16) mag of the exact one, and
// bincount's within 1.01 u count mag (every partial sum is at most mag).  Labels of a few hundred cells -- the rasters of many
// small polygons that reach this route -- pass unless their sum lies within 10^-9 of a float32 tie: 49 152 polygons of 16 x 16
// cells of zero-mean noise used to send the ~2 000 of them with a cell next to zero through a gather and six sorting passes
// (470 us a call).  A label that passes loses its flag (the float32 of its sum is the answer, by monotonicity); the verdict is
// counted again over those that keep it.
__global__ __launch_bounds__(256) void zonal_exact_lazy_decide_kernel(const ExactRange* __restrict__ range, const double* __restrict__ sums,
                                                                      const double* __restrict__ mags, int n_labels, int statistic,
                                                                      unsigned char* __restrict__ inexact, ExactVerdict* __restrict__ verdict,
                                                                      int packed_counts = 0, double steps_per_wave = 0.0,
                                                                      float* __restrict__ out = nullptr, long long* __restrict__ counts = nullptr) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n_labels) return;
  if (out) {  // (zonal_exact_result_kernel's work in this launch: the results as if the label passes)
    const unsigned long long cells_l = packed_counts ? range[l].count & ((1ull << 40) - 1) : range[l].count;
    counts[l] = (long long)cells_l;
    out[l] = cells_l == 0 ? std::numeric_limits<float>::quiet_NaN()
                          : (statistic == Z_MEAN ? (float)(sums[l] / (double)cells_l) : (float)sums[l]);
  }
  if (!inexact[l]) return;
  const ExactRange r = range[l];
  // (after PHASE 3 the count carries the number of additions onto the label's sum in its bits 40 and up: one per run.  With
  //  them the sum's own error is u (additions + steps of a wave's stretch + 16) mag instead of u (2 count + 16) mag -- a lane's
  //  chain over a run adds at most one result per step, the reduction across a wave six levels, every atomic one result --,
  //  which for polygons whose rows are a step or two is a hundred times less)
  const unsigned long long cells = packed_counts ? r.count & ((1ull << 40) - 1) : r.count;
  const double S = sums[l], mag = mags[l], n = (double)cells;
  const double additions = packed_counts ? (double)(r.count >> 40) + steps_per_wave : 2.0 * n;
  bool pass = false;
  if (!r.bad && cells != 0 && mag < 1.0e300) {  // (false for NaN)
    constexpr double u = 1.1102230246251565e-16;
    const double tau = 1.5 * u * (additions + 16.0) * mag;
    const double E = 1.01 * u * (n * mag + 4.0 * mag + (2.0 * n + 4.0) * tau) + tau + (mag > 0.0 ? 1.0e-300 : 0.0);
    double below = S - E, above = S + E;
    if (statistic == Z_MEAN) {
      below /= n;
      above /= n;
    }
    const float lowest = (float)below, highest = (float)above;
    pass = __float_as_uint(lowest) == __float_as_uint(highest) && fabsf(lowest) < __builtin_huge_valf();
  }
  if (pass || cells == 0) {  // (a label without an active cell: NaN, whatever is asked)
    inexact[l] = 0;
  } else {
    const unsigned slot = atomicAdd(&verdict->labels, 1u);
    atomicAdd(&verdict->cells, cells);
    if (packed_counts) inexact[l] = (unsigned char)(min(slot, 254u) + 1u);  // (its slot among the labels zonal_few_* add in order)
  }
}

// When only a few labels need their cells in order, only their cells are sorted: the active cells of the flagged labels are
// appended to a list as (label << 32 | cell index, value) -- in any order: the sort by that key restores the raster order
// inside every label --, and the sequential sums run on the list.
template <typename K>
__global__ __launch_bounds__(256) void zonal_gather_kernel(const unsigned* __restrict__ keys, const K* __restrict__ payload, size_t cells,
                                                           int n_labels, const unsigned char* __restrict__ inexact,
                                                           unsigned long long* __restrict__ list_keys, K* __restrict__ list_payload,
                                                           unsigned* __restrict__ cursor) {
  const int lane = threadIdx.x & 63;
  const size_t stride = (size_t)gridDim.x * 256, rounded = (cells + 63) & ~(size_t)63;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < rounded; i += stride) {  // (whole waves enter every step)
    const unsigned label = i < cells ? keys[i] : (unsigned)n_labels;
    const bool take = label < (unsigned)n_labels && inexact[label];
    const unsigned long long mask = __ballot(take);
    if (mask == 0) continue;
    unsigned base = 0;
    if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(cursor, (unsigned)__popcll(mask));
    base = __builtin_amdgcn_readlane(base, __ffsll((long long)mask) - 1);
    if (take) {
      const unsigned at = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
      list_keys[at] = ((unsigned long long)label << 32) | (unsigned long long)i;
      list_payload[at] = payload[i];
    }
  }
}
__global__ __launch_bounds__(256) void zonal_list_bounds_kernel(const unsigned long long* __restrict__ keys, unsigned count, int n_labels,
                                                                unsigned* __restrict__ first) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l > n_labels) return;
  const unsigned long long bound = (unsigned long long)l << 32;
  unsigned lo = 0, hi = count;
  while (lo < hi) {
    const unsigned mid = lo + (hi - lo) / 2;
    if (keys[mid] < bound) lo = mid + 1;
    else hi = mid;
  }
  first[l] = lo;
}

// ---------------------------------------------------------------------------------
// the label sort as a stable partition (up to ZP_BINS - 1 labels)
// ---------------------------------------------------------------------------------
// np.lexsort / bincount's order needs the cells of every label in raster order: a STABLE sort by label.  A general radix
// sort takes two 8-bit passes for the 11 bits of 1024 labels, each moving key and payload (2 x 103 us on the 12.6 M
// cells of the default raster limit, + 30 us of histograms).  With a few thousand labels at most the sort is one
// counting pass instead:
//   count    a workgroup per tile of ZP_TILE consecutive cells: the tile's histogram (LDS) -> counts[tile][bin] (where the
//            keys come from zonal_exact_kernel, that kernel has gathered the counts on its way: one atomic per run);
//   scan     a wave per bin: the exclusive prefix over the tiles -> offsets[tile][bin], totals[bin];
//   first    one workgroup: the exclusive prefix of the totals = first[] (what zonal_bounds_kernel searches for);
//   scatter  a workgroup per tile: every wave owns ZP_WAVE consecutive cells of it, histograms them into its own LDS row,
//            the rows become the waves' write cursors (first[bin] + offsets[tile][bin] + the rows of the waves before),
//            and the wave walks its cells again in order, 64 at a time: rank among the lanes holding the same label
//            (match_label: one ballot per label bit; label rasters are runs, and a step whose 64 cells share one label
//            takes a short cut), cursor read, cells stored at cursor + rank, cursor advanced by the last such lane.
// Every cell is read twice (the second time from the L2) and written once.
constexpr int ZP_BINS = 2048, ZP_GROUP = 8;  // (a wave of an NW-wave workgroup owns ZP_TILE / NW cells)

// among the lanes with valid set: how many hold this lane's label, and how many of those sit below it
__device__ __forceinline__ void match_label(unsigned label, bool valid, int bits, unsigned* rank, unsigned* count) {
  unsigned long long mask = __ballot(valid);
  for (int b = 0; b < bits; ++b) {
    const bool one = (label >> b) & 1u;
    const unsigned long long ones = __ballot(one);
    mask &= one ? ones : ~ones;
  }
  *rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
  *count = (unsigned)__popcll(mask);
}

// One step of 64 keys (~0u: past the end) into `hist` (LDS).  A step holds few distinct keys as a rule -- one label, or one label
// and the idle key of its no-data cells: every distinct key is ONE addition of its lanes' count (a ballot per key); a step of
// more than ZP_DISTINCT keys finishes with an LDS atomic per remaining lane (the order of the additions is immaterial here).
constexpr int ZP_DISTINCT = 4;
__device__ __forceinline__ void histogram_step(unsigned key, unsigned* hist, int lane) {
  unsigned long long open = __ballot(key != ~0u);
#pragma unroll 1
  for (int round = 0; open && round < ZP_DISTINCT; ++round) {
    const unsigned lead = (unsigned)__builtin_amdgcn_readlane((int)key, __ffsll((long long)open) - 1);
    const unsigned long long same = __ballot(key == lead);
    if (lane == 0) atomicAdd(&hist[lead], (unsigned)__popcll(same));
    open &= ~same;
  }
  if (open && ((open >> lane) & 1ull)) atomicAdd(&hist[key], 1u);
}
// histogram of the WAVE_CELLS cells from `begin` on into `hist` (LDS)
template <int WAVE_CELLS>
__device__ __forceinline__ void partition_histogram(const unsigned* __restrict__ keys, unsigned begin, unsigned cells, unsigned* hist,
                                                    int lane) {
  constexpr int STEPS = WAVE_CELLS / 64, GROUP = STEPS < 8 ? STEPS : 8;
  for (int step0 = 0; step0 < STEPS; step0 += GROUP) {
    unsigned label[GROUP];
#pragma unroll
    for (int k = 0; k < GROUP; ++k) {
      const unsigned i = begin + (unsigned)(step0 + k) * 64u + (unsigned)lane;
      label[k] = keys[min(i, cells - 1)];  // (unconditional: a branch around a load makes the compiler wait for it)
      if (i >= cells) label[k] = ~0u;
    }
#pragma unroll
    for (int k = 0; k < GROUP; ++k) histogram_step(label[k], hist, lane);
  }
}

// zonal_keys_kernel and the counting pass in one (where the keys do not come from zonal_exact_kernel): a workgroup per tile
// turns its cells into (key, payload) and histograms the keys on the way
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void zonal_keys_count_kernel(const T* __restrict__ values, const int* __restrict__ labels,
                                                                   unsigned cells, int n_labels, int has_ndv, T ndv,
                                                                   const float* __restrict__ thresholds, unsigned* __restrict__ keys,
                                                                   typename Ordered<T>::Key* __restrict__ payload,
                                                                   unsigned char* __restrict__ present, unsigned* __restrict__ counts) {
  __shared__ unsigned hist[ZP_BINS];
  const int bins = n_labels + 1;
  for (int b = threadIdx.x; b < bins; b += 64 * NW) hist[b] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int WAVE_CELLS = ZP_TILE / NW, STEPS = WAVE_CELLS / 64, GROUP = STEPS < 8 ? STEPS : 8;
  const unsigned begin = blockIdx.x * (unsigned)ZP_TILE + (unsigned)wave * WAVE_CELLS;
  int marked = INT_MIN;  // (mark_present)
  for (int step0 = 0; step0 < STEPS; step0 += GROUP) {
    T v[GROUP];
    int label[GROUP];
#pragma unroll
    for (int k = 0; k < GROUP; ++k) {
      const unsigned i = min(begin + (unsigned)(step0 + k) * 64u + (unsigned)lane, cells - 1);
      v[k] = values[i];
      label[k] = labels[i];
    }
#pragma unroll
    for (int k = 0; k < GROUP; ++k) {
      const unsigned i = begin + (unsigned)(step0 + k) * 64u + (unsigned)lane;
      const bool inside = i < cells;
      const bool labelled = (unsigned)label[k] < (unsigned)n_labels;
      mark_present(present, inside ? label[k] : -1, inside && labelled, &marked);
      bool active = labelled && !(has_ndv && v[k] == ndv);
      if (thresholds && active) {
        const float t = thresholds[label[k]];
        active = !(t != t) && (double)v[k] >= (double)t;
      }
      const unsigned key = inside ? (active ? (unsigned)label[k] : (unsigned)n_labels) : ~0u;
      if (inside) {
        keys[i] = key;
        payload[i] = Ordered<T>::encode(v[k]);
      }
      histogram_step(key, hist, lane);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += 64 * NW) counts[(size_t)blockIdx.x * bins + b] = hist[b];
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void zonal_part_count_kernel(const unsigned* __restrict__ keys, unsigned cells, int bins,
                                                                   unsigned* __restrict__ counts) {
  __shared__ unsigned hist[ZP_BINS];
  for (int b = threadIdx.x; b < bins; b += 64 * NW) hist[b] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  partition_histogram<ZP_TILE / NW>(keys, blockIdx.x * (unsigned)ZP_TILE + (unsigned)wave * (ZP_TILE / NW), cells, hist, lane);
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += 64 * NW) counts[(size_t)blockIdx.x * bins + b] = hist[b];
}

__global__ __launch_bounds__(256) void zonal_part_scan_kernel(const unsigned* __restrict__ counts, unsigned tiles, int bins,
                                                              unsigned* __restrict__ offsets, unsigned* __restrict__ totals) {
  const int bin = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (bin >= bins) return;
  unsigned carry = 0;
  for (unsigned t0 = 0; t0 < tiles; t0 += 64) {
    const unsigned t = t0 + (unsigned)lane;
    const unsigned c = t < tiles ? counts[(size_t)t * bins + bin] : 0u;
    unsigned inclusive = c;
#pragma unroll
    for (int offset = 1; offset < 64; offset <<= 1) {
      const unsigned below = __shfl_up(inclusive, offset);
      if (lane >= offset) inclusive += below;
    }
    if (t < tiles) offsets[(size_t)t * bins + bin] = carry + inclusive - c;
    carry += __shfl(inclusive, 63);
  }
  if (lane == 0) totals[bin] = carry;
}

__global__ __launch_bounds__(256) void zonal_part_first_kernel(const unsigned* __restrict__ totals, int bins, unsigned* __restrict__ first) {
  __shared__ unsigned partial[256];
  const int per = (bins + 255) / 256;
  unsigned local = 0;
  for (int k = 0; k < per; ++k) {
    const int b = threadIdx.x * per + k;
    if (b < bins) local += totals[b];
  }
  partial[threadIdx.x] = local;
  __syncthreads();
  for (int offset = 1; offset < 256; offset <<= 1) {
    const unsigned below = (int)threadIdx.x >= offset ? partial[threadIdx.x - offset] : 0u;
    __syncthreads();
    partial[threadIdx.x] += below;
    __syncthreads();
  }
  unsigned run = partial[threadIdx.x] - local;
  for (int k = 0; k < per; ++k) {
    const int b = threadIdx.x * per + k;
    if (b < bins) {
      first[b] = run;
      run += totals[b];
    }
  }
}

template <typename K, int NW>
__global__ __launch_bounds__(64 * NW) void zonal_part_scatter_kernel(const unsigned* __restrict__ keys, const K* __restrict__ payload,
                                                                     unsigned cells, int bins, int bits,
                                                                     const unsigned* __restrict__ offsets,
                                                                     const unsigned* __restrict__ first, unsigned* __restrict__ out_keys,
                                                                     K* __restrict__ out_payload) {
  constexpr int WAVE_CELLS = ZP_TILE / NW, STEPS = WAVE_CELLS / 64, GROUP = STEPS < ZP_GROUP ? STEPS : ZP_GROUP;
  extern __shared__ unsigned cursor[];  // [NW][bins]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < NW * bins; i += 64 * NW) cursor[i] = 0;
  __syncthreads();
  unsigned* const mine = cursor + wave * bins;  // (a wave's LDS operations complete in order: no barrier around its own row)
  const unsigned begin = blockIdx.x * (unsigned)ZP_TILE + (unsigned)wave * WAVE_CELLS;
  partition_histogram<WAVE_CELLS>(keys, begin, cells, mine, lane);
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += 64 * NW) {
    unsigned at = first[b] + offsets[(size_t)blockIdx.x * bins + b];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const unsigned c = cursor[w * bins + b];
      cursor[w * bins + b] = at;
      at += c;
    }
  }
  __syncthreads();
  for (int step0 = 0; step0 < STEPS; step0 += GROUP) {
    unsigned label[GROUP];
    K cell[GROUP];
#pragma unroll
    for (int k = 0; k < GROUP; ++k) {
      const unsigned i = begin + (unsigned)(step0 + k) * 64u + (unsigned)lane;
      const unsigned at = min(i, cells - 1);
      label[k] = keys[at];
      cell[k] = payload[at];
      if (i >= cells) label[k] = ~0u;
    }
#pragma unroll
    for (int k = 0; k < GROUP; ++k) {
      // the step's distinct keys one after the other (one label, or one label and the idle key, as a rule): the lanes of a key
      // rank themselves with one ballot; a step of more than ZP_DISTINCT keys finishes with match_label (a ballot per label bit)
      unsigned long long open = __ballot(label[k] != ~0u);
#pragma unroll 1
      for (int round = 0; open && round < ZP_DISTINCT; ++round) {
        const unsigned lead = (unsigned)__builtin_amdgcn_readlane((int)label[k], __ffsll((long long)open) - 1);
        const unsigned long long same = __ballot(label[k] == lead);
        const unsigned base = mine[lead];  // (one address for the wave)
        if (lane == 0) mine[lead] = base + (unsigned)__popcll(same);
        if (label[k] == lead) {
          const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0u));
          if (out_keys) out_keys[base + rank] = lead;  // (the select path only wants the payloads)
          out_payload[base + rank] = cell[k];
        }
        open &= ~same;
      }
      if (open == 0) continue;  // (wave-uniform)
      const bool valid = ((open >> lane) & 1ull) != 0;
      unsigned rank, count;
      match_label(label[k], valid, bits, &rank, &count);
      if (valid) {
        const unsigned base = mine[label[k]];
        if (rank + 1 == count) mine[label[k]] = base + count;
        if (out_keys) out_keys[base + rank] = label[k];
        out_payload[base + rank] = cell[k];
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// stable radix sort of (key, value) pairs: passes of the counting pass above over digits of up to 11 bits
// ---------------------------------------------------------------------------------
// What the partition cannot take in one pass -- more than ZP_BINS - 1 labels, a sort by VALUE (32- or 64-bit keys), the
// (label << 32 | cell) list of the labels that need their cells in order -- is sorted least significant digit first: every pass
// is the partition's count -> scan -> first -> scatter with digit = (key >> shift) & mask as the bin, moving key and value;
// each pass is stable, so the whole is.  (Rounds 1-4 called rocPRIM's device radix sort here.)
constexpr int RS_NW = 8;
template <typename KeyT>
__device__ __forceinline__ unsigned radix_digit(KeyT key, int shift, unsigned mask) {
  return (unsigned)(key >> shift) & mask;
}
template <typename KeyT, int NW>
__global__ __launch_bounds__(64 * NW) void radix_count_kernel(const KeyT* __restrict__ keys, unsigned cells, int shift, unsigned mask,
                                                              int bins, unsigned* __restrict__ counts) {
  __shared__ unsigned hist[ZP_BINS];
  for (int b = threadIdx.x; b < bins; b += 64 * NW) hist[b] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int WAVE_CELLS = ZP_TILE / NW, STEPS = WAVE_CELLS / 64, GROUP = STEPS < 8 ? STEPS : 8;
  const unsigned begin = blockIdx.x * (unsigned)ZP_TILE + (unsigned)wave * WAVE_CELLS;
  for (int step0 = 0; step0 < STEPS; step0 += GROUP) {
    unsigned digit[GROUP];
#pragma unroll
    for (int k = 0; k < GROUP; ++k) {
      const unsigned i = begin + (unsigned)(step0 + k) * 64u + (unsigned)lane;
      digit[k] = radix_digit(keys[min(i, cells - 1)], shift, mask);  // (unconditional load: see partition_histogram)
      if (i >= cells) digit[k] = ~0u;
    }
#pragma unroll
    for (int k = 0; k < GROUP; ++k) {
      const unsigned f = (unsigned)__builtin_amdgcn_readfirstlane((int)digit[k]);
      if (__all(digit[k] == f)) {
        if (lane == 0 && f != ~0u) atomicAdd(&hist[f], 64u);
      } else if (digit[k] != ~0u) {
        atomicAdd(&hist[digit[k]], 1u);
      }
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += 64 * NW) counts[(size_t)blockIdx.x * bins + b] = hist[b];
}

template <typename KeyT, typename ValT, int NW>
__global__ __launch_bounds__(64 * NW) void radix_scatter_kernel(const KeyT* __restrict__ keys, const ValT* __restrict__ values,
                                                                unsigned cells, int shift, unsigned mask, int bins, int bits,
                                                                const unsigned* __restrict__ offsets, const unsigned* __restrict__ first,
                                                                KeyT* __restrict__ out_keys, ValT* __restrict__ out_values) {
  constexpr int WAVE_CELLS = ZP_TILE / NW, STEPS = WAVE_CELLS / 64, GROUP = STEPS < ZP_GROUP ? STEPS : ZP_GROUP;
  extern __shared__ unsigned cursor[];  // [NW][bins]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < NW * bins; i += 64 * NW) cursor[i] = 0;
  __syncthreads();
  unsigned* const mine = cursor + wave * bins;  // (a wave's LDS operations complete in order: no barrier around its own row)
  const unsigned begin = blockIdx.x * (unsigned)ZP_TILE + (unsigned)wave * WAVE_CELLS;
  for (int step0 = 0; step0 < STEPS; step0 += GROUP) {  // the wave's own histogram
    unsigned digit[GROUP];
#pragma unroll
    for (int k = 0; k < GROUP; ++k) {
      const unsigned i = begin + (unsigned)(step0 + k) * 64u + (unsigned)lane;
      digit[k] = radix_digit(keys[min(i, cells - 1)], shift, mask);
      if (i >= cells) digit[k] = ~0u;
    }
#pragma unroll
    for (int k = 0; k < GROUP; ++k) {
      const unsigned f = (unsigned)__builtin_amdgcn_readfirstlane((int)digit[k]);
      if (__all(digit[k] == f)) {
        if (lane == 0 && f != ~0u) atomicAdd(&mine[f], 64u);
      } else if (digit[k] != ~0u) {
        atomicAdd(&mine[digit[k]], 1u);
      }
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += 64 * NW) {  // the rows become the waves' write cursors
    unsigned at = first[b] + offsets[(size_t)blockIdx.x * bins + b];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const unsigned c = cursor[w * bins + b];
      cursor[w * bins + b] = at;
      at += c;
    }
  }
  __syncthreads();
  for (int step0 = 0; step0 < STEPS; step0 += GROUP) {
    unsigned digit[GROUP];
    KeyT key[GROUP];
    ValT cell[GROUP];
#pragma unroll
    for (int k = 0; k < GROUP; ++k) {
      const unsigned i = begin + (unsigned)(step0 + k) * 64u + (unsigned)lane;
      const unsigned at = min(i, cells - 1);
      key[k] = keys[at];
      cell[k] = values[at];
      digit[k] = i < cells ? radix_digit(key[k], shift, mask) : ~0u;
    }
#pragma unroll
    for (int k = 0; k < GROUP; ++k) {
      const unsigned f = (unsigned)__builtin_amdgcn_readfirstlane((int)digit[k]);
      if (__all(digit[k] == f)) {
        if (f == ~0u) continue;
        const unsigned base = mine[f];
        if (lane == 0) mine[f] = base + 64u;
        out_keys[base + lane] = key[k];
        out_values[base + lane] = cell[k];
        continue;
      }
      const bool valid = digit[k] != ~0u;
      unsigned rank, count;
      match_label(digit[k], valid, bits, &rank, &count);
      if (valid) {
        const unsigned base = mine[digit[k]];
        if (rank + 1 == count) mine[digit[k]] = base + count;
        out_keys[base + rank] = key[k];
        out_values[base + rank] = cell[k];
      }
    }
  }
}

struct RadixTemp {  // counts / offsets of a pass: tiles x bins each; totals, first: bins (+ 1)
  unsigned *counts = nullptr, *offsets = nullptr, *totals = nullptr, *first = nullptr;
  size_t tiles = 0;
};
static size_t radix_temp_bytes(size_t n) {
  const size_t tiles = (n + ZP_TILE - 1) / ZP_TILE;
  return sizeof(unsigned) * (2 * tiles * ZP_BINS + 2 * ZP_BINS + 2) + 64;
}
// Sorts the n pairs by bits [0, end_bit) of their keys; the sorted pairs end up in (keys_b, values_b), the contents of
// (keys_a, values_a) are destroyed.  temp: radix_temp_bytes(n) bytes.
template <typename KeyT, typename ValT>
int radix_sort_pairs(void* temp, KeyT* keys_a, KeyT* keys_b, ValT* values_a, ValT* values_b, size_t n, int end_bit, hipStream_t s) {
  if (n == 0) return GM_OK;
  if (n >= (1ull << 32) - ZP_TILE) return gm::fail(GM_ERR_UNSUPPORTED, "gm_zonal_statistic: %zu cells in one sort", n);
  const size_t tiles = (n + ZP_TILE - 1) / ZP_TILE;
  unsigned* counts = (unsigned*)temp;
  unsigned* offsets = counts + tiles * ZP_BINS;
  unsigned* totals = offsets + tiles * ZP_BINS;
  unsigned* first = totals + ZP_BINS;
  if (end_bit < 1) end_bit = 1;
  const int passes = (end_bit + 10) / 11, width = (end_bit + passes - 1) / passes;
  KeyT *kin = keys_a, *kout = keys_b;
  ValT *vin = values_a, *vout = values_b;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = pass * width, bits = std::min(width, end_bit - shift), bins = 1 << bits;
    const unsigned mask = (unsigned)bins - 1u;
    const size_t lds = sizeof(unsigned) * RS_NW * (size_t)bins;
    if (lds > 65536)
      GM_HIP(hipFuncSetAttribute((const void*)radix_scatter_kernel<KeyT, ValT, RS_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((radix_count_kernel<KeyT, RS_NW>), dim3((unsigned)tiles), dim3(64 * RS_NW), 0, s, kin, (unsigned)n, shift, mask, bins,
                       counts);
    hipLaunchKernelGGL(zonal_part_scan_kernel, dim3((unsigned)(((size_t)bins * 64 + 255) / 256)), dim3(256), 0, s, counts, (unsigned)tiles,
                       bins, offsets, totals);
    hipLaunchKernelGGL(zonal_part_first_kernel, dim3(1), dim3(256), 0, s, totals, bins, first);
    hipLaunchKernelGGL((radix_scatter_kernel<KeyT, ValT, RS_NW>), dim3((unsigned)tiles), dim3(64 * RS_NW), lds, s, kin, vin, (unsigned)n,
                       shift, mask, bins, bits, offsets, first, kout, vout);
    std::swap(kin, kout);
    std::swap(vin, vout);
  }
  GM_HIP(hipGetLastError());
  if (kin != keys_b) {  // (an even number of passes ended in the a-buffers)
    GM_HIP(hipMemcpyAsync(keys_b, kin, sizeof(KeyT) * n, hipMemcpyDeviceToDevice, s));
    GM_HIP(hipMemcpyAsync(values_b, vin, sizeof(ValT) * n, hipMemcpyDeviceToDevice, s));
  }
  return GM_OK;
}

// ---------------------------------------------------------------------------------
// medians and percentiles by radix SELECT (round 3; up to ZP_BINS - 1 labels)
// ---------------------------------------------------------------------------------
// lexsort((values, labels)) costs two full radix sorts (0.44 + 0.24 ms for 12.6 M float32 cells) to read two cells per
// label.  Instead: the cells are partitioned by label (the counting pass above; their order inside a label is immaterial
// here), and every label finds the key of its wanted rank digit by digit from the top: per 8-bit digit one pass over the
// partitioned keys builds, per label, the histogram of that digit among the cells that match the digits chosen so far
// (a wave keeps the histogram of the label it is in in LDS and sends it to the label's 256 global counters when the label
// changes -- the array is sorted by label, so that is once per label and wave), and a lane per label picks the digit that
// holds the rank.  sizeof(T) passes (the order-preserving keys of narrow types have zero upper bytes); then, for a label
// that also needs the element one rank up (an even count, a fractional percentile index) and has no second cell equal to
// the selected key, one more pass finds the smallest key above it.
struct SelectState {
  unsigned long long prefix;  // the digits chosen so far, in place
  unsigned rank;              // rank of the wanted cell among the cells that match the prefix
  unsigned n_eq;              // (after the last pass) cells equal to the selected key
  unsigned want_next;         // the cell one rank up is wanted as well
  unsigned count;             // active cells of the label
};
constexpr int SEL_WAVE = 4096;  // cells a wave walks per pass

__global__ __launch_bounds__(256) void zonal_select_init_kernel(const unsigned* __restrict__ first, int n_labels, int statistic,
                                                                double qval, SelectState* __restrict__ state,
                                                                unsigned long long* __restrict__ next) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n_labels) return;
  const unsigned n = first[l + 1] - first[l];
  SelectState st{0ull, 0u, 0u, 0u, n};
  if (n != 0) {
    if (statistic == Z_MEDIAN) {  // scipy.ndimage._measurements._select: the two middle cells
      const unsigned step = (n - 1) / 2;
      st.rank = step;
      st.want_next = (n - 1 - step) != step ? 1u : 0u;
    } else {  // measurements.percentile: floor and ceiling of the fractional index
      const double frac = (double)(n - 1) * (qval / 100.0);
      st.rank = (unsigned)floor(frac);
      st.want_next = ceil(frac) != floor(frac) ? 1u : 0u;
    }
  }
  state[l] = st;
  next[l] = ~0ull;
}

// NEXT = false: the histogram pass for the digit at `shift`; NEXT = true: the smallest key above the selected one.
// A wave walks SEL_WAVE consecutive cells of the partitioned array, label segment by label segment (first[] says where the
// labels change: the label array itself is not read).
template <typename K, bool NEXT>
__global__ __launch_bounds__(256) void zonal_select_pass_kernel(const unsigned* __restrict__ first, const K* __restrict__ keys,
                                                                int n_labels, int shift, const SelectState* __restrict__ state,
                                                                unsigned* __restrict__ hist, unsigned long long* __restrict__ next) {
  __shared__ unsigned local[4][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned* const mine = local[wave];  // (a wave's LDS operations complete in order: no barrier around its own row)
  const unsigned active = first[n_labels];
  const unsigned begin = (unsigned)(((size_t)blockIdx.x * 4 + wave) * SEL_WAVE);
  if (begin >= active) return;
  const unsigned end = min(begin + (unsigned)SEL_WAVE, active);
  // the label that holds cell `begin`: the last one that starts at or before it
  int cur = 0;
  {
    int lo = 0, hi = n_labels;  // first[lo] <= begin < first[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) / 2;
      if (first[mid] <= begin) lo = mid;
      else hi = mid;
    }
    cur = lo;
  }
  if constexpr (!NEXT) {
#pragma unroll
    for (int k = 0; k < 4; ++k) mine[lane + 64 * k] = 0;
  }
  unsigned pos = begin;
  while (pos < end) {
    while (first[cur + 1] <= pos) ++cur;  // (labels without cells; wave-uniform)
    const SelectState st = state[cur];
    const unsigned long long prefix = st.prefix;
    const unsigned stop = min(end, first[cur + 1]);
    if constexpr (!NEXT) {
      const bool top = shift + 8 >= 64;
      for (unsigned i0 = pos; i0 < stop; i0 += 256) {  // (four steps' loads together)
        unsigned long long key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) key[u] = (unsigned long long)keys[min(i0 + 64u * u + (unsigned)lane, stop - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool inside = i0 + 64u * u + (unsigned)lane < stop;
          const bool match = top || (key[u] >> ((shift + 8) & 63)) == (prefix >> ((shift + 8) & 63));
          if (inside && match) atomicAdd(&mine[(unsigned)(key[u] >> shift) & 255u], 1u);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned c = mine[lane + 64 * k];
        if (c) atomicAdd(&hist[(size_t)cur * 256 + lane + 64 * k], c);
        mine[lane + 64 * k] = 0;
      }
    } else {
      if (st.want_next && !(st.rank + 1 < st.n_eq)) {  // (wave-uniform)
        unsigned long long best = ~0ull;
        for (unsigned i0 = pos; i0 < stop; i0 += 256) {
          unsigned long long key[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) key[u] = (unsigned long long)keys[min(i0 + 64u * u + (unsigned)lane, stop - 1)];
#pragma unroll
          for (int u = 0; u < 4; ++u)  // (a clamped re-read is a cell of the segment: harmless for a minimum)
            if (key[u] > prefix && key[u] < best) best = key[u];
        }
#pragma unroll
        for (int offset = 32; offset >= 1; offset >>= 1) {
          const unsigned lo = __shfl_xor((unsigned)best, offset), hi = __shfl_xor((unsigned)(best >> 32), offset);
          const unsigned long long other = ((unsigned long long)hi << 32) | lo;
          best = other < best ? other : best;
        }
        if (lane == 0 && best != ~0ull) atomicMin(&next[cur], best);
      }
    }
    pos = stop;
  }
}

// a wave per label: lane l holds the bins 4 l .. 4 l + 3, the wave scans the 64 lane totals and the lane whose bins hold the
// rank picks the digit
__global__ __launch_bounds__(256) void zonal_select_step_kernel(int n_labels, int shift, SelectState* __restrict__ state,
                                                                unsigned* __restrict__ hist) {
  const int l = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (l >= n_labels) return;
  SelectState st = state[l];
  if (st.count == 0) return;  // (wave-uniform)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4* const h = reinterpret_cast<u32x4*>(hist + (size_t)l * 256) + lane;
  const u32x4 c = *h;
  *h = u32x4{0u, 0u, 0u, 0u};
  const unsigned mine = c[0] + c[1] + c[2] + c[3];
  unsigned inclusive = mine;
#pragma unroll
  for (int offset = 1; offset < 64; offset <<= 1) {
    const unsigned below = __shfl_up(inclusive, offset);
    if (lane >= offset) inclusive += below;
  }
  unsigned before = inclusive - mine;
  const bool holds = st.rank >= before && st.rank < inclusive;  // (exactly one lane: the counts add up to more than the rank)
  if (holds) {
    int digit = 4 * lane;
    unsigned inside = c[0];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (st.rank >= before + inside) {
        before += inside;
        digit += 1;
        inside = c[k + 1];
      }
    }
    st.prefix |= (unsigned long long)digit << shift;
    st.rank -= before;
    st.n_eq = inside;
    state[l] = st;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void zonal_select_result_kernel(const SelectState* __restrict__ state,
                                                                  const unsigned long long* __restrict__ next, int n_labels,
                                                                  int statistic, double qval, float* __restrict__ out,
                                                                  long long* __restrict__ counts) {
  typedef typename Ordered<T>::Key Key;
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n_labels) return;
  const SelectState st = state[l];
  const unsigned n = st.count;
  counts[l] = n;
  if (n == 0) {
    out[l] = std::numeric_limits<float>::quiet_NaN();
    return;
  }
  const Key k1 = (Key)st.prefix;
  const Key k2 = st.want_next ? ((st.rank + 1 < st.n_eq) ? k1 : (Key)next[l]) : k1;
  const T x = Ordered<T>::decode(k1), y = Ordered<T>::decode(k2);
  float result;
  if (statistic == Z_MEDIAN) {  // (as zonal_pick_kernel)
    if constexpr (std::is_same<T, float>::value) result = (x + y) / 2.0f;
    else if constexpr (std::is_same<T, double>::value) result = (float)((x + y) / 2.0);
    else result = (float)(((double)x + (double)y) / 2.0);
  } else {
    const double frac = (double)(n - 1) * (qval / 100.0);
    const double below = floor(frac);
    const T diff = (T)(y - x);
    result = (float)((double)x + (frac - below) * (double)diff);
  }
  out[l] = result;
}

struct PartitionBuffers {
  unsigned *counts = nullptr, *offsets = nullptr, *totals = nullptr;
  bool counted = false;  // counts[] already holds this frame's tile histograms (zonal_exact_kernel gathered them)
};

__global__ __launch_bounds__(256) void zonal_exact_result_kernel(const ExactRange* __restrict__ range, const double* __restrict__ sums,
                                                                 int n_labels, int statistic, float* __restrict__ out,
                                                                 long long* __restrict__ counts, int packed_counts = 0) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n_labels) return;
  const unsigned long long n = packed_counts ? range[l].count & ((1ull << 40) - 1) : range[l].count;  // (PHASE 3 of zonal_exact_kernel)
  counts[l] = (long long)n;
  const double total = sums[l];
  out[l] = n == 0 ? std::numeric_limits<float>::quiet_NaN() : (statistic == Z_MEAN ? (float)(total / (double)n) : (float)total);
}

// ---------------------------------------------------------------------------------------------------------------------
// sum / mean without the sort (round 5): chunk records.
//
// bincount adds a label's cells in raster order.  Cut the raster into stretches of consecutive cells, one per wave: the cells
// of label l inside stretch w -- its CHUNK (w, l) -- are consecutive in l's sequence, and the chunks of l in the order of w are
// that sequence.  Pass 1 (zonal_chunk_kernel, the only pass over the raster: 8 bytes per cell, 4 cells per lane and step) leaves
// per chunk what certifies it and what it adds up to when certified: the count, the highest exponent and the lowest set bit
// among its cells, the extreme partial sums on the way, a tree sum, the steps it lies in; and a note of the same kind per step of 256
// cells.  Pass 2 (zonal_chunk_combine_kernel, a wave per label) walks the label's chunks in order: a chunk none of whose
// additions can round, given the running sum that enters it -- every intermediate value a multiple of 2^lo below 2^(lo + 53)
// -- is ONE addition of its tree sum; any other chunk is replayed: its cells read from the raster again and added one after
// the other (chunks of one or two steps), or step by step from the notes (certified steps one addition each, the others cell
// after cell).  Nothing is sorted, nothing is written per cell.  A wave finds its chunks through one byte per (label,
// stretch): the slot of the chunk among the stretch's records.
// Label rasters that are not made of runs (more than ZC_SLOTS labels inside a stretch, or more than ZC_LABELS_PER_STEP in one
// step) raise `overflow` and the sums take the sorting route above.
constexpr int ZC_LANE = 4, ZC_STEP = 64 * ZC_LANE;  // cells per lane and step / per step
constexpr int ZC_SLOTS = 64;                        // chunk records per stretch (a lane each)
constexpr int ZC_MAX_STEPS = 64;                    // steps per stretch
constexpr int ZC_LABELS_PER_STEP = 32;
constexpr int ZC_LIST = 4096;                       // chunks of a label listed at a time (one byte-row round of the table)
struct ChunkRec {  // 32 bytes
  double total;    // tree sum of the chunk's cells (meaningful when the chunk is certified)
  float low, high;  // the lowest / highest partial sum while the chunk's cells are added in order, relative to the sum entering
                    // the chunk and to within the four cells of a lane (rounded outwards; low <= 0 <= high)
  unsigned count;  // active cells
  int label;
  short hi, lo;    // top exponent / exponent of the lowest set bit over the non-zero cells (SHRT_MIN / SHRT_MAX: none)
  unsigned char first, last;  // the steps of the stretch that hold the chunk's cells
  unsigned char bad, pad;     // bad: a NaN or an infinity among the cells
};
struct StepRec {  // 24 bytes, per step of ZC_STEP cells
  double total;
  float low, high;  // as in ChunkRec
  int label;        // >= 0: the one label of the step's active cells; -1: no active cell; -2: add it literally (several labels, a non-finite cell)
  short hi, lo;     // as in ChunkRec
};
struct ChunkVerdict {
  unsigned overflow;  // pass 1: the label raster is not made of runs, or its stretches hold more labels than records
  unsigned hard;      // pass 2: a label whose chunks had to be replayed too often (a long label, or a sum that keeps rounding)
  unsigned not_runs;  // pass 1: ... a step with more labels than a raster of runs has (shorter stretches would not help)
};
// A label is given up (`hard`: the whole call takes the sorting route, whose block pipeline is built for such sums) after this
// many replayed chunks -- replays read the raster again, a few microseconds each -- or, once ZL_LONG of its cells have gone
// by, at the first chunk that has to be walked step by step.
constexpr unsigned ZC_REPLAYS = 64, ZC_LONG_WALKS = 0;
// ... or after this many steps added cell after cell (a chunk scattered over a stretch of mixed steps is a dozen of them)
constexpr unsigned ZC_LITERAL_STEPS = 160;
template <typename T>
struct alignas(4 * sizeof(T) > 16 ? 16 : 4 * sizeof(T)) CellQuad {
  T c[ZC_LANE];
};
struct alignas(16) LabelQuad {
  int c[ZC_LANE];
};

template <typename T>
__device__ __forceinline__ bool cell_active(T v, int label, int n_labels, int has_ndv, T ndv, const float* __restrict__ thresholds) {
  bool active = (unsigned)label < (unsigned)n_labels && !(has_ndv && v == ndv);
  if (thresholds && active) {
    const float t = thresholds[label];
    active = !(t != t) && (double)v >= (double)t;
  }
  return active;
}
__device__ __forceinline__ double readlane_f64(double x, int lane) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, lane), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// No addition of `cells` cells -- multiples of 2^low, below 2^(top + 1) each, their partial sums in order inside [below, above]
// -- onto acc can round: every intermediate value is a multiple of 2^g, g = min(low, lowest set bit of acc), and lies inside
// acc + [below, above]; below 2^(g + 53) in magnitude it is a float64.  Two bounds, either suffices: by the exponents
// (zonal_sum_kernel's), by the partial sums (zonal_group_exact_kernel's: tighter by what a sum of mixed signs does not grow;
// above - below < 2^(52 + g) makes sure that every RANGE of the cells sums without rounding, so that the partial sums -- float64
// sums themselves, hence the 0.99 -- and the tree sum are the exact ones).
__device__ __forceinline__ bool additions_exact(double acc, int top, int low, unsigned cells, double below, double above) {
  if (top == INT_MIN) return true;  // nothing but zeros
  if (!(fabs(acc) <= 1.7976931348623157e308)) return false;
  int acc_top = INT_MIN, acc_low = INT_MAX;
  if (acc != 0.0) exponent_span(acc, &acc_top, &acc_low);
  const int g = min(low, acc_low);
  const int growth = cells <= 1 ? 0 : 32 - __clz((int)(cells - 1));  // ceil(log2(cells))
  const long long hi = max((long long)acc_top, (long long)top + growth) + 2;
  if (hi - g <= 53 && hi < 1023) return true;
  if (!(g > -1000 && g < 900)) return false;
  const double reach = fmax(fabs(acc + below), fabs(acc + above));
  return reach < ldexp(0.99, 53 + g) && above - below < ldexp(0.99, 52 + g);  // (false for NaN / inf bounds)
}
// Inclusive sum over the lanes in lane order (data-parallel primitives: shifts inside the rows of 16 lanes -- a lane without a
// source adds 0.0 --, then the row totals are passed on like in wave_sum_exact)
__device__ __forceinline__ double wave_scan_f64(double v) {
  v += dpp_move_f64<0x111>(0.0, v);        // row_shr:1
  v += dpp_move_f64<0x112>(0.0, v);        // row_shr:2
  v += dpp_move_f64<0x114>(0.0, v);        // row_shr:4
  v += dpp_move_f64<0x118>(0.0, v);        // row_shr:8
  v += dpp_move_f64<0x142, 0xA>(0.0, v);   // row_bcast15 into rows 1 and 3
  v += dpp_move_f64<0x143, 0xC>(0.0, v);   // row_bcast31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
  const double inf = __builtin_huge_val();
  v = fmin(v, dpp_move_f64<0xB1>(v, v));
  v = fmin(v, dpp_move_f64<0x4E>(v, v));
  v = fmin(v, dpp_move_f64<0x141>(v, v));
  v = fmin(v, dpp_move_f64<0x140>(v, v));
  v = fmin(v, dpp_move_f64<0x142, 0xA>(inf, v));
  v = fmin(v, dpp_move_f64<0x143, 0xC>(inf, v));
  return readlane_f64(v, 63);
}
__device__ __forceinline__ double wave_max_f64(double v) { return -wave_min_f64(-v); }
__device__ __forceinline__ float wave_min_f32(float v) {
  const int inf = 0x7f800000;
  v = fminf(v, __int_as_float(dpp_move<0xB1>(__float_as_int(v), __float_as_int(v))));
  v = fminf(v, __int_as_float(dpp_move<0x4E>(__float_as_int(v), __float_as_int(v))));
  v = fminf(v, __int_as_float(dpp_move<0x141>(__float_as_int(v), __float_as_int(v))));
  v = fminf(v, __int_as_float(dpp_move<0x140>(__float_as_int(v), __float_as_int(v))));
  v = fminf(v, __int_as_float(dpp_move<0x142, 0xA>(inf, __float_as_int(v))));
  v = fminf(v, __int_as_float(dpp_move<0x143, 0xC>(inf, __float_as_int(v))));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f32(float v) { return -wave_min_f32(-v); }
// float32 bounds that contain the float64 ones (inf for what float32 cannot hold: such a chunk is never certified by them)
__device__ __forceinline__ float outward_low(double x) { return (float)(fmin(x, 0.0) * 1.000001 - 1.0e-30); }
__device__ __forceinline__ float outward_high(double x) { return (float)(fmax(x, 0.0) * 1.000001 + 1.0e-30); }
// acc + the first n of the doubles parked in `mine`, in order, n <= ZC_STEP; the caller has zeroed up to the next multiple of
// 16 (the wave as one sequential adder: zonal_sum_kernel's replay)
__device__ __forceinline__ double chain_add_parked(const double* mine, double acc, unsigned n) {
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  constexpr int ROUND = 16;
  f64x2 now[ROUND / 2], next[ROUND / 2];
#pragma unroll
  for (int k = 0; k < ROUND / 2; ++k) now[k] = *reinterpret_cast<const f64x2*>(mine + 2 * k);
  for (unsigned base = 0; base < n; base += ROUND) {  // (wave-uniform)
    // the next round's reads are issued before this round's 16 additions (past n: the zeros, or stale values never added)
#pragma unroll
    for (int k = 0; k < ROUND / 2; ++k) next[k] = *reinterpret_cast<const f64x2*>(mine + base + ROUND + 2 * k);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < ROUND / 2; ++k) {
      acc += now[k].x;
      acc += now[k].y;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < ROUND / 2; ++k) now[k] = next[k];
  }
  return acc;
}

// Medians / percentiles under A FEW labels (round 6): the radix select straight off the raster.  The partition that brings every
// label's cells together costs as much as four select passes over the raster itself (12.6 M cells: 47 + 11 + 5 + 45 us against
// ~22 us a pass), and a handful of labels needs no partition to keep their histograms apart: a workgroup keeps 256 bins per label
// in LDS (up to ZR_LABELS labels), a lane adds its cell's digit to its label's row, a workgroup flushes once.  The first pass
// counts every active cell (its histogram's totals are the labels' counts: zonal_select_raw_counts_kernel turns them into the
// `first` array the partitioned route gets from its sort, so init / step / result are that route's kernels); later passes count
// the cells that match the digits chosen so far; the last one looks for the smallest key above the selected one.  Keys of 4
// bytes (or less): four or fewer digit passes.  One label of 12.6 M float32 cells: median 244 -> ~150 us a call.
constexpr int ZR_LABELS = 32;
template <typename T, bool NEXT>
__global__ __launch_bounds__(256) void zonal_select_raw_pass_kernel(const T* __restrict__ values, const int* __restrict__ labels, size_t cells,
                                                                    size_t cells_per_wave, int n_labels, int has_ndv, T ndv,
                                                                    const float* __restrict__ thresholds,
                                                                    int shift, int first_pass, const SelectState* __restrict__ state,
                                                                    unsigned* __restrict__ hist, unsigned long long* __restrict__ next,
                                                                    unsigned char* __restrict__ present) {
  typedef typename Ordered<T>::Key Key;
  static_assert(sizeof(Key) == 4, "keys of up to 4 bytes");
  __shared__ unsigned bins[NEXT ? 1 : ZR_LABELS * 256];
  __shared__ unsigned prefix_of[ZR_LABELS], wanted[ZR_LABELS], best_of[ZR_LABELS];
  for (int i = threadIdx.x; i < n_labels; i += 256) {
    const SelectState st = first_pass ? SelectState{0ull, 0u, 0u, 0u, 0u} : state[i];
    prefix_of[i] = (unsigned)st.prefix;
    wanted[i] = (st.want_next && !(st.rank + 1 < st.n_eq) && st.count != 0) ? 1u : 0u;
    best_of[i] = ~0u;
  }
  if constexpr (!NEXT) {
    for (int i = threadIdx.x; i < n_labels * 256; i += 256) bins[i] = 0u;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  int marked = INT_MIN;
  // (a wave walks a stretch of its own with its loads four steps ahead, like the count kernel: a grid-stride loop of one cell
  //  per lane and turn took 29-47 us a pass where the raster's bytes take 20)
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const size_t begin = min(wave * cells_per_wave, cells), end = min(begin + cells_per_wave, cells);
  walk_cells<T>(values, labels, begin, end, lane, [&](T v, int label, bool inside, size_t i) {
    (void)i;
    const bool labelled = inside && (unsigned)label < (unsigned)n_labels;
    if (first_pass) mark_present(present, label, labelled, &marked);
    const bool active = labelled && cell_active<T>(v, label, n_labels, has_ndv, ndv, thresholds);
    const unsigned key = (unsigned)Ordered<T>::encode(v);
    if constexpr (!NEXT) {
      const bool match = active && (first_pass || shift + 8 >= 32 || (key >> (shift + 8)) == (prefix_of[label] >> (shift + 8)));
      const unsigned long long mask = __ballot(match);
      if (mask == 0) return;  // (wave-uniform)
      const unsigned bin = match ? (unsigned)label * 256u + ((key >> shift) & 255u) : ~0u;
      // (cells of one label in one bin -- a raster whose values share their top digit -- go in one addition per wave)
      const unsigned lead = (unsigned)__builtin_amdgcn_readlane((int)bin, __ffsll((long long)mask) - 1);
      if (__all(!match || bin == lead)) {
        if (lane == __ffsll((long long)mask) - 1) atomicAdd(&bins[lead], (unsigned)__popcll(mask));
      } else if (match) {
        atomicAdd(&bins[bin], 1u);
      }
    } else {
      // (a plain look at the best so far first: after a few steps hardly a cell is below it, and a median has half of the
      //  label's cells above the selected key -- an LDS atomic each took the pass from 21 to 42 us)
      if (active && wanted[label] && key > prefix_of[label] && key < *(volatile unsigned*)&best_of[label]) atomicMin(&best_of[label], key);
    }
  });
  __syncthreads();
  if constexpr (!NEXT) {
    for (int i = threadIdx.x; i < n_labels * 256; i += 256) {
      const unsigned c = bins[i];
      if (c) atomicAdd(&hist[i], c);
    }
  } else {
    for (int i = threadIdx.x; i < n_labels; i += 256)
      if (best_of[i] != ~0u) atomicMin(&next[i], (unsigned long long)best_of[i]);
  }
}
// first[l] = the active cells of the labels before l (what the partitioned route reads off its sorted keys): from the totals of
// the first pass's histograms, which stay where they are for the first step
__global__ __launch_bounds__(64) void zonal_select_raw_counts_kernel(const unsigned* __restrict__ hist, int n_labels, unsigned* __restrict__ first) {
  const int lane = threadIdx.x;
  unsigned running = 0;
  for (int l = 0; l < n_labels; ++l) {
    unsigned c = 0;
    for (int k = lane; k < 256; k += 64) c += hist[(size_t)l * 256 + k];
    c = wave_sum_u32(c);
    if (lane == 0) first[l] = running;
    running += c;
  }
  if (lane == 0) first[n_labels] = running;
}

// The few labels of a raster of thousands of small ones that the float32 certificate of the parallel sums turns away (a sum that
// cancels to nothing, a sum within 10^-9 of a float32 tie: a handful of 6 144 or 49 152 polygons per call) -- added in bincount's
// order without a sort of the raster: zonal_exact_lazy_decide_kernel has given each a slot (up to ZF_SLOTS); their active cells
// are gathered from the raster itself as (cell index, float64 value) into their slot's list (up to ZF_CAP cells each); a
// workgroup per slot sorts its list by cell index in LDS (a bitonic network) and one lane adds it, one cell after the other.
// Both kernels leave at once when the verdict counted no label (they are queued before the host knows); more labels than slots
// or a label longer than a list raise `failed_host`, and the frame goes on to the routes that sort.
constexpr unsigned ZF_CAP = 4096, ZF_SLOTS = 64;
template <typename T>
__global__ __launch_bounds__(256) void zonal_few_gather_kernel(const T* __restrict__ values, const int* __restrict__ labels, size_t cells,
                                                               int n_labels, int has_ndv, T ndv, const float* __restrict__ thresholds,
                                                               const unsigned char* __restrict__ inexact,
                                                               const ExactVerdict* __restrict__ verdict,
                                                               unsigned* __restrict__ list_index, double* __restrict__ list_values,
                                                               int* __restrict__ slot_label, unsigned* __restrict__ slot_cursor) {
  if (verdict->labels == 0 || verdict->labels > ZF_SLOTS) return;  // (wave-uniform: nothing to add in order / too many)
  const int lane = threadIdx.x & 63;
  const size_t stride = (size_t)gridDim.x * 256, rounded = (cells + 63) & ~(size_t)63;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < rounded; i += stride) {  // (whole waves enter every step)
    const int label = i < cells ? labels[i] : -1;
    unsigned slot = 0;
    T v = T(0);
    if ((unsigned)label < (unsigned)n_labels) slot = inexact[label];
    if (slot > ZF_SLOTS) slot = 0;
    if (slot) {
      v = values[i];
      if (!cell_active<T>(v, label, n_labels, has_ndv, ndv, thresholds)) slot = 0;
    }
    // one atomic per wave and slot (a row of a polygon is 16-64 lanes of one label: a lane each on one counter took 174 us
    // for forty labels of 1 800 cells)
    unsigned long long open = __ballot(slot != 0);
    while (open) {  // (wave-uniform)
      const unsigned leader = (unsigned)__builtin_amdgcn_readlane((int)slot, __ffsll((long long)open) - 1);
      const unsigned long long peers = __ballot(slot == leader);
      unsigned base = 0;
      if (lane == __ffsll((long long)peers) - 1) base = atomicAdd(&slot_cursor[leader - 1], (unsigned)__popcll(peers));
      base = (unsigned)__builtin_amdgcn_readlane((int)base, __ffsll((long long)peers) - 1);
      if (slot == leader) {
        const unsigned at = base + (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
        if (at == 0) slot_label[slot - 1] = label;
        if (at < ZF_CAP) {
          list_index[(slot - 1) * ZF_CAP + at] = (unsigned)i;  // (fewer than 2^31 cells)
          list_values[(slot - 1) * ZF_CAP + at] = (double)v;   // what bincount adds
        }
      }
      open &= ~peers;
    }
  }
}
__global__ __launch_bounds__(1024) void zonal_few_ordered_kernel(const unsigned* __restrict__ list_index, const double* __restrict__ list_values,
                                                                 const int* __restrict__ slot_label, const unsigned* __restrict__ slot_cursor,
                                                                 const ExactVerdict* __restrict__ verdict, int statistic,
                                                                 float* __restrict__ out, long long* __restrict__ counts,
                                                                 unsigned* __restrict__ failed_host) {
  __shared__ unsigned key[ZF_CAP];
  __shared__ double val[ZF_CAP];
  const unsigned n_failed = verdict->labels, slot = blockIdx.x;
  if (n_failed == 0) return;
  if (n_failed > ZF_SLOTS) {
    if (slot == 0 && threadIdx.x == 0) *failed_host = 1u;
    return;
  }
  if (slot >= n_failed) return;
  const unsigned m = slot_cursor[slot];
  if (m > ZF_CAP || m == 0) {  // (m == 0 cannot be: a label that failed has active cells)
    if (threadIdx.x == 0) *failed_host = 1u;
    return;
  }
  unsigned P = 64;
  while (P < m) P <<= 1;
  for (unsigned i = threadIdx.x; i < P; i += 1024) {
    key[i] = i < m ? list_index[slot * ZF_CAP + i] : ~0u;
    val[i] = i < m ? list_values[slot * ZF_CAP + i] : 0.0;
  }
  __syncthreads();
  for (unsigned size = 2; size <= P; size <<= 1) {
    for (unsigned gap = size >> 1; gap > 0; gap >>= 1) {
      for (unsigned t = threadIdx.x; t < P / 2; t += 1024) {
        const unsigned lo = (t / gap) * 2 * gap + t % gap, hi = lo + gap;
        const bool up = (lo & size) == 0;
        const unsigned a = key[lo], b = key[hi];
        if ((a > b) == up) {
          key[lo] = b;
          key[hi] = a;
          const double x = val[lo];
          val[lo] = val[hi];
          val[hi] = x;
        }
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    double acc = 0.0;
    for (unsigned j = 0; j < m; ++j) acc += val[j];  // np.bincount: one cell after the other, in raster order
    const int label = slot_label[slot];
    counts[label] = (long long)m;
    out[label] = statistic == Z_MEAN ? (float)(acc / (double)m) : (float)acc;
  }
}

// gm_zonal_statistic_host: the results on their way to the host BEFORE the one wait of a route that has a verdict to wait for.
// The routes of the sums end in "kernels queued -> wait for the verdict word -> (the caller copies the results: a second wait)";
// with the copy queued behind the kernels the verdict's wait covers it (one wake-up of the host per call instead of two: ~20 us
// of a call of 80-120).  The copy is only trusted when the verdict says the route answered; anything else is copied again.
struct HostCopy {
  void* pinned = nullptr;     // the thread's pinned staging buffer
  const void* dev = nullptr;  // the device range that holds counts | out | present of the call
  size_t bytes = 0;
  bool queued = false, valid = false;
};
static thread_local HostCopy* t_host_copy = nullptr;
constexpr size_t kHostCopyBytes = 1u << 20;  // (49 152 labels: 639 KB)
static void* host_copy_staging() {
  static thread_local void* buffer = nullptr;
  if (!buffer && hipHostMalloc(&buffer, kHostCopyBytes, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    buffer = nullptr;
  }
  return buffer;
}
// (call sites: right before the event of a verdict is recorded / once the verdict is known)
static void host_copy_queue(int64_t frames, hipStream_t s) {
  HostCopy* const hc = t_host_copy;
  if (!hc || frames != 1 || !hc->pinned) return;
  hc->valid = false;
  hc->queued = hipMemcpyAsync(hc->pinned, hc->dev, hc->bytes, hipMemcpyDeviceToHost, s) == hipSuccess;
  if (!hc->queued) (void)hipGetLastError();
}
static void host_copy_settle(bool answered) {
  HostCopy* const hc = t_host_copy;
  if (!hc) return;
  hc->valid = hc->queued && answered;
  hc->queued = false;
}

// (the second kernel of zonal_tables_lazy_kernel: a wave per label over the workgroups' rows)
template <int PER>  // rows per lane and batch: 4, or 16 where a label has thousands of rows (a batch is one round of wave reductions)
__global__ __launch_bounds__(256) void zonal_tables_lazy_label_kernel(const LazyRow* __restrict__ rows, unsigned groups, int n_labels,
                                                                      double steps_per_wave, int statistic, int verbose,
                                                                      float* __restrict__ out, long long* __restrict__ counts,
                                                                      unsigned* __restrict__ failed) {
  const int l = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (l >= n_labels) return;
  double S = 0.0, M = 0.0, A = 0.0, W = 0.0, V = 0.0;
  unsigned long long n = 0;
  unsigned batches = 0;
  bool bad = false;
  // (the rows of the next 256 workgroups are requested before this batch's are composed: a wave per label is one memory round
  //  trip after the other otherwise -- 8192 rows of one label 31 us)
  auto fetch = [&](unsigned g0, LazyRow (&into)[PER]) {
#pragma unroll
    for (int c = 0; c < PER; ++c) {
      const unsigned g = g0 + (unsigned)PER * (unsigned)lane + (unsigned)c;
      LazyRow r{0.0, 0.0, 0u, 0u};
      if (g < groups) r = rows[(size_t)g * n_labels + l];
      into[c] = r;
    }
  };
  LazyRow ahead[PER];
  fetch(0u, ahead);
  for (unsigned g0 = 0; g0 < groups; g0 += 64u * PER) {  // the workgroups in raster order, PER per lane
    LazyRow four[PER];
    double run = 0.0, lane_mag = 0.0;
    unsigned lane_count = 0u;
#pragma unroll
    for (int c = 0; c < PER; ++c) four[c] = ahead[c];
    if (g0 + 64u * PER < groups) fetch(g0 + 64u * PER, ahead);  // (wave-uniform)
#pragma unroll
    for (int c = 0; c < PER; ++c) {
      const LazyRow r = four[c];
      run += r.sum;
      lane_mag += r.mag;
      lane_count += r.count;
      bad = bad || (r.adds & 0x80000000u) != 0;
    }
    double entering = S + (wave_scan_f64(run) - run), lane_weight = 0.0, lane_chain = 0.0, lane_reach = 0.0;
#pragma unroll
    for (int c = 0; c < PER; ++c) {
      const double reach = fabs(entering) + four[c].mag;
      lane_weight += (double)four[c].count * reach;
      lane_chain += ((double)(four[c].adds & 0x7fffffffu) + steps_per_wave + 32.0) * four[c].mag;
      if (four[c].count) lane_reach = fmax(lane_reach, reach);
      entering += four[c].sum;
    }
    M = fmax(M, wave_max_f64(lane_reach));
    W += wave_sum_exact(lane_weight);
    V += wave_sum_exact(lane_chain);
    S += wave_sum_exact(run);
    A += wave_sum_exact(lane_mag);
    n += wave_sum_u32(lane_count);
    ++batches;
  }
  if (n == 0 && __ballot(bad) == 0) {
    if (lane == 0) {
      counts[l] = 0;
      out[l] = std::numeric_limits<float>::quiet_NaN();
    }
    return;
  }
  if (__ballot(bad) == 0 && A < 1.0e300 && W < 1.0e300 && V < 1.0e300) {  // (wave-uniform; false for NaN)
    constexpr double u = 1.1102230246251565e-16;
    const double tau = 1.5 * u * (V + (double)(batches + 8u) * M);
    const double E = 1.01 * u * (W + 4.0 * M + (2.0 * (double)n + 4.0) * tau) + tau + (A > 0.0 ? 1.0e-300 : 0.0);
    double below = S - E, above = S + E;
    if (statistic == Z_MEAN) {
      below /= (double)n;
      above /= (double)n;
    }
    const float lowest = (float)below, highest = (float)above;
    if (verbose && lane == 0)
      printf("[gmhip] float32 certificate (tables), label %d: n %llu S %.17g E %.3g (W %.3g V %.3g M %.3g A %.3g tau %.3g) -> %.9g .. %.9g\n", l, n, S,
             E, W, V, M, A, tau, (double)lowest, (double)highest);
    if (__float_as_uint(lowest) == __float_as_uint(highest) && fabsf(lowest) < __builtin_huge_valf()) {
      if (lane == 0) {
        counts[l] = (long long)n;
        out[l] = lowest;
      }
      return;
    }
  }
  if (lane == 0) *failed = 1u;
}

// The verdict of pass 1 where the host can read it without a copy: a pinned word the kernel writes, and the event behind the
// kernel -- the host waits for pass 1 only, pass 2 is already queued behind it.
struct HostFlag {
  unsigned* word = nullptr;
  hipEvent_t event = nullptr;
  int device = -1;
};
static HostFlag* host_flag() {
  static thread_local HostFlag f;
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess) return nullptr;
  if (!f.word && hipHostMalloc((void**)&f.word, 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    f.word = nullptr;
    return nullptr;
  }
  if (f.device != device) {
    if (f.event) (void)hipEventDestroy(f.event);
    f.event = nullptr;
    if (hipEventCreateWithFlags(&f.event, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      f.event = nullptr;
      f.device = -1;
      return nullptr;
    }
    f.device = device;
  }
  return &f;
}

template <typename T>
__global__ __launch_bounds__(256) void zonal_chunk_kernel(const T* __restrict__ values, const int* __restrict__ labels, size_t cells,
                                                          size_t cells_per_wave, unsigned stretches, unsigned row_bytes, int n_labels,
                                                          int has_ndv, T ndv, const float* __restrict__ thresholds,
                                                          ChunkRec* __restrict__ records, unsigned char* __restrict__ slot_of,
                                                          StepRec* __restrict__ step_records, ChunkVerdict* __restrict__ verdict,
                                                          unsigned* __restrict__ verdict_host, unsigned char* __restrict__ present) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (wave >= stretches) return;
  const size_t begin = wave * cells_per_wave, end = min(begin + cells_per_wave, cells);
  if (begin >= end) return;
  const unsigned steps_per_wave = (unsigned)(cells_per_wave / ZC_STEP);
  const unsigned steps = (unsigned)((end - begin + ZC_STEP - 1) / ZC_STEP), full = (unsigned)((end - begin) / ZC_STEP);
  // the stretch's chunk records, a lane each
  int slot_label = -1, n_slots = 0;
  double r_total = 0.0;
  float r_low = 0.0f, r_high = 0.0f;
  unsigned r_count = 0, r_bad = 0, r_first = 255u, r_last = 0u;
  int r_hi = INT_MIN, r_lo = INT_MAX;
  bool dead = false;     // this label raster is not for chunk records (wave-uniform)
  bool scattered = false;  // the first step's lanes mostly hold several labels each (wave-uniform)
  int marked = INT_MIN;  // the label whose `present` flag the step before has set (INT_MIN: none / several)
  // the cells of the full steps: unconditional 16-byte loads, one step ahead
  auto fetch_v = [&](unsigned step) -> CellQuad<T> {
    return *reinterpret_cast<const CellQuad<T>*>(values + begin + (size_t)step * ZC_STEP + ZC_LANE * (size_t)lane);
  };
  auto fetch_l = [&](unsigned step) -> LabelQuad {
    return *reinterpret_cast<const LabelQuad*>(labels + begin + (size_t)step * ZC_STEP + ZC_LANE * (size_t)lane);
  };
  CellQuad<T> next_v{};
  LabelQuad next_l{};
  if (full) {
    next_v = fetch_v(0);
    next_l = fetch_l(0);
  }
  for (unsigned step = 0; step < steps; ++step) {
    // somebody else has found the raster unfit for this pass (more labels in a stretch than records, scattered labels): every
    // wave used to walk on until it found out for itself -- stripes 25 cells wide: 135 us for an attempt that one wave had
    // given up after a third of its stretch.  A look at the verdict every fourth step (a scalar load that hits the L2).
    if ((step & 3u) == 3u && __builtin_amdgcn_readfirstlane((int)__atomic_load_n(&verdict->overflow, __ATOMIC_RELAXED))) return;
    T v[ZC_LANE];
    int label[ZC_LANE];
    bool inside[ZC_LANE];
    if (step < full) {  // (wave-uniform)
#pragma unroll
      for (int k = 0; k < ZC_LANE; ++k) {
        v[k] = next_v.c[k];
        label[k] = next_l.c[k];
        inside[k] = true;
      }
      if (step + 1 < full) {
        next_v = fetch_v(step + 1);
        next_l = fetch_l(step + 1);
      }
    } else {  // the stretch's ragged last step (the raster's end): cell by cell
#pragma unroll
      for (int k = 0; k < ZC_LANE; ++k) {
        const size_t i = begin + (size_t)step * ZC_STEP + ZC_LANE * (size_t)lane + k;
        inside[k] = i < end;
        v[k] = inside[k] ? values[i] : T(0);
        label[k] = inside[k] ? labels[i] : -1;
      }
    }
    // `present[label] = 1` for every labelled cell; a step of one label that the step before has marked says nothing new
    if (present) {
      const int lead = __builtin_amdgcn_readfirstlane(label[0]);
      bool same = true;
#pragma unroll
      for (int k = 0; k < ZC_LANE; ++k) same = same && label[k] == lead;
      const bool uniform = __all(same);
      if (!(uniform && lead == marked)) {
#pragma unroll
        for (int k = 0; k < ZC_LANE; ++k)
          if ((unsigned)label[k] < (unsigned)n_labels) present[label[k]] = 1;
        marked = uniform ? lead : INT_MIN;
      }
    }
    bool remaining[ZC_LANE];
#pragma unroll
    for (int k = 0; k < ZC_LANE; ++k) remaining[k] = inside[k] && cell_active<T>(v[k], label[k], n_labels, has_ndv, ndv, thresholds);
    if (step == 0 && full) {
      // A label raster made of runs keeps a lane's four consecutive cells under one label nearly everywhere (stripes 25 cells
      // wide: 16 % of the lanes see two labels, polygons of 16 x 16 cells 25 %); labels scattered cell by cell do not (two
      // labels at random: 87 %, twenty: all).  Round 6: a wave whose FIRST step has more than half of its lanes on mixed
      // labels says so at once -- such a raster used to walk all of its steps here with a loop turn per label and step
      // (20 labels at random: 424 us) only for the second kernel to find every step marked "add it literally".
      // (... with MORE THAN EIGHT labels in that step: under a handful of labels the loop below is cheap, and integer cells
      //  under 2 labels at random are answered here in 133 us against 338 by the certificates of the parallel sums)
      bool mixed = false;
#pragma unroll
      for (int k = 1; k < ZC_LANE; ++k) mixed = mixed || (label[k] != label[0]);
      scattered = __popcll(__ballot(mixed)) > 32;
    }
    // what a cell contributes whatever its label: the float64 bincount adds, its top exponent (INT_MAX: not finite; INT_MIN:
    // zero) and the exponent of its lowest set bit -- once per cell, not once per label of the step
    double cell_x[ZC_LANE];
    int cell_hi[ZC_LANE], cell_lo[ZC_LANE];
#pragma unroll
    for (int k = 0; k < ZC_LANE; ++k) {
      cell_x[k] = (double)v[k];
      cell_hi[k] = INT_MIN;
      cell_lo[k] = INT_MAX;
      if constexpr (std::is_same<T, float>::value) {
        // float32: the exponent field; the lowest set bit e - 150 + ctz(mantissa with its hidden bit) -- one less than the
        // truth for subnormals, which only makes the test stricter (their top exponent -126: an upper bound)
        const unsigned magnitude_bits = __float_as_uint(v[k]) & 0x7fffffffu;
        if (magnitude_bits >= 0x7f800000u) {
          cell_hi[k] = INT_MAX;
        } else if (magnitude_bits != 0u) {
          cell_hi[k] = max((int)(magnitude_bits >> 23), 1) - 127;
          cell_lo[k] = (int)(magnitude_bits >> 23) + (__ffs((int)(magnitude_bits | 0x800000u)) - 1) - 150;
        }
      } else {
        if (!(fabs(cell_x[k]) <= 1.7976931348623157e308)) {
          cell_hi[k] = INT_MAX;
        } else if (cell_x[k] != 0.0) {
          cell_span<T>(v[k], &cell_hi[k], &cell_lo[k]);
        }
      }
    }
    StepRec note{0.0, 0.0f, 0.0f, -1, SHRT_MIN, SHRT_MAX};
    int found = 0;
    while (true) {
      const unsigned long long open = __ballot(remaining[0] | remaining[1] | remaining[2] | remaining[3]);
      if (!open) break;
      if (found == ZC_LABELS_PER_STEP) {
        dead = true;
        if (lane == 0) {
          verdict->not_runs = 1u;
          if (verdict_host) verdict_host[2] = 1u;
        }
        break;
      }
      const int mine = remaining[0] ? label[0] : remaining[1] ? label[1] : remaining[2] ? label[2] : label[3];
      const int lead = __builtin_amdgcn_readlane(mine, __ffsll((long long)open) - 1);  // (a uniform lane index)
      unsigned n = 0;
      int hi = INT_MIN, lo = INT_MAX;  // (hi == INT_MAX: a NaN or an infinity among the cells)
      double sum = 0.0, magnitude = 0.0;
#pragma unroll
      for (int k = 0; k < ZC_LANE; ++k) {
        const bool taken = remaining[k] && label[k] == lead;
        n += (unsigned)__popcll(__ballot(taken));  // (wave-uniform: the cells of this label in the step)
        if (taken) {
          remaining[k] = false;
          hi = max(hi, cell_hi[k]);
          lo = min(lo, cell_lo[k]);
          sum += cell_x[k];  // what bincount adds
          magnitude += fabs(cell_x[k]);
        }
      }
      hi = wave_max(hi);
      lo = wave_min(lo);
      const unsigned bad = hi == INT_MAX ? 1u : 0u;
      if (bad) hi = INT_MIN;
      // the stretch's record of this label (a lane's)
      const unsigned long long hit = __ballot(slot_label == lead);
      int slot;
      if (hit) {
        slot = __ffsll((long long)hit) - 1;
      } else if (n_slots < ZC_SLOTS) {
        slot = n_slots++;
      } else {
        dead = true;
        break;
      }
      // The partial sums in raster order, to within a lane's four cells: the sum before the lane -/+ the lane's magnitudes,
      // relative to the sum entering the CHUNK (the record's total so far); their extremes as float32 rounded outwards --
      // minima and maxima of float32 values do not round.
      const double inclusive = wave_scan_f64(sum);
      const double chunk_so_far = hit ? readlane_f64(r_total, slot) : 0.0;
      const double before = chunk_so_far + (inclusive - sum);
      const float lowest = wave_min_f32(outward_low(before - magnitude)), highest = wave_max_f32(outward_high(before + magnitude));
      sum = readlane_f64(inclusive, 63);  // the step's tree sum (only used when certified exact)
      if (found == 0) {
        note.total = sum;
        note.low = outward_low((double)lowest - chunk_so_far);  // (relative to the sum entering the step)
        note.high = outward_high((double)highest - chunk_so_far);
        note.label = bad ? -2 : lead;
        note.hi = (short)max(hi, (int)SHRT_MIN);  // (float64 exponents fit; INT_MIN -> SHRT_MIN)
        note.lo = (short)min(lo, (int)SHRT_MAX);
      } else {
        note.label = -2;
      }
      ++found;
      if (lane == slot) {
        slot_label = lead;
        r_low = fminf(r_low, lowest);
        r_high = fmaxf(r_high, highest);
        r_total += sum;
        r_count += n;
        r_bad |= bad;
        r_hi = max(r_hi, hi);
        r_lo = min(r_lo, lo);
        r_first = min(r_first, step);
        r_last = max(r_last, step);
      }
    }
    if (step == 0 && scattered && found > 8 && !dead) {
      dead = true;
      if (lane == 0) {
        verdict->not_runs = 1u;
        if (verdict_host) verdict_host[2] = 1u;
      }
    }
    if (dead) {
      if (lane == 0) {  // (plain stores: a system-scope atomic per wave on the pinned word cost 5 ms for 4096 waves)
        verdict->overflow = 1u;
        if (verdict_host) *verdict_host = 1u;
      }
      return;  // (whichever route answers instead marks `present` itself)
    }
    if (lane == 0) step_records[wave * steps_per_wave + step] = note;
  }
  if (lane < n_slots) {
    ChunkRec rec;
    rec.total = r_total;
    rec.low = r_low;
    rec.high = r_high;
    rec.count = r_count;
    rec.label = slot_label;
    rec.hi = (short)max(r_hi, (int)SHRT_MIN);
    rec.lo = (short)min(r_lo, (int)SHRT_MAX);
    rec.first = (unsigned char)r_first;
    rec.last = (unsigned char)r_last;
    rec.bad = (unsigned char)r_bad;
    rec.pad = 0;
    records[wave * ZC_SLOTS + lane] = rec;
    slot_of[(size_t)slot_label * row_bytes + wave] = (unsigned char)(lane + 1);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void zonal_chunk_combine_kernel(const T* __restrict__ values, const int* __restrict__ labels, size_t cells,
                                                                  size_t cells_per_wave, unsigned stretches, unsigned row_bytes, int n_labels,
                                                                  int has_ndv, T ndv, const float* __restrict__ thresholds,
                                                                  const ChunkRec* __restrict__ records,
                                                                  const unsigned char* __restrict__ slot_of,
                                                                  const StepRec* __restrict__ step_records,
                                                                  ChunkVerdict* __restrict__ verdict, unsigned* __restrict__ verdict_host,
                                                                  unsigned long_from, unsigned patience, int statistic, int lazy,
                                                                  float* __restrict__ out, long long* __restrict__ counts) {
  __shared__ __align__(16) double parked[4][2][ZC_STEP + 32];
  __shared__ unsigned listed[4][ZC_LIST];
  const int l = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (l >= n_labels) return;
  if (*(const volatile unsigned*)&verdict->overflow || *(const volatile unsigned*)&verdict->hard) return;  // (the sorting route will answer)
  double* const mine = parked[threadIdx.x >> 6][0];
  double* const mine_b = parked[threadIdx.x >> 6][1];
  unsigned* const list = listed[threadIdx.x >> 6];
  const unsigned steps_per_wave = (unsigned)(cells_per_wave / ZC_STEP);
  double acc = 0.0;
  unsigned long long n_total = 0;
  unsigned replays = 0, walks = 0;
  auto give_up = [&]() {
    if (lane == 0) {
      verdict->hard = 1u;
      if (verdict_host) verdict_host[1] = 1u;
    }
  };
  // One step of a stretch, cell after cell: the cells of this label parked in order, then the chain of dependent additions.
  struct StepCells {
    T v[ZC_LANE];
    int label[ZC_LANE];
  };
  auto load_step = [&](size_t begin, size_t end, unsigned step) -> StepCells {
    StepCells c;
#pragma unroll
    for (int k = 0; k < ZC_LANE; ++k) {
      const size_t i = begin + (size_t)step * ZC_STEP + ZC_LANE * (size_t)lane + k;
      c.v[k] = values[min(i, cells - 1)];
      c.label[k] = i < end ? labels[min(i, cells - 1)] : -1;
    }
    return c;
  };
  // the cells of this label among a step's, parked in raster order (then zeros up to a whole round of the chain): how many
  auto park = [&](const StepCells& c, double* where) -> unsigned {
    bool take[ZC_LANE];
    unsigned below = 0, n = 0;
#pragma unroll
    for (int k = 0; k < ZC_LANE; ++k) {
      take[k] = c.label[k] == l && cell_active<T>(c.v[k], c.label[k], n_labels, has_ndv, ndv, thresholds);
      const unsigned long long m = __ballot(take[k]);
      below += __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));  // taken in the lanes below
      n += (unsigned)__popcll(m);
    }
    // rank of cell (lane, k) in raster order: the cells of the lanes below, then this lane's earlier ones
    unsigned rank = below;
#pragma unroll
    for (int k = 0; k < ZC_LANE; ++k) {
      if (take[k]) where[rank++] = (double)c.v[k];
    }
    if (lane < 16) where[n + lane] = 0.0;  // (+0.0 changes nothing: the sum starts from +0.0 and can never become -0.0)
    return n;
  };
  unsigned literal_steps = 0;
  auto literally = [&](const StepCells& c) {
    ++literal_steps;
    const unsigned n = park(c, mine);
    if (n) acc = chain_add_parked(mine, acc, n);  // (wave-uniform)
  };
  // the label's chunks of a round of the table in order: (stretch << 8 | slot); a lane reads 64 stretches' bytes
  auto list_round = [&](unsigned round0) -> unsigned {
    unsigned n_list = 0;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned at = round0 + 64u * (unsigned)lane;
    u32x4 q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = u32x4{0u, 0u, 0u, 0u};
    if (at < row_bytes) {
      const u32x4* const row = reinterpret_cast<const u32x4*>(slot_of + (size_t)l * row_bytes + at);
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = row[i];
    }
    unsigned mine_n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned word = q[i][c];
        mine_n += (word & 0xffu ? 1u : 0u) + (word & 0xff00u ? 1u : 0u) + (word & 0xff0000u ? 1u : 0u) + (word & 0xff000000u ? 1u : 0u);
      }
    unsigned inclusive = mine_n;
#pragma unroll
    for (int offset = 1; offset < 64; offset <<= 1) {
      const unsigned other = __shfl_up(inclusive, offset);
      if (lane >= offset) inclusive += other;
    }
    n_list = (unsigned)__builtin_amdgcn_readlane((int)inclusive, 63);
    if (n_list == 0) return 0u;  // (wave-uniform)
    unsigned pos = inclusive - mine_n;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int byte = 0; byte < 4; ++byte) {
          const unsigned slot = (q[i][c] >> (8 * byte)) & 0xffu;
          if (slot) list[pos++] = ((at + 16u * i + 4u * c + byte) << 8) | slot;
        }
    return n_list;
  };
  // Round 6: the float32 answer without the order.  What the caller keeps of a sum is its rounding into the float32 `agg`
  // (aggregate.py:187-203), and bincount's sequential float64 sum s_n lies close to ANY float64 sum of the same cells: with
  // X_k the exact partial sums and M >= |X_k| for all k, s_k = (s_(k-1) + x_k)(1 + d_k), |d_k| <= u = 2^-53, gives
  // the errors e_k = s_k - X_k obey |e_k| <= (1 + u) |e_(k-1)| + u |X_k|, hence |e_n| <= (1 + u)^n u sum_k |X_k| <= 1.000001 u
  // sum_k |X_k| (n < 2^31).  The chunk records hold what bounds the |X_k| -- the extreme partial sums inside every chunk,
  // relative to the sum entering it: W = sum over the chunks of (cells) x (largest |partial sum|) -- and tree sums that differ
  // from the exact ones by at most tau: a rounding is at most u times the result of its addition, and nothing amplifies it on
  // the way, so tau <= u x (the sum of the |results| of all the additions that lead to S below).  Level by level: a lane's 4
  // additions and the 6 across the wave in zonal_chunk_kernel: every cell enters at most one result per level, 10 A in all
  // (A >= the sum of the |cells|: count x 2^(hi + 1) per chunk); the additions of a chunk's steps onto its record: one per
  // step that holds a cell, each result a partial sum relative to the chunk's start, at most r = max(-low, high); the chunk
  // totals (each at most r) through a lane's 4 additions and the 6 across the wave here: 10 levels again -- together V = sum
  // over the chunks of (min(cells, 64) + 10) r; and the running S, once per batch, at most M (the largest |partial sum|).
  // tau = 1.5 u (10 A + V + (batches + 8) M).  So s_n lies inside S -/+ E, E = 1.01 u
  // (W + 2 n tau) + tau -- plus 4 u (M + tau) for the roundings of S -/+ E themselves -- and when both ends of S -/+ E --
  // divided by n for the mean: a division and a conversion are monotone -- round to the same float32, that float32 is what
  // bincount's sum rounds to, whatever it rounded on the way.  Zero-mean noise under a few long labels, whose sum sheds and
  // takes up low bits a few hundred times (each such event a replay or a step of the block pipeline, one after the other:
  // three labels over 12.6 M cells 1.4 ms), passes with a margin of 10^-2 of a float32 unit in the last place; a sum that
  // does land next to a float32 tie -- sums of a few cells of 2^-24 beside 1.0 in the tests -- fails and is added in order below.
  if (lazy) {
    const double inf = __builtin_huge_val();
    double S = 0.0, M = 0.0, A = 0.0;
    unsigned long long n = 0;
    unsigned batches = 0;
    bool bad = false;
    double W = 0.0;  // sum over the chunks of (cells of the chunk) x (the largest |partial sum| inside it)
    double V = 0.0;  // ... of (steps of the chunk that hold its cells, at most, + 10) x (the largest |partial sum - the sum entering the chunk|)
    for (unsigned round0 = 0; round0 < row_bytes; round0 += ZC_LIST) {
      const unsigned n_list = list_round(round0);
      for (unsigned b0 = 0; b0 < n_list; b0 += 256u) {  // (wave-uniform)
        ChunkRec four[4];
        double run = 0.0, lane_abs = 0.0;
        unsigned lane_count = 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned i = b0 + 4u * (unsigned)lane + (unsigned)c;
          const unsigned e = i < n_list ? list[i] : 0u;
          ChunkRec r{};
          r.hi = SHRT_MIN;
          if (e) r = records[(size_t)(e >> 8) * ZC_SLOTS + (e & 0xffu) - 1];
          four[c] = r;
          run += r.total;
          lane_count += r.count;
          bad = bad || r.bad != 0;
          if (r.hi != SHRT_MIN) lane_abs += ldexp((double)r.count, (int)r.hi + 1);  // (every |cell| < 2^(hi + 1))
        }
        double entering = S + (wave_scan_f64(run) - run), lane_weight = 0.0, lane_chain = 0.0, lane_reach = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double reach = fmax(fabs(entering + (double)four[c].low), fabs(entering + (double)four[c].high));
          lane_weight += (double)four[c].count * reach;
          lane_chain += (double)(min(four[c].count, (unsigned)ZC_MAX_STEPS) + 10u) * fmax(-(double)four[c].low, (double)four[c].high);
          lane_reach = fmax(lane_reach, reach);
          entering += four[c].total;
        }
        M = fmax(M, wave_max_f64(lane_reach));
        W += wave_sum_exact(lane_weight);
        V += wave_sum_exact(lane_chain);
        S += wave_sum_exact(run);
        A += wave_sum_exact(lane_abs);
        n += wave_sum_u32(lane_count);
        ++batches;
      }
    }
    if (n != 0 && __ballot(bad) == 0 && A < 1.0e300 && W < 1.0e300 && V < 1.0e300) {  // (wave-uniform; false for NaN)
      constexpr double u = 1.1102230246251565e-16;
      const double tau = 1.5 * u * (10.0 * A + V + (double)(batches + 8u) * M);
      // (+ 1e-300 wherever a cell is not zero: no claim rests on what an adder or an atomic unit does with float64 subnormals)
      const double E = 1.01 * u * (W + 4.0 * M + (2.0 * (double)n + 4.0) * tau) + tau + (A > 0.0 ? 1.0e-300 : 0.0);
      double below = S - E, above = S + E;
      if (statistic == Z_MEAN) {
        below /= (double)n;
        above /= (double)n;
      }
      if (lazy == 2 && lane == 0)
        printf("[gmhip] float32 certificate, label %d: n %llu S %.17g E %.3g (W %.3g V %.3g M %.3g A %.3g tau %.3g, %u batches) -> %.9g .. %.9g\n", l,
               n, S, E, W, V, M, A, tau, batches, (double)(float)below, (double)(float)above);
      const float lowest = (float)below, highest = (float)above;
      if (__float_as_uint(lowest) == __float_as_uint(highest) && fabsf(lowest) < inf) {
        if (lane == 0) {
          counts[l] = (long long)n;
          out[l] = lowest;
        }
        return;
      }
    }
  }
  for (unsigned round0 = 0; round0 < row_bytes; round0 += ZC_LIST) {
    const unsigned n_list = list_round(round0);
    if (n_list == 0) continue;  // (wave-uniform)
    // (a batch's records are requested while the batch before it is worked on: one memory round trip per label, not per batch)
    auto fetch_batch = [&](unsigned b0, unsigned* entry_out) -> ChunkRec {
      const unsigned entry = b0 + (unsigned)lane < n_list ? list[b0 + lane] : 0u;
      ChunkRec rec{};
      rec.hi = SHRT_MIN;
      rec.lo = SHRT_MAX;
      if (entry) rec = records[(size_t)(entry >> 8) * ZC_SLOTS + (entry & 0xffu) - 1];
      *entry_out = entry;
      return rec;
    };
    unsigned b0 = 0;
    // Labels of many chunks (a few labels covering the raster) first try 256 chunks per test, a lane holding FOUR consecutive
    // ones: the lane composes its four in order -- running total, extreme partial sums -- and the wave's reductions, which are
    // most of a test's ~3 000 cycles, run once per 256 chunks.  (One label of positive cells over 12.6 M cells, 4096 chunks: the
    // wave of this kernel 90 -> 25 us.)  The first such batch that does not go as one leaves the rest to the batches of 64 below.
    if (n_list >= 128u) {
      auto fetch_four = [&](unsigned at, ChunkRec (&four)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned i = at + 4u * (unsigned)lane + (unsigned)c;
          const unsigned e = i < n_list ? list[i] : 0u;
          ChunkRec r{};
          r.hi = SHRT_MIN;
          r.lo = SHRT_MAX;
          if (e) r = records[(size_t)(e >> 8) * ZC_SLOTS + (e & 0xffu) - 1];
          four[c] = r;
        }
      };
      ChunkRec four_next[4];
      fetch_four(0u, four_next);
      for (; b0 < n_list; b0 += 256u) {
        ChunkRec four[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) four[c] = four_next[c];
        if (b0 + 256u < n_list) fetch_four(b0 + 256u, four_next);  // (wave-uniform)
        const double inf = __builtin_huge_val();
        double run = 0.0, lane_below = inf, lane_above = -inf;
        int lane_low = INT_MAX, lane_top = INT_MIN;
        unsigned lane_count = 0u;
        bool lane_bad = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          lane_below = fmin(lane_below, run + (double)four[c].low);
          lane_above = fmax(lane_above, run + (double)four[c].high);
          run += four[c].total;
          if (four[c].lo != SHRT_MAX) lane_low = min(lane_low, (int)four[c].lo);
          if (four[c].hi != SHRT_MIN) lane_top = max(lane_top, (int)four[c].hi);
          lane_count += four[c].count;
          lane_bad = lane_bad || four[c].bad != 0;
        }
        const int low = wave_min(lane_low), top = wave_max(lane_top);
        const double before = wave_scan_f64(run) - run;
        const double below = wave_min_f64(before + lane_below), above = wave_max_f64(before + lane_above);
        if (__ballot(lane_bad) != 0 || !additions_exact(acc, top, low, 0x7fffffffu, below, above)) break;  // (wave-uniform)
        acc += wave_sum_exact(run);
        n_total += wave_sum_u32(lane_count);
      }
    }
    unsigned entry_next = 0u;
    ChunkRec rec_next{};
    if (b0 < n_list) rec_next = fetch_batch(b0, &entry_next);  // (wave-uniform)
    for (; b0 < n_list; b0 += 64) {
      const unsigned entry = entry_next;
      const ChunkRec rec = rec_next;
      if (b0 + 64 < n_list) rec_next = fetch_batch(b0 + 64, &entry_next);  // (wave-uniform)
      const unsigned in_batch = min(64u, n_list - b0);
      {
        // all the batch's chunks at once: their cells are multiples of 2^low, the partial sums stay inside [below, above]
        const int low = wave_min(rec.lo == SHRT_MAX ? INT_MAX : (int)rec.lo);
        const int top = wave_max(rec.hi == SHRT_MIN ? INT_MIN : (int)rec.hi);
        const double before = wave_scan_f64(rec.total) - rec.total;
        const double below = wave_min_f64(before + (double)rec.low), above = wave_max_f64(before + (double)rec.high);
        // (the exponent rule would need the batch's cell count: pass a count that rules it out unless everything is tiny)
        if (__ballot(rec.bad != 0) == 0 && additions_exact(acc, top, low, 0x7fffffffu, below, above)) {  // (wave-uniform)
          acc += wave_sum_exact(rec.total);
          n_total += wave_sum_u32(rec.count);
          continue;
        }
      }
      // (a chunk that has to be replayed is often followed by another -- a running sum that has taken up a very fine cell fails
      //  the test until it has grown past it: the cells of the NEXT chunk are requested before this one's chain of additions)
      bool have_ahead = false;
      StepCells ahead_one{}, ahead_two{};
      for (unsigned j = 0; j < in_batch; ++j) {
        if ((j & 15u) == 0) {
          // the batch did not go as one: its chunks sixteen at a time, the same test (a label that is being replayed fails
          // in a few places, not everywhere: most sixteens still go in one addition)
          const bool member = (unsigned)lane >= j && (unsigned)lane < j + 16u && (unsigned)lane < in_batch;
          const int low = wave_min(member && rec.lo != SHRT_MAX ? (int)rec.lo : INT_MAX);
          const int top = wave_max(member && rec.hi != SHRT_MIN ? (int)rec.hi : INT_MIN);
          const double mine_total = member ? rec.total : 0.0;
          const double before = wave_scan_f64(mine_total) - mine_total;
          const double inf = __builtin_huge_val();
          const double below = wave_min_f64(member ? before + (double)rec.low : inf);
          const double above = wave_max_f64(member ? before + (double)rec.high : -inf);
          if (__ballot(member && rec.bad != 0) == 0 && additions_exact(acc, top, low, 0x7fffffffu, below, above)) {  // (wave-uniform)
            acc += wave_sum_exact(mine_total);
            n_total += wave_sum_u32(member ? rec.count : 0u);
            have_ahead = false;
            j += 15u;
            continue;
          }
        }
        const double total = readlane_f64(rec.total, (int)j);
        const double below = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(rec.low), (int)j));
        const double above = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(rec.high), (int)j));
        const unsigned count = (unsigned)__builtin_amdgcn_readlane((int)rec.count, (int)j);
        const unsigned packed = (unsigned)__builtin_amdgcn_readlane((int)((unsigned)(unsigned short)rec.hi | ((unsigned)(unsigned short)rec.lo << 16)), (int)j);
        const unsigned range = (unsigned)__builtin_amdgcn_readlane((int)((unsigned)rec.first | ((unsigned)rec.last << 8) | ((unsigned)rec.bad << 16)), (int)j);
        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)entry, (int)j) >> 8;
        const int hi16 = (int)(short)(packed & 0xffffu), lo16 = (int)(short)(packed >> 16);
        const int hi = hi16 == SHRT_MIN ? INT_MIN : hi16, lo = lo16 == SHRT_MAX ? INT_MAX : lo16;
        const unsigned first = range & 0xffu, last = (range >> 8) & 0xffu, bad = range >> 16;
        n_total += count;
        if (!bad && additions_exact(acc, hi, lo, count, below, above)) {
          acc += total;
          have_ahead = false;
          continue;
        }
        if (++replays > patience || (patience != ~0u && literal_steps > ZC_LITERAL_STEPS) ||
            (n_total >= long_from && last - first >= 2 && ++walks > ZC_LONG_WALKS)) {  // (wave-uniform)
          give_up();
          return;
        }
        const size_t begin = (size_t)w * cells_per_wave, end = min(begin + cells_per_wave, cells);
        if (last - first < 2) {  // a chunk of one or two steps: its cells, one after the other
          StepCells one, two;
          if (have_ahead) {
            one = ahead_one;
            two = ahead_two;
          } else {
            one = load_step(begin, end, first);
            two = load_step(begin, end, last);
          }
          const unsigned n_one = park(one, mine), n_two = last != first ? park(two, mine_b) : 0u;
          // (this chunk's cells are in LDS: the next chunk's are requested now and arrive during the chains below)
          have_ahead = false;
          if (j + 1 < in_batch) {
            const unsigned next_range = (unsigned)__builtin_amdgcn_readlane((int)((unsigned)rec.first | ((unsigned)rec.last << 8)), (int)j + 1);
            const unsigned next_first = next_range & 0xffu, next_last = next_range >> 8;
            if (next_last - next_first < 2) {
              const unsigned next_w = (unsigned)__builtin_amdgcn_readlane((int)entry, (int)j + 1) >> 8;
              const size_t next_begin = (size_t)next_w * cells_per_wave, next_end = min(next_begin + cells_per_wave, cells);
              ahead_one = load_step(next_begin, next_end, next_first);
              ahead_two = load_step(next_begin, next_end, next_last);
              have_ahead = true;
            }
          }
          if (n_one) acc = chain_add_parked(mine, acc, n_one);
          if (n_two) acc = chain_add_parked(mine_b, acc, n_two);
          continue;
        }
        have_ahead = false;
        // a longer chunk step by step, from the notes (a lane reads the note of a step)
        StepRec note{0.0, 0.0f, 0.0f, -1, SHRT_MIN, SHRT_MAX};
        if (first + (unsigned)lane <= last) note = step_records[(size_t)w * steps_per_wave + first + lane];
        unsigned long long wanted = __ballot(note.label == l || note.label == -2);
        while (wanted) {
          const int t = __ffsll((long long)wanted) - 1;
          wanted &= wanted - 1;
          const int label = __builtin_amdgcn_readlane(note.label, t);
          if (label == l) {
            const double step_total = readlane_f64(note.total, t);
            const double step_below = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(note.low), t));
            const double step_above = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(note.high), t));
            const unsigned span = (unsigned)__builtin_amdgcn_readlane((int)((unsigned)(unsigned short)note.hi | ((unsigned)(unsigned short)note.lo << 16)), t);
            const int top16 = (int)(short)(span & 0xffffu), low16 = (int)(short)(span >> 16);
            if (additions_exact(acc, top16 == SHRT_MIN ? INT_MIN : top16, low16 == SHRT_MAX ? INT_MAX : low16, (unsigned)ZC_STEP, step_below, step_above)) {
              acc += step_total;
              continue;
            }
          }
          literally(load_step(begin, end, first + (unsigned)t));
        }
      }
    }
  }
  if (lane == 0) {
    counts[l] = (long long)n_total;
    const float nan = std::numeric_limits<float>::quiet_NaN();
    // ndimage.mean: sum / count in float64; the assignment into the float32 `agg` rounds once
    out[l] = n_total == 0 ? nan : (statistic == Z_MEAN ? (float)(acc / (double)n_total) : (float)acc);
  }
}

template <typename T>
T scalar_from_bits(uint64_t bits) {
  gm_scalar s;
  s.u64 = bits;
  if constexpr (std::is_floating_point<T>::value) return (T)s.f64;
  else if constexpr (std::is_same<T, unsigned long long>::value) return (T)s.u64;
  else return (T)s.i64;
}

// The contiguous stretch of cells a wave of the direct kernels walks: at least 2048 cells, at most 16 waves per CU (more,
// shorter stretches keep more loads in flight -- 1024-label sums 428 -> 377 us, integer sums 201 -> 150 us -- but every
// wave ends with atomics on its last label's counters, which one label covering the raster serialises: 265 -> 302 us).
static int64_t stretch() { static const int64_t v = getenv("GM_ZONAL_STRETCH") ? atoll(getenv("GM_ZONAL_STRETCH")) : 2048; return v; }
static int64_t wave_cap() { static const int64_t v = getenv("GM_ZONAL_WAVECAP") ? atoll(getenv("GM_ZONAL_WAVECAP")) : 16; return v; }

// Medians / percentiles under MANY labels (round 6).  Beyond the counting partition's 2047 labels the two sorts of lexsort were
// all there was: 12.6 M cells of float32 noise under 49 152 polygons of 16 x 16 cells 1.4 ms, five radix passes, three of them
// over digits of noise.  What such rasters have instead is SHORT labels: the cells are sorted by label alone (the label's bits:
// two passes over keys that come in runs), and a workgroup then sorts every label's segment by value in LDS (a bitonic network
// over the next power of two, padded with the largest key) and writes it back -- what zonal_pick_kernel reads its ranks from.
// A label of more than ZS_MAX cells sets `too_long` and is left as it is: the caller falls back to the two sorts.
constexpr unsigned ZS_MAX = 4096, ZS_SMALL = 1024;
// THREADS lanes per label; labels of (FROM, CAP] cells (the launch for the short labels is one wave per label: its barriers cost
// nothing; the second launch finds the few longer ones)
template <typename K, int THREADS, unsigned FROM, unsigned CAP>
__global__ __launch_bounds__(THREADS) void zonal_segment_sort_kernel(K* __restrict__ payload, const unsigned* __restrict__ first,
                                                                     int n_labels, unsigned* __restrict__ too_long) {
  __shared__ K seg[CAP];
  const int l = blockIdx.x;
  if (l >= n_labels) return;
  const unsigned a = first[l], n = first[l + 1] - a;
  if (n < 2 || n <= FROM) return;
  if (n > CAP) {
    if (CAP == ZS_MAX && threadIdx.x == 0) *too_long = 1u;
    return;
  }
  unsigned P = 2;
  while (P < n) P <<= 1;
  for (unsigned i = threadIdx.x; i < P; i += THREADS) seg[i] = i < n ? payload[a + i] : ~(K)0;
  __syncthreads();
  for (unsigned k = 2; k <= P; k <<= 1) {
    for (unsigned j = k >> 1; j > 0; j >>= 1) {
      for (unsigned t = threadIdx.x; t < P / 2; t += THREADS) {
        // the t-th pair of this stage: i has bit j clear
        const unsigned i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), partner = i | j;
        const K x = seg[i], y = seg[partner];
        const bool ascending = (i & k) == 0;
        if ((x > y) == ascending) {
          seg[i] = y;
          seg[partner] = x;
        }
      }
      __syncthreads();
    }
  }
  for (unsigned i = threadIdx.x; i < n; i += THREADS) payload[a + i] = seg[i];
}

// which attempt of the chunk records answered a label raster last time (see zonal_run)
struct ChunkHint {
  const void* labels;
  int n_labels;
  int64_t cells;
  int attempt;
};
static std::mutex g_chunk_hint_mu;
static ChunkHint g_chunk_hints[8];
static unsigned g_chunk_hint_turn = 0;
static int chunk_hint_get(const void* labels, int n_labels, int64_t cells) {
  std::lock_guard<std::mutex> lk(g_chunk_hint_mu);
  for (const ChunkHint& h : g_chunk_hints)
    if (h.labels == labels && h.n_labels == n_labels && h.cells == cells) return h.attempt;
  return 0;
}
static void chunk_hint_set(const void* labels, int n_labels, int64_t cells, int attempt) {
  std::lock_guard<std::mutex> lk(g_chunk_hint_mu);
  for (ChunkHint& h : g_chunk_hints)
    if (h.labels == labels && h.n_labels == n_labels && h.cells == cells) {
      h.attempt = attempt;
      return;
    }
  g_chunk_hints[g_chunk_hint_turn++ % 8] = ChunkHint{labels, n_labels, cells, attempt};
}

struct Scratch {  // device buffers of one call, returned to the pool on every exit path
  void* p[32] = {nullptr};
  int n = 0;
  int take(void** out, size_t bytes) {
    if (n == 32) return gm::fail(GM_ERR_INVALID, "gm_zonal_statistic: scratch table full");
    const int rc = gm_malloc(out, bytes ? bytes : 1);
    if (rc == GM_OK) p[n++] = *out;
    return rc;
  }
  ~Scratch() {
    for (int i = 0; i < n; ++i) gm_free(p[i]);
  }
};

template <typename T>
int zonal_run(const void* values_v, int64_t frames, int64_t cells, int has_ndv, uint64_t ndv_bits, const int* labels,
              int n_labels, const float* thresholds, int statistic, double qval, float* out, long long* counts,
              unsigned char* present, hipStream_t s) {
  typedef typename Ordered<T>::Key Key;
  const T* values = (const T*)values_v;
  const T ndv = scalar_from_bits<T>(ndv_bits);
  Scratch scratch;
  if (statistic == Z_COUNT || statistic == Z_MIN || statistic == Z_MAX) {
    unsigned long long* tally;
    Key* extreme;
    int rc;
    if ((rc = scratch.take((void**)&tally, sizeof(unsigned long long) * n_labels))) return rc;
    if ((rc = scratch.take((void**)&extreme, sizeof(Key) * n_labels))) return rc;
    // every wave walks one contiguous stretch of cells (label runs continue from step to step)
    const int64_t waves = std::max<int64_t>(1, std::min<int64_t>((cells + stretch() - 1) / stretch(), (int64_t)gm::current_cu_count() * wave_cap()));
    const size_t cells_per_wave = (size_t)(((cells + waves - 1) / waves + 63) / 64 * 64);
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    // label counts whose table fits a workgroup's LDS take zonal_tables_kernel (GM_ZONAL_TABLES=0: never -- A/B runs, and the
    // tests compare the two; =2: the old kernel only for rasters the new one cannot take)
    static const int tables_mode = getenv("GM_ZONAL_TABLES") ? atoi(getenv("GM_ZONAL_TABLES")) : 1;
    const size_t t_lds = (statistic == Z_COUNT ? 0 : sizeof(Key) * (size_t)((n_labels + 1) & ~1)) + sizeof(unsigned) * (size_t)n_labels;
    const bool tables = tables_mode != 0 && t_lds <= ZT_LDS_BYTES;
    // (two workgroups of 16 waves per compute unit at most; a wave walks at least 4 steps)
    const int64_t t_waves_wanted = std::max<int64_t>(1, std::min<int64_t>((cells + 255) / 256, (int64_t)gm::current_cu_count() * 2 * (ZT_THREADS / 64)));
    const unsigned t_blocks = (unsigned)((t_waves_wanted + ZT_THREADS / 64 - 1) / (ZT_THREADS / 64));
    const size_t t_cells_per_wave = (size_t)(((cells + (int64_t)t_blocks * (ZT_THREADS / 64) - 1) / ((int64_t)t_blocks * (ZT_THREADS / 64)) + 63) / 64 * 64);
    for (int64_t f = 0; f < frames; ++f) {
      unsigned char* seen = f == 0 ? present : (unsigned char*)nullptr;
      hipLaunchKernelGGL(zonal_direct_init_kernel<Key>, dim3((n_labels + 255) / 256), dim3(256), 0, s, tally, extreme,
                         statistic == Z_MIN ? ~(Key)0 : (Key)0, seen, n_labels);
      float* out_f = out + (size_t)f * n_labels;
      long long* counts_f = counts + (size_t)f * n_labels;
#define GM_ZD(STAT)                                                                                                        \
  do {                                                                                                                     \
    if (tables)                                                                                                            \
      hipLaunchKernelGGL((zonal_tables_kernel<T, STAT>), dim3(t_blocks), dim3(ZT_THREADS), t_lds, s, values + (size_t)f * cells, \
                         labels, (size_t)cells, t_cells_per_wave, n_labels, has_ndv, ndv, thresholds, tally, extreme, seen); \
    else                                                                                                                   \
      hipLaunchKernelGGL((zonal_direct_kernel<T, STAT>), dim3(blocks), dim3(256), 0, s, values + (size_t)f * cells, labels, \
                         (size_t)cells, cells_per_wave, n_labels, has_ndv, ndv, thresholds, tally, extreme, seen);         \
    hipLaunchKernelGGL((zonal_direct_result_kernel<T, STAT>), dim3((n_labels + 255) / 256), dim3(256), 0, s, tally, extreme, \
                       n_labels, out_f, counts_f);                                                                        \
  } while (0)
      if (statistic == Z_COUNT) GM_ZD(Z_COUNT);
      else if (statistic == Z_MIN) GM_ZD(Z_MIN);
      else GM_ZD(Z_MAX);
#undef GM_ZD
      gm::route_hit(GM_ROUTE_ZONAL, tables ? GM_RZ_LDS_TABLES : GM_RZ_DIRECT);
      GM_HIP(hipGetLastError());
    }
    return GM_OK;
  }
  const bool by_value = statistic == Z_MEDIAN || statistic == Z_PERCENTILE;
  // every wave of the direct kernels walks one contiguous stretch of cells
  const int64_t waves = std::max<int64_t>(1, std::min<int64_t>((cells + stretch() - 1) / stretch(), (int64_t)gm::current_cu_count() * wave_cap()));
  const size_t cells_per_wave = (size_t)(((cells + waves - 1) / waves + 63) / 64 * 64);
  const unsigned direct_blocks = (unsigned)((waves + 3) / 4);
  int rc;
  ExactRange* range = nullptr;
  double* sums = nullptr;
  unsigned char* inexact = nullptr;  // per label: its sum needs the cells in order
  ExactVerdict* verdict_dev = nullptr;
  unsigned* cursor = nullptr;
  if (!by_value && !getenv("GM_ZONAL_SORTED")) {
    if ((rc = scratch.take((void**)&range, sizeof(ExactRange) * n_labels))) return rc;
    if ((rc = scratch.take((void**)&sums, sizeof(double) * n_labels))) return rc;
    if ((rc = scratch.take((void**)&inexact, (size_t)n_labels))) return rc;
    if ((rc = scratch.take((void**)&verdict_dev, sizeof(ExactVerdict)))) return rc;
    if ((rc = scratch.take((void**)&cursor, sizeof(unsigned)))) return rc;
  }
  unsigned *keys_a = nullptr, *keys_b = nullptr, *first = nullptr;
  double* mags = nullptr;  // (the sums of the magnitudes beside the parallel sums: the float32 certificate)
  LazyRow* lazy_rows = nullptr;
  SelectState* raw_state = nullptr;  // (the radix select off the raster: zonal_select_raw_pass_kernel)
  unsigned *raw_hist = nullptr, *raw_first = nullptr;
  unsigned long long* raw_next = nullptr;
  unsigned* few_index = nullptr;  // (the labels the parallel sums' certificate turns away, added in order: zonal_few_*)
  double* few_values = nullptr;
  unsigned* few_slots = nullptr;
  double *blk_sum = nullptr, *blk_pred = nullptr, *blk_abs = nullptr;  // (the block pipeline of the long labels)
  int* blk_gexp = nullptr;
  GroupExact* grp_exact = nullptr;
  int* blk_label = nullptr;
  BlockInfo* blk_info = nullptr;
  GroupInfo* grp_info = nullptr;
  Key *pay_a = nullptr, *pay_b = nullptr;
  void* temp = nullptr;
  size_t temp_bytes = 0;
  unsigned label_bits = 1;
  while ((1ull << label_bits) <= (unsigned long long)n_labels) ++label_bits;
  // (GM_ZONAL_PART=0: the general radix sort for every label count -- A/B runs, and the tests compare the two)
  static const bool partition_allowed = !(getenv("GM_ZONAL_PART") && atoi(getenv("GM_ZONAL_PART")) == 0);
  // (sorts that start from cells in value order -- medians, percentiles -- see the labels as noise, where the partition's
  // per-lane scatter loses to the radix sort's exchange through LDS: 985 against 748 us; GM_ZONAL_PART=1 forces it, for the tests)
  // medians / percentiles: partition by label, then a radix select per label (GM_ZONAL_SELECT=0: the two sorts of lexsort)
  static const bool select_allowed = !(getenv("GM_ZONAL_SELECT") && atoi(getenv("GM_ZONAL_SELECT")) == 0);
  const bool select_path = by_value && select_allowed && partition_allowed && n_labels + 1 <= ZP_BINS;
  const bool partition = partition_allowed && n_labels + 1 <= ZP_BINS && (!by_value || select_path || getenv("GM_ZONAL_PART"));
  // medians / percentiles beyond the partition's label count: sort by label, then every label's segment by value in LDS
  // (zonal_segment_sort_kernel; GM_ZONAL_SEGSORT=0: the two sorts of lexsort -- A/B runs, and the tests compare the two)
  static const bool segsort_allowed = !(getenv("GM_ZONAL_SEGSORT") && atoi(getenv("GM_ZONAL_SEGSORT")) == 0);
  const bool segsort_path = by_value && !select_path && segsort_allowed && n_labels + 1 > ZP_BINS;
  SelectState* sel_state = nullptr;
  unsigned* sel_hist = nullptr;
  unsigned long long* sel_next = nullptr;
  const size_t part_bins = (size_t)n_labels + 1, part_tiles = ((size_t)cells + ZP_TILE - 1) / ZP_TILE;
  PartitionBuffers part;
  struct {
    ChunkRec* records = nullptr;
    unsigned char* slot_of = nullptr;
    StepRec* steps = nullptr;
    ChunkVerdict* verdict = nullptr;
    // what each buffer holds room for: the three do not grow together (attempt 1 has four times the stretches of attempt 0
    // but a quarter of the steps per stretch: a later frame's attempt 0 can need MORE step records than attempt 1 took)
    size_t records_for = 0, table_for = 0, steps_for = 0;
  } chunk;
  const unsigned blocks = (unsigned)std::min<int64_t>((cells + 255) / 256, (int64_t)gm::current_cu_count() * 32);
  for (int64_t f = 0; f < frames; ++f) {
    const T* frame = values + (size_t)f * cells;
    float* out_f = out + (size_t)f * n_labels;
    long long* counts_f = counts + (size_t)f * n_labels;
    unsigned char* seen = f == 0 ? present : (unsigned char*)nullptr;
    // the keys / payloads the first pass writes are taken up front; the sort's double buffers and its temporary
    // storage -- 8 to 12 more bytes per cell plus the sort's scratch, gigabytes for a 16384^2 frame -- only once a
    // verdict asks for the full sort: sums that the two-pass parallel path answers never allocate them
    auto key_buffers = [&]() -> int {
      if (keys_a) return GM_OK;
      int rc2;
      if ((rc2 = scratch.take((void**)&keys_a, sizeof(unsigned) * cells))) return rc2;
      if ((rc2 = scratch.take((void**)&pay_a, sizeof(Key) * cells))) return rc2;
      return scratch.take((void**)&first, sizeof(unsigned) * ((size_t)n_labels + 2));
    };
    auto sort_buffers = [&]() -> int {  // (what this frame's sort needs and earlier frames have not taken yet)
      int rc2;
      if (!keys_b) {
        if ((rc2 = scratch.take((void**)&keys_b, sizeof(unsigned) * cells))) return rc2;
        if ((rc2 = scratch.take((void**)&pay_b, sizeof(Key) * cells))) return rc2;
      }
      size_t temp_value = 0, temp_label = 0;
      if (by_value && !select_path) temp_value = radix_temp_bytes((size_t)cells);
      if (select_path && !sel_state) {
        if ((rc2 = scratch.take((void**)&sel_state, sizeof(SelectState) * n_labels))) return rc2;
        if ((rc2 = scratch.take((void**)&sel_hist, sizeof(unsigned) * 256 * (size_t)n_labels))) return rc2;
        if ((rc2 = scratch.take((void**)&sel_next, sizeof(unsigned long long) * n_labels))) return rc2;
      }
      if (!partition) {
        temp_label = radix_temp_bytes((size_t)cells);
      } else if (!part.offsets) {
        if (!part.counts && (rc2 = scratch.take((void**)&part.counts, sizeof(unsigned) * part_tiles * part_bins))) return rc2;
        if ((rc2 = scratch.take((void**)&part.offsets, sizeof(unsigned) * part_tiles * part_bins))) return rc2;
        if ((rc2 = scratch.take((void**)&part.totals, sizeof(unsigned) * part_bins))) return rc2;
      }
      const size_t need = temp_value > temp_label ? temp_value : temp_label;
      if (need > temp_bytes) {  // (at most twice per call: the earlier, smaller one goes back with the rest of the scratch)
        if ((rc2 = scratch.take(&temp, need))) return rc2;
        temp_bytes = need;
      }
      return GM_OK;
    };
    // the stable sort by label of (keys, payload): one counting pass for up to ZP_BINS - 1 labels (which also yields
    // first[]), the general radix sort + the search for the bounds otherwise
    auto sort_by_label = [&](const unsigned* keys_in, const Key* pay_in, unsigned* keys_out, Key* pay_out) -> int {
      gm::route_hit(GM_ROUTE_ZONAL, partition ? GM_RZ_SORT_PARTITION : GM_RZ_SORT_RADIX);
      if (partition) {
        static const int nw = getenv("GM_ZONAL_PART_WAVES") ? atoi(getenv("GM_ZONAL_PART_WAVES")) : 8;
        const size_t lds = sizeof(unsigned) * part_bins;
#define GM_ZP(NW)                                                                                                              \
  do {                                                                                                                         \
    if (NW * lds > 65536)                                                                                                      \
      GM_HIP(hipFuncSetAttribute((const void*)zonal_part_scatter_kernel<Key, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                 (int)(NW * lds)));                                                                            \
    if (!part.counted)                                                                                                         \
      hipLaunchKernelGGL(zonal_part_count_kernel<NW>, dim3((unsigned)part_tiles), dim3(64 * NW), 0, s, keys_in, (unsigned)cells, \
                         (int)part_bins, part.counts);                                                                         \
    hipLaunchKernelGGL(zonal_part_scan_kernel, dim3((unsigned)((part_bins * 64 + 255) / 256)), dim3(256), 0, s, part.counts,   \
                       (unsigned)part_tiles, (int)part_bins, part.offsets, part.totals);                                       \
    hipLaunchKernelGGL(zonal_part_first_kernel, dim3(1), dim3(256), 0, s, part.totals, (int)part_bins, first);                 \
    hipLaunchKernelGGL((zonal_part_scatter_kernel<Key, NW>), dim3((unsigned)part_tiles), dim3(64 * NW), NW * lds, s, keys_in,   \
                       pay_in, (unsigned)cells, (int)part_bins, (int)label_bits, part.offsets, first, keys_out, pay_out);      \
  } while (0)
        if (nw == 4) GM_ZP(4);
        else if (nw == 16) GM_ZP(16);
        else GM_ZP(8);
#undef GM_ZP
      } else {
        // (the sort is free to use its input as a buffer: nobody reads the unsorted pairs again)
        if (int rc3 = radix_sort_pairs<unsigned, Key>(temp, const_cast<unsigned*>(keys_in), keys_out, const_cast<Key*>(pay_in), pay_out,
                                                      (size_t)cells, (int)label_bits, s))
          return rc3;
        hipLaunchKernelGGL(zonal_bounds_kernel, dim3((n_labels + 1 + 255) / 256), dim3(256), 0, s, keys_out, (unsigned)cells, n_labels, first);
      }
      GM_HIP(hipGetLastError());
      return GM_OK;
    };
    // The float32 certificate over workgroup LDS tables (zonal_tables_lazy_kernel + zonal_tables_lazy_label_kernel): 1 = every label
    // answered, 0 = not (a label failed, too many labels for a table), < 0 = -error.  GM_ZONAL_LAZY_TABLES=0: never (A/B runs, tests)
    static const bool lazy_tables_allowed = !(getenv("GM_ZONAL_LAZY_TABLES") && atoi(getenv("GM_ZONAL_LAZY_TABLES")) == 0);
    static const int lazy_mode = getenv("GM_ZONAL_LAZY") ? atoi(getenv("GM_ZONAL_LAZY")) : 1;
    bool lazy_tried = false;
    auto lazy_tables = [&]() -> int {
      lazy_tried = true;
      if (!lazy_mode || !lazy_tables_allowed || (size_t)n_labels * sizeof(LazyRow) > ZT_LDS_BYTES) return 0;
      HostFlag* const flag = host_flag();
      if (!flag) return 0;
      // workgroups: as many as a table of 16 MB holds rows for (shorter ranges per workgroup: tighter bounds) -- up to 8192 of one
      // wave for a few labels, up to 2048 of four waves, of sixteen when the rows of many labels limit their number
      const int64_t fit = (int64_t)((16u << 20) / ((size_t)n_labels * sizeof(LazyRow)));
      static const int forced_waves = getenv("GM_ZONAL_LAZY_WAVES") ? atoi(getenv("GM_ZONAL_LAZY_WAVES")) : 0;  // (1, 4, 16: A/B runs)
      const int waves_per_group = forced_waves ? forced_waves : fit >= 8192 ? 1 : fit >= 1024 ? 4 : 16;
      int64_t groups = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(fit, waves_per_group == 1 ? 8192 : 2048),
                                                                (cells + 64 * waves_per_group - 1) / (64 * waves_per_group)));
      const size_t lazy_cells_per_wave = (size_t)((cells + groups * waves_per_group - 1) / (groups * waves_per_group) + 63) / 64 * 64;
      groups = (int64_t)(((size_t)cells + lazy_cells_per_wave * waves_per_group - 1) / (lazy_cells_per_wave * waves_per_group));
      if (!lazy_rows) {
        if (int rc2 = scratch.take((void**)&lazy_rows, sizeof(LazyRow) * (size_t)groups * n_labels)) return -rc2;
      }
      flag->word[5] = 0u;
      const size_t lds = (size_t)n_labels * sizeof(LazyRow);
#define GM_ZL(THREADS)                                                                                                              \
  hipLaunchKernelGGL((zonal_tables_lazy_kernel<T, THREADS>), dim3((unsigned)groups), dim3(THREADS), lds, s, frame, labels, (size_t)cells, \
                     lazy_cells_per_wave, n_labels, has_ndv, ndv, thresholds, lazy_rows, seen)
      if (waves_per_group == 1) GM_ZL(64);
      else if (waves_per_group == 4) GM_ZL(256);
      else GM_ZL(1024);
#undef GM_ZL
      if (groups > 2048)
        hipLaunchKernelGGL(zonal_tables_lazy_label_kernel<16>, dim3((unsigned)(((size_t)n_labels * 64 + 255) / 256)), dim3(256), 0, s, lazy_rows,
                           (unsigned)groups, n_labels, (double)(lazy_cells_per_wave / 64), statistic, lazy_mode == 2 ? 1 : 0, out_f, counts_f,
                           flag->word + 5);
      else
        hipLaunchKernelGGL(zonal_tables_lazy_label_kernel<4>, dim3((unsigned)(((size_t)n_labels * 64 + 255) / 256)), dim3(256), 0, s, lazy_rows,
                           (unsigned)groups, n_labels, (double)(lazy_cells_per_wave / 64), statistic, lazy_mode == 2 ? 1 : 0, out_f, counts_f,
                           flag->word + 5);
      hipError_t e = hipGetLastError();
      host_copy_queue(frames, s);
      if (e == hipSuccess) e = hipEventRecord(flag->event, s);
      if (e == hipSuccess) e = hipEventSynchronize(flag->event);
      if (e != hipSuccess) return -gm::hip_fail(e, "zonal_tables_lazy_kernel");
      const bool answered = ((volatile unsigned*)flag->word)[5] == 0u;
      host_copy_settle(answered);
      if (!answered) return 0;
      gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_LDS_LAZY);
      return 1;
    };
    // That route comes BEFORE the chunk records where a wave is a workgroup of its own (up to 85 labels: rows for 8192 stretches,
    // bounds nearly as tight as the chunk records') -- its pass over the raster keeps a step of one label in registers and is half
    // the chunk kernel's: one label of 12.6 M cells 126 -> 77 us a call.  A label that fails there takes the chunk records next.
    // (GM_ZONAL_LAZY_FIRST=0: never first; =1: first for every label count -- A/B runs)
    static const int lazy_first_mode = getenv("GM_ZONAL_LAZY_FIRST") ? atoi(getenv("GM_ZONAL_LAZY_FIRST")) : -1;
    // (... and where the chunk records' first attempt overflowed last time -- more labels to a stretch than records: stripes, small
    //  parcels -- and the tables answered instead: remembered with the label raster's address, hint 3)
    const int label_hint = chunk_hint_get(labels, n_labels, cells);
    const bool lazy_first = lazy_first_mode < 0 ? ((size_t)n_labels * sizeof(LazyRow) * 8192 <= (16u << 20) || label_hint == 3) : lazy_first_mode != 0;
    // (a raster one of whose labels failed the tables' bound for this statistic -- the chunk records' tighter bound answered it, or
    //  the order did -- is remembered with its cells' address and goes there at once next time: two attempts per call otherwise)
    const void* const tables_key = (const char*)frame + 1;
    const int tables_tag = n_labels ^ (statistic << 24);
    if (range && lazy_first && chunk_hint_get(tables_key, tables_tag, cells) == 0) {
      const int answered = lazy_tables();
      if (answered < 0) return -answered;
      if (answered) continue;
      if (lazy_tried && lazy_mode) chunk_hint_set(tables_key, tables_tag, cells, 1);
    }
    // Rasters of thousands of SMALL labels (parcels: more labels than a workgroup's table holds, a few thousand cells each at
    // most on average): the parallel sums first -- zonal_exact_kernel PHASE 3: sums, sums of magnitudes and counts, one atomic
    // each per run -- and the float32 certificate in its crudest form on every label (zonal_exact_lazy_decide_kernel: for a
    // label of n cells E ~ 4 n u mag, which clears a float32 rounding boundary by orders of magnitude while n is small).  One
    // pass and one wait; 49 152 polygons of 16 x 16 cells 309 -> ~120 us a call (round 5: 4.9 ms), 6 592 polygons of 48 x 40
    // 278 -> ~120.  A label that fails (a long one among the small, a sum that cancels to nothing) sends the frame on to the
    // routes below.  (GM_ZONAL_PARALLEL_FIRST=0: never -- A/B runs, and the tests compare)
    static const bool parallel_first_allowed = !(getenv("GM_ZONAL_PARALLEL_FIRST") && atoi(getenv("GM_ZONAL_PARALLEL_FIRST")) == 0);
    // (a raster whose labels failed it in numbers -- zero-mean noise under thousands of labels: a few dozen sums within the
    //  crude bound of a float32 tie -- is remembered with its cells' address and skips the attempt next time)
    const void* const parallel_key = (const char*)frame + 2;
    if (range && lazy_mode && parallel_first_allowed && (size_t)n_labels * sizeof(LazyRow) > ZT_LDS_BYTES &&
        cells / n_labels <= 4096 && chunk_hint_get(parallel_key, n_labels, cells) == 0) {
      HostFlag* const flag = host_flag();
      if (flag) {
        if (!mags && (rc = scratch.take((void**)&mags, sizeof(double) * n_labels))) return rc;
        flag->word[6] = 0u;
        if (!few_index) {
          if ((rc = scratch.take((void**)&few_index, sizeof(unsigned) * ZF_CAP * ZF_SLOTS))) return rc;
          if ((rc = scratch.take((void**)&few_values, sizeof(double) * ZF_CAP * ZF_SLOTS))) return rc;
          if ((rc = scratch.take((void**)&few_slots, (sizeof(int) + sizeof(unsigned)) * ZF_SLOTS))) return rc;
        }
        hipLaunchKernelGGL(zonal_exact_init_kernel, dim3((std::max<int>(n_labels, 2 * (int)ZF_SLOTS) + 255) / 256), dim3(256), 0, s, range, sums,
                           n_labels, verdict_dev, cursor, mags, inexact, few_slots, 2 * (int)ZF_SLOTS);
        hipLaunchKernelGGL((zonal_exact_kernel<T, 3>), dim3(direct_blocks), dim3(256), 0, s, frame, labels, (size_t)cells, cells_per_wave,
                           n_labels, has_ndv, ndv, thresholds, range, sums, seen, (unsigned*)nullptr, (Key*)nullptr, (unsigned*)nullptr, mags);
        // (with the results as if every label passed; the few that do not -- up to ZF_SLOTS labels of up to ZF_CAP cells -- are
        //  added in order behind it; anything else raises the flag and the frame goes on)
        hipLaunchKernelGGL(zonal_exact_lazy_decide_kernel, dim3((n_labels + 255) / 256), dim3(256), 0, s, range, sums, mags, n_labels, statistic,
                           inexact, verdict_dev, 1, (double)(cells_per_wave / 64), out_f, counts_f);
        hipLaunchKernelGGL(zonal_few_gather_kernel<T>, dim3(blocks), dim3(256), 0, s, frame, labels, (size_t)cells, n_labels, has_ndv, ndv,
                           thresholds, inexact, verdict_dev, few_index, few_values, (int*)few_slots, (unsigned*)few_slots + ZF_SLOTS);
        hipLaunchKernelGGL(zonal_few_ordered_kernel, dim3(ZF_SLOTS), dim3(1024), 0, s, few_index, few_values, (const int*)few_slots,
                           (const unsigned*)few_slots + ZF_SLOTS, verdict_dev, statistic, out_f, counts_f, flag->word + 6);
        GM_HIP(hipGetLastError());
        host_copy_queue(frames, s);
        GM_HIP(hipEventRecord(flag->event, s));
        GM_HIP(hipEventSynchronize(flag->event));
        const bool answered = ((volatile unsigned*)flag->word)[6] == 0u;
        host_copy_settle(answered);
        if (answered) {
          gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_PARALLEL_LAZY);
          continue;
        }
        chunk_hint_set(parallel_key, n_labels, cells, 1);
      }
    }
    // sums first try the route that sorts nothing (chunk records: zonal_chunk_kernel); GM_ZONAL_CHUNKS=0 switches it off and
    // GM_ZONAL_SORTED=1 skips it together with the other parallel sums (A/B runs, and the tests compare the routes)
    // (GM_ZONAL_CHUNKS=2: also the labels of ZL_LONG cells and more stay on this route however often they are replayed -- the tests do)
    static const int chunks_mode = getenv("GM_ZONAL_CHUNKS") ? atoi(getenv("GM_ZONAL_CHUNKS")) : 1;
    const bool chunks_allowed = chunks_mode != 0;
    // (GM_ZONAL_LAZY=0: no label is answered by the float32 certificate of zonal_chunk_combine_kernel -- every sum in bincount's
    //  order, as before round 6: A/B runs, and the tests compare the two)
    //  (=2: the kernel prints every label's bounds)
    static const int lazy_allowed = getenv("GM_ZONAL_LAZY") ? atoi(getenv("GM_ZONAL_LAZY")) : 1;
    bool chunk_gave_up_hard = false;  // (a sum that keeps rounding: the parallel sums below cannot answer it either)
    bool chunk_not_runs = false;      // (labels scattered cell by cell: the certificates through the workgroups' LDS tables)
    if (range && chunks_allowed && cells >= ZC_LANE && (uintptr_t)frame % sizeof(CellQuad<T>) == 0 &&
        (uintptr_t)labels % sizeof(LabelQuad) == 0) {
      // stretches of at least `stretch()` cells, whole steps, at most 64 steps each (more, shorter stretches for large rasters).
      // A second attempt in stretches a quarter as long when the first one only found stretches with more labels than records
      // (small polygons, vertical stripes: runs all the same) and the table of the shorter stretches still fits.
      bool answered = false, by_tables = false;
      // (a label raster whose first attempt overflowed is remembered -- by address, label count and size: a hint, never more --
      //  and starts with the shorter stretches next time: the other statistics of the same polygons, the next frames)
      // (... and a raster whose SUMS gave the pass up -- a long label that keeps rounding: hard -- is remembered with its cells'
      //  address: the mean after the sum, the next request for the same tile, go to the sorting route at once)
      // (a key of its own kind per hint -- the pool hands the block of one request's label raster to the next request's cells:
      //  a hint about labels read as a hint about cells sent a service's every other request to the sorting route)
      const void* const hard_key = (const char*)frame + 3;
      const int first_attempt = std::max(label_hint == 3 ? 1 : label_hint, chunk_hint_get(hard_key, n_labels, cells));
      if (first_attempt >= 2) chunk_gave_up_hard = true;
      for (int attempt = first_attempt; attempt < 2 && !answered; ++attempt) {
        const int64_t shortest = attempt == 0 ? stretch() : std::max<int64_t>(ZC_STEP, stretch() / 4);
        const int64_t most = (int64_t)gm::current_cu_count() * wave_cap() * (attempt == 0 ? 1 : 4);
        int64_t n_stretch = std::max<int64_t>(1, std::min<int64_t>((cells + shortest - 1) / shortest, most));
        size_t chunk_cells = (size_t)(((cells + n_stretch - 1) / n_stretch + ZC_STEP - 1) / ZC_STEP * ZC_STEP);
        if (chunk_cells > ZC_MAX_STEPS * (size_t)ZC_STEP) chunk_cells = ZC_MAX_STEPS * (size_t)ZC_STEP;
        n_stretch = (int64_t)(((size_t)cells + chunk_cells - 1) / chunk_cells);
        const size_t row_bytes = ((size_t)n_stretch + 63) / 64 * 64;  // a label's row of the slot table: whole 64-byte groups
        const size_t table_bytes = (size_t)n_labels * row_bytes;
        // (one byte per label and stretch, cleared by one fill and read once by the labels' waves.  128 MB: 6592 polygons of 48 x 40
        //  cells in stretches of 768 -- 108 MB -- 735 -> 466 us; 49 152 polygons of 16 x 16 cells -- 201 MB, and 192 labels to a
        //  stretch: an attempt for nothing -- stay with the per-label certificates, 465 us against 650.  GM_ZONAL_TABLE_MB for A/B runs)
        static const size_t table_limit = (size_t)(getenv("GM_ZONAL_TABLE_MB") ? atoll(getenv("GM_ZONAL_TABLE_MB")) : 128) << 20;
        if (table_bytes > table_limit) break;
        const size_t n_records = ZC_SLOTS * (size_t)n_stretch, n_steps = (chunk_cells / ZC_STEP) * (size_t)n_stretch;
        if (!chunk.records || chunk.records_for < n_records) {
          if ((rc = scratch.take((void**)&chunk.records, sizeof(ChunkRec) * n_records))) return rc;
          chunk.records_for = n_records;
        }
        if (!chunk.slot_of || chunk.table_for < table_bytes) {
          if ((rc = scratch.take((void**)&chunk.slot_of, table_bytes + 64))) return rc;  // (the verdict behind the table: one fill)
          chunk.table_for = table_bytes;
        }
        if (!chunk.steps || chunk.steps_for < n_steps) {
          if ((rc = scratch.take((void**)&chunk.steps, sizeof(StepRec) * n_steps))) return rc;
          chunk.steps_for = n_steps;
        }
        chunk.verdict = (ChunkVerdict*)(chunk.slot_of + table_bytes);
        gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_CHUNK_ATTEMPTS);
        GM_HIP(hipMemsetAsync(chunk.slot_of, 0, table_bytes + 64, s));
        HostFlag* const flag = host_flag();
        if (flag) flag->word[0] = flag->word[1] = flag->word[2] = 0u;
        static const unsigned chunk_long = getenv("GM_ZONAL_LONG") ? (unsigned)std::max(2 * (int)ZB, atoi(getenv("GM_ZONAL_LONG"))) : ZL_LONG;
        hipLaunchKernelGGL(zonal_chunk_kernel<T>, dim3((unsigned)((n_stretch + 3) / 4)), dim3(256), 0, s, frame, labels, (size_t)cells, chunk_cells,
                           (unsigned)n_stretch, (unsigned)row_bytes, n_labels, has_ndv, ndv, thresholds, chunk.records, chunk.slot_of,
                           chunk.steps, chunk.verdict, flag ? flag->word : (unsigned*)nullptr, seen);
        hipLaunchKernelGGL(zonal_chunk_combine_kernel<T>, dim3((unsigned)(((size_t)n_labels * 64 + 255) / 256)), dim3(256), 0, s, frame, labels,
                           (size_t)cells, chunk_cells, (unsigned)n_stretch, (unsigned)row_bytes, n_labels, has_ndv, ndv, thresholds,
                           chunk.records, chunk.slot_of, chunk.steps, chunk.verdict,
                           flag ? flag->word : (unsigned*)nullptr, chunks_mode == 2 ? ~0u : chunk_long,
                           chunks_mode == 2 ? ~0u : ZC_REPLAYS, statistic, lazy_allowed, out_f, counts_f);
        GM_HIP(hipGetLastError());
        ChunkVerdict chunk_verdict{1u, 0u, 1u};
        if (flag) {  // (the one host decision of this route: three words in pinned memory, read once the kernels are through)
          host_copy_queue(frames, s);
          GM_HIP(hipEventRecord(flag->event, s));
          GM_HIP(hipEventSynchronize(flag->event));
          chunk_verdict.overflow = ((volatile unsigned*)flag->word)[0];
          chunk_verdict.hard = ((volatile unsigned*)flag->word)[1];
          chunk_verdict.not_runs = ((volatile unsigned*)flag->word)[2];
          host_copy_settle(!chunk_verdict.overflow && !chunk_verdict.hard);
        } else if ((rc = gm_d2h(&chunk_verdict, chunk.verdict, sizeof(chunk_verdict)))) {
          return rc;
        }
        if (!chunk_verdict.overflow && !chunk_verdict.hard) {
          answered = true;
          if (attempt != first_attempt || attempt == 1) chunk_hint_set(labels, n_labels, cells, attempt);
          break;
        }
        // (a sum that keeps rounding fails the certificate of the parallel sums below; and they pay an atomic per quantity and
        //  run, which only long runs repay -- the rasters the chunk records take: 12.6 M cells under 40 labels at random 31-50 ms
        //  there, 1 ms through the sort; in stripes 25 cells wide 5.8 ms against 0.9, in polygons of 48 x 40 cells 5 ms against 0.6)
        // ... where the sort is the counting partition.  Beyond its 2047 labels the atomics spread over thousands of counters and
        // the sort is the radix sort: 6592 polygons of 48 x 40 int32 cells 240 us through the parallel sums, 590 through the sort.
        chunk_gave_up_hard = chunk_verdict.hard != 0 || n_labels + 1 <= ZP_BINS;
        chunk_not_runs = chunk_verdict.not_runs != 0 && chunk_verdict.hard == 0;
        if (chunk_verdict.hard) chunk_hint_set(hard_key, n_labels, cells, 2);
        if (attempt == 0 && chunk_verdict.overflow && !chunk_verdict.hard && !chunk_verdict.not_runs && !lazy_tried) {
          // more labels to a stretch than records (vertical stripes 25 cells wide, small parcels): before the shorter stretches --
          // four times the records, a pass of 145-200 us -- the tables of zonal_tables_lazy_kernel, whose pass does not care
          // (stripes under 164 labels 291 -> 150 us a call); what answered is remembered
          const int tables = lazy_tables();
          if (tables < 0) return -tables;
          if (tables) {
            chunk_hint_set(labels, n_labels, cells, 3);
            answered = by_tables = true;
            break;
          }
        }
        if (chunk_verdict.hard || chunk_verdict.not_runs || chunk_cells <= (size_t)ZC_STEP) break;  // (shorter stretches would not help)
      }
      gm::route_hit(GM_ROUTE_ZONAL, answered && !by_tables ? GM_RZ_CHUNK_ANSWERED : GM_RZ_CHUNK_GAVE_UP);
      if (answered) continue;
    }
    unsigned longest = ~0u;  // (cells of the longest label, where a verdict has counted them)
    // labels scattered cell by cell: the exactness certificates and, where every label passes, the sums -- through tables in
    // LDS (zonal_tables_exact_kernel; GM_ZONAL_TABLES_EXACT=0: never -- A/B runs, and the tests compare the two)
    // ... first the float32 certificate through the same kind of tables (zonal_tables_lazy_kernel: one pass, no sum in order; a label
    // that fails sends the frame on)
    if (range && chunk_not_runs && !lazy_tried) {
      const int answered = lazy_tables();
      if (answered < 0) return -answered;
      if (answered) continue;
    }
    static const bool tables_exact_allowed = !(getenv("GM_ZONAL_TABLES_EXACT") && atoi(getenv("GM_ZONAL_TABLES_EXACT")) == 0);
    if (range && chunk_not_runs && tables_exact_allowed && (size_t)n_labels * 16 <= ZT_LDS_BYTES) {
      const size_t lds = (size_t)n_labels * 16;
      const int64_t waves_wanted = std::max<int64_t>(1, std::min<int64_t>((cells + 255) / 256, (int64_t)gm::current_cu_count() * 2 * (ZX_THREADS / 64)));
      const unsigned x_blocks = (unsigned)((waves_wanted + ZX_THREADS / 64 - 1) / (ZX_THREADS / 64));
      const size_t x_cells_per_wave = (size_t)(((cells + (int64_t)x_blocks * (ZX_THREADS / 64) - 1) / ((int64_t)x_blocks * (ZX_THREADS / 64)) + 63) / 64 * 64);
      hipLaunchKernelGGL(zonal_exact_init_kernel, dim3((n_labels + 255) / 256), dim3(256), 0, s, range, sums, n_labels, verdict_dev, cursor);
      hipLaunchKernelGGL((zonal_tables_exact_kernel<T, 0>), dim3(x_blocks), dim3(ZX_THREADS), lds, s, frame, labels, (size_t)cells, x_cells_per_wave,
                         n_labels, has_ndv, ndv, thresholds, range, sums, seen);
      hipLaunchKernelGGL(zonal_exact_decide_kernel, dim3((n_labels + 255) / 256), dim3(256), 0, s, range, n_labels, inexact, verdict_dev);
      ExactVerdict verdict{1u, 0u, (unsigned long long)cells};
      if ((rc = gm_d2h(&verdict, verdict_dev, sizeof(verdict)))) return rc;
      if (verdict.labels == 0) {
        hipLaunchKernelGGL((zonal_tables_exact_kernel<T, 1>), dim3(x_blocks), dim3(ZX_THREADS), (size_t)n_labels * 8, s, frame, labels, (size_t)cells,
                           x_cells_per_wave, n_labels, has_ndv, ndv, thresholds, range, sums, (unsigned char*)nullptr);
        hipLaunchKernelGGL(zonal_exact_result_kernel, dim3((n_labels + 255) / 256), dim3(256), 0, s, range, sums, n_labels, statistic,
                           out_f, counts_f);
        GM_HIP(hipGetLastError());
        gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_LDS_EXACT);
        continue;
      }
      longest = verdict.longest;
    }
    // medians / percentiles under a few labels of a raster of 4-byte cells (or narrower): the radix select straight off the
    // raster (zonal_select_raw_pass_kernel; GM_ZONAL_RAW_SELECT=0: never -- A/B runs, and the tests compare the two)
    static const bool raw_select_allowed = !(getenv("GM_ZONAL_RAW_SELECT") && atoi(getenv("GM_ZONAL_RAW_SELECT")) == 0);
    if constexpr (sizeof(Key) == 4) {
      if (by_value && select_allowed && raw_select_allowed && n_labels <= ZR_LABELS) {
        if (!raw_state) {
          if ((rc = scratch.take((void**)&raw_state, sizeof(SelectState) * n_labels))) return rc;
          if ((rc = scratch.take((void**)&raw_hist, sizeof(unsigned) * 256 * (size_t)n_labels))) return rc;
          if ((rc = scratch.take((void**)&raw_next, sizeof(unsigned long long) * n_labels))) return rc;
          if ((rc = scratch.take((void**)&raw_first, sizeof(unsigned) * ((size_t)n_labels + 2)))) return rc;
        }
        const unsigned label_grid = (unsigned)((n_labels + 255) / 256), step_grid = (unsigned)(((size_t)n_labels * 64 + 255) / 256);
        const int64_t raw_waves = std::max<int64_t>(1, std::min<int64_t>((cells + 255) / 256, (int64_t)gm::current_cu_count() * 32));
        const size_t raw_cells_per_wave = (size_t)(((cells + raw_waves - 1) / raw_waves + 63) / 64 * 64);
        const unsigned raw_grid = (unsigned)((raw_waves + 3) / 4);
        GM_HIP(hipMemsetAsync(raw_hist, 0, sizeof(unsigned) * 256 * (size_t)n_labels, s));
        gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_RAW_SELECT);
        const int top = 8 * ((int)sizeof(T) - 1);
        for (int shift = top; shift >= 0; shift -= 8) {
          hipLaunchKernelGGL((zonal_select_raw_pass_kernel<T, false>), dim3(raw_grid), dim3(256), 0, s, frame, labels, (size_t)cells,
                             raw_cells_per_wave, n_labels, has_ndv, ndv, thresholds, shift, shift == top ? 1 : 0, raw_state, raw_hist,
                             raw_next, seen);
          if (shift == top) {
            hipLaunchKernelGGL(zonal_select_raw_counts_kernel, dim3(1), dim3(64), 0, s, raw_hist, n_labels, raw_first);
            hipLaunchKernelGGL(zonal_select_init_kernel, dim3(label_grid), dim3(256), 0, s, raw_first, n_labels, statistic, qval, raw_state,
                               raw_next);
          }
          hipLaunchKernelGGL(zonal_select_step_kernel, dim3(step_grid), dim3(256), 0, s, n_labels, shift, raw_state, raw_hist);
        }
        hipLaunchKernelGGL((zonal_select_raw_pass_kernel<T, true>), dim3(raw_grid), dim3(256), 0, s, frame, labels, (size_t)cells,
                           raw_cells_per_wave, n_labels, has_ndv, ndv, thresholds, 0, 0, raw_state, raw_hist, raw_next,
                           (unsigned char*)nullptr);
        hipLaunchKernelGGL(zonal_select_result_kernel<T>, dim3(label_grid), dim3(256), 0, s, raw_state, raw_next, n_labels, statistic, qval,
                           out_f, counts_f);
        GM_HIP(hipGetLastError());
        continue;
      }
    }
    if ((rc = key_buffers())) return rc;
    bool keys_written = false;
    part.counted = false;
    if (range && !chunk_gave_up_hard) {
      if (partition) {  // (3 MB for the default raster limit and 1024 labels)
        if (!part.counts && (rc = scratch.take((void**)&part.counts, sizeof(unsigned) * part_tiles * part_bins))) return rc;
        GM_HIP(hipMemsetAsync(part.counts, 0, sizeof(unsigned) * part_tiles * part_bins, s));
        part.counted = true;
      }
      hipLaunchKernelGGL(zonal_exact_init_kernel, dim3((n_labels + 255) / 256), dim3(256), 0, s, range, sums, n_labels, verdict_dev, cursor);
      hipLaunchKernelGGL((zonal_exact_kernel<T, 0>), dim3(direct_blocks), dim3(256), 0, s, frame, labels, (size_t)cells, cells_per_wave,
                         n_labels, has_ndv, ndv, thresholds, range, sums, seen, keys_a, pay_a, part.counted ? part.counts : (unsigned*)nullptr);
      hipLaunchKernelGGL(zonal_exact_decide_kernel, dim3((n_labels + 255) / 256), dim3(256), 0, s, range, n_labels, inexact, verdict_dev);
      ExactVerdict verdict{1u, 0u, (unsigned long long)cells};
      if ((rc = gm_d2h(&verdict, verdict_dev, sizeof(verdict)))) return rc;  // (the one host decision of a sum: through pinned memory)
      // every label exact: the parallel sums are the answer.  A few labels not (they hold at most 1/8 of the cells): the
      // parallel sums answer all the others, and only the cells of those few are sorted and added in order.
      // (round 6) labels the exponents turn away: the parallel sums with the sums of the magnitudes beside them, and the float32
      // certificate on those labels (zonal_exact_lazy_decide_kernel); what still needs its cells in order is counted again
      bool summed = false;
      if (verdict.labels != 0 && lazy_allowed) {
        if (!mags && (rc = scratch.take((void**)&mags, sizeof(double) * n_labels))) return rc;
        GM_HIP(hipMemsetAsync(mags, 0, sizeof(double) * n_labels, s));
        GM_HIP(hipMemsetAsync(verdict_dev, 0, sizeof(ExactVerdict), s));
        hipLaunchKernelGGL((zonal_exact_kernel<T, 2>), dim3(direct_blocks), dim3(256), 0, s, frame, labels, (size_t)cells,
                           cells_per_wave, n_labels, has_ndv, ndv, thresholds, range, sums, (unsigned char*)nullptr,
                           (unsigned*)nullptr, (Key*)nullptr, (unsigned*)nullptr, mags);
        hipLaunchKernelGGL(zonal_exact_lazy_decide_kernel, dim3((n_labels + 255) / 256), dim3(256), 0, s, range, sums, mags, n_labels, statistic,
                           inexact, verdict_dev);
        ExactVerdict again{1u, 0u, (unsigned long long)cells};
        if ((rc = gm_d2h(&again, verdict_dev, sizeof(again)))) return rc;
        if (again.labels < verdict.labels) gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_EXACT_LAZY, verdict.labels - again.labels);
        verdict.labels = again.labels;
        verdict.cells = again.cells;
        summed = true;
      }
      const bool partial = verdict.labels != 0 && verdict.cells != 0 && verdict.cells <= (unsigned long long)cells / 8;
      if (verdict.labels == 0 || partial) {
        gm::route_hit(GM_ROUTE_ZONAL, partial ? GM_RZ_EXACT_PARTIAL : GM_RZ_EXACT_ALL);
        if (!summed)
          hipLaunchKernelGGL((zonal_exact_kernel<T, 1>), dim3(direct_blocks), dim3(256), 0, s, frame, labels, (size_t)cells,
                             cells_per_wave, n_labels, has_ndv, ndv, thresholds, range, sums, (unsigned char*)nullptr,
                             (unsigned*)nullptr, (Key*)nullptr, (unsigned*)nullptr);
        hipLaunchKernelGGL(zonal_exact_result_kernel, dim3((n_labels + 255) / 256), dim3(256), 0, s, range, sums, n_labels, statistic,
                           out_f, counts_f);
        if (partial) {
          const size_t m = (size_t)verdict.cells;
          Scratch local;  // the list's buffers: sized by this frame's verdict
          unsigned long long *list_a, *list_b;
          Key *value_a, *value_b;
          void* list_temp;
          if ((rc = local.take((void**)&list_a, sizeof(unsigned long long) * m))) return rc;
          if ((rc = local.take((void**)&list_b, sizeof(unsigned long long) * m))) return rc;
          if ((rc = local.take((void**)&value_a, sizeof(Key) * m))) return rc;
          if ((rc = local.take((void**)&value_b, sizeof(Key) * m))) return rc;
          if ((rc = local.take(&list_temp, radix_temp_bytes(m)))) return rc;
          hipLaunchKernelGGL(zonal_gather_kernel<Key>, dim3(blocks), dim3(256), 0, s, keys_a, pay_a, (size_t)cells, n_labels, inexact,
                             list_a, value_a, cursor);
          if ((rc = radix_sort_pairs<unsigned long long, Key>(list_temp, list_a, list_b, value_a, value_b, m, 32 + (int)label_bits, s))) return rc;
          hipLaunchKernelGGL(zonal_list_bounds_kernel, dim3((n_labels + 1 + 255) / 256), dim3(256), 0, s, list_b, (unsigned)m, n_labels, first);
          hipLaunchKernelGGL(zonal_short_sum_kernel<T>, dim3((n_labels + 255) / 256), dim3(256), 0, s, value_b, first, n_labels, statistic,
                             inexact, out_f, counts_f);
          hipLaunchKernelGGL(zonal_sum_kernel<T>, dim3((unsigned)(((size_t)n_labels * 64 + 255) / 256)), dim3(256), 0, s, value_b, first,
                             n_labels, statistic, inexact, out_f, counts_f);
        }
        GM_HIP(hipGetLastError());
        continue;
      }
      keys_written = true;  // (and `present` is recorded)
      longest = verdict.longest;
    }
    if ((rc = sort_buffers())) return rc;
    if (!keys_written) {
      if (partition && (!by_value || select_path)) {  // (keys and the tile histograms of the partition that follows, in one pass)
        hipLaunchKernelGGL((zonal_keys_count_kernel<T, 8>), dim3((unsigned)part_tiles), dim3(512), 0, s, frame, labels, (unsigned)cells, n_labels,
                           has_ndv, ndv, thresholds, keys_a, pay_a, seen, part.counts);
        part.counted = true;
      } else {
        hipLaunchKernelGGL(zonal_keys_kernel<T>, dim3(blocks), dim3(256), 0, s, frame, labels, (size_t)cells, n_labels, has_ndv, ndv,
                           thresholds, keys_a, pay_a, seen);
      }
    }
    const unsigned* sorted_keys;
    const Key* sorted_pay;
    if (select_path) {
      // the cells of every label together (in any order), then the wanted ranks by radix select: no sort by value
      if ((rc = sort_by_label(keys_a, pay_a, (unsigned*)nullptr, pay_b))) return rc;
      const unsigned label_grid = (unsigned)((n_labels + 255) / 256);
      const unsigned pass_grid = (unsigned)(((size_t)cells + 4 * SEL_WAVE - 1) / (4 * SEL_WAVE));
      GM_HIP(hipMemsetAsync(sel_hist, 0, sizeof(unsigned) * 256 * (size_t)n_labels, s));
      gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_SELECT);
      hipLaunchKernelGGL(zonal_select_init_kernel, dim3(label_grid), dim3(256), 0, s, first, n_labels, statistic, qval, sel_state, sel_next);
      for (int shift = 8 * ((int)sizeof(T) - 1); shift >= 0; shift -= 8) {
        hipLaunchKernelGGL((zonal_select_pass_kernel<Key, false>), dim3(pass_grid), dim3(256), 0, s, first, pay_b, n_labels, shift, sel_state,
                           sel_hist, sel_next);
        hipLaunchKernelGGL(zonal_select_step_kernel, dim3((unsigned)(((size_t)n_labels * 64 + 255) / 256)), dim3(256), 0, s, n_labels, shift,
                           sel_state, sel_hist);
      }
      hipLaunchKernelGGL((zonal_select_pass_kernel<Key, true>), dim3(pass_grid), dim3(256), 0, s, first, pay_b, n_labels, 0, sel_state,
                         sel_hist, sel_next);
      hipLaunchKernelGGL(zonal_select_result_kernel<T>, dim3(label_grid), dim3(256), 0, s, sel_state, sel_next, n_labels, statistic, qval,
                         out_f, counts_f);
      GM_HIP(hipGetLastError());
      continue;
    }
    if (segsort_path) {
      HostFlag* const flag = host_flag();
      if (flag) {
        flag->word[4] = 0u;
        if ((rc = sort_by_label(keys_a, pay_a, keys_b, pay_b))) return rc;
        hipLaunchKernelGGL((zonal_segment_sort_kernel<Key, 64, 0u, ZS_SMALL>), dim3((unsigned)n_labels), dim3(64), 0, s, pay_b, first, n_labels,
                           flag->word + 4);
        hipLaunchKernelGGL((zonal_segment_sort_kernel<Key, 256, ZS_SMALL, ZS_MAX>), dim3((unsigned)n_labels), dim3(256), 0, s, pay_b, first,
                           n_labels, flag->word + 4);
        hipLaunchKernelGGL(zonal_pick_kernel<T>, dim3((n_labels + 255) / 256), dim3(256), 0, s, pay_b, first, n_labels, statistic,
                           qval, out_f, counts_f);
        GM_HIP(hipGetLastError());
        GM_HIP(hipEventRecord(flag->event, s));
        GM_HIP(hipEventSynchronize(flag->event));
        if (((volatile unsigned*)flag->word)[4] == 0u) {
          gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_SEGMENT_SORT);
          continue;
        }
        // a label longer than a workgroup sorts: the keys again (the sort above used them up), then the two sorts
        hipLaunchKernelGGL(zonal_keys_kernel<T>, dim3(blocks), dim3(256), 0, s, frame, labels, (size_t)cells, n_labels, has_ndv, ndv,
                           thresholds, keys_a, pay_a, (unsigned char*)nullptr);
      }
    }
    if (by_value) {
      gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_LEXSORT);
      // lexsort((values, labels)): by value, then stably by label
      if ((rc = radix_sort_pairs<Key, unsigned>(temp, pay_a, pay_b, keys_a, keys_b, (size_t)cells, 8 * (int)sizeof(Key), s))) return rc;
      if ((rc = sort_by_label(keys_b, pay_b, keys_a, pay_a))) return rc;
      sorted_keys = keys_a;
      sorted_pay = pay_a;
    } else {
      if ((rc = sort_by_label(keys_a, pay_a, keys_b, pay_b))) return rc;
      sorted_keys = keys_b;
      sorted_pay = pay_b;
    }
    if (statistic == Z_SUM || statistic == Z_MEAN) {
      // labels of ZL_LONG cells and more: block-parallel (zonal_compose_kernel and its three helpers); GM_ZONAL_BLOCKS=0
      // keeps every label on the wave-per-label kernel (A/B runs, and the tests compare the two)
      static const bool block_pipeline = !(getenv("GM_ZONAL_BLOCKS") && atoi(getenv("GM_ZONAL_BLOCKS")) == 0);
      static const unsigned long_from = getenv("GM_ZONAL_LONG") ? (unsigned)std::max(2 * (int)ZB, atoi(getenv("GM_ZONAL_LONG"))) : ZL_LONG;
      const bool long_labels = block_pipeline && (uint64_t)cells >= long_from && longest >= long_from;
      gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_ORDERED_SUMS);
      if (long_labels) {
        gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_BLOCK_PIPELINE);
        const size_t n_blocks = ((size_t)cells + ZB - 1) / ZB;
        if (!blk_sum) {
          if ((rc = scratch.take((void**)&blk_sum, sizeof(double) * n_blocks))) return rc;
          if ((rc = scratch.take((void**)&blk_pred, sizeof(double) * n_blocks))) return rc;
          if ((rc = scratch.take((void**)&blk_abs, sizeof(double) * n_blocks))) return rc;
          if ((rc = scratch.take((void**)&blk_gexp, sizeof(int) * n_blocks))) return rc;
          if ((rc = scratch.take((void**)&grp_exact, sizeof(GroupExact) * ((n_blocks + ZG - 1) / ZG)))) return rc;
          if ((rc = scratch.take((void**)&blk_label, sizeof(int) * n_blocks))) return rc;
          if ((rc = scratch.take((void**)&blk_info, sizeof(BlockInfo) * n_blocks))) return rc;
          if ((rc = scratch.take((void**)&grp_info, sizeof(GroupInfo) * ((n_blocks + ZG - 1) / ZG)))) return rc;
        }
        const unsigned block_grid = (unsigned)((n_blocks * 64 + 255) / 256), label_grid = (unsigned)(((size_t)n_labels * 64 + 255) / 256);
        hipLaunchKernelGGL(zonal_block_sum_kernel<T>, dim3(block_grid), dim3(256), 0, s, sorted_pay, sorted_keys, (unsigned)cells, n_labels,
                           blk_sum, blk_label, blk_abs, blk_gexp);
        hipLaunchKernelGGL(zonal_block_prefix_kernel<T>, dim3((unsigned)n_labels), dim3(64 * ZPRE_WAVES), 0, s, sorted_pay, first, n_labels, blk_sum,
                           blk_pred, long_from);
        hipLaunchKernelGGL(zonal_block_info_kernel<T>, dim3(block_grid), dim3(256), 0, s, sorted_pay, first, (unsigned)cells, blk_label,
                           blk_pred, blk_info, long_from);
        {
          const unsigned n_groups = (unsigned)((n_blocks + ZG - 1) / ZG);
          hipLaunchKernelGGL(zonal_group_info_kernel, dim3((unsigned)(((size_t)n_groups * 64 + 255) / 256)), dim3(256), 0, s, blk_label, blk_info,
                             first, (unsigned)((size_t)cells / ZB), long_from, grp_info, n_groups);
          hipLaunchKernelGGL(zonal_group_exact_kernel, dim3((unsigned)(((size_t)n_groups * 64 + 255) / 256)), dim3(256), 0, s, blk_label, blk_sum,
                             blk_abs, blk_gexp, first, (unsigned)((size_t)cells / ZB), long_from, grp_exact, n_groups);
        }
        static const bool verbose = getenv("GM_ZONAL_VERBOSE") != nullptr;
        static const bool certify = !(getenv("GM_ZONAL_CERTIFY") && atoi(getenv("GM_ZONAL_CERTIFY")) == 0);  // (0: A/B runs, tests)
        unsigned* replayed = nullptr;
        if (verbose) {
          if ((rc = gm_malloc((void**)&replayed, 8 * sizeof(unsigned)))) return rc;
          GM_HIP(hipMemsetAsync(replayed, 0, 8 * sizeof(unsigned), s));
        }
        hipLaunchKernelGGL(zonal_compose_kernel<T>, dim3(label_grid), dim3(256), 0, s, sorted_pay, first, n_labels, statistic, blk_label,
                           blk_info, grp_info, out_f, counts_f, replayed, long_from, blk_sum, blk_abs,
                           certify ? blk_gexp : (const int*)nullptr, certify ? grp_exact : (const GroupExact*)nullptr);
        if (verbose) {
          unsigned host[8] = {0};
          GM_HIP(hipMemcpyAsync(host, replayed, sizeof(host), hipMemcpyDeviceToHost, s));
          GM_HIP(hipStreamSynchronize(s));
          fprintf(stderr, "[gmhip] zonal block pipeline: %zu blocks, %u of them added literally (%u with their cells requested ahead), "
                          "%u turned away by the binade analysis but certified exact, %u groups of 256 certified exact, %u taken in one "
                          "step by the binade analysis, %u walked in %u compositions\n", n_blocks, host[0], host[1], host[2], host[3], host[6],
                  host[5], host[4]);
          (void)gm_free(replayed);
        }
      }
      hipLaunchKernelGGL(zonal_short_sum_kernel<T>, dim3((n_labels + 255) / 256), dim3(256), 0, s, sorted_pay, first, n_labels,
                         statistic, (const unsigned char*)nullptr, out_f, counts_f);
      hipLaunchKernelGGL(zonal_sum_kernel<T>, dim3((unsigned)(((size_t)n_labels * 64 + 255) / 256)), dim3(256), 0, s, sorted_pay, first,
                         n_labels, statistic, (const unsigned char*)nullptr, out_f, counts_f, long_labels ? long_from : 0u);
    }
    else
      hipLaunchKernelGGL(zonal_pick_kernel<T>, dim3((n_labels + 255) / 256), dim3(256), 0, s, sorted_pay, first, n_labels, statistic,
                         qval, out_f, counts_f);
    GM_HIP(hipGetLastError());
  }
  // the scratch buffers go back to the pool: the pool is stream-ordered like every other buffer of this library
  return GM_OK;
}

}  // namespace

extern "C" {

int gm_zonal_statistic(const void* values, int dtype, int64_t frames, int64_t cells, int has_ndv, uint64_t ndv_bits,
                       const int32_t* labels, int32_t n_labels, const float* thresholds, int statistic, double percentile,
                       float* out, int64_t* counts, uint8_t* present) {
  if (!values || !labels || !out || !counts) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (frames < 0 || cells < 0 || n_labels < 0) return gm::fail(GM_ERR_INVALID, "negative extent");
  if (statistic < Z_SUM || statistic > Z_PERCENTILE) return gm::fail(GM_ERR_INVALID, "gm_zonal_statistic: statistic %d", statistic);
  if (statistic == Z_PERCENTILE && !(percentile >= 0.0 && percentile <= 100.0))
    return gm::fail(GM_ERR_INVALID, "gm_zonal_statistic: percentile %g", percentile);
  if (statistic == Z_PERCENTILE && dtype == GM_BOOL)
    return gm::fail(GM_ERR_UNSUPPORTED, "gm_zonal_statistic: percentile of a boolean raster (NumPy has no boolean subtraction)");
  if (cells >= (int64_t)1 << 31) return gm::fail(GM_ERR_UNSUPPORTED, "gm_zonal_statistic: %lld cells (fewer than 2^31)", (long long)cells);
  if (n_labels == INT32_MAX) return gm::fail(GM_ERR_UNSUPPORTED, "gm_zonal_statistic: too many labels");
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  gm::route_hit(GM_ROUTE_ZONAL, GM_RZ_CALLS);
  // (count / min / max clear `present` in the launch that clears their tallies: zonal_direct_init_kernel)
  const bool direct = (statistic == Z_COUNT || statistic == Z_MIN || statistic == Z_MAX) && frames > 0 && cells > 0;
  if (present && n_labels && !direct) GM_HIP(hipMemsetAsync(present, 0, (size_t)n_labels, s));
  if (frames == 0 || n_labels == 0) return GM_OK;
  if (cells == 0) {
    // nothing is active: every label stays NaN with a count of zero
    const size_t n = (size_t)frames * n_labels;
    GM_HIP(hipMemsetAsync(out, 0xFF, n * sizeof(float), s));  // (all-ones is a quiet NaN)
    GM_HIP(hipMemsetAsync(counts, 0, n * sizeof(int64_t), s));
    return GM_OK;
  }
#define GM_Z(T) \
  return zonal_run<T>(values, frames, cells, has_ndv, ndv_bits, labels, n_labels, thresholds, statistic, percentile, out, (long long*)counts, present, s)
  switch (dtype) {
    case GM_BOOL: case GM_U8: GM_Z(unsigned char);
    case GM_I8: GM_Z(signed char);
    case GM_U16: GM_Z(unsigned short);
    case GM_I16: GM_Z(short);
    case GM_U32: GM_Z(unsigned int);
    case GM_I32: GM_Z(int);
    case GM_U64: GM_Z(unsigned long long);
    case GM_I64: GM_Z(long long);
    case GM_F32: GM_Z(float);
    case GM_F64: GM_Z(double);
    default: return gm::fail(GM_ERR_INVALID, "gm_zonal_statistic: dtype %d", dtype);
  }
#undef GM_Z
}

int gm_zonal_statistic_host(const void* values, int dtype, int64_t frames, int64_t cells, int has_ndv, uint64_t ndv_bits,
                            const int32_t* labels, int32_t n_labels, const float* thresholds, int statistic, double percentile,
                            float* out_host, int64_t* counts_host, uint8_t* present_host) {
  if (!out_host || !counts_host) return gm::fail(GM_ERR_INVALID, "null pointer");
  if (frames < 0 || n_labels < 0) return gm::fail(GM_ERR_INVALID, "negative extent");
  const size_t n = (size_t)frames * (size_t)n_labels;
  const size_t bytes = 12 * n + (size_t)n_labels;
  if (bytes == 0) return gm_zonal_statistic(values, dtype, frames, cells, has_ndv, ndv_bits, labels, n_labels, thresholds, statistic, percentile,
                                            (float*)out_host, counts_host, present_host);
  hipStream_t s = gm::current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  // one device range for the three results: counts (8-byte aligned) | out | present
  unsigned char* dev = nullptr;
  if (int rc = gm_malloc((void**)&dev, bytes)) return rc;
  HostCopy hc;
  hc.pinned = bytes <= kHostCopyBytes ? host_copy_staging() : nullptr;
  hc.dev = dev;
  hc.bytes = bytes;
  t_host_copy = &hc;
  int rc = gm_zonal_statistic(values, dtype, frames, cells, has_ndv, ndv_bits, labels, n_labels, thresholds, statistic, percentile,
                              (float*)(dev + 8 * n), (int64_t*)dev, dev + 12 * n);
  t_host_copy = nullptr;
  if (rc == GM_OK && !hc.valid) {
    if (hc.pinned) {
      hipError_t e = hipMemcpyAsync(hc.pinned, dev, bytes, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) rc = gm::hip_fail(e, "gm_zonal_statistic_host: copy");
      else hc.valid = true;
    } else {  // (more than the staging buffer holds: the three ranges straight into the caller's arrays)
      rc = gm_d2h(counts_host, dev, 8 * n);
      if (rc == GM_OK) rc = gm_d2h(out_host, dev + 8 * n, 4 * n);
      if (rc == GM_OK && present_host) rc = gm_d2h(present_host, dev + 12 * n, (size_t)n_labels);
    }
  }
  if (rc == GM_OK && hc.valid) {
    const unsigned char* from = (const unsigned char*)hc.pinned;
    memcpy(counts_host, from, 8 * n);
    memcpy(out_host, from + 8 * n, 4 * n);
    if (present_host) memcpy(present_host, from + 12 * n, (size_t)n_labels);
  }
  (void)gm_free(dev);
  return rc;
}

}  // extern "C"
