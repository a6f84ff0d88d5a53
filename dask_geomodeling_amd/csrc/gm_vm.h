// K1: the fused element-wise program kernel ("VM") for gfx950.
//
// One launch evaluates a straight-line accumulator program (gmhip.h: gm_program)
// over one tile.  HBM traffic = one read of every distinct leaf tile + one write of
// the result; everything in between lives in VGPRs.
//
// Execution shape (wave64, 256-thread workgroups, grid-stride):
//   * a lane owns U sub-vectors of 4 consecutive elements per block-iteration; for
//     every load instruction lane l reads 16 B (4-byte types) at base + l*16, i.e.
//     each wave-instruction covers one contiguous 1 KiB span (coalescing rule of
//     cdna_hip_programming.md section 2);
//   * all leaf tiles of the program are loaded first (NIN*U independent loads in
//     flight per lane), converted to the register class and tagged with their
//     native-dtype no-data bits; then the program is interpreted from SGPR-resident
//     step descriptors (kernarg -> s_load), every dispatch branch is wave-uniform
//     (scalar branches, no divergence, no dynamic VGPR indexing);
//   * every step re-applies the reference rule of raster/elemwise.py:263-296:
//     no-data = OR(operand == its ndv), compute in the step dtype, non-finite -> fill,
//     no-data -> fill.
//
// No MFMA: the path is element-wise and HBM-bound (12 B/px for the canonical chain).
#pragma once
#include <cmath>
#include <type_traits>

#include "gm_common.h"

namespace gmvm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef long long i64x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));

struct VmArgs {
  gm_program prog;
  const void* in[GM_MAX_INPUTS];
  void* out;
  unsigned long long n;         // total elements
  unsigned long long n_groups;  // 4-element groups handled by the vector path
  unsigned long long n_iters;   // block-iterations of the vector path
  unsigned long long tail_start;  // first element of the scalar tail path
  unsigned int xcd_chunking;
  unsigned int nontemporal;  // bit0 loads, bit1 stores
};

template <typename T>
__device__ __forceinline__ T ld(const T* p, bool nt) {
  return nt ? __builtin_nontemporal_load(p) : *p;
}
template <typename T>
__device__ __forceinline__ void st(T* p, T v, bool nt) {
  if (nt)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// ---- register-class traits ---------------------------------------------------
template <typename RC> struct is_float_class { static constexpr bool value = false; };
template <> struct is_float_class<float> { static constexpr bool value = true; };
template <> struct is_float_class<double> { static constexpr bool value = true; };

template <typename RC> __device__ __forceinline__ RC scalar_of(const gm_scalar& s);
template <> __device__ __forceinline__ float scalar_of<float>(const gm_scalar& s) { return s.f32; }
template <> __device__ __forceinline__ double scalar_of<double>(const gm_scalar& s) { return s.f64; }
template <> __device__ __forceinline__ int scalar_of<int>(const gm_scalar& s) { return s.i32; }
template <> __device__ __forceinline__ long long scalar_of<long long>(const gm_scalar& s) { return (long long)s.i64; }

// ---- typed loads of N (1 or 4) consecutive elements starting at element e ----
// returns the native-dtype no-data bits (bit k: element k == ndv)
template <typename RC, int N>
__device__ __forceinline__ unsigned load_elems(const void* base, int dtype, unsigned long long e, bool nt,
                                               bool has_ndv, const gm_scalar& ndv, RC (&v)[N]) {
  unsigned nd = 0;
#define GM_LOAD_CASE(DT, T, VT4, NATIVE_T, NDVEXPR)                                  \
  case DT: {                                                                         \
    NATIVE_T nv = (NATIVE_T)(NDVEXPR);                                               \
    if constexpr (N == 4) {                                                          \
      VT4 x = ld((const VT4*)((const T*)base + e), nt);                              \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                \
        T t = x[k];                                                                  \
        v[k] = (RC)t;                                                                \
        if (has_ndv && (NATIVE_T)t == nv) nd |= 1u << k;                             \
      }                                                                              \
    } else {                                                                         \
      T t = ld((const T*)base + e, nt);                                              \
      v[0] = (RC)t;                                                                  \
      if (has_ndv && (NATIVE_T)t == nv) nd |= 1u;                                    \
    }                                                                                \
  } break;
#define GM_LOAD_CASE8(DT, T, VT2, NATIVE_T, NDVEXPR)                                 \
  case DT: {                                                                         \
    NATIVE_T nv = (NATIVE_T)(NDVEXPR);                                               \
    if constexpr (N == 4) {                                                          \
      VT2 x0 = ld((const VT2*)((const T*)base + e), nt);                             \
      VT2 x1 = ld((const VT2*)((const T*)base + e + 2), nt);                         \
      T t[4] = {(T)x0[0], (T)x0[1], (T)x1[0], (T)x1[1]};                             \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                \
        v[k] = (RC)t[k];                                                             \
        if (has_ndv && (NATIVE_T)t[k] == nv) nd |= 1u << k;                          \
      }                                                                              \
    } else {                                                                         \
      T t = ld((const T*)base + e, nt);                                              \
      v[0] = (RC)t;                                                                  \
      if (has_ndv && (NATIVE_T)t == nv) nd |= 1u;                                    \
    }                                                                                \
  } break;
  switch (dtype) {
    GM_LOAD_CASE(GM_F32, float, f32x4, float, ndv.f64)
    GM_LOAD_CASE8(GM_F64, double, f64x2, double, ndv.f64)
    GM_LOAD_CASE(GM_I32, int, i32x4, long long, ndv.i64)
    GM_LOAD_CASE(GM_U32, unsigned int, u32x4, long long, ndv.i64)
    GM_LOAD_CASE8(GM_I64, long long, i64x2, long long, ndv.i64)
    GM_LOAD_CASE8(GM_U64, unsigned long long, i64x2, unsigned long long, ndv.u64)
    GM_LOAD_CASE(GM_U16, unsigned short, u16x4, long long, ndv.i64)
    GM_LOAD_CASE(GM_I16, short, u16x4, long long, ndv.i64)
    GM_LOAD_CASE(GM_U8, unsigned char, u8x4, long long, ndv.i64)
    GM_LOAD_CASE(GM_I8, signed char, u8x4, long long, ndv.i64)
    GM_LOAD_CASE(GM_BOOL, unsigned char, u8x4, long long, ndv.i64)
    default: break;
  }
#undef GM_LOAD_CASE
#undef GM_LOAD_CASE8
  return nd;
}

template <typename RC, int N>
__device__ __forceinline__ void store_elems(void* base, int dtype, unsigned long long e, bool nt, const RC (&v)[N]) {
#define GM_STORE_CASE(DT, T, VT4)                                                    \
  case DT: {                                                                         \
    if constexpr (N == 4) {                                                          \
      VT4 x;                                                                         \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) x[k] = (T)v[k];                  \
      st((VT4*)((T*)base + e), x, nt);                                               \
    } else {                                                                         \
      st((T*)base + e, (T)v[0], nt);                                                 \
    }                                                                                \
  } break;
#define GM_STORE_CASE8(DT, T, VT2)                                                   \
  case DT: {                                                                         \
    if constexpr (N == 4) {                                                          \
      VT2 x0, x1;                                                                    \
      x0[0] = (T)v[0]; x0[1] = (T)v[1]; x1[0] = (T)v[2]; x1[1] = (T)v[3];            \
      st((VT2*)((T*)base + e), x0, nt);                                              \
      st((VT2*)((T*)base + e + 2), x1, nt);                                          \
    } else {                                                                         \
      st((T*)base + e, (T)v[0], nt);                                                 \
    }                                                                                \
  } break;
  switch (dtype) {
    GM_STORE_CASE(GM_F32, float, f32x4)
    GM_STORE_CASE8(GM_F64, double, f64x2)
    GM_STORE_CASE(GM_I32, int, i32x4)
    GM_STORE_CASE(GM_U32, unsigned int, u32x4)
    GM_STORE_CASE8(GM_I64, long long, i64x2)
    GM_STORE_CASE8(GM_U64, long long, i64x2)
    GM_STORE_CASE(GM_U16, unsigned short, u16x4)
    GM_STORE_CASE(GM_I16, unsigned short, u16x4)
    GM_STORE_CASE(GM_U8, unsigned char, u8x4)
    GM_STORE_CASE(GM_I8, unsigned char, u8x4)
    GM_STORE_CASE(GM_BOOL, unsigned char, u8x4)
    default: break;
  }
#undef GM_STORE_CASE
#undef GM_STORE_CASE8
}

// ---- arithmetic in the register class ------------------------------------------
template <typename RC> __device__ __forceinline__ RC op_add(RC a, RC b) { return a + b; }
template <typename RC> __device__ __forceinline__ RC op_sub(RC a, RC b) { return a - b; }
template <typename RC> __device__ __forceinline__ RC op_mul(RC a, RC b) { return a * b; }
// integers wrap silently like NumPy under errstate(all="ignore")
template <> __device__ __forceinline__ int op_add<int>(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
template <> __device__ __forceinline__ int op_sub<int>(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
template <> __device__ __forceinline__ int op_mul<int>(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
template <> __device__ __forceinline__ long long op_add<long long>(long long a, long long b) {
  return (long long)((unsigned long long)a + (unsigned long long)b);
}
template <> __device__ __forceinline__ long long op_sub<long long>(long long a, long long b) {
  return (long long)((unsigned long long)a - (unsigned long long)b);
}
template <> __device__ __forceinline__ long long op_mul<long long>(long long a, long long b) {
  return (long long)((unsigned long long)a * (unsigned long long)b);
}

template <typename RC> __device__ __forceinline__ RC op_div(RC a, RC b) {
  if constexpr (is_float_class<RC>::value) return a / b;
  else return 0;  // Divide is always a float step (elemwise.py:377-380)
}

// np.power: float classes go through float64 pow and one rounding (<= 1 ulp of any
// faithful powf); integer classes use exponentiation by squaring with wrap-around.
template <typename RC> __device__ __forceinline__ RC op_pow(RC a, RC b) {
  if constexpr (is_float_class<RC>::value) {
    return (RC)pow((double)a, (double)b);
  } else {
    typedef typename std::conditional<sizeof(RC) == 4, unsigned, unsigned long long>::type U;
    U base = (U)a, r = 1;
    unsigned long long e = (unsigned long long)b;  // host guarantees b >= 0
    while (e) {
      if (e & 1) r *= base;
      base *= base;
      e >>= 1;
    }
    return (RC)r;
  }
}

template <typename RC> __device__ __forceinline__ RC op_unary(int opcode, RC a) {
  if constexpr (is_float_class<RC>::value) {
    double x = (double)a;
    double r = opcode == GM_OP_EXP ? exp(x) : (opcode == GM_OP_LOG ? log(x) : log10(x));
    return (RC)r;
  } else {
    return a;
  }
}

template <typename RC> __device__ __forceinline__ bool finite_rc(RC v) {
  if constexpr (is_float_class<RC>::value) return isfinite(v);
  else return true;
}

// np.isclose(x, y) of NumPy 2.2 as used by utils.get_index (utils.py:61-64):
//   (|x - y| <= atol + rtol*|y|) & isfinite(y) | (x == y); evaluated in f32 or f64
template <typename RC>
__device__ __forceinline__ bool is_close(RC x, double y, double tol, bool in_f64, bool y_finite) {
  if (in_f64) {
    double xd = (double)x;
    return ((fabs(xd - y) <= tol) && y_finite) || (xd == y);
  } else {
    float xf = (float)x, yf = (float)y;
    return ((fabsf(xf - yf) <= (float)tol) && y_finite) || (xf == yf);
  }
}

// ---- the interpreter -------------------------------------------------------------
template <typename RC, int N, int NIN, bool HEAVY>
struct Machine {
  RC in[NIN][N];
  unsigned in_nd[NIN];
  RC tmp[GM_MAX_TEMPS][N];
  RC acc[N];

  // operand fetch: value + no-data bits; every branch below is wave-uniform
  __device__ __forceinline__ unsigned fetch(const gm_operand& o, RC (&v)[N]) const {
    unsigned nd = 0;
    switch (o.kind) {
      case GM_OPD_INPUT: {
        switch (o.index) {
#define GM_IN_CASE(K)                                                    \
  case K:                                                                \
    if constexpr (K < NIN) {                                             \
      _Pragma("unroll") for (int k = 0; k < N; ++k) v[k] = in[K][k];     \
      nd = in_nd[K];                                                     \
    }                                                                    \
    break;
          GM_IN_CASE(0) GM_IN_CASE(1) GM_IN_CASE(2) GM_IN_CASE(3)
          GM_IN_CASE(4) GM_IN_CASE(5) GM_IN_CASE(6) GM_IN_CASE(7)
#undef GM_IN_CASE
          default: break;
        }
      } break;
      case GM_OPD_SCALAR: {
        RC s = scalar_of<RC>(o.value);
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = s;
      } break;
      case GM_OPD_TEMP: {
        switch (o.index) {
#define GM_TMP_CASE(K)                                                   \
  case K:                                                                \
    _Pragma("unroll") for (int k = 0; k < N; ++k) v[k] = tmp[K][k];      \
    break;
          GM_TMP_CASE(0) GM_TMP_CASE(1) GM_TMP_CASE(2) GM_TMP_CASE(3)
#undef GM_TMP_CASE
          default: break;
        }
        if (o.has_ndv) {
          RC ndv = scalar_of<RC>(o.value);
#pragma unroll
          for (int k = 0; k < N; ++k)
            if (v[k] == ndv) nd |= 1u << k;
        }
      } break;
      case GM_OPD_ACC: {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = acc[k];
        if (o.has_ndv) {
          RC ndv = scalar_of<RC>(o.value);
#pragma unroll
          for (int k = 0; k < N; ++k)
            if (v[k] == ndv) nd |= 1u << k;
        }
      } break;
      default: {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = (RC)0;
      } break;
    }
    return nd;
  }

  __device__ __forceinline__ void run(const gm_program& P) {
    const unsigned n_steps = P.n_steps;
    for (unsigned s = 0; s < n_steps; ++s) {
      const gm_step& S = P.steps[s];
      RC a[N], b[N], r[N];
      unsigned nda = fetch(S.a, a);
      unsigned ndb = fetch(S.b, b);
      unsigned nd = nda | ndb;
      const RC fill = scalar_of<RC>(S.fill);
      const int opcode = S.opcode;
      switch (opcode) {
        case GM_OP_ADD:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = op_add<RC>(a[k], b[k]);
          break;
        case GM_OP_SUB:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = op_sub<RC>(a[k], b[k]);
          break;
        case GM_OP_MUL:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = op_mul<RC>(a[k], b[k]);
          break;
        case GM_OP_DIV:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = op_div<RC>(a[k], b[k]);
          break;
        case GM_OP_POW:
          if constexpr (HEAVY) {
#pragma unroll
            for (int k = 0; k < N; ++k) r[k] = op_pow<RC>(a[k], b[k]);
          }
          break;
        case GM_OP_EQ:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)(a[k] == b[k]);
          break;
        case GM_OP_NE:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)(a[k] != b[k]);
          break;
        case GM_OP_GT:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)(a[k] > b[k]);
          break;
        case GM_OP_GE:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)(a[k] >= b[k]);
          break;
        case GM_OP_LT:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)(a[k] < b[k]);
          break;
        case GM_OP_LE:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)(a[k] <= b[k]);
          break;
        case GM_OP_AND:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)((a[k] != (RC)0) && (b[k] != (RC)0));
          break;
        case GM_OP_OR:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)((a[k] != (RC)0) || (b[k] != (RC)0));
          break;
        case GM_OP_XOR:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)((a[k] != (RC)0) != (b[k] != (RC)0));
          break;
        case GM_OP_EXP:
        case GM_OP_LOG:
        case GM_OP_LOG10:
          if constexpr (HEAVY) {
#pragma unroll
            for (int k = 0; k < N; ++k) r[k] = op_unary<RC>(opcode, a[k]);
          }
          break;
        case GM_OP_INVERT:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)(a[k] == (RC)0);
          nd = 0;
          break;
        case GM_OP_ISDATA:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)(((nda >> k) & 1u) == 0u);
          nd = 0;
          break;
        case GM_OP_ISNODATA:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (RC)((nda >> k) & 1u);
          nd = 0;
          break;
        case GM_OP_CLIP: {
          // misc.py:115-123: drop where the mask source is False (bool) / no-data
          unsigned drop = ndb;
          if (S.b.is_bool) {
#pragma unroll
            for (int k = 0; k < N; ++k)
              if (b[k] == (RC)0) drop |= 1u << k;
          }
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = a[k];
          nd = drop;  // fill == the store's own no-data value
        } break;
        case GM_OP_MASK: {
          // misc.py:213-221: value where get_index() says data, else fill
          const RC value = scalar_of<RC>(S.aux0);
          unsigned nodata_bits;
          if (S.flags & 4u) {
            nodata_bits = nda;  // integer tiles: exact !=
          } else {
            nodata_bits = 0;
            const double y = S.aux2.f64;
            const double tol = S.aux1.f64;
            const bool in_f64 = (S.flags & 1u) != 0, y_finite = (S.flags & 2u) != 0;
#pragma unroll
            for (int k = 0; k < N; ++k)
              if (is_close<RC>(a[k], y, tol, in_f64, y_finite)) nodata_bits |= 1u << k;
          }
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = value;
          nd = nodata_bits;
        } break;
        case GM_OP_COPY:
        default:
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = a[k];
          break;
      }
      // step dtype narrower than the register class
      if constexpr (std::is_same<RC, double>::value) {
        if (S.round_f32) {
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (double)(float)r[k];
        }
      }
      if constexpr (std::is_same<RC, long long>::value) {
        if (S.wrap_i32) {
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = (long long)(int)r[k];
        }
      }
      // elemwise.py:292-296: non-finite -> fill, then input no-data -> fill
      if constexpr (is_float_class<RC>::value) {
        if (S.check_finite) {
#pragma unroll
          for (int k = 0; k < N; ++k)
            if (!finite_rc<RC>(r[k])) nd |= 1u << k;
        }
      }
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] = ((nd >> k) & 1u) ? fill : r[k];
      if (S.save_temp) {
        switch (S.save_temp - 1) {
#define GM_SAVE_CASE(K)                                                  \
  case K:                                                                \
    _Pragma("unroll") for (int k = 0; k < N; ++k) tmp[K][k] = acc[k];    \
    break;
          GM_SAVE_CASE(0) GM_SAVE_CASE(1) GM_SAVE_CASE(2) GM_SAVE_CASE(3)
#undef GM_SAVE_CASE
          default: break;
        }
      }
    }
  }
};

// XCD-aware iteration order: workgroups are dealt round-robin over the 8 XCDs
// (MI355X_MICROARCH.md "Workgroup dispatch"); with chunking on, the blocks that share
// an XCD walk one contiguous eighth of the tile so each XCD's L2/fabric port streams
// its own address range.  Placement affects speed only, never correctness.
__device__ __forceinline__ unsigned long long first_iter(unsigned long long n_iters, unsigned xcd_chunking,
                                                        unsigned long long* stride, unsigned long long* end) {
  if (xcd_chunking && gridDim.x >= 8 && (gridDim.x % 8) == 0) {
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned long long per = (n_iters + 7) / 8;
    const unsigned long long lo = per * xcd;
    unsigned long long hi = lo + per;
    if (hi > n_iters) hi = n_iters;
    *stride = gridDim.x / 8;
    *end = hi;
    return lo + (blockIdx.x >> 3);
  }
  *stride = gridDim.x;
  *end = n_iters;
  return blockIdx.x;
}

// The kernel.  HEAVY instantiations carry pow/exp/log (their register appetite would
// otherwise cap the occupancy of every program); light programs use HEAVY = false.
template <typename RC, int NIN, int U, bool HEAVY>
__global__ __launch_bounds__(256) void vm_kernel(const VmArgs A) {
  const gm_program& P = A.prog;
  const bool nt_ld = (A.nontemporal & 1u) != 0, nt_st = (A.nontemporal & 2u) != 0;
  const unsigned n_in = P.n_inputs;
  const int out_dtype = P.out_dtype;
  constexpr int N = 4 * U;

  unsigned long long stride, end;
  unsigned long long it = first_iter(A.n_iters, A.xcd_chunking, &stride, &end);
  for (; it < end; it += stride) {
    Machine<RC, N, NIN, HEAVY> M;
    const unsigned long long g0 = it * (256ull * U) + threadIdx.x;  // sub-vector u is group g0 + 256*u
    // phase 1: all leaf loads in flight (NIN*U independent 16-byte loads per lane)
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      if (k < (int)n_in) {
        const gm_input_desc& D = P.inputs[k];
        if (D.broadcast) {
          RC one[1];
          unsigned nd1 = load_elems<RC, 1>(A.in[k], D.dtype, 0, false, D.has_ndv, D.ndv, one);
#pragma unroll
          for (int q = 0; q < N; ++q) M.in[k][q] = one[0];
          M.in_nd[k] = nd1 ? ((1u << N) - 1u) : 0u;
        } else {
          unsigned nd = 0;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            RC v4[4] = {(RC)0, (RC)0, (RC)0, (RC)0};
            unsigned nd4 = 0;
            const unsigned long long g = g0 + 256ull * u;
            if (g < A.n_groups) nd4 = load_elems<RC, 4>(A.in[k], D.dtype, g * 4ull, nt_ld, D.has_ndv, D.ndv, v4);
#pragma unroll
            for (int q = 0; q < 4; ++q) M.in[k][4 * u + q] = v4[q];
            nd |= nd4 << (4 * u);
          }
          M.in_nd[k] = nd;
        }
      }
    }
    // phase 2: interpret; phase 3: store
    M.run(P);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long g = g0 + 256ull * u;
      if (g < A.n_groups) {
        RC v4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v4[q] = M.acc[4 * u + q];
        store_elems<RC, 4>(A.out, out_dtype, g * 4ull, nt_st, v4);
      }
    }
  }

  // scalar tail: elements the 16-byte vector path cannot take (n % 4, misaligned views)
  const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long e = A.tail_start + tid; e < A.n; e += nthreads) {
    Machine<RC, 1, NIN, HEAVY> M1;
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      if (k < (int)n_in) {
        const gm_input_desc& D = P.inputs[k];
        M1.in_nd[k] = load_elems<RC, 1>(A.in[k], D.dtype, D.broadcast ? 0ull : e, false, D.has_ndv, D.ndv, M1.in[k]);
      }
    }
    M1.run(P);
    store_elems<RC, 1>(A.out, out_dtype, e, false, M1.acc);
  }
}

// host-side launcher of one register class (defined once per class in gm_vm_<class>.hip)
template <typename RC>
int launch_class(const VmArgs& args, int n_inputs, int U, bool heavy, dim3 grid, hipStream_t stream);

#define GM_VM_DEFINE_LAUNCHER(RC)                                                                  \
  template <>                                                                                      \
  int launch_class<RC>(const VmArgs& args, int n_inputs, int U, bool heavy, dim3 grid,             \
                       hipStream_t stream) {                                                       \
    const bool small = n_inputs <= 4;                                                              \
    if (heavy) {                                                                                   \
      if (small) hipLaunchKernelGGL((vm_kernel<RC, 4, 1, true>), grid, dim3(256), 0, stream, args); \
      else hipLaunchKernelGGL((vm_kernel<RC, 8, 1, true>), grid, dim3(256), 0, stream, args);      \
    } else if (U == 2) {                                                                           \
      if (small) hipLaunchKernelGGL((vm_kernel<RC, 4, 2, false>), grid, dim3(256), 0, stream, args); \
      else hipLaunchKernelGGL((vm_kernel<RC, 8, 2, false>), grid, dim3(256), 0, stream, args);     \
    } else {                                                                                       \
      if (small) hipLaunchKernelGGL((vm_kernel<RC, 4, 1, false>), grid, dim3(256), 0, stream, args); \
      else hipLaunchKernelGGL((vm_kernel<RC, 8, 1, false>), grid, dim3(256), 0, stream, args);     \
    }                                                                                              \
    hipError_t e = hipGetLastError();                                                              \
    if (e != hipSuccess) return gm::hip_fail(e, "vm_kernel launch");                               \
    return GM_OK;                                                                                  \
  }

}  // namespace gmvm
