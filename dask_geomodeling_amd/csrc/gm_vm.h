// K1: the fused element-wise program kernel ("VM") for gfx950.
//
// One launch evaluates a straight-line accumulator program (gmhip.h: gm_program) over a batch of
// up to GM_MAX_BATCH tiles (blockIdx.y = tile).  HBM traffic = one read of every distinct leaf
// tile + one write of the result; everything in between lives in VGPRs.
//
// Execution shape (wave64, 256-thread workgroups), chosen from measurements on MI355X
// (tools/ceiling*.hip):
//   * FULL GRID, no grid-stride loop: a workgroup owns 256*G groups of 4 consecutive elements and
//     ends.  A 2-read/1-write float4 stream runs 8-10 % faster this way than with persistent
//     workgroups (fresh waves issue their loads at once; the dispatcher balances the CUs);
//   * for every load instruction lane l reads 16 B at base + 16*l: one contiguous 1 KiB span per
//     wave-instruction; loads and stores carry the non-temporal hint (the tiles are streamed once;
//     +10 % on the same measurement);
//   * all leaf loads are issued first, then the program is interpreted.  The lowered program
//     (VmCode) lives in device memory, uploaded once per distinct program, and is read with scalar
//     loads (s_load_dwordx8 per micro-op); every dispatch branch is wave-uniform.  Decode is a
//     two-level compare tree (operand source, then operation), about 10 SALU instructions per
//     micro-op, amortised over the 4*G*64 elements a wave owns;
//   * every step re-applies the reference rule of raster/elemwise.py:263-296:
//     no-data = OR(operand == its ndv), compute in the step dtype, non-finite -> fill,
//     no-data -> fill.
//
// No MFMA: the path is element-wise and HBM-bound (12 B/px for the canonical chain).
#pragma once
#ifdef __HIPCC_RTC__
// Compiled at run time by hiprtc (gm_spec.hip: kernels specialised per program): no host headers there.
// hiprtc predeclares the device runtime (math functions, __umulhi, threadIdx ...); what this file uses
// of the standard library is declared here.
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t;
#define offsetof(t, m) __builtin_offsetof(t, m)
namespace std {
template <typename A, typename B> struct is_same { static constexpr bool value = false; };
template <typename A> struct is_same<A, A> { static constexpr bool value = true; };
template <bool C, typename A, typename B> struct conditional { typedef A type; };
template <typename A, typename B> struct conditional<false, A, B> { typedef B type; };
template <typename T> struct is_integral { static constexpr bool value = false; };
template <> struct is_integral<signed char> { static constexpr bool value = true; };
template <> struct is_integral<unsigned char> { static constexpr bool value = true; };
template <> struct is_integral<short> { static constexpr bool value = true; };
template <> struct is_integral<unsigned short> { static constexpr bool value = true; };
template <> struct is_integral<int> { static constexpr bool value = true; };
template <> struct is_integral<unsigned int> { static constexpr bool value = true; };
template <> struct is_integral<long long> { static constexpr bool value = true; };
template <> struct is_integral<unsigned long long> { static constexpr bool value = true; };
}  // namespace std
#include "gmhip.h"
#else
#include <cmath>
#include <cstddef>
#include <type_traits>

#include "gm_common.h"
#endif

namespace gmvm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef long long i64x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));
typedef signed char i8x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// Micro-ops: gm_elemwise_run lowers every gm_step into 1-3 of these.  The four arithmetic
// ops and Clip -- the bulk of real views -- get "fast routes": one flat wave-uniform switch per
// micro-op selects a fully static body (operand arrays named at compile time: no operand copies,
// no per-operand dispatch).  Everything else runs the generic step interpreter (UOP_GENERIC).
enum { UOP_GENERIC = 0, UOP_LD = 1, UOP_ST = 2, UOP_ADD = 3, UOP_SUB = 4, UOP_RSUB = 5, UOP_MUL = 6,
       UOP_DIV = 7, UOP_RDIV = 8, UOP_CLIP = 9,
       // comparisons acc OP b (a comparison feeding Clip is the idiomatic masking pattern)
       UOP_EQ = 10, UOP_NE = 11, UOP_GT = 12, UOP_GE = 13, UOP_LT = 14, UOP_LE = 15,
       // unary ops on the accumulator whose parameters fit the 32-byte micro-op (no second fetch of
       // the 80-byte step, no walk through the generic opcode switch):
       //   DIGITIZE arg: bit0 acc has ndv, bit1 right, bit2 int64 edges, bits 8-15 edges, 16-17 table;
       //            acc_ndv = ndv of the accumulator, fill = fill
       //   LOOKUP   arg: bit0 select, bits 8-11 exception keys, 16-17 table; acc_ndv.u64 = window
       //            slots, b_val.i64 = kmin, fill = value of a miss under select
       //   MASKI    (integer tiles) arg: bit0 acc has ndv; acc_ndv = ndv, b_val = value, fill = fill
       UOP_DIGITIZE = 16, UOP_LOOKUP = 17, UOP_MASKI = 18,
       // Clip(acc, y CMP literal) in one micro-op -- the comparison whose only use is the mask of a
       // Clip never materialises (saves LD y, CMP, ST temp and the operand fetch of the CLIP):
       //   arg: bit0 y has ndv, bit1 a no-data y drops the cell (the comparison's fill is False),
       //        bits 8-10 comparison (0..5 = EQ NE GT GE LT LE); acc_ndv = ndv of y, b_val = literal,
       //        fill = the store's no-data value
       UOP_CLIPCMP = 19 };
// operand source of a micro-op: 0 = literal, 1 + k = input k, 1 + 8 + k = temp k
constexpr unsigned UOP_SRC_SCALAR = 0, UOP_SRC_INPUT = 1, UOP_SRC_TEMP = 1 + GM_MAX_INPUTS;
constexpr int GM_MAX_UOPS = 40;

struct gm_uop {
  unsigned int route;  // kind | src << 8
  unsigned int arg;    // GENERIC: step index; fast routes: bit0 acc has ndv, bit1 b has ndv, bit2 b is
                       // bool, bit3 round_f32, bit4 wrap_i32
  gm_scalar acc_ndv;   // ndv of the value in the accumulator (register-class repr)
  gm_scalar b_val;     // literal, or ndv of the b operand
  gm_scalar fill;
};

constexpr int GM_MAX_BATCH = 16;

// The lowered program: device resident, one per distinct gm_program (cached by content on the host).
struct VmCode {
  unsigned long long tables[GM_MAX_TABLES];  // device addresses of the program's lookup tables
  gm_input_desc inputs[GM_MAX_INPUTS];
  gm_uop uops[GM_MAX_UOPS];
  gm_step steps[GM_MAX_STEPS];  // bodies of the UOP_GENERIC micro-ops
};

// Launch arguments (kernarg segment, 1.3 KB).
struct VmLaunch {
  unsigned long long code;        // device address of the VmCode (an integer: the kernel turns it straight
                                  // into a constant-address-space pointer, so the program is read with s_load)
  unsigned long long n;           // elements per tile
  unsigned long long n_groups;    // 4-element groups per tile handled by the vector kernel
  unsigned long long tail_start;  // first element of the scalar tail kernel
  unsigned int n_uops;
  unsigned int n_inputs;
  unsigned int out_dtype;
  unsigned int flags;             // bit2 native: every input and the output are non-broadcast tiles of the
                                  // class' own dtype
  unsigned int lds_table;         // TABLES kernels: 1 + index of the LOOKUP table staged in LDS (0 = none)
  unsigned int lds_words;         //   its dense window, in 8-byte words
  unsigned int spans;             //   spans of 256*G groups a workgroup walks (amortises the staging)
  unsigned int cols;              // strided inputs (windows of a larger resident raster): tile width in elements
  // group -> row of a strided tile: g / (cols / 4) as multiply-high + shifts (division by an invariant,
  // Granlund & Montgomery 1994, fig. 4.1: exact for every 32-bit g), instead of a ~25-instruction
  // 32-bit division per group
  unsigned int div_magic, div_sh1, div_sh2, reserved0;
  unsigned int pitch[GM_MAX_BATCH][GM_MAX_INPUTS];  // row pitch in elements, 0 = contiguous
  unsigned int out_pitch[GM_MAX_BATCH];             // the same for the result (a window of a mosaic)
  const void* in[GM_MAX_BATCH][GM_MAX_INPUTS];
  void* out[GM_MAX_BATCH];
};
constexpr unsigned VM_FLAG_NATIVE = 4u, VM_FLAG_STRIDED = 8u;

// Launch arguments and code are read through the constant address space with explicit typed
// loads -> s_load_dword*; the by-value kernel parameter itself is never touched, so the compiler
// cannot decide to keep a private copy of it in scratch.
typedef const __attribute__((address_space(4))) char* kargs_t;
template <typename T>
__device__ __forceinline__ T kload(kargs_t base, size_t offset) {
  // word-typed loads: the natural alignment of the words is what lets the compiler pick SMEM
  // (a byte-typed memcpy from a pointer of unknown alignment becomes vector loads + readfirstlane)
  if constexpr (sizeof(T) % 8 == 0) {
    typedef const __attribute__((address_space(4))) unsigned long long* w64_t;
    union { T v; unsigned long long w[sizeof(T) / 8]; } u;
#pragma unroll
    for (size_t j = 0; j < sizeof(T) / 8; ++j) u.w[j] = ((w64_t)(base + offset))[j];
    return u.v;
  } else {
    static_assert(sizeof(T) % 4 == 0, "kload: 4-byte granularity");
    typedef const __attribute__((address_space(4))) unsigned int* w32_t;
    union { T v; unsigned int w[sizeof(T) / 4]; } u;
#pragma unroll
    for (size_t j = 0; j < sizeof(T) / 4; ++j) u.w[j] = ((w32_t)(base + offset))[j];
    return u.v;
  }
}
#define GM_KARG(base, member) kload<decltype(VmLaunch::member)>(base, offsetof(VmLaunch, member))

// Tiles are streamed exactly once: every tile load and store carries the non-temporal hint
// (+10 % on a 2-read/1-write float4 stream, tools/ceiling2.hip).  `nt` is a compile-time
// constant at every call site (a run-time flag gets if-converted and the hint is lost).
// Tile pointers come out of the kernarg tables as integers, i.e. as generic pointers: without the
// address-space cast they are flat_load / flat_store, which also count against lgkmcnt (every
// scalar load of a micro-op then waits for the tile loads in flight).
template <typename T>
__device__ __forceinline__ T ld(const T* p, bool nt) {
  typedef const __attribute__((address_space(1))) T* global_t;
  const global_t g = (global_t)(uintptr_t)p;
  return nt ? __builtin_nontemporal_load(g) : *g;
}
template <typename T>
__device__ __forceinline__ void st(T* p, T v, bool nt) {
  typedef __attribute__((address_space(1))) T* global_t;
  const global_t g = (global_t)(uintptr_t)p;
  if (nt)
    __builtin_nontemporal_store(v, g);
  else
    *g = v;
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

// a literal of a generated kernel: the 8 bytes of a gm_scalar -> the register class' value
template <typename RC> __device__ __forceinline__ RC lit(unsigned long long bits) {
  gm_scalar s;
  s.u64 = bits;
  return scalar_of<RC>(s);
}
// packs of such literals (gm_spec.hip hands tables over as template arguments), and a DIGITIZE edge
template <unsigned long long... V> struct Bits {};
template <typename E> __device__ __forceinline__ E edge_of(unsigned long long bits) {
  if constexpr (std::is_same<E, float>::value) return __builtin_bit_cast(float, (unsigned)bits);
  else if constexpr (std::is_same<E, double>::value) return __builtin_bit_cast(double, bits);
  else return (E)(long long)bits;
}

// ---- typed loads of N (1 or 4) consecutive elements starting at element e ----
// `native` (64-bit integer tiles in the F64 class only): cells equal to nv -- compared as
// integers -- are replaced by the sentinel 0.5 (gmhip.h: gm_input_desc.native_ndv)
template <typename RC, int N>
__device__ __forceinline__ void load_elems(const void* base, int dtype, unsigned long long e, bool nt, RC (&v)[N],
                                           bool native = false, long long nv = 0) {
#define GM_LOAD_CASE(DT, T, VT4)                                                     \
  case DT: {                                                                         \
    if constexpr (N == 4) {                                                          \
      VT4 x = ld((const VT4*)((const T*)base + e), nt);                              \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) v[k] = (RC)(T)x[k];              \
    } else {                                                                         \
      v[0] = (RC)ld((const T*)base + e, nt);                                         \
    }                                                                                \
  } break;
#define GM_LOAD_CASE8(DT, T, VT2)                                                    \
  case DT: {                                                                         \
    if constexpr (N == 4) {                                                          \
      VT2 x0 = ld((const VT2*)((const T*)base + e), nt);                             \
      VT2 x1 = ld((const VT2*)((const T*)base + e + 2), nt);                         \
      const T t[4] = {(T)x0[0], (T)x0[1], (T)x1[0], (T)x1[1]};                       \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) v[k] = (RC)t[k];                 \
      if constexpr (std::is_same<RC, double>::value && std::is_integral<T>::value) { \
        if (native) {                                                                \
          _Pragma("unroll") for (int k = 0; k < 4; ++k)                              \
            if ((long long)t[k] == nv) v[k] = (RC)0.5;                               \
        }                                                                            \
      }                                                                              \
    } else {                                                                         \
      const T t = ld((const T*)base + e, nt);                                        \
      v[0] = (RC)t;                                                                  \
      if constexpr (std::is_same<RC, double>::value && std::is_integral<T>::value) { \
        if (native && (long long)t == nv) v[0] = (RC)0.5;                            \
      }                                                                              \
    }                                                                                \
  } break;
  switch (dtype) {
    GM_LOAD_CASE(GM_F32, float, f32x4)
    GM_LOAD_CASE8(GM_F64, double, f64x2)
    GM_LOAD_CASE(GM_I32, int, i32x4)
    GM_LOAD_CASE(GM_U32, unsigned int, u32x4)
    GM_LOAD_CASE8(GM_I64, long long, i64x2)
    GM_LOAD_CASE8(GM_U64, unsigned long long, u64x2)
    GM_LOAD_CASE(GM_U16, unsigned short, u16x4)
    GM_LOAD_CASE(GM_I16, short, i16x4)
    GM_LOAD_CASE(GM_U8, unsigned char, u8x4)
    GM_LOAD_CASE(GM_I8, signed char, i8x4)
    GM_LOAD_CASE(GM_BOOL, unsigned char, u8x4)
    default: break;
  }
#undef GM_LOAD_CASE
#undef GM_LOAD_CASE8
}

// float64 -> int64 of a result: the one value that matters beyond 2^63 - 1024 is the int64 fill value
// (the dtype's maximum, which float64 holds as 2^63): it must come out as INT64_MAX, like the reference's
// np.full(shape, fillvalue, int64) -- a plain cast of 2^63 is undefined
template <typename T, typename RC>
__device__ __forceinline__ T to_stored(RC v) {
  if constexpr (std::is_same<RC, double>::value && std::is_same<T, long long>::value) {
    return v >= 9223372036854775807.0 ? 9223372036854775807ll : (long long)v;
  } else if constexpr (std::is_same<RC, double>::value && std::is_same<T, unsigned long long>::value) {
    // (likewise the uint64 fill value, 2^64 as a float64; values from 2^63 on -- reduce_rasters of uint64 rasters works in
    //  float64 -- must not go through int64)
    return v >= 18446744073709551615.0 ? 18446744073709551615ull : (unsigned long long)v;
  } else if constexpr (is_float_class<RC>::value && std::is_integral<T>::value && sizeof(T) < 4) {
    // float -> 1- / 2-byte integer: through int32 and a truncation, like NumPy's cast on x86 (700.0 -> 188).
    // A direct float -> uint8 conversion is undefined for values outside the type, and the compiler then
    // packs the four bytes of a group without masking: an out-of-range value would spill into its neighbours
    return (T)(int)v;
  } else {
    return (T)v;
  }
}

template <typename RC, int N>
__device__ __forceinline__ void store_elems(void* base, int dtype, unsigned long long e, bool nt, const RC (&v)[N]) {
#define GM_STORE_CASE(DT, T, VT4)                                                    \
  case DT: {                                                                         \
    if constexpr (N == 4) {                                                          \
      VT4 x;                                                                         \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) x[k] = to_stored<T, RC>(v[k]);   \
      st((VT4*)((T*)base + e), x, nt);                                               \
    } else {                                                                         \
      st((T*)base + e, to_stored<T, RC>(v[0]), nt);                                  \
    }                                                                                \
  } break;
#define GM_STORE_CASE8(DT, T, VT2)                                                   \
  case DT: {                                                                         \
    if constexpr (N == 4) {                                                          \
      VT2 x0, x1;                                                                    \
      x0[0] = to_stored<T, RC>(v[0]); x0[1] = to_stored<T, RC>(v[1]);                \
      x1[0] = to_stored<T, RC>(v[2]); x1[1] = to_stored<T, RC>(v[3]);                \
      st((VT2*)((T*)base + e), x0, nt);                                              \
      st((VT2*)((T*)base + e + 2), x1, nt);                                          \
    } else {                                                                         \
      st((T*)base + e, to_stored<T, RC>(v[0]), nt);                                  \
    }                                                                                \
  } break;
  switch (dtype) {
    GM_STORE_CASE(GM_F32, float, f32x4)
    GM_STORE_CASE8(GM_F64, double, f64x2)
    GM_STORE_CASE(GM_I32, int, i32x4)
    GM_STORE_CASE(GM_U32, unsigned int, u32x4)
    GM_STORE_CASE8(GM_I64, long long, i64x2)
    GM_STORE_CASE8(GM_U64, unsigned long long, u64x2)  /* (from the I64 class: the bit pattern it travelled as) */
    GM_STORE_CASE(GM_U16, unsigned short, u16x4)
    GM_STORE_CASE(GM_I16, short, i16x4)
    GM_STORE_CASE(GM_U8, unsigned char, u8x4)
    GM_STORE_CASE(GM_I8, signed char, i8x4)
    GM_STORE_CASE(GM_BOOL, unsigned char, u8x4)
    default: break;
  }
#undef GM_STORE_CASE
#undef GM_STORE_CASE8
}

// ---- arithmetic in the register class ------------------------------------------
// integers wrap silently like NumPy under errstate(all="ignore")
template <typename RC> __device__ __forceinline__ RC op_add(RC a, RC b) {
  if constexpr (std::is_same<RC, int>::value) return (int)((unsigned)a + (unsigned)b);
  else if constexpr (std::is_same<RC, long long>::value) return (long long)((unsigned long long)a + (unsigned long long)b);
  else return a + b;
}
template <typename RC> __device__ __forceinline__ RC op_sub(RC a, RC b) {
  if constexpr (std::is_same<RC, int>::value) return (int)((unsigned)a - (unsigned)b);
  else if constexpr (std::is_same<RC, long long>::value) return (long long)((unsigned long long)a - (unsigned long long)b);
  else return a - b;
}
template <typename RC> __device__ __forceinline__ RC op_mul(RC a, RC b) {
  if constexpr (std::is_same<RC, int>::value) return (int)((unsigned)a * (unsigned)b);
  else if constexpr (std::is_same<RC, long long>::value) return (long long)((unsigned long long)a * (unsigned long long)b);
  else return a * b;
}
template <typename RC> __device__ __forceinline__ RC op_div(RC a, RC b) {
  if constexpr (is_float_class<RC>::value) return a / b;
  else return 0;  // Divide is always a float step (elemwise.py:377-380)
}

// generated by tools/gen_pow_tables.py: {1 / c, log2(c)} for the 128 intervals of [0x3f338000, 2x) as float32 bits
static __device__ const double POWF_LOG2_TAB[128][2] = {
    {0x1.6c16c16c16c17p+0, -0x1.042bd4b9a7c99p-1}, {0x1.6a13cd1537290p+0, -0x1.0014332be0032p-1},
    {0x1.6816816816817p+0, -0x1.f804ae8d0cd04p-2}, {0x1.661ec6a5122f9p+0, -0x1.efec61b011f85p-2},
    {0x1.642c8590b2164p+0, -0x1.e7df5fe538ab3p-2}, {0x1.623fa77016240p+0, -0x1.dfdd89d586e2cp-2},
    {0x1.6058160581606p+0, -0x1.d7e6c0abc357bp-2}, {0x1.5e75bb8d015e7p+0, -0x1.cffae611ad12ap-2},
    {0x1.5c9882b931057p+0, -0x1.c819dc2d45fe4p-2}, {0x1.5ac056b015ac0p+0, -0x1.c043859e2fdb2p-2},
    {0x1.58ed2308158edp+0, -0x1.b877c57b1b06fp-2}, {0x1.571ed3c506b3ap+0, -0x1.b0b67f4f46812p-2},
    {0x1.5555555555555p+0, -0x1.a8ff971810a5dp-2}, {0x1.5390948f40febp+0, -0x1.a152f142981b5p-2},
    {0x1.51d07eae2f815p+0, -0x1.99b072a96c6b2p-2}, {0x1.5015015015015p+0, -0x1.921800924dd3bp-2},
    {0x1.4e5e0a72f0539p+0, -0x1.8a8980abfbd30p-2}, {0x1.4cab88725af6ep+0, -0x1.8304d90c11fd1p-2},
    {0x1.4afd6a052bf5bp+0, -0x1.7b89f02cf2aafp-2}, {0x1.49539e3b2d067p+0, -0x1.7418acebbf18fp-2},
    {0x1.47ae147ae147bp+0, -0x1.6cb0f6865c8ebp-2}, {0x1.460cbc7f5cf9ap+0, -0x1.6552b49986277p-2},
    {0x1.446f86562d9fbp+0, -0x1.5dfdcf1eeae0fp-2}, {0x1.42d6625d51f87p+0, -0x1.56b22e6b578e5p-2},
    {0x1.4141414141414p+0, -0x1.4f6fbb2cec598p-2}, {0x1.3fb013fb013fbp+0, -0x1.48365e695d797p-2},
    {0x1.3e22cbce4a902p+0, -0x1.4106017c3eca0p-2}, {0x1.3c995a47babe7p+0, -0x1.39de8e1559f6ep-2},
    {0x1.3b13b13b13b14p+0, -0x1.32bfee370ee6ap-2}, {0x1.3991c2c187f63p+0, -0x1.2baa0c34be1ebp-2},
    {0x1.3813813813814p+0, -0x1.249cd2b13cd6fp-2}, {0x1.3698df3de0748p+0, -0x1.1d982c9d5270ap-2},
    {0x1.3521cfb2b78c1p+0, -0x1.169c05363f157p-2}, {0x1.33ae45b57bcb2p+0, -0x1.0fa848044b352p-2},
    {0x1.323e34a2b10bfp+0, -0x1.08bce0d95fa36p-2}, {0x1.30d190130d190p+0, -0x1.01d9bbcfa61d4p-2},
    {0x1.2f684bda12f68p+0, -0x1.f5fd8a9063e32p-3}, {0x1.2e025c04b8097p+0, -0x1.e857d3d361368p-3},
    {0x1.2c9fb4d812ca0p+0, -0x1.dac22d3e441d6p-3}, {0x1.2b404ad012b40p+0, -0x1.cd3c712d31106p-3},
    {0x1.29e4129e4129ep+0, -0x1.bfc67a7fff4cap-3}, {0x1.288b01288b013p+0, -0x1.b2602497d534ap-3},
    {0x1.27350b8812735p+0, -0x1.a5094b54d2828p-3}, {0x1.25e22708092f1p+0, -0x1.97c1cb13c7ec0p-3},
    {0x1.2492492492492p+0, -0x1.8a8980abfbd30p-3}, {0x1.23456789abcdfp+0, -0x1.7d60496cfbb4cp-3},
    {0x1.21fb78121fb78p+0, -0x1.7046031c79f84p-3}, {0x1.20b470c67c0d9p+0, -0x1.633a8bf437ce6p-3},
    {0x1.1f7047dc11f70p+0, -0x1.563dc29ffacafp-3}, {0x1.1e2ef3b3fb874p+0, -0x1.494f863b8df32p-3},
    {0x1.1cf06ada2811dp+0, -0x1.3c6fb650cde51p-3}, {0x1.1bb4a4046ed29p+0, -0x1.2f9e32d5bfdd1p-3},
    {0x1.1a7b9611a7b96p+0, -0x1.22dadc2ab3496p-3}, {0x1.19453808ca29cp+0, -0x1.162593186da70p-3},
    {0x1.1811811811812p+0, -0x1.097e38ce6064ep-3}, {0x1.16e0689427379p+0, -0x1.f9c95dc1d1167p-4},
    {0x1.15b1e5f75270dp+0, -0x1.e0b1ae8f2fd56p-4}, {0x1.1485f0e0acd3bp+0, -0x1.c7b528b70f1bcp-4},
    {0x1.135c81135c811p+0, -0x1.aed391ab6674ap-4}, {0x1.12358e75d3033p+0, -0x1.960caf9abb7c1p-4},
    {0x1.1111111111111p+0, -0x1.7d60496cfbb4bp-4}, {0x1.0fef010fef011p+0, -0x1.64ce26c067157p-4},
    {0x1.0ecf56be69c90p+0, -0x1.4c560fe68af8bp-4}, {0x1.0db20a88f4696p+0, -0x1.33f7cde14cf63p-4},
    {0x1.0c9714fbcda3bp+0, -0x1.1bb32a60054a2p-4}, {0x1.0b7e6ec259dc8p+0, -0x1.0387efbca86a7p-4},
    {0x1.0a6810a6810a7p+0, -0x1.d6ebd1f1fec14p-5}, {0x1.0953f39010954p+0, -0x1.a6f9c377dd31dp-5},
    {0x1.0842108421084p+0, -0x1.77394c9d958d0p-5}, {0x1.073260a47f7c6p+0, -0x1.47aa07357703cp-5},
    {0x1.0624dd2f1a9fcp+0, -0x1.184b8e4c56afcp-5}, {0x1.05197f7d73404p+0, -0x1.d23afc49139f1p-6},
    {0x1.0410410410410p+0, -0x1.743ee861f353fp-6}, {0x1.03091b51f5e1ap+0, -0x1.16a21e20a0a29p-6},
    {0x1.0204081020408p+0, -0x1.72c7ba20f731cp-7}, {0x1.0101010101010p+0, -0x1.720d9c06a8348p-8},
    {0x1.0000000000000p+0, 0x0.0p+0}, {0x1.fc07f01fc07f0p-1, 0x1.6fe50b6ef085dp-7},
    {0x1.f81f81f81f820p-1, 0x1.6e79685c2d212p-6}, {0x1.f44659e4a4271p-1, 0x1.11cd1d513341bp-5},
    {0x1.f07c1f07c1f08p-1, 0x1.6bad3758efd81p-5}, {0x1.ecc07b301ecc0p-1, 0x1.c4dfab90aab6ap-5},
    {0x1.e9131abf0b767p-1, 0x1.0eb389fa29f9dp-4}, {0x1.e573ac901e574p-1, 0x1.3aa2fdd27f1bfp-4},
    {0x1.e1e1e1e1e1e1ep-1, 0x1.663f6fac91318p-4}, {0x1.de5d6e3f8868ap-1, 0x1.918a16e46335ep-4},
    {0x1.dae6076b981dbp-1, 0x1.bc84240adabb9p-4}, {0x1.d77b654b82c34p-1, 0x1.e72ec117fa5adp-4},
    {0x1.d41d41d41d41dp-1, 0x1.08c588cda79e5p-3}, {0x1.d0cb58f6ec074p-1, 0x1.1dcd197552b7dp-3},
    {0x1.cd85689039b0bp-1, 0x1.32ae9e278ae19p-3}, {0x1.ca4b3055ee191p-1, 0x1.476a9f983f74dp-3},
    {0x1.c71c71c71c71cp-1, 0x1.5c01a39fbd68bp-3}, {0x1.c3f8f01c3f8f0p-1, 0x1.70742d4ef0280p-3},
    {0x1.c0e070381c0e0p-1, 0x1.84c2bd02f03b6p-3}, {0x1.bdd2b899406f7p-1, 0x1.98edd077e70e1p-3},
    {0x1.bacf914c1bad0p-1, 0x1.acf5e2db4ec91p-3}, {0x1.b7d6c3dda338bp-1, 0x1.c0db6cdd94defp-3},
    {0x1.b4e81b4e81b4fp-1, 0x1.d49ee4c32596cp-3}, {0x1.b2036406c80d9p-1, 0x1.e840be74e6a4dp-3},
    {0x1.af286bca1af28p-1, 0x1.fbc16b902680dp-3}, {0x1.ac5701ac5701bp-1, 0x1.0790adbb03009p-2},
    {0x1.a98ef606a63bep-1, 0x1.11307dad30b74p-2}, {0x1.a6d01a6d01a6dp-1, 0x1.1ac05b291f070p-2},
    {0x1.a41a41a41a41ap-1, 0x1.24407ab0e073ap-2}, {0x1.a16d3f97a4b02p-1, 0x1.2db10fc4d9aaep-2},
    {0x1.9ec8e951033d9p-1, 0x1.37124cea4cdedp-2}, {0x1.9c2d14ee4a102p-1, 0x1.406463b1b0448p-2},
    {0x1.999999999999ap-1, 0x1.49a784bcd1b8ap-2}, {0x1.970e4f80cb872p-1, 0x1.52dbdfc4c96b5p-2},
    {0x1.948b0fcd6e9e0p-1, 0x1.5c01a39fbd689p-2}, {0x1.920fb49d0e229p-1, 0x1.6518fe4677ba6p-2},
    {0x1.8f9c18f9c18fap-1, 0x1.6e221cd9d0cddp-2}, {0x1.8d3018d3018d3p-1, 0x1.771d2ba7efb3cp-2},
    {0x1.8acb90f6bf3aap-1, 0x1.800a563161c53p-2}, {0x1.886e5f0abb04ap-1, 0x1.88e9c72e0b224p-2},
    {0x1.8618618618618p-1, 0x1.91bba891f170ap-2}, {0x1.83c977ab2beddp-1, 0x1.9a802391e2330p-2},
    {0x1.8181818181818p-1, 0x1.a33760a7f6051p-2}, {0x1.7f405fd017f40p-1, 0x1.abe18797f1f4ap-2},
    {0x1.7d05f417d05f4p-1, 0x1.b47ebf73882a1p-2}, {0x1.7ad2208e0ecc3p-1, 0x1.bd0f2e9e79032p-2},
    {0x1.78a4c8178a4c8p-1, 0x1.c592fad295b57p-2}, {0x1.767dce434a9b1p-1, 0x1.ce0a4923a587dp-2},
    {0x1.745d1745d1746p-1, 0x1.d6753e032ea0ep-2}, {0x1.724287f46debcp-1, 0x1.ded3fd442364cp-2},
    {0x1.702e05c0b8170p-1, 0x1.e726aa1e754d3p-2}, {0x1.6e1f76b4337c7p-1, 0x1.ef6d67328e220p-2},
};
static __device__ const double POWF_EXP2_TAB[32] = {
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0,
};
// log2(1 + r) = r (C1 + C2 r + ... + C7 r^6) + O(r^8),  2^r = 1 + r (L1 + ... + L5 r^4) + O(r^6): Taylor coefficients
static __device__ const double POWF_LOG2_C[7] = {0x1.71547652b82fep+0, -0x1.71547652b82fep-1, 0x1.ec709dc3a03fdp-2, -0x1.71547652b82fep-2, 0x1.2776c50ef9bfep-2, -0x1.ec709dc3a03fdp-3, 0x1.a61762a7aded9p-3};
static __device__ const double POWF_EXP2_L[5] = {0x1.62e42fefa39efp-1, 0x1.ebfbdff82c58fp-3, 0x1.c6b08d704a0c0p-5, 0x1.3b2ab6fba4e77p-7, 0x1.5d87fe78a6731p-10};
// ln(2), log10(2)
#define POWF_LN2 0x1.62e42fefa39efp-1
#define POWF_LOG10_2 0x1.34413509f79ffp-2

// The tables in LDS (round 5): the common case of np.power / np.log / np.log10 reads them there -- two gathers per value that no
// longer share the vector-memory path with the tile's own loads and stores.  A workgroup of the VM_MATH kernels copies them in
// while its tile loads are in flight (vm_body); the full routines (special values, the float64 class) keep the global tables.
struct PowfLds {
  double log2_tab[128][2];
  double exp2_tab[32];
};
__device__ __forceinline__ PowfLds* powf_lds() {
  __shared__ __align__(16) PowfLds tables;
  return &tables;
}
__device__ __forceinline__ void powf_lds_stage() {  // (the caller synchronises the workgroup afterwards)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  PowfLds* const T = powf_lds();
  const unsigned t = threadIdx.x;
  if (t < 128u) ((u32x4*)T->log2_tab)[t] = ((const u32x4*)POWF_LOG2_TAB)[t];
  else if (t < 144u) ((u32x4*)T->exp2_tab)[t - 128u] = ((const u32x4*)POWF_EXP2_TAB)[t - 128u];
}

// The two halves of the float32 transcendentals, in float64 arithmetic (round 4):
//   log2 of a positive finite float32 = e + log2(c) + log2(1 + r), r = z / c - 1, by a table of 128 mantissa intervals (the one
//     around 1.0 has c = 1: values near 1 keep their RELATIVE precision) and five or seven Taylor terms (|r| <= 2^-8);
//   2^(z + z_lo) -> float32 by 32 table steps and five Taylor terms (|r| <= 2^-6: error < 2^-48), z clamped to the range in which
//     the float32 result is not 0 / inf anyway.
template <int TERMS>  // 5: good to ~2^-42 of the value at the interval edges (power); 7: ~2^-56 (log / log10)
__device__ __forceinline__ double log2_of_positive_f32(unsigned ax) {
  const bool tiny = ax < 0x00800000u;  // subnormal: scaled by 2^23 first
  const unsigned nx = tiny ? __float_as_uint(__uint_as_float(ax) * 0x1p23f) : ax;
  const unsigned tmp = nx - 0x3f338000u;
  const int i = (int)((tmp >> 16) & 127u);
  const int e = ((int)tmp >> 23) - (tiny ? 23 : 0);
  const double z = (double)__uint_as_float(nx - (tmp & 0xff800000u));  // in [0.7, 1.4)
  const double r = fma(z, POWF_LOG2_TAB[i][0], -1.0);
  double p = POWF_LOG2_C[TERMS - 1];
#pragma unroll
  for (int k = TERMS - 2; k >= 2; --k) p = fma(r, p, POWF_LOG2_C[k]);
  p = fma(r, p, POWF_LOG2_C[1]);
  p = fma(r, p, POWF_LOG2_C[0]);
  return fma(r, p, (double)e + POWF_LOG2_TAB[i][1]);
}
// ... of a NORMAL float32 (no scaling of subnormals): the common case of the routes below, which check for it
template <int TERMS>
__device__ __forceinline__ double log2_of_normal_f32(unsigned ax, const double (*tab)[2]) {
  const unsigned tmp = ax - 0x3f338000u;
  const int i = (int)((tmp >> 16) & 127u);
  const int e = (int)tmp >> 23;
  const double z = (double)__uint_as_float(ax - (tmp & 0xff800000u));  // in [0.7, 1.4)
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  const f64x2 entry = *(const f64x2*)tab[i];  // {1 / c, log2(c)}: one 16-byte read
  const double r = fma(z, entry.x, -1.0);
  double p = POWF_LOG2_C[TERMS - 1];
#pragma unroll
  for (int k = TERMS - 2; k >= 2; --k) p = fma(r, p, POWF_LOG2_C[k]);
  p = fma(r, p, POWF_LOG2_C[1]);
  p = fma(r, p, POWF_LOG2_C[0]);
  return fma(r, p, (double)e + entry.y);
}
__device__ __forceinline__ float exp2_to_f32(double z, double z_lo) {
  z = fmin(fmax(z, -160.0), 130.0);  // (beyond: 0 / inf as float32 either way; keeps 2^k a normal double)
  const double kd = rint(z * 32.0);
  const int ki = (int)kd;
  const double rr = fma(kd, -0x1p-5, z) + z_lo;
  double q = fma(rr, POWF_EXP2_L[4], POWF_EXP2_L[3]);
  q = fma(rr, q, POWF_EXP2_L[2]);
  q = fma(rr, q, POWF_EXP2_L[1]);
  q = fma(rr, q, POWF_EXP2_L[0]);
  q = fma(rr, q, 1.0);
  const long long scaled = __double_as_longlong(POWF_EXP2_TAB[ki & 31]) + ((long long)(ki >> 5) << 52);
  return (float)(__longlong_as_double(scaled) * q);
}

// np.log / np.log10 on the float32 class: the float64 ocml calls rounded once cost 204 us per 8192^2 tile (84 us of streaming);
// log2 through the half above times ln(2) / log10(2): 161 us, and still the correctly rounded float32 -- the float64 value is
// good to ~2^-52, so a float32 rounding boundary lies inside the error only once in ~2^27 arguments
// (tests/test_gpu_parity.py::test_exp_log_are_correctly_rounded, tools/ulp_measure.py).  np.exp stays on ocml's float64 exp:
// the same construction (x log2(e) in two pieces, the 2^x half with six terms) was measured at 162 us against its 110.
// The common case first (round 5): a positive normal argument needs none of the selects of logf_via_f64 below; zero, subnormal,
// negative, infinite and NaN arguments are one unsigned compare away.  `special` = the value still needs the full routine (the
// caller -- Machine::apply_logf -- runs the common case for ALL its values first, so that their table reads are in flight
// together, and then asks once per wave whether any value was special).  +-0 is settled here (-inf: rasters are full of zeros).
__device__ __forceinline__ float logf_common(float xf, double scale, bool& special, const PowfLds* T) {
  const unsigned ix = __float_as_uint(xf);
  const bool ordinary = ix - 0x00800000u < 0x7f800000u - 0x00800000u;  // positive, normal, finite
  const bool zero = (ix << 1) == 0u;
  const float out = (float)(log2_of_normal_f32<7>(ordinary ? ix : 0x3f800000u, T->log2_tab) * scale);
  special = !(ordinary | zero);
  return zero ? -__builtin_inff() : out;
}
__device__ __forceinline__ float logf_via_f64(float xf, double scale) {  // scale: ln(2) or log10(2)
  const unsigned ix = __float_as_uint(xf), ax = ix & 0x7fffffffu;
  float out = (float)(log2_of_positive_f32<7>(ax) * scale);
  out = ax == 0x7f800000u ? xf : out;                   // +inf (-inf: below)
  out = (ix >> 31) != 0 ? __builtin_nanf("") : out;     // negative
  out = ax == 0 ? -__builtin_inff() : out;              // +-0
  return ax > 0x7f800000u ? xf : out;                   // NaN
}

// np.power on the float32 class (round 4).  float64 pow() rounded once cost ~130 float64 operations per value (a
// 4096^2 tile: 109 us against 22 us of streaming); float32 needs far less than a float64 pow delivers.  Here, in
// float64 arithmetic: 2^(y log2|x|) by the two halves above (values near 1 keep their RELATIVE precision in the logarithm,
// which is what a large exponent magnifies).  The
// exponent argument is good to ~2^-35 absolutely over the float32 range, so the float64 result rounds to the correctly
// rounded float32 except within 2^-11 ulp of a tie -- there it is the other neighbour, i.e. <= 1 ulp from any faithful
// powf (glibc's: 0.52 ulp), the budget of tests/test_gpu_parity.py.  Signs, zeros, infinities and NaN follow C99 F.9.4.4
// = NumPy.  (tools/ulp_measure.py: measured against the extended-precision value.)
// The common case first (round 5): a positive normal base, a finite exponent and a result well inside float32's normal range
// need none of the C99 special cases of powf_via_f64 below, no scaling of subnormals and no clamp -- the same operations on the
// same values, hence the same bits (rint(32 z) by the 1.5 * 2^52 shift: the same integer).  (+0) ** (a positive finite y) = +0 is
// settled here as well.  Anything else (a negative or subnormal base, -0, inf / NaN anywhere, |y log2 x| beyond 125) is
// `special`: Machine::apply_powf runs the common case for ALL its values first and then asks once per wave.
__device__ __forceinline__ float powf_common(float xf, float yf, bool& special, const PowfLds* T) {
  const unsigned ix = __float_as_uint(xf), ay = __float_as_uint(yf) & 0x7fffffffu;
  const bool base_ok = ix - 0x00800000u < 0x7f800000u - 0x00800000u;  // positive, normal, finite
  const double zz = (double)yf * log2_of_normal_f32<5>(base_ok ? ix : 0x3f800000u, T->log2_tab);
  const bool ordinary = base_ok & (ay < 0x7f800000u) & (fabs(zz) <= 125.0);  // (NaN compares false)
  const bool zero = (ix == 0u) & (yf > 0.0f) & (ay < 0x7f800000u);
  // 2^zz as in exp2_to_f32 (a zz out of range gives garbage that `special` replaces; the table index is masked)
  const double km = fma(zz, 32.0, 0x1.8p52);
  const int ki = (int)(unsigned)__double_as_longlong(km);
  const double rr = fma(km - 0x1.8p52, -0x1p-5, zz);
  double q = fma(rr, POWF_EXP2_L[4], POWF_EXP2_L[3]);
  q = fma(rr, q, POWF_EXP2_L[2]);
  q = fma(rr, q, POWF_EXP2_L[1]);
  q = fma(rr, q, POWF_EXP2_L[0]);
  q = fma(rr, q, 1.0);
  const long long scaled = __double_as_longlong(T->exp2_tab[ki & 31]) + ((long long)(ki >> 5) << 52);
  const float out = (float)(__longlong_as_double(scaled) * q);
  special = !(ordinary | zero);
  return zero ? 0.0f : out;
}
__device__ __forceinline__ float powf_via_f64(float xf, float yf) {
  const unsigned ix = __float_as_uint(xf), iy = __float_as_uint(yf);
  const unsigned ax = ix & 0x7fffffffu, ay = iy & 0x7fffffffu;
  // y: 0 = not an integer, 1 = an odd integer, 2 = an even one (everything from 2^24 on and +-inf included)
  const int ey = (int)(ay >> 23);
  const unsigned unit = 1u << ((150 - ey) & 31);  // the bit worth 1 when 127 <= ey <= 150
  int yint = (ay & (unit - 1)) ? 0 : ((ay & unit) ? 1 : 2);
  yint = ey > 150 ? 2 : yint;
  yint = ey < 127 ? (ay == 0 ? 2 : 0) : yint;
  double l2 = log2_of_positive_f32<5>(ax);  // (ax == 0 / inf / NaN: garbage, replaced below)
  l2 = ax == 0 ? -__builtin_huge_val() : l2;
  l2 = ax == 0x7f800000u ? __builtin_huge_val() : l2;
  double zz = (double)yf * l2;
  zz = ax == 0x3f800000u ? 0.0 : zz;                 // |x| = 1: 1 whatever y (0 * inf otherwise)
  float out = exp2_to_f32(zz, 0.0);
  const bool negative = (ix >> 31) != 0;
  out = (negative && yint == 1) ? -out : out;
  out = (negative && yint == 0 && ax != 0 && ax < 0x7f800000u) ? __builtin_nanf("") : out;  // finite x < 0, y not an integer
  out = (ax > 0x7f800000u || ay > 0x7f800000u) ? __builtin_nanf("") : out;
  return (ay == 0 || ix == 0x3f800000u) ? 1.0f : out;
}

// np.power: the float32 class by powf_via_f64 above, float64 through pow() (<= 1 ulp);
// integer classes use exponentiation by squaring with wrap-around.
template <typename RC> __device__ __forceinline__ RC op_pow(RC a, RC b) {
  if constexpr (std::is_same<RC, float>::value) {
    return powf_via_f64(a, b);
  } else if constexpr (is_float_class<RC>::value) {
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
    return ((fabs(xd - y) <= tol) & y_finite) | (xd == y);
  } else {
    float xf = (float)x, yf = (float)y;
    return ((fabsf(xf - yf) <= (float)tol) & y_finite) | (xf == yf);
  }
}

// ---- descriptor decoding (all wave-uniform: SALU on s_load'ed words) ----------------
__device__ __forceinline__ unsigned opd_kind(const gm_operand& o) { return o.bits & 0xffu; }
__device__ __forceinline__ unsigned opd_index(const gm_operand& o) { return (o.bits >> 8) & 0xffu; }
__device__ __forceinline__ bool opd_has_ndv(const gm_operand& o) { return ((o.bits >> 16) & 1u) != 0; }
__device__ __forceinline__ bool opd_is_bool(const gm_operand& o) { return ((o.bits >> 17) & 1u) != 0; }
__device__ __forceinline__ unsigned step_opcode(const gm_step& S) { return S.bits & 0xffu; }
__device__ __forceinline__ unsigned step_save_temp(const gm_step& S) { return (S.bits >> 8) & 0xffu; }
__device__ __forceinline__ unsigned step_flags(const gm_step& S) { return (S.bits >> 16) & 0xffu; }
__device__ __forceinline__ bool step_round_f32(const gm_step& S) { return ((S.bits >> 24) & 1u) != 0; }
__device__ __forceinline__ bool step_wrap_i32(const gm_step& S) { return ((S.bits >> 25) & 1u) != 0; }
__device__ __forceinline__ bool step_check_finite(const gm_step& S) { return ((S.bits >> 26) & 1u) != 0; }
__device__ __forceinline__ int in_dtype(const gm_input_desc& D) { return (int)(D.bits & 0xffu); }
__device__ __forceinline__ bool in_broadcast(const gm_input_desc& D) { return ((D.bits >> 8) & 1u) != 0; }
__device__ __forceinline__ bool in_native(const gm_input_desc& D) { return ((D.bits >> 9) & 1u) != 0; }

// ---- the interpreter -------------------------------------------------------------
// FLAVOR of an instantiation: rarely used, register-hungry opcodes live in their own kernels so that
// they do not cap the occupancy of ordinary programs
constexpr int VM_LIGHT = 0;   // everything else
constexpr int VM_MATH = 1;    // + pow / exp / log / log10
constexpr int VM_TABLES = 2;  // + DIGITIZE / LOOKUP
// N elements per lane, NIN input slots, NT temps.  HEAVY instantiations carry
// pow/exp/log (their register appetite would otherwise cap every program's occupancy).
template <typename RC, int N, int NIN, int NT, int FLAVOR>
struct Machine {
  static constexpr bool HEAVY = FLAVOR == VM_MATH;
  static constexpr bool TABLES = FLAVOR == VM_TABLES;
  static constexpr bool FOLD_LD = FLAVOR == VM_LIGHT && NIN <= 2;  // kernels that accept "LD folded into the op"
  RC in[NIN][N];
  RC tmp0[N], tmp1[N], tmp2[N], tmp3[N];  // separate members, selected at compile time only: an
  RC acc[N];                               // array of arrays ends up in scratch for NT >= 2

  template <int K> __device__ __forceinline__ RC (&T())[N] {
    if constexpr (K == 0) return tmp0;
    else if constexpr (K == 1) return tmp1;
    else if constexpr (K == 2) return tmp2;
    else return tmp3;
  }
  template <int K> __device__ __forceinline__ const RC (&T() const)[N] {
    if constexpr (K == 0) return tmp0;
    else if constexpr (K == 1) return tmp1;
    else if constexpr (K == 2) return tmp2;
    else return tmp3;
  }

  // operand fetch; every branch below is wave-uniform
  __device__ __forceinline__ void fetch(const gm_operand& o, RC (&v)[N]) const {
    switch (opd_kind(o)) {
      case GM_OPD_INPUT: {
        switch (opd_index(o)) {
#define GM_IN_CASE(K)                                                    \
  case K:                                                                \
    if constexpr (K < NIN) {                                             \
      _Pragma("unroll") for (int k = 0; k < N; ++k) v[k] = in[K][k];     \
    }                                                                    \
    break;
          GM_IN_CASE(0) GM_IN_CASE(1) GM_IN_CASE(2) GM_IN_CASE(3)
          GM_IN_CASE(4) GM_IN_CASE(5) GM_IN_CASE(6) GM_IN_CASE(7)
          GM_IN_CASE(8) GM_IN_CASE(9) GM_IN_CASE(10) GM_IN_CASE(11)
          GM_IN_CASE(12) GM_IN_CASE(13) GM_IN_CASE(14) GM_IN_CASE(15)
#undef GM_IN_CASE
          default: break;
        }
        static_assert(GM_MAX_INPUTS == 16, "one GM_IN_CASE per input");
      } break;
      case GM_OPD_SCALAR: {
        const RC s = scalar_of<RC>(o.value);
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = s;
      } break;
      case GM_OPD_TEMP: {
        switch (opd_index(o)) {
#define GM_TMP_CASE(K)                                                   \
  case K:                                                                \
    if constexpr (K < NT) {                                              \
      _Pragma("unroll") for (int k = 0; k < N; ++k) v[k] = T<K>()[k];    \
    }                                                                    \
    break;
          GM_TMP_CASE(0) GM_TMP_CASE(1) GM_TMP_CASE(2) GM_TMP_CASE(3)
#undef GM_TMP_CASE
          default: break;
        }
      } break;
      case GM_OPD_ACC: {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = acc[k];
      } break;
      default: {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = (RC)0;
      } break;
    }
  }

  // In-place step body: acc[k] = bad ? fill : f(acc[k], b[k])
  //   bad = acc[k] is no-data | b[k] is no-data | !finite(result)
  // The left operand is always the accumulator (step() loads it there first).  B_SCALAR:
  // the right operand is a literal held in an SGPR -- no copy, no no-data test.
  // Float classes: an operand without ndv is compared against NaN (== is never true), so the
  // lane masks need no extra s_and; integer classes gate the compare with the has_ndv bit.
  template <bool B_SCALAR, typename F>
  __device__ __forceinline__ void apply(const gm_step& S, const RC (&b)[N], F f) {
    const bool a_has = opd_has_ndv(S.a), b_has = opd_has_ndv(S.b);
    RC a_ndv = scalar_of<RC>(S.a.value), b_ndv = scalar_of<RC>(S.b.value);
    const RC b_val = b_ndv;  // B_SCALAR: the literal itself
    if constexpr (is_float_class<RC>::value) {
      // once per step, on the scalar unit: an operand without ndv compares against NaN
      a_ndv = a_has ? a_ndv : (RC)__builtin_nanf("");
      b_ndv = b_has ? b_ndv : (RC)__builtin_nanf("");
    }
    const RC fill = scalar_of<RC>(S.fill);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const RC x = acc[k];
      const RC y = B_SCALAR ? b_val : b[k];
      bool bad;  // bitwise (not short-circuit) logic: v_cmp into lane masks, no branches
      if constexpr (is_float_class<RC>::value) {
        bad = (x == a_ndv);
        if constexpr (!B_SCALAR) bad = bad | (y == b_ndv);
      } else {
        bad = a_has & (x == a_ndv);
        if constexpr (!B_SCALAR) bad = bad | (b_has & (y == b_ndv));
      }
      const RC r = f(x, y);
      if constexpr (is_float_class<RC>::value) bad = bad | !finite_rc<RC>(r);
      acc[k] = bad ? fill : r;
    }
  }

  // np.power / np.log / np.log10 on the float32 class (round 5): the common case of powf_common / logf_common for all N values
  // first -- straight-line code, the N table reads of each half in flight together -- with no-data cells evaluated as 1.0 so that
  // a raster's no-data value never sends a wave into the full routine; then one wave-level test for values that need it.
  template <bool B_SCALAR>
  __device__ __forceinline__ void apply_powf(const gm_step& S, const RC (&b)[N]) {
    if constexpr (std::is_same<RC, float>::value) {
      const float a_ndv = opd_has_ndv(S.a) ? scalar_of<RC>(S.a.value) : __builtin_nanf("");
      const float b_lit = scalar_of<RC>(S.b.value);
      const float b_ndv = opd_has_ndv(S.b) ? b_lit : __builtin_nanf("");
      const float fill = scalar_of<RC>(S.fill);
      const PowfLds* const tables = powf_lds();
      float xs[N], ys[N], r[N];
      bool bad[N], special[N], any = false;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        bad[k] = (acc[k] == a_ndv);
        if constexpr (!B_SCALAR) bad[k] = bad[k] | (b[k] == b_ndv);
        xs[k] = bad[k] ? 1.0f : acc[k];
        ys[k] = B_SCALAR ? b_lit : (bad[k] ? 1.0f : b[k]);
        r[k] = powf_common(xs[k], ys[k], special[k], tables);
        any = any | special[k];
      }
      if (__builtin_amdgcn_ballot_w64(any)) {
#pragma unroll
        for (int k = 0; k < N; ++k)
          if (special[k]) r[k] = powf_via_f64(xs[k], ys[k]);
      }
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] = (bad[k] | !finite_rc<RC>(r[k])) ? fill : r[k];
    }
  }
  __device__ __forceinline__ void apply_logf(const gm_step& S, const double scale) {
    if constexpr (std::is_same<RC, float>::value) {
      const float a_ndv = opd_has_ndv(S.a) ? scalar_of<RC>(S.a.value) : __builtin_nanf("");
      const float fill = scalar_of<RC>(S.fill);
      const PowfLds* const tables = powf_lds();
      float xs[N], r[N];
      bool bad[N], special[N], any = false;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        bad[k] = (acc[k] == a_ndv);
        xs[k] = bad[k] ? 1.0f : acc[k];
        r[k] = logf_common(xs[k], scale, special[k], tables);
        any = any | special[k];
      }
      if (__builtin_amdgcn_ballot_w64(any)) {
#pragma unroll
        for (int k = 0; k < N; ++k)
          if (special[k]) r[k] = logf_via_f64(xs[k], scale);
      }
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] = (bad[k] | !finite_rc<RC>(r[k])) ? fill : r[k];
    }
  }

  template <typename F>
  __device__ __forceinline__ void apply2(const gm_step& S, bool b_scalar, const RC (&b)[N], F f) {
    if (b_scalar) apply<true>(S, b, f);
    else apply<false>(S, b, f);
  }

  // the same with the step-dtype narrowing of the wide classes applied to f's result
  template <typename F>
  __device__ __forceinline__ void math(const gm_step& S, bool b_scalar, const RC (&b)[N], F f) {
    if constexpr (std::is_same<RC, double>::value) {
      if (step_round_f32(S)) {  // float32 step inside the F64 class: round after the op
        apply2(S, b_scalar, b, [f](RC x, RC y) { return (RC)(float)f(x, y); });
        return;
      }
    }
    if constexpr (std::is_same<RC, long long>::value) {
      if (step_wrap_i32(S)) {  // int32 step inside the I64 class: wrap to 32 bits
        apply2(S, b_scalar, b, [f](RC x, RC y) { return (RC)(int)f(x, y); });
        return;
      }
    }
    apply2(S, b_scalar, b, f);
  }

  // ---- table ops (TABLES kernels): shared by the generic step and the packed micro-ops -----------
  // misc.py:387-399.  Edge-major linear count: every edge is one scalar load (wave-uniform),
  // v_cmp + v_addc per element; <= GM_MAX_LINEAR_EDGES edges.  Comparisons in the type NumPy's
  // promoted searchsorted uses: F32 class -> float32 thresholds prepared by the host with
  // directed rounding (same decisions as float64 compares), F64 -> float64 edges, integer
  // classes -> int64 edges, or float64 edges against (double)value.
  __device__ __forceinline__ void op_digitize(kargs_t C, const unsigned table, const unsigned nb, const bool right,
                                              const bool int_edges, const bool a_has, const RC a_ndv, const RC fill) {
    const kargs_t tbl = (kargs_t)(uintptr_t)kload<unsigned long long>(C, offsetof(VmCode, tables) + 8 * table);
    int cnt[N];
#pragma unroll
    for (int k = 0; k < N; ++k) cnt[k] = 0;
    // count the edges below (or at) the value: one compare + one add-with-carry per edge and element.
    // `right` is decided once, outside the loops (inside it would cost a select per compare), and the
    // edges arrive four per scalar load.
    auto count = [&](auto below) {
      typedef typename std::conditional<std::is_same<RC, float>::value, float, double>::type E;
      if constexpr (std::is_same<RC, float>::value || std::is_same<RC, double>::value) {
        for (unsigned b0 = 0; b0 < nb; b0 += 4) {
          struct Quad { E e[4]; };
          const Quad q = kload<Quad>(tbl, (unsigned)sizeof(E) * b0);  // (tables are padded to 4 edges by the host)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (b0 + j < nb) {
#pragma unroll
              for (int k = 0; k < N; ++k) cnt[k] += below(q.e[j], acc[k]) ? 1 : 0;
            }
          }
        }
      }
    };
    if constexpr (std::is_same<RC, float>::value || std::is_same<RC, double>::value) {
      if (right) count([](RC e, RC x) { return e < x; });
      else count([](RC e, RC x) { return e <= x; });
    } else {
      for (unsigned b = 0; b < nb; ++b) {
        if (int_edges) {
          const long long e = kload<long long>(tbl, 8 * b);
#pragma unroll
          for (int k = 0; k < N; ++k) cnt[k] += (right ? (e < (long long)acc[k]) : (e <= (long long)acc[k])) ? 1 : 0;
        } else {
          const double e = kload<double>(tbl, 8 * b);
#pragma unroll
          for (int k = 0; k < N; ++k) cnt[k] += (right ? (e < (double)acc[k]) : (e <= (double)acc[k])) ? 1 : 0;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      int c = cnt[k];
      if constexpr (is_float_class<RC>::value) c = (acc[k] != acc[k]) ? (int)nb : c;  // NaN sorts to the end
      acc[k] = (a_has & (acc[k] == a_ndv)) ? fill : (RC)c;
    }
  }

  // misc.py:505-514.  Dense window of pre-resolved targets (entries in the register class' own
  // width) + keys outside the window -- typically only the merged-in no-data key -- matched against
  // scalar registers.  The accumulator holds an exact integer in every class.
  __device__ __forceinline__ void op_lookup(kargs_t C, const unsigned table, const long long kmin64,
                                            const unsigned long long range64, const unsigned n_exc, const bool select,
                                            const RC fill) {
    typedef typename std::conditional<sizeof(RC) == 8, unsigned long long, unsigned int>::type W;
    typedef typename std::conditional<sizeof(RC) == 8, long long, int>::type KT;
    const KT kmin = (KT)kmin64;
    const W range = (W)range64;
    const unsigned long long addr = kload<unsigned long long>(C, offsetof(VmCode, tables) + 8 * table);
    const kargs_t exc = (kargs_t)(uintptr_t)(addr + (((unsigned long long)range * sizeof(RC) + 7ull) & ~7ull));
    // (window index, looked-up target) are the only per-element state besides the accumulator;
    // an exception key k matches where the index equals k - kmin (wrapping arithmetic)
    W idx[N];
    RC hit[N];
#pragma unroll
    for (int k = 0; k < N; ++k) idx[k] = (W)(KT)acc[k] - (W)kmin;
    {
      // the window was staged in LDS behind the tile loads (the host stages the table of the
      // program's LOOKUP; one per program): one ds_read per element.  A gather from device
      // memory instead costs the whole CU ~64 cycles per wave-instruction and 2 VGPRs of
      // address per element.
      extern __shared__ __align__(16) unsigned long long vm_lds[];
      const W* const lds = (const W*)vm_lds;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const W w = lds[idx[k] < range ? idx[k] : (W)0];
        RC h;
        __builtin_memcpy(&h, &w, sizeof(RC));
        hit[k] = h;
      }
    }
    if (select) {
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] = fill;  // a miss: fill; else the value passes through
    }
    for (unsigned x = 0; x < n_exc; ++x) {  // exception-major: each pair is fetched once
      const W ei = (W)(KT)kload<long long>(exc, 16 * x) - (W)kmin;
      const RC ev = kload<RC>(exc, 16 * x + 8);
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] = (idx[k] == ei) ? ev : acc[k];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = idx[k] < range ? hit[k] : acc[k];
  }

  // ---- the table ops with their tables as LITERALS (kernels specialised per program, gm_spec.hip):
  // the same decisions as op_digitize / op_lookup above, without table loads, LDS staging or barrier.
  // The literals travel as packs of bit patterns (Bits<...>), expanded with fold expressions: a local
  // array of them would be an alloca, which the backend may "promote" to LDS and index dynamically.
  template <typename E, unsigned long long... EB>
  __device__ __forceinline__ void op_digitize_lit(Bits<EB...>, const bool right, const bool a_has, const RC a_ndv,
                                                  const RC fill) {
    constexpr int NB = (int)sizeof...(EB);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const E x = (E)acc[k];
      int c = 0;
      ((c += (right ? (edge_of<E>(EB) < x) : (edge_of<E>(EB) <= x)) ? 1 : 0), ...);
      if constexpr (is_float_class<RC>::value) c = (acc[k] != acc[k]) ? NB : c;  // NaN sorts to the end
      acc[k] = (a_has & (acc[k] == a_ndv)) ? fill : (RC)c;
    }
  }

  // DIGITIZE followed by LOOKUP (Reclassify(Classify(x)), the class raster idiom): the lookup's result is a
  // function of the edge count, tabulated by the specialiser -- M0 for no edge below the value, the k-th MB
  // once the k-th edge is below it (edges ascend, so the edges below a value are a prefix and the last
  // satisfied compare selects the count; the specialiser drops edges that do not change the result),
  // map_nan for NaN, map_fill for a no-data cell.  One compare + one select per remaining edge.
  template <typename E, unsigned long long M0, unsigned long long... EB, unsigned long long... MB>
  __device__ __forceinline__ void op_digimap_lit(Bits<EB...>, Bits<M0, MB...>, const bool right, const bool a_has,
                                                 const RC a_ndv, const RC map_nan, const RC map_fill) {
    static_assert(sizeof...(EB) == sizeof...(MB), "one result per edge");
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const E x = (E)acc[k];
      RC r = lit<RC>(M0);
      ((r = (right ? (edge_of<E>(EB) < x) : (edge_of<E>(EB) <= x)) ? lit<RC>(MB) : r), ...);
      if constexpr (is_float_class<RC>::value) r = (acc[k] != acc[k]) ? map_nan : r;  // NaN sorts to the end
      acc[k] = (a_has & (acc[k] == a_ndv)) ? map_fill : r;
    }
  }

  // small dense windows (<= 16 slots): a select chain over the slots instead of a read from LDS
  template <unsigned long long... WB, unsigned long long... EK, unsigned long long... EV>
  __device__ __forceinline__ void op_lookup_lit(Bits<WB...>, Bits<EK...>, Bits<EV...>, const long long kmin64,
                                                const bool select, const RC fill) {
    static_assert(sizeof...(EK) == sizeof...(EV), "one target per exception key");
    typedef typename std::conditional<sizeof(RC) == 8, unsigned long long, unsigned int>::type W;
    typedef typename std::conditional<sizeof(RC) == 8, long long, int>::type KT;
    const KT kmin = (KT)kmin64;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const W idx = (W)(KT)acc[k] - (W)kmin;
      RC r = select ? fill : acc[k];
      ((r = (idx == (W)(KT)(long long)EK - (W)kmin) ? lit<RC>(EV) : r), ...);
      W j = 0;
      ((r = (idx == j++) ? lit<RC>(WB) : r), ...);
      acc[k] = r;
    }
  }

  __device__ __forceinline__ void step(kargs_t C, const gm_step& S) {
    RC b[N];
    const unsigned b_kind = opd_kind(S.b);
    const bool b_scalar = (b_kind == GM_OPD_SCALAR) | (b_kind == GM_OPD_NONE);
    if (!b_scalar) fetch(S.b, b);                       // (also covers b == ACC: copied before acc changes)
    if (opd_kind(S.a) != GM_OPD_ACC) fetch(S.a, acc);   // left operand -> accumulator
    const int opcode = (int)step_opcode(S);
    switch (opcode) {
      case GM_OP_ADD: math(S, b_scalar, b, [](RC x, RC y) { return op_add<RC>(x, y); }); break;
      case GM_OP_SUB: math(S, b_scalar, b, [](RC x, RC y) { return op_sub<RC>(x, y); }); break;
      case GM_OP_MUL: math(S, b_scalar, b, [](RC x, RC y) { return op_mul<RC>(x, y); }); break;
      case GM_OP_DIV: math(S, b_scalar, b, [](RC x, RC y) { return op_div<RC>(x, y); }); break;
      case GM_OP_POW:
        if constexpr (HEAVY) {
          if constexpr (std::is_same<RC, float>::value) {
            if (b_scalar) apply_powf<true>(S, b);
            else apply_powf<false>(S, b);
          } else {
            math(S, b_scalar, b, [](RC x, RC y) { return op_pow<RC>(x, y); });
          }
        }
        break;
      case GM_OP_EXP:
        if constexpr (HEAVY && is_float_class<RC>::value) math(S, true, b, [](RC x, RC) { return (RC)exp((double)x); });
        break;
      case GM_OP_LOG:
        if constexpr (HEAVY && std::is_same<RC, float>::value) apply_logf(S, POWF_LN2);
        else if constexpr (HEAVY && is_float_class<RC>::value) math(S, true, b, [](RC x, RC) { return (RC)log((double)x); });
        break;
      case GM_OP_LOG10:
        if constexpr (HEAVY && std::is_same<RC, float>::value) apply_logf(S, POWF_LOG10_2);
        else if constexpr (HEAVY && is_float_class<RC>::value) math(S, true, b, [](RC x, RC) { return (RC)log10((double)x); });
        break;
      case GM_OP_EQ: apply2(S, b_scalar, b, [](RC x, RC y) { return (RC)(x == y); }); break;
      case GM_OP_NE: apply2(S, b_scalar, b, [](RC x, RC y) { return (RC)(x != y); }); break;
      case GM_OP_GT: apply2(S, b_scalar, b, [](RC x, RC y) { return (RC)(x > y); }); break;
      case GM_OP_GE: apply2(S, b_scalar, b, [](RC x, RC y) { return (RC)(x >= y); }); break;
      case GM_OP_LT: apply2(S, b_scalar, b, [](RC x, RC y) { return (RC)(x < y); }); break;
      case GM_OP_LE: apply2(S, b_scalar, b, [](RC x, RC y) { return (RC)(x <= y); }); break;
      case GM_OP_AND: apply2(S, b_scalar, b, [](RC x, RC y) { return (RC)((x != (RC)0) & (y != (RC)0)); }); break;
      case GM_OP_OR: apply2(S, b_scalar, b, [](RC x, RC y) { return (RC)((x != (RC)0) | (y != (RC)0)); }); break;
      case GM_OP_XOR: apply2(S, b_scalar, b, [](RC x, RC y) { return (RC)((x != (RC)0) != (y != (RC)0)); }); break;
      case GM_OP_INVERT:
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = (RC)(acc[k] == (RC)0);
        break;
      case GM_OP_ISDATA:
      case GM_OP_ISNODATA: {
        const bool a_has = opd_has_ndv(S.a), want = opcode == GM_OP_ISNODATA;
        const RC a_ndv = scalar_of<RC>(S.a.value);
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = (RC)((a_has & (acc[k] == a_ndv)) == want);
      } break;
      case GM_OP_CLIP: {
        // misc.py:115-123: drop where the mask source is no-data / False (bool source);
        // fill == the store's own no-data value
        const bool b_has = opd_has_ndv(S.b), b_bool = opd_is_bool(S.b);
        const RC b_ndv = scalar_of<RC>(S.b.value), fill = scalar_of<RC>(S.fill);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const bool drop = (b_has & (b[k] == b_ndv)) | (b_bool & (b[k] == (RC)0));
          acc[k] = drop ? fill : acc[k];
        }
      } break;
      case GM_OP_MASK: {
        // misc.py:213-221: value where get_index() says data, else fill
        const RC value = scalar_of<RC>(S.aux0), fill = scalar_of<RC>(S.fill);
        const unsigned flags = step_flags(S);
        if (flags & 4u) {  // integer tiles: exact !=
          const bool a_has = opd_has_ndv(S.a);
          const RC a_ndv = scalar_of<RC>(S.a.value);
#pragma unroll
          for (int k = 0; k < N; ++k) acc[k] = (a_has & (acc[k] == a_ndv)) ? fill : value;
        } else {
          const double y = S.aux2.f64, tol = S.aux1.f64;
          const bool in_f64 = (flags & 1u) != 0, y_finite = (flags & 2u) != 0;
#pragma unroll
          for (int k = 0; k < N; ++k) acc[k] = is_close<RC>(acc[k], y, tol, in_f64, y_finite) ? fill : value;
        }
      } break;
      case GM_OP_FILLNODATA: {
        // elemwise.py:749-755: take b where b is data (utils.get_index: isclose for float tiles)
        const unsigned flags = step_flags(S);
        if (flags & 4u) {
          const bool b_has = opd_has_ndv(S.b);
          const RC b_ndv = scalar_of<RC>(S.b.value);
#pragma unroll
          for (int k = 0; k < N; ++k) acc[k] = (b_has & (b[k] == b_ndv)) ? acc[k] : b[k];
        } else {
          const double y = S.aux2.f64, tol = S.aux1.f64;
          const bool in_f64 = (flags & 1u) != 0, y_finite = (flags & 2u) != 0;
#pragma unroll
          for (int k = 0; k < N; ++k) acc[k] = is_close<RC>(b[k], y, tol, in_f64, y_finite) ? acc[k] : b[k];
        }
      } break;
      case GM_OP_MASKBELOW: {
        // misc.py:249-250: values[values < value] = no_data_value
        const RC thr = scalar_of<RC>(S.aux0), fill = scalar_of<RC>(S.fill);
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = (acc[k] < thr) ? fill : acc[k];
      } break;
      case GM_OP_STEP: {
        // misc.py:314-326: left / at / right around `location`, no-data cells kept
        const RC loc = scalar_of<RC>(S.aux0), left = scalar_of<RC>(S.aux1), at = scalar_of<RC>(S.aux2),
                 right = scalar_of<RC>(S.aux3), fill = scalar_of<RC>(S.fill);
        const bool a_has = opd_has_ndv(S.a);
        const RC a_ndv = scalar_of<RC>(S.a.value);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const RC x = acc[k];
          RC r = x;  // NaN compares false three times and stays
          r = (x > loc) ? right : r;
          r = (x == loc) ? at : r;
          r = (x < loc) ? left : r;
          acc[k] = (a_has & (x == a_ndv)) ? fill : r;
        }
      } break;
      case GM_OP_DIGITIZE:
        if constexpr (TABLES) {
          const unsigned flags = step_flags(S);
          op_digitize(C, (unsigned)S.aux0.u64, (unsigned)S.aux1.u64, (flags & 1u) != 0, (flags & 2u) != 0,
                      opd_has_ndv(S.a), scalar_of<RC>(S.a.value), scalar_of<RC>(S.fill));
        }
        break;
      case GM_OP_LOOKUP:
        if constexpr (TABLES) {
          op_lookup(C, (unsigned)S.aux0.u64, (long long)S.aux1.i64, S.aux2.u64, (unsigned)S.aux3.u64,
                    (step_flags(S) & 1u) != 0, scalar_of<RC>(S.fill));
        }
        break;
      case GM_OP_NANIFNODATA:
        if constexpr (is_float_class<RC>::value) {
          // reduction.py:102-107: stack_array[i, index] = values[index], NaN elsewhere
          const unsigned flags = step_flags(S);
          const RC nan = (RC)__builtin_nanf("");
          if (flags & 4u) {  // integer tile (held exactly in the float class): exact !=
            const bool a_has = opd_has_ndv(S.a);
            const RC a_ndv = scalar_of<RC>(S.a.value);
#pragma unroll
            for (int k = 0; k < N; ++k) acc[k] = (a_has & (acc[k] == a_ndv)) ? nan : acc[k];
          } else {
            const double y = S.aux2.f64, tol = S.aux1.f64;
            const bool in_f64 = (flags & 1u) != 0, y_finite = (flags & 2u) != 0;
#pragma unroll
            for (int k = 0; k < N; ++k) acc[k] = is_close<RC>(acc[k], y, tol, in_f64, y_finite) ? nan : acc[k];
          }
          if (flags & 8u) {
            // the work dtype is float16 and the raster is wider than it (a 1-byte output dtype over int32 / float
            // rasters): the assignment into the float16 stack rounds the value -- once, to nearest even, from
            // the class's own precision (float64 -> float16 directly, like NumPy's cast) -- and overflows to inf
#pragma unroll
            for (int k = 0; k < N; ++k) {
              if constexpr (std::is_same<RC, double>::value) {
                // one rounding, straight from the float64 value (npy_double_to_half): float64 -> float32 rounded to
                // odd keeps the sticky information the float32 -> float16 rounding needs
                const double d = acc[k];
                float f = __double2float_rz(d);
                if ((double)f != d && d == d && fabs(d) <= 3.0e38) f = __uint_as_float(__float_as_uint(f) | 1u);
                acc[k] = (RC)(float)(_Float16)f;
              } else {
                acc[k] = (RC)(_Float16)acc[k];
              }
            }
          }
        }
        break;
      case GM_OP_FMAX:
        if constexpr (is_float_class<RC>::value) {
#pragma unroll
          // (a tie -- the two zeros -- goes to the later operand, like the minima / maxima of gm_reduce.hip and gm_temporal.hip:
          //  NumPy's loops on x86 hosts end in maxps, which hands back its second operand when the two compare equal)
          for (int k = 0; k < N; ++k) acc[k] = ((acc[k] > b[k]) | (b[k] != b[k])) ? acc[k] : b[k];
        }
        break;
      case GM_OP_NANTOFILL:
        if constexpr (is_float_class<RC>::value) {
          const RC fill = scalar_of<RC>(S.fill);
#pragma unroll
          for (int k = 0; k < N; ++k) acc[k] = (acc[k] != acc[k]) ? fill : acc[k];
        }
        break;
      case GM_OP_COPY:
        if constexpr (is_float_class<RC>::value) {
          if (step_flags(S) & 1u) {
            // the copy feeds a float -> integer store that has to come out like NumPy's cast on x86 (cvttss2si /
            // cvttsd2si): whatever lies outside the 32-bit range -- inf and NaN included -- converts to the
            // "integer indefinite" value, INT32_MIN (its low byte / half: 0).  reduce_rasters assigns float16 /
            // float32 statistics into integer rasters without a finite check (reduction.py:115): a sum that
            // overflowed float16 reads 0 in a uint8 result, not the 255 a saturating conversion gives
#pragma unroll
            for (int k = 0; k < N; ++k) acc[k] = (acc[k] >= (RC)-2147483648.0 && acc[k] < (RC)2147483648.0) ? acc[k] : (RC)-2147483648.0;
          }
        }
        break;
      default:
        break;  // the accumulator already holds the operand
    }
    if constexpr (NT > 0) {
      const unsigned save = step_save_temp(S);
      if (save) {
        switch (save - 1) {
#define GM_SAVE_CASE(K)                                                    \
  case K:                                                                  \
    if constexpr (K < NT) {                                                \
      _Pragma("unroll") for (int k = 0; k < N; ++k) T<K>()[k] = acc[k];    \
    }                                                                      \
    break;
          GM_SAVE_CASE(0) GM_SAVE_CASE(1) GM_SAVE_CASE(2) GM_SAVE_CASE(3)
#undef GM_SAVE_CASE
          default: break;
        }
      }
    }
  }

  // ---- fast routes ---------------------------------------------------------------------
  // acc[k] = bad ? fill : f(acc[k], y) with y = literal (B_SCALAR) or b[k]; all-VALU select
  // chain (v_cmp straight into v_cndmask, no lane-mask arithmetic on the shared scalar unit)
  template <bool B_SCALAR, typename F>
  __device__ __forceinline__ void fast_body(const gm_uop& U, const RC* b, F f) {
    const unsigned arg = U.arg;
    const bool a_has = (arg & 1u) != 0, b_has = (arg & 2u) != 0;
    RC a_ndv = scalar_of<RC>(U.acc_ndv), b_ndv = scalar_of<RC>(U.b_val);
    const RC b_lit = b_ndv, fill = scalar_of<RC>(U.fill);
    if constexpr (is_float_class<RC>::value) {
      a_ndv = a_has ? a_ndv : (RC)__builtin_nanf("");
      b_ndv = b_has ? b_ndv : (RC)__builtin_nanf("");
      // no-data tests first (lane masks), then the operation IN PLACE: the old accumulator value is dead
      // when its result is born, so the result can keep its register (with the test after the operation
      // both are live at once, the result lands elsewhere, and every micro-op ends in N register moves)
      bool nodata[N];
#pragma unroll
      for (int k = 0; k < N; ++k) {
        nodata[k] = acc[k] == a_ndv;
        if constexpr (!B_SCALAR) nodata[k] = nodata[k] | (b[k] == b_ndv);
      }
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] = f(acc[k], B_SCALAR ? b_lit : b[k]);
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] = (nodata[k] | !finite_rc<RC>(acc[k])) ? fill : acc[k];
    } else {
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const RC x = acc[k];
        const RC y = B_SCALAR ? b_lit : b[k];
        bool bad = a_has & (x == a_ndv);
        if constexpr (!B_SCALAR) bad = bad | (b_has & (y == b_ndv));
        acc[k] = bad ? fill : f(x, y);
      }
    }
  }

  template <bool B_SCALAR, typename F>
  __device__ __forceinline__ void fast_math(const gm_uop& U, const RC* b, F f) {
    if constexpr (std::is_same<RC, double>::value) {
      if (U.arg & 8u) {
        fast_body<B_SCALAR>(U, b, [f](RC x, RC y) { return (RC)(float)f(x, y); });
        return;
      }
    }
    if constexpr (std::is_same<RC, long long>::value) {
      if (U.arg & 16u) {
        fast_body<B_SCALAR>(U, b, [f](RC x, RC y) { return (RC)(int)f(x, y); });
        return;
      }
    }
    fast_body<B_SCALAR>(U, b, f);
  }

  template <bool B_SCALAR>
  __device__ __forceinline__ void fast_op(const gm_uop& U, const unsigned kind, const RC* b) {
    switch (kind) {  // `kind` is a literal at every call site: the switch folds away
      case UOP_ADD: fast_math<B_SCALAR>(U, b, [](RC x, RC y) { return op_add<RC>(x, y); }); break;
      case UOP_SUB: fast_math<B_SCALAR>(U, b, [](RC x, RC y) { return op_sub<RC>(x, y); }); break;
      case UOP_RSUB: fast_math<B_SCALAR>(U, b, [](RC x, RC y) { return op_sub<RC>(y, x); }); break;
      case UOP_MUL: fast_math<B_SCALAR>(U, b, [](RC x, RC y) { return op_mul<RC>(x, y); }); break;
      case UOP_DIV: fast_math<B_SCALAR>(U, b, [](RC x, RC y) { return op_div<RC>(x, y); }); break;
      case UOP_RDIV: fast_math<B_SCALAR>(U, b, [](RC x, RC y) { return op_div<RC>(y, x); }); break;
      case UOP_EQ: fast_body<B_SCALAR>(U, b, [](RC x, RC y) { return (RC)(x == y); }); break;
      case UOP_NE: fast_body<B_SCALAR>(U, b, [](RC x, RC y) { return (RC)(x != y); }); break;
      case UOP_GT: fast_body<B_SCALAR>(U, b, [](RC x, RC y) { return (RC)(x > y); }); break;
      case UOP_GE: fast_body<B_SCALAR>(U, b, [](RC x, RC y) { return (RC)(x >= y); }); break;
      case UOP_LT: fast_body<B_SCALAR>(U, b, [](RC x, RC y) { return (RC)(x < y); }); break;
      case UOP_LE: fast_body<B_SCALAR>(U, b, [](RC x, RC y) { return (RC)(x <= y); }); break;
      case UOP_CLIP:
        if constexpr (!B_SCALAR) {
          // misc.py:115-123: acc (the store) -> its own ndv where the mask source has no data
          const bool b_has = (U.arg & 2u) != 0, b_bool = (U.arg & 4u) != 0;
          const RC b_ndv = scalar_of<RC>(U.b_val), fill = scalar_of<RC>(U.fill);
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const bool drop = (b_has & (b[k] == b_ndv)) | (b_bool & (b[k] == (RC)0));
            acc[k] = drop ? fill : acc[k];
          }
        }
        break;
      default: break;
    }
  }

  // operand select of a micro-op: y = input k / temp k (wave-uniform compare tree, static indices)
  __device__ __forceinline__ void select(const unsigned src, RC (&y)[N]) const {
    switch (src) {
    // (the empty asm keeps the compiler from running one case's copy speculatively and overriding it
    //  with another's: exactly one copy of N registers is executed)
#define GM_SEL_IN(K)                                                    \
  case UOP_SRC_INPUT + K:                                               \
    if constexpr (K < NIN) {                                            \
      asm volatile("");                                                 \
      _Pragma("unroll") for (int k = 0; k < N; ++k) y[k] = in[K][k];    \
    }                                                                   \
    break;
#define GM_SEL_TMP(K)                                                   \
  case UOP_SRC_TEMP + K:                                                \
    if constexpr (K < NT) {                                             \
      asm volatile("");                                                 \
      _Pragma("unroll") for (int k = 0; k < N; ++k) y[k] = T<K>()[k];   \
    }                                                                   \
    break;
      GM_SEL_IN(0) GM_SEL_IN(1) GM_SEL_IN(2) GM_SEL_IN(3) GM_SEL_IN(4) GM_SEL_IN(5) GM_SEL_IN(6) GM_SEL_IN(7)
      GM_SEL_IN(8) GM_SEL_IN(9) GM_SEL_IN(10) GM_SEL_IN(11) GM_SEL_IN(12) GM_SEL_IN(13) GM_SEL_IN(14) GM_SEL_IN(15)
      GM_SEL_TMP(0) GM_SEL_TMP(1) GM_SEL_TMP(2) GM_SEL_TMP(3)
#undef GM_SEL_IN
#undef GM_SEL_TMP
      default:
#pragma unroll
        for (int k = 0; k < N; ++k) y[k] = (RC)0;
        break;
    }
  }

  // One micro-op: two-level wave-uniform decode (operand source, then operation).  The operand
  // copy costs N v_mov; what it buys is a decode of ~10 scalar instructions instead of a linear
  // chain over every (operation, source) pair -- with 4-8 elements per lane the decode, not the
  // VALU work, is what a micro-op costs.
  __device__ __forceinline__ void exec(kargs_t C, const gm_uop& U) {
    const unsigned kind = U.route & 0xffu, src = (U.route >> 8) & 0xffu;
    if (kind >= UOP_DIGITIZE) {
      const unsigned arg = U.arg;
      if (kind == UOP_CLIPCMP) {
        RC y[N];
        select(src, y);
        const bool y_has = (arg & 1u) != 0, ndv_drops = (arg & 2u) != 0;
        const RC y_ndv = scalar_of<RC>(U.acc_ndv), lit = scalar_of<RC>(U.b_val), fill = scalar_of<RC>(U.fill);
        auto clip_if = [&](auto keep) {
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const bool nodata = y_has & (y[k] == y_ndv);
            const bool drop = nodata ? ndv_drops : !keep(y[k], lit);
            acc[k] = drop ? fill : acc[k];
          }
        };
        switch ((arg >> 8) & 7u) {
          case 0: clip_if([](RC a, RC b) { return a == b; }); break;
          case 1: clip_if([](RC a, RC b) { return a != b; }); break;
          case 2: clip_if([](RC a, RC b) { return a > b; }); break;
          case 3: clip_if([](RC a, RC b) { return a >= b; }); break;
          case 4: clip_if([](RC a, RC b) { return a < b; }); break;
          default: clip_if([](RC a, RC b) { return a <= b; }); break;
        }
      } else if (kind == UOP_MASKI) {
        const bool a_has = (arg & 1u) != 0;
        const RC a_ndv = scalar_of<RC>(U.acc_ndv), value = scalar_of<RC>(U.b_val), fill = scalar_of<RC>(U.fill);
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = (a_has & (acc[k] == a_ndv)) ? fill : value;
      } else if constexpr (TABLES) {
        if (kind == UOP_DIGITIZE)
          op_digitize(C, (arg >> 16) & 3u, (arg >> 8) & 0xffu, (arg & 2u) != 0, (arg & 4u) != 0, (arg & 1u) != 0,
                      scalar_of<RC>(U.acc_ndv), scalar_of<RC>(U.fill));
        else
          op_lookup(C, (arg >> 16) & 3u, (long long)U.b_val.i64, U.acc_ndv.u64, (arg >> 8) & 0xfu, (arg & 1u) != 0,
                    scalar_of<RC>(U.fill));
      }
    } else if (kind >= UOP_ADD) {
      // bits 16-23 of the route: a tile / temp to load into the accumulator first (a leading LD folded
      // into the operation by lower(): one micro-op fetch and decode less)
      // (only the small LIGHT kernels: in the others the extra operand select costs registers -- the
      //  4-input LIGHT kernel went from 95 to 97 VGPRs, i.e. from 5 to 4 waves per SIMD)
      if constexpr (FOLD_LD) {
        const unsigned first = (U.route >> 16) & 0xffu;
        if (first) select(first, acc);
      }
      if (src == UOP_SRC_SCALAR) {
        fast_op<true>(U, kind, acc);
      } else {
        RC y[N];
        select(src, y);
        fast_op<false>(U, kind, y);
      }
    } else if (kind == UOP_LD) {
      if (src == UOP_SRC_SCALAR) {
        const RC v = scalar_of<RC>(U.b_val);
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = v;
      } else {
        select(src, acc);
      }
    } else if (kind == UOP_ST) {
      switch (src) {
#define GM_ST_TMP(K)                                                     \
  case K:                                                                \
    if constexpr (K < NT) {                                              \
      _Pragma("unroll") for (int k = 0; k < N; ++k) T<K>()[k] = acc[k];  \
    }                                                                    \
    break;
        GM_ST_TMP(0) GM_ST_TMP(1) GM_ST_TMP(2) GM_ST_TMP(3)
#undef GM_ST_TMP
        default: break;
      }
    } else {  // UOP_GENERIC
      const gm_step S = kload<gm_step>(C, offsetof(VmCode, steps) + sizeof(gm_step) * U.arg);
      step(C, S);
    }
  }

  __device__ __forceinline__ void run(kargs_t C, const unsigned n_uops) {
    for (unsigned i = 0; i < n_uops; ++i) {
      const gm_uop U = kload<gm_uop>(C, offsetof(VmCode, uops) + sizeof(gm_uop) * i);
      exec(C, U);
    }
  }
};

template <typename RC> struct native_vec;
template <> struct native_vec<float> { typedef f32x4 type; };
template <> struct native_vec<int> { typedef i32x4 type; };

// vector kernel: blockIdx.y = tile of the batch, blockIdx.x = span of 256*G groups of 4 elements
// LOOKUP window -> LDS: 4 loads of a lane are in flight before the first LDS write (a load -> wait ->
// write loop would pay one L2 round trip per 2 KiB of window)
__device__ __forceinline__ void stage_window(unsigned long long* lds, const unsigned long long* __restrict__ src,
                                             unsigned words) {
  for (unsigned i = threadIdx.x; i < words; i += 1024) {
    unsigned long long v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned k = i + 256u * u;
      v[u] = src[k < words ? k : words - 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned k = i + 256u * u;
      if (k < words) lds[k] = v[u];
    }
  }
}

// Where the program comes from.  PROG = InterpProg: read from device memory at run time (one kernel per
// register class / size / flavour serves every program).  gm_spec.hip generates, per hot program, a PROG
// whose members are literals: the same body and the same Machine then compile into a kernel with the
// decode, the operand copies and the dtype dispatch folded away (the arithmetic is this file's, so the
// results are the interpreter's bit for bit).
struct InterpProg {
  static __device__ __forceinline__ unsigned flags(kargs_t K) { return GM_KARG(K, flags); }
  static __device__ __forceinline__ unsigned n_inputs(kargs_t K) { return GM_KARG(K, n_inputs); }
  static __device__ __forceinline__ int out_dtype(kargs_t K) { return (int)GM_KARG(K, out_dtype); }
  static __device__ __forceinline__ unsigned lds_table(kargs_t K) { return GM_KARG(K, lds_table); }
  static __device__ __forceinline__ gm_input_desc input(kargs_t C, int k) {
    return kload<gm_input_desc>(C, offsetof(VmCode, inputs) + sizeof(gm_input_desc) * k);
  }
  template <typename M>
  static __device__ __forceinline__ void run(M& m, kargs_t K, kargs_t C) { m.run(C, GM_KARG(K, n_uops)); }
};

template <typename RC, int G, int NIN, int NT, int FLAVOR, typename PROG>
__device__ __forceinline__ void vm_body() {
  const kargs_t K = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int N = 4 * G;
  const unsigned tile = blockIdx.y;
  const unsigned flags = PROG::flags(K);
  const unsigned long long n_groups = GM_KARG(K, n_groups);
  const unsigned n_in = PROG::n_inputs(K);
  const kargs_t C = (kargs_t)(uintptr_t)GM_KARG(K, code);
  const size_t in_base = offsetof(VmLaunch, in) + sizeof(void*) * GM_MAX_INPUTS * tile;
  constexpr bool nt_ld = true, nt_st = true;
  // the whole pointer row of this tile in one scalar load (slots past n_inputs are null, never used)
  const void* in_ptr[NIN];
#pragma unroll
  for (int k = 0; k < NIN; ++k) in_ptr[k] = kload<const void*>(K, in_base + sizeof(void*) * k);
  bool native = false;
  if constexpr (sizeof(RC) == 4) native = (flags & VM_FLAG_NATIVE) != 0;
  void* const out = kload<void*>(K, offsetof(VmLaunch, out) + sizeof(void*) * tile);

  // TABLES kernels walk `spans` consecutive spans so that the LDS staging of a LOOKUP window is
  // amortised; every other kernel does exactly one span (full grid)
  // (a span loop in the other flavours was measured: it never helps them -- 5 % slower at 2 spans, 11 % at
  //  4 on the canonical chain -- and its mere presence costs the 4-input LIGHT kernel 10 %)
  unsigned spans = 1;
  if constexpr (FLAVOR == VM_TABLES) spans = GM_KARG(K, spans);
  for (unsigned span = 0; span < spans; ++span) {
  const unsigned long long g0 = ((unsigned long long)blockIdx.x * spans + span) * (256ull * G) + threadIdx.x;  // sub-vector u: group g0 + 256*u

  // Strided inputs -- zero-copy windows of a larger raster resident in HBM (a tile request into a
  // MemorySource): group g of the tile sits at row g / (cols/4), column 4 * (g % (cols/4)); the
  // element offset of input k is row * pitch_k + column (4 g for contiguous inputs).  One 32-bit
  // division per group, shared by all inputs.
  const bool strided = (flags & VM_FLAG_STRIDED) != 0;
  unsigned row_of[G], col_of[G];
  if (strided) {
    const unsigned cols4 = GM_KARG(K, cols) >> 2;
    const unsigned magic = GM_KARG(K, div_magic), sh1 = GM_KARG(K, div_sh1), sh2 = GM_KARG(K, div_sh2);
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const unsigned g32 = (unsigned)min(g0 + 256ull * u, n_groups - 1);  // (clamped like the loads below)
      const unsigned t = __umulhi(g32, magic);
      row_of[u] = (t + ((g32 - t) >> sh1)) >> sh2;
      col_of[u] = (g32 - row_of[u] * cols4) << 2;
    }
  }
  const size_t pitch_base = offsetof(VmLaunch, pitch) + sizeof(unsigned) * GM_MAX_INPUTS * tile;
  auto elem_offset = [&](int k, int u, unsigned long long g) -> unsigned long long {
    if (strided) {
      const unsigned pitch = kload<unsigned>(K, pitch_base + sizeof(unsigned) * k);
      if (pitch) return (unsigned long long)row_of[u] * pitch + col_of[u];
    }
    return g * 4ull;
  };

  Machine<RC, N, NIN, NT, FLAVOR> M;
  // phase 1: all leaf loads in flight (NIN*G independent 16-byte loads per lane)
  if (native) {
    // every tile already has the class' dtype: straight 16-byte loads, no dtype dispatch
    if constexpr (sizeof(RC) == 4) {
      typedef typename native_vec<RC>::type V4;
#pragma unroll
      for (int k = 0; k < NIN; ++k) {
        if (k < (int)n_in) {
          const V4* const p = (const V4*)in_ptr[k];
#pragma unroll
          for (int u = 0; u < G; ++u) {
            // (groups past the tile re-read its last group: an unconditional load -- no zero fill, no
            //  branch around it -- whose result is computed on and never stored)
            const unsigned long long g = min(g0 + 256ull * u, n_groups - 1);
            const V4 x = ld((const V4*)((const RC*)p + elem_offset(k, u, g)), nt_ld);
#pragma unroll
            for (int q = 0; q < 4; ++q) M.in[k][4 * u + q] = x[q];
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      if (k < (int)n_in) {
        const gm_input_desc D = PROG::input(C, k);
        const void* const in_k = in_ptr[k];
        if (in_broadcast(D)) {
          RC one[1];
          load_elems<RC, 1>(in_k, in_dtype(D), 0, false, one, in_native(D), (long long)D.ndv.i64);
#pragma unroll
          for (int q = 0; q < N; ++q) M.in[k][q] = one[0];
        } else {
#pragma unroll
          for (int u = 0; u < G; ++u) {
            RC v4[4] = {(RC)0, (RC)0, (RC)0, (RC)0};
            const unsigned long long g = g0 + 256ull * u;
            if (g < n_groups)
              load_elems<RC, 4>(in_k, in_dtype(D), elem_offset(k, u, g), nt_ld, v4, in_native(D), (long long)D.ndv.i64);
#pragma unroll
            for (int q = 0; q < 4; ++q) M.in[k][4 * u + q] = v4[q];
          }
        }
      }
    }
  }
  if constexpr (FLAVOR == VM_MATH && std::is_same<RC, float>::value) {
    if (span == 0) {  // the tables of np.power / np.log into LDS while the tile loads are in flight
      powf_lds_stage();
      __syncthreads();
    }
  }
  if constexpr (FLAVOR == VM_TABLES) {
    const unsigned lds_table = PROG::lds_table(K);
    if (lds_table) {
      if (span == 0) {  // stage the LOOKUP window while the tile loads are in flight
        extern __shared__ __align__(16) unsigned long long vm_lds[];
        const unsigned words = GM_KARG(K, lds_words);
        const unsigned long long* const src =
            (const unsigned long long*)(uintptr_t)kload<unsigned long long>(C, offsetof(VmCode, tables) + 8 * (lds_table - 1));
        stage_window(vm_lds, src, words);
        __syncthreads();
      }
    }
  }
  // phase 2: interpret
  PROG::run(M, K, C);
  // phase 3: store (the result may itself be a window: a tile written straight into its mosaic)
  unsigned out_pitch = 0;
  if (strided) out_pitch = kload<unsigned>(K, offsetof(VmLaunch, out_pitch) + sizeof(unsigned) * tile);
  if constexpr (G == 2 && sizeof(RC) == 4) {
    // 1-byte results, both groups of every lane of this wave inside the tile: the 4 bytes of a group are
    // a 4-byte store per lane as they stand.  The wave transposes its 2 x 64 words through 512 bytes of
    // its own LDS instead (2 ds_write_b32 + 1 ds_read_b64 per lane, no barrier: one wave, in-order LDS)
    // and lane l stores the 8 consecutive pixels 8 l .. 8 l + 7 of the 256-pixel run it belongs to.
    const int od = PROG::out_dtype(K);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long wave_g0 = g0 - lane;
    if ((od == GM_U8 || od == GM_I8 || od == GM_BOOL) && !out_pitch && wave_g0 + 256ull + 63ull < n_groups) {
      __shared__ unsigned vm_swap[4][128];
      unsigned* const mine = vm_swap[threadIdx.x >> 6];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        unsigned word = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const RC r = M.acc[4 * u + q];
          const unsigned byte = (unsigned)(unsigned char)(int)r;  // (through int32: see to_stored)
          word |= byte << (8 * q);
        }
        mine[64 * u + lane] = word;
      }
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 pair = *(const u32x2*)(mine + 2 * lane);
      const unsigned long long px = (wave_g0 + 256ull * (lane >> 5)) * 4ull + 8ull * (lane & 31u);
      st((u32x2*)((unsigned char*)out + px), pair, nt_st);
      continue;  // next span
    }
  }
#pragma unroll
  for (int u = 0; u < G; ++u) {
    const unsigned long long g = g0 + 256ull * u;
    if (g < n_groups) {
      const unsigned long long at = out_pitch ? (unsigned long long)row_of[u] * out_pitch + col_of[u] : g * 4ull;
      if (native) {
        if constexpr (sizeof(RC) == 4) {
          typedef typename native_vec<RC>::type V4;
          V4 x;
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] = M.acc[4 * u + q];
          st((V4*)((RC*)out + at), x, nt_st);
        }
      } else {
        RC v4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v4[q] = M.acc[4 * u + q];
        store_elems<RC, 4>(out, PROG::out_dtype(K), at, nt_st, v4);
      }
    }
  }
  }  // spans
}

template <typename RC, int G, int NIN, int NT, int FLAVOR>
__global__ __launch_bounds__(256) void vm_kernel(const VmLaunch) {
  vm_body<RC, G, NIN, NT, FLAVOR, InterpProg>();
}

// scalar tail kernel: the elements the 16-byte vector path cannot take (n % 4, or
// everything when a pointer is not 16-byte aligned); one element per lane, grid-stride
template <typename RC, int NIN, int NT, int FLAVOR>
__global__ __launch_bounds__(256) void vm_tail_kernel(const VmLaunch) {
  const kargs_t K = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
  const kargs_t C = (kargs_t)(uintptr_t)GM_KARG(K, code);
  const unsigned tile = blockIdx.y;
  const unsigned n_in = GM_KARG(K, n_inputs);
  const unsigned n_uops = GM_KARG(K, n_uops);
  const int out_dtype = (int)GM_KARG(K, out_dtype);
  const unsigned long long n_total = GM_KARG(K, n);
  const size_t in_base = offsetof(VmLaunch, in) + sizeof(void*) * GM_MAX_INPUTS * tile;
  void* const out = kload<void*>(K, offsetof(VmLaunch, out) + sizeof(void*) * tile);
  const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
  if constexpr (FLAVOR == VM_MATH && std::is_same<RC, float>::value) {
    powf_lds_stage();
    __syncthreads();
  }
  if constexpr (FLAVOR == VM_TABLES) {
    const unsigned lds_table = GM_KARG(K, lds_table);
    if (lds_table) {  // the LOOKUP window lives in LDS here as well
      extern __shared__ __align__(16) unsigned long long vm_lds[];
      const unsigned words = GM_KARG(K, lds_words);
      const unsigned long long* const src =
          (const unsigned long long*)(uintptr_t)kload<unsigned long long>(C, offsetof(VmCode, tables) + 8 * (lds_table - 1));
      stage_window(vm_lds, src, words);
      __syncthreads();
    }
  }
  const bool strided = (GM_KARG(K, flags) & VM_FLAG_STRIDED) != 0;
  const unsigned long long cols = GM_KARG(K, cols);
  const size_t pitch_base = offsetof(VmLaunch, pitch) + sizeof(unsigned) * GM_MAX_INPUTS * tile;
  for (unsigned long long e = GM_KARG(K, tail_start) + tid; e < n_total; e += nthreads) {
    Machine<RC, 1, NIN, NT, FLAVOR> M1;
    unsigned long long row = 0, col = 0;
    if (strided) {
      row = e / cols;
      col = e - row * cols;
    }
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      if (k < (int)n_in) {
        const gm_input_desc D = kload<gm_input_desc>(C, offsetof(VmCode, inputs) + sizeof(gm_input_desc) * k);
        const void* const in_k = kload<const void*>(K, in_base + sizeof(void*) * k);
        unsigned long long at = e;
        if (strided) {
          const unsigned pitch = kload<unsigned>(K, pitch_base + sizeof(unsigned) * k);
          if (pitch) at = row * pitch + col;
        }
        load_elems<RC, 1>(in_k, in_dtype(D), in_broadcast(D) ? 0ull : at, false, M1.in[k], in_native(D),
                          (long long)D.ndv.i64);
      }
    }
    M1.run(C, n_uops);
    unsigned long long out_at = e;
    if (strided) {
      const unsigned out_pitch = kload<unsigned>(K, offsetof(VmLaunch, out_pitch) + sizeof(unsigned) * tile);
      if (out_pitch) out_at = row * out_pitch + col;
    }
    store_elems<RC, 1>(out, out_dtype, out_at, false, M1.acc);
  }
}

#ifndef __HIPCC_RTC__
// host-side launcher of one register class (defined once per class in gm_vm_<class>.hip).
// size: 0 = S (<= 2 inputs, <= 1 temp), 1 = M (<= 4 inputs, <= 2 temps), 2 = L (8 inputs, 4 temps)
// flavor: VM_LIGHT / VM_MATH / VM_TABLES; grid.y = tail_grid.y = tiles of the batch
template <typename RC>
int launch_class(const VmLaunch& args, int size, int G, int flavor, dim3 grid, dim3 tail_grid, hipStream_t stream);

#define GM_VM_LAUNCH(RC, G_, NIN_, NT_, FLAVOR_)                                                          \
  do {                                                                                                   \
    if (grid.x)                                                                                          \
      hipLaunchKernelGGL((vm_kernel<RC, G_, NIN_, NT_, FLAVOR_>), grid, dim3(256),                       \
                         (FLAVOR_ == VM_TABLES && args.lds_table) ? (size_t)args.lds_words * 8 : 0, stream, args); \
    if (tail_grid.x)                                                                                     \
      hipLaunchKernelGGL((vm_tail_kernel<RC, NIN_, NT_, FLAVOR_>), tail_grid, dim3(256),                 \
                         (FLAVOR_ == VM_TABLES && args.lds_table) ? (size_t)args.lds_words * 8 : 0, stream, args); \
  } while (0)

#define GM_VM_LAUNCH_SIZES(RC, FLAVOR_, G_S, G_M)                     \
  do {                                                               \
    if (size == 0) {                                                 \
      if (G >= 2 && G_S >= 2) GM_VM_LAUNCH(RC, G_S, 2, 1, FLAVOR_);  \
      else GM_VM_LAUNCH(RC, 1, 2, 1, FLAVOR_);                       \
    } else if (size == 1) {                                          \
      if (G >= 2 && G_M >= 2) GM_VM_LAUNCH(RC, G_M, 4, 2, FLAVOR_);  \
      else GM_VM_LAUNCH(RC, 1, 4, 2, FLAVOR_);                       \
    } else if (size == 2) {                                          \
      GM_VM_LAUNCH(RC, 1, 8, 4, FLAVOR_);                            \
    } else {                                                         \
      GM_VM_LAUNCH(RC, 1, 16, 4, FLAVOR_);                           \
    }                                                                \
  } while (0)

#define GM_VM_DEFINE_LAUNCHER(RC)                                                                  \
  template <>                                                                                      \
  int launch_class<RC>(const VmLaunch& args, int size, int G, int flavor, dim3 grid, dim3 tail_grid, \
                       hipStream_t stream) {                                                       \
    if (flavor == VM_MATH) GM_VM_LAUNCH_SIZES(RC, VM_MATH, 1, 1);                                  \
    else if (flavor == VM_TABLES) GM_VM_LAUNCH_SIZES(RC, VM_TABLES, 2, 2);                         \
    else GM_VM_LAUNCH_SIZES(RC, VM_LIGHT, 2, 2);                                                   \
    hipError_t e = hipGetLastError();                                                              \
    if (e != hipSuccess) return gm::hip_fail(e, "vm_kernel launch");                               \
    return GM_OK;                                                                                  \
  }

#endif  // !__HIPCC_RTC__

}  // namespace gmvm
