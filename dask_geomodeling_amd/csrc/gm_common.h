// Shared internals of libgmhip (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "gmhip.h"

namespace gm {

// thread-local error text behind gm_last_error()
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

// current device's compute stream (creates device state on first use)
hipStream_t current_stream();
int current_cu_count();

// Route counters behind gm_route_stats(): every host-side choice between kernels of one entry point counts the route it took,
// so that a test which forces a route can assert that the route ran (a threshold that moves cannot turn it vacuous).
// Families and slots are the GM_ROUTE_* constants of gmhip.h.
void route_hit(int family, int slot, unsigned long long n = 1);

#define GM_HIP(call)                                  \
  do {                                                \
    hipError_t e_ = (call);                           \
    if (e_ != hipSuccess) return gm::hip_fail(e_, #call); \
  } while (0)

// out[i] = table[in[i]] over 1-byte tiles (gm_tables.hip): n_tiles <= 16 tiles of rows x cols elements,
// pitches in elements (0 = contiguous); table: 256 entries of out_size bytes on the device.
// Requirements (checked by the caller): 16-byte aligned tile pointers; strided tiles have cols and
// pitches divisible by 16.
int lut8_launch(int n_tiles, const void* const* in, const uint32_t* in_pitch, void* const* out, const uint32_t* out_pitch,
                size_t rows, size_t cols, const void* table, int out_size, hipStream_t stream);

// A program that is nothing but LOOKUP(int32 tile) (gm_tables.hip): out = (E)(window hit | exception |
// fill / pass-through), exactly Machine::op_lookup of the int32 class.  window: `range` int32 entries on
// the device; exceptions: <= 8 (key, value) pairs.  16-byte aligned tile pointers.
struct Lookup32 {
  const int* window;
  unsigned range;
  int kmin;
  int n_exc;
  int exc_key[8], exc_val[8];
  int select;
  int fill;
};
// in_pitch / out_pitch: elements per row of strided tiles (zero-copy windows), nullptr / 0 = contiguous; strided
// tiles need cols % 16 == 0, in_pitch % 4 == 0, out_pitch % 16 == 0 and fewer than 2^32 elements.
int lookup32_launch(int n_tiles, const void* const* in, const uint32_t* in_pitch, void* const* out,
                    const uint32_t* out_pitch, size_t rows, size_t cols, const Lookup32& table, int out_size,
                    hipStream_t stream);

inline int dtype_size(int dt) {
  switch (dt) {
    case GM_BOOL: case GM_U8: case GM_I8: return 1;
    case GM_U16: case GM_I16: return 2;
    case GM_U32: case GM_I32: case GM_F32: return 4;
    case GM_U64: case GM_I64: case GM_F64: return 8;
    default: return 0;
  }
}
inline bool dtype_is_float(int dt) { return dt == GM_F32 || dt == GM_F64; }

}  // namespace gm
