// Internal interface of the program specialiser (gm_spec.hip); not part of the C-ABI.
//
// The interpreter kernels of gm_vm.h serve every program and pay for it per micro-op (scalar decode,
// operand copies, run-time dtype dispatch).  A program that turns out to be HOT -- it has been asked to
// process many pixels -- is compiled once, at run time, into a kernel of its own: the specialiser prints
// the lowered program as literal arguments of the very same Machine / vm_body templates and hands that
// source to hiprtc (gfx950), so the arithmetic is the interpreter's, bit for bit, and what disappears is
// the interpretation.  Code objects are cached in memory per device and on disk by content.
#pragma once
#include <string>
#include <vector>

#include "gm_vm.h"

namespace gmspec {

struct Request {
  const gmvm::VmCode* host = nullptr;   // lowered program (uops, steps, inputs)
  unsigned n_uops = 0;
  unsigned n_inputs = 0;
  unsigned reg_class = 0;
  unsigned out_dtype = 0;
  int flavor = 0;                       // gmvm::VM_LIGHT / VM_MATH / VM_TABLES
  int n_temps = 0;                      // temps the program uses
  bool native = false;
  bool strided = false;                 // launch shape: tiles are windows (row pitches)
  int G = 1;                            // 4-element groups per lane
  const std::vector<unsigned long long>* tables = nullptr;  // device image of all tables (words)
  std::vector<size_t> table_offset;     // word offset of table t in *tables
};

struct Kernel {
  void* function = nullptr;             // hipFunction_t
  bool stages_lds = false;              // the kernel still stages a LOOKUP window in LDS
};

// mode: 0 = off, 1 = hot programs only (default), 2 = every program at its first launch (tests)
int mode();
unsigned long long hot_elements();      // mode 1: elements a program must have been asked for before it is compiled

// The HIP source of the kernel for `r` (no device needed).  `literal_lookup` reports whether every LOOKUP
// window became literals (no LDS staging).
std::string generate(const Request& r, bool* literal_lookup, std::string* kernel_name = nullptr);

// Compile `source` for gfx950 with hiprtc (no device needed); the code object lands in `code`.
int compile(const std::string& source, std::vector<char>* code, std::string* log);

// What identifies a requested kernel once its source has been generated (cheap to poll with).
struct Ticket {
  bool valid = false;
  int device = 0;
  unsigned long long hash = 0;
  char kernel_name[40] = {0};
};

// Kernel for `r` on the current device: memory cache -> disk cache -> compile.  Returns GM_OK and a null
// function when specialisation is unavailable (no hiprtc) -- the caller keeps the interpreter.
// async: the disk read / compilation / module load happens on the library's worker thread; the call returns at
// once with a null function and *pending = true, and the caller keeps launching the interpreter until
// poll_kernel(ticket) hands the kernel over.  The global lock is never held across a compilation.
int get_kernel(const Request& r, Kernel* out, bool async = false, Ticket* ticket = nullptr, bool* pending = nullptr);
int poll_kernel(const Ticket& t, Kernel* out, bool* pending);

int launch(const Kernel& k, const gmvm::VmLaunch& args, unsigned grid_x, unsigned grid_y, size_t shmem, hipStream_t stream);

}  // namespace gmspec
