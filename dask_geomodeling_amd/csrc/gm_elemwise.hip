// K1 host side: validates a gm_program and launches the VM kernel of its register
// class (kernels: gm_vm.h, instantiated per class in gm_vm_<class>.hip).
#include "gm_vm.h"

using namespace gmvm;

namespace {

gm_vm_tuning g_tuning = {8, 0, 0, 0};

int validate_operand(const gm_operand& o, unsigned n_inputs, const char* which, unsigned step) {
  const unsigned kind = o.bits & 0xffu, index = (o.bits >> 8) & 0xffu;
  switch (kind) {
    case GM_OPD_NONE: case GM_OPD_ACC: case GM_OPD_SCALAR: return GM_OK;
    case GM_OPD_INPUT:
      if (index >= n_inputs) return gm::fail(GM_ERR_INVALID, "step %u operand %s: input %u of %u", step, which, index, n_inputs);
      return GM_OK;
    case GM_OPD_TEMP:
      if (index >= GM_MAX_TEMPS) return gm::fail(GM_ERR_INVALID, "step %u operand %s: temp %u", step, which, index);
      return GM_OK;
    default:
      return gm::fail(GM_ERR_INVALID, "step %u operand %s: kind %u", step, which, kind);
  }
}

}  // namespace

extern "C" {

int gm_vm_set_tuning(const gm_vm_tuning* t) {
  if (!t) return gm::fail(GM_ERR_INVALID, "null tuning");
  gm_vm_tuning v = *t;
  if (v.vec != 4 && v.vec != 8 && v.vec != 16) v.vec = 8;
  if (v.blocks_per_cu < 0 || v.blocks_per_cu > 4096) v.blocks_per_cu = 0;
  g_tuning = v;
  return GM_OK;
}

int gm_vm_get_tuning(gm_vm_tuning* t) {
  if (!t) return gm::fail(GM_ERR_INVALID, "null tuning");
  *t = g_tuning;
  return GM_OK;
}

int gm_elemwise_run(const gm_program* prog, const void* const* inputs, void* out, size_t n) {
  if (!prog || !out || (!inputs && prog->n_inputs)) return gm::fail(GM_ERR_INVALID, "null argument");
  if (prog->n_steps == 0 || prog->n_steps > GM_MAX_STEPS) return gm::fail(GM_ERR_INVALID, "n_steps %u", prog->n_steps);
  if (prog->n_inputs > GM_MAX_INPUTS) return gm::fail(GM_ERR_INVALID, "n_inputs %u", prog->n_inputs);
  if (prog->reg_class > GM_CLASS_I64) return gm::fail(GM_ERR_INVALID, "reg_class %u", prog->reg_class);
  if (gm::dtype_size(prog->out_dtype) == 0) return gm::fail(GM_ERR_INVALID, "out_dtype %u", prog->out_dtype);
  for (unsigned k = 0; k < prog->n_inputs; ++k) {
    if (gm::dtype_size(prog->inputs[k].bits & 0xffu) == 0) return gm::fail(GM_ERR_INVALID, "input %u dtype %u", k, prog->inputs[k].bits & 0xffu);
    if (!inputs[k]) return gm::fail(GM_ERR_INVALID, "input %u is null", k);
  }
  for (unsigned s = 0; s < prog->n_steps; ++s) {
    const gm_step& S = prog->steps[s];
    const unsigned opcode = S.bits & 0xffu, save_temp = (S.bits >> 8) & 0xffu;
    if (opcode == GM_OP_NOP || opcode >= GM_OP_COUNT) return gm::fail(GM_ERR_INVALID, "step %u opcode %u", s, opcode);
    if (save_temp > GM_MAX_TEMPS) return gm::fail(GM_ERR_INVALID, "step %u save_temp %u", s, save_temp);
    int rc = validate_operand(S.a, prog->n_inputs, "a", s);
    if (rc) return rc;
    rc = validate_operand(S.b, prog->n_inputs, "b", s);
    if (rc) return rc;
    if (opcode == GM_OP_DIV && prog->reg_class >= GM_CLASS_I32)
      return gm::fail(GM_ERR_INVALID, "step %u: DIV in an integer class", s);
  }
  if (n == 0) return GM_OK;
  hipStream_t stream = gm::current_stream();
  if (!stream) return GM_ERR_NO_DEVICE;

  VmArgs args;
  memset(&args, 0, sizeof(args));
  args.prog = *prog;
  bool aligned = ((uintptr_t)out % 16) == 0;
  for (unsigned k = 0; k < prog->n_inputs; ++k) {
    args.in[k] = inputs[k];
    if (!((prog->inputs[k].bits >> 8) & 1u) && ((uintptr_t)inputs[k] % 16) != 0) aligned = false;
  }
  args.out = out;
  args.n = n;
  bool heavy = false;
  unsigned max_in = prog->n_inputs, max_tmp = 0;
  for (unsigned s = 0; s < prog->n_steps; ++s) {
    const gm_step& S = prog->steps[s];
    const unsigned op = S.bits & 0xffu, save_temp = (S.bits >> 8) & 0xffu;
    if (op == GM_OP_POW || op == GM_OP_EXP || op == GM_OP_LOG || op == GM_OP_LOG10) heavy = true;
    if (save_temp > max_tmp) max_tmp = save_temp;
  }
  const int size = (max_in <= 2 && max_tmp <= 1) ? 0 : ((max_in <= 4 && max_tmp <= 2) ? 1 : 2);
  int G = g_tuning.vec >= 16 ? 4 : (g_tuning.vec >= 8 ? 2 : 1);
  if (heavy || size == 2) G = 1;
  else if (size == 1 && G > 2) G = 2;
  args.n_groups = aligned ? n / 4 : 0;
  args.tail_start = args.n_groups * 4;
  const unsigned long long per_iter = 256ull * G;
  args.n_iters = (args.n_groups + per_iter - 1) / per_iter;
  args.xcd_chunking = g_tuning.xcd_chunking;
  args.nontemporal = g_tuning.nontemporal;

  const int cus = gm::current_cu_count();
  unsigned long long cap = (unsigned long long)(cus > 0 ? cus : 256) * (unsigned long long)g_tuning.blocks_per_cu;
  unsigned long long blocks = args.n_iters;
  if (cap && blocks > cap) blocks = cap;
  if (blocks > 0x7fffffffull) blocks = 0x7fffffffull;
  if (args.xcd_chunking && blocks >= 8) blocks -= blocks % 8;
  const unsigned long long tail = n - args.tail_start;
  unsigned long long tail_blocks = (tail + 255) / 256;
  const unsigned long long tail_cap = (unsigned long long)(cus > 0 ? cus : 256) * 8;
  if (tail_blocks > tail_cap) tail_blocks = tail_cap;
  dim3 grid((unsigned)blocks), tail_grid((unsigned)tail_blocks);

  switch (prog->reg_class) {
    case GM_CLASS_F32: return launch_class<float>(args, size, G, heavy, grid, tail_grid, stream);
    case GM_CLASS_F64: return launch_class<double>(args, size, G, heavy, grid, tail_grid, stream);
    case GM_CLASS_I32: return launch_class<int>(args, size, G, heavy, grid, tail_grid, stream);
    default: return launch_class<long long>(args, size, G, heavy, grid, tail_grid, stream);
  }
}

}  // extern "C"

// ABI self-check used by the Python binding: sizes of the program structs.
extern "C" size_t gm_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(gm_program);
    case 1: return sizeof(gm_step);
    case 2: return sizeof(gm_operand);
    case 3: return sizeof(gm_input_desc);
    default: return 0;
  }
}
