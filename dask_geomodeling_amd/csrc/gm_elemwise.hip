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

// Lower the program into micro-ops (gm_vm.h).  A step whose accumulator-side operand is the
// running value and whose opcode is +, -, *, / or Clip becomes a fast route; a leading non-ACC
// left operand becomes an explicit LD; save_temp becomes ST.  Anything else stays GENERIC.
int lower(const gm_program* prog, VmArgs* args) {
  unsigned n = 0;
  auto push = [&](unsigned kind, unsigned src, unsigned arg, gm_scalar acc_ndv, gm_scalar b_val, gm_scalar fill) -> bool {
    if (n >= (unsigned)GM_MAX_UOPS) return false;
    gm_uop& u = args->uops[n++];
    u.route = kind | (src << 8);
    u.arg = arg;
    u.acc_ndv = acc_ndv;
    u.b_val = b_val;
    u.fill = fill;
    return true;
  };
  const bool float_class = prog->reg_class <= GM_CLASS_F64;
  gm_scalar zero;
  zero.u64 = 0;
  for (unsigned s = 0; s < prog->n_steps; ++s) {
    const gm_step& S = prog->steps[s];
    const unsigned opcode = S.bits & 0xffu, save = (S.bits >> 8) & 0xffu;
    const unsigned round_f32 = (S.bits >> 24) & 1u, wrap_i32 = (S.bits >> 25) & 1u, chk = (S.bits >> 26) & 1u;
    auto kind_of = [](const gm_operand& o) { return o.bits & 0xffu; };
    auto src_of = [&](const gm_operand& o) -> int {  // -1: not addressable as a fast-route source
      const unsigned k = kind_of(o), i = (o.bits >> 8) & 0xffu;
      if (k == GM_OPD_SCALAR) return (int)UOP_SRC_SCALAR;
      if (k == GM_OPD_INPUT) return (int)(UOP_SRC_INPUT + i);
      if (k == GM_OPD_TEMP) return (int)(UOP_SRC_TEMP + i);
      return -1;
    };
    auto has_ndv = [](const gm_operand& o) { return (o.bits >> 16) & 1u; };
    auto is_bool = [](const gm_operand& o) { return (o.bits >> 17) & 1u; };
    bool fast = false;
    const bool arith = opcode == GM_OP_ADD || opcode == GM_OP_SUB || opcode == GM_OP_MUL ||
                       (opcode == GM_OP_DIV && float_class);
    // the finite check is part of every float fast body: only lower float math steps that ask for it
    if (arith && (!float_class || chk)) {
      const gm_operand *x = &S.a, *y = &S.b;  // x: accumulator side, y: other side
      bool reversed = false, need_ld = false;
      if (kind_of(S.a) == GM_OPD_ACC && kind_of(S.b) != GM_OPD_ACC) {
      } else if (kind_of(S.b) == GM_OPD_ACC && kind_of(S.a) != GM_OPD_ACC) {
        x = &S.b; y = &S.a; reversed = true;
      } else if (kind_of(S.a) != GM_OPD_ACC && kind_of(S.b) != GM_OPD_ACC && kind_of(S.a) != GM_OPD_NONE) {
        need_ld = true;
      } else {
        x = nullptr;
      }
      if (x && src_of(*y) >= 0 && (!need_ld || src_of(*x) >= 0)) {
        unsigned kind = UOP_ADD;
        if (opcode == GM_OP_SUB) kind = reversed ? UOP_RSUB : UOP_SUB;
        else if (opcode == GM_OP_MUL) kind = UOP_MUL;
        else if (opcode == GM_OP_DIV) kind = reversed ? UOP_RDIV : UOP_DIV;
        if (need_ld && !push(UOP_LD, (unsigned)src_of(*x), 0, zero, x->value, zero)) return -1;
        const unsigned x_has = kind_of(*x) == GM_OPD_SCALAR ? 0u : has_ndv(*x);
        const unsigned y_has = kind_of(*y) == GM_OPD_SCALAR ? 0u : has_ndv(*y);
        const unsigned arg = x_has | (y_has << 1) | (round_f32 << 3) | (wrap_i32 << 4);
        if (!push(kind, (unsigned)src_of(*y), arg, x->value, y->value, S.fill)) return -1;
        fast = true;
      }
    } else if (opcode == GM_OP_CLIP && src_of(S.b) > 0 && kind_of(S.a) != GM_OPD_NONE) {
      bool ok = true;
      if (kind_of(S.a) != GM_OPD_ACC) {
        if (src_of(S.a) >= 0) ok = push(UOP_LD, (unsigned)src_of(S.a), 0, zero, S.a.value, zero);
        else ok = false;
        if (!ok && n >= (unsigned)GM_MAX_UOPS) return -1;
      }
      if (ok) {
        const unsigned arg = (has_ndv(S.b) << 1) | (is_bool(S.b) << 2);
        if (!push(UOP_CLIP, (unsigned)src_of(S.b), arg, zero, S.b.value, S.fill)) return -1;
        fast = true;
      }
    }
    if (fast) {
      if (save && !push(UOP_ST, save - 1, 0, zero, zero, zero)) return -1;
    } else {
      if (!push(UOP_GENERIC, 0, s, zero, zero, zero)) return -1;  // its save_temp is handled by step()
    }
  }
  args->n_uops = n;
  return 0;
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
  if (lower(prog, &args) != 0) return gm::fail(GM_ERR_INVALID, "program does not fit %d micro-ops", GM_MAX_UOPS);
  {
    const int native_code = prog->reg_class == GM_CLASS_F32 ? GM_F32 : (prog->reg_class == GM_CLASS_I32 ? GM_I32 : -1);
    bool native = native_code >= 0 && (int)prog->out_dtype == native_code;
    for (unsigned k = 0; k < prog->n_inputs && native; ++k)
      native = ((int)(prog->inputs[k].bits & 0xffu) == native_code) && !((prog->inputs[k].bits >> 8) & 3u);
    args.native = native ? 1u : 0u;
  }
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
