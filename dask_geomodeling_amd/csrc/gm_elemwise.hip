// K1 host side: validates a gm_program and launches the VM kernel of its register
// class (kernels: gm_vm.h, instantiated per class in gm_vm_<class>.hip).
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include <chrono>

#include "gm_spec.h"
#include "gm_vm.h"

using namespace gmvm;

namespace {

// (measurement knob: GM_VM_VEC=4|8 presets the elements per lane, like gm_vm_set_tuning)
gm_vm_tuning initial_tuning() {
  gm_vm_tuning t = {0, 0, {0, 0}};
  if (const char* e = getenv("GM_VM_VEC")) {
    const int v = atoi(e);
    if (v == 4 || v == 8) t.vec = v;
  }
  return t;
}
gm_vm_tuning g_tuning = initial_tuning();

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
int lower(const gm_program* prog, VmCode* code, unsigned* n_uops) {
  unsigned n = 0;
  auto push = [&](unsigned kind, unsigned src, unsigned arg, gm_scalar acc_ndv, gm_scalar b_val, gm_scalar fill) -> bool {
    if (n >= (unsigned)GM_MAX_UOPS) return false;
    gm_uop& u = code->uops[n++];
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
  // A comparison "tile CMP literal" whose only use is the boolean mask of one later Clip (the idiomatic
  // masking pattern, Clip(x, Greater(D, 20))) is folded into that Clip: fold_into[q] = s, folded[s] = true.
  int fold_into[GM_MAX_STEPS];
  bool folded[GM_MAX_STEPS];
  for (unsigned s = 0; s < prog->n_steps; ++s) { fold_into[s] = -1; folded[s] = false; }
  for (unsigned s = 0; s + 1 < prog->n_steps; ++s) {
    const gm_step& S = prog->steps[s];
    const unsigned opcode = S.bits & 0xffu, save = (S.bits >> 8) & 0xffu;
    auto kind = [](const gm_operand& o) { return o.bits & 0xffu; };
    auto index = [](const gm_operand& o) { return (o.bits >> 8) & 0xffu; };
    if (opcode < GM_OP_EQ || opcode > GM_OP_LE || !save) continue;
    if (kind(S.a) != GM_OPD_INPUT || kind(S.b) != GM_OPD_SCALAR) continue;
    const gm_step& N1 = prog->steps[s + 1];  // the comparison's value must not be consumed from the accumulator
    if (kind(N1.a) == GM_OPD_ACC || kind(N1.b) == GM_OPD_ACC || kind(N1.a) == GM_OPD_NONE) continue;
    int user = -1, uses = 0;
    bool resaved = false;
    for (unsigned q = s + 1; q < prog->n_steps; ++q) {
      const gm_step& Q = prog->steps[q];
      for (const gm_operand* o : {&Q.a, &Q.b})
        if (kind(*o) == GM_OPD_TEMP && index(*o) == save - 1 && !resaved) { ++uses; user = (int)q; }
      if (((Q.bits >> 8) & 0xffu) == save) resaved = true;  // the slot is reused from here on
    }
    if (uses != 1) continue;
    const gm_step& Q = prog->steps[user];
    const bool is_clip_mask = (Q.bits & 0xffu) == GM_OP_CLIP && kind(Q.b) == GM_OPD_TEMP && index(Q.b) == save - 1 &&
                              ((Q.b.bits >> 17) & 1u);
    if (!is_clip_mask || (kind(Q.a) == GM_OPD_TEMP && index(Q.a) == save - 1)) continue;
    // the tile must still be addressable at the Clip: inputs always are
    fold_into[user] = (int)s;
    folded[s] = true;
  }
  for (unsigned s = 0; s < prog->n_steps; ++s) {
    if (folded[s]) continue;
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
    const bool compare = opcode >= GM_OP_EQ && opcode <= GM_OP_LE;
    if (compare && kind_of(S.a) == GM_OPD_ACC && kind_of(S.b) != GM_OPD_ACC && src_of(S.b) >= 0) {
      // acc OP b with b a literal / input / temp (the finite check of the float fast body is a no-op on 0 / 1)
      const unsigned kind = UOP_EQ + (opcode - GM_OP_EQ);
      const unsigned x_has = has_ndv(S.a), y_has = kind_of(S.b) == GM_OPD_SCALAR ? 0u : has_ndv(S.b);
      if (!push(kind, (unsigned)src_of(S.b), x_has | (y_has << 1), S.a.value, S.b.value, S.fill)) return -1;
      fast = true;
    } else if (compare && kind_of(S.a) != GM_OPD_ACC && kind_of(S.a) != GM_OPD_NONE && src_of(S.a) >= 0 &&
               kind_of(S.b) != GM_OPD_ACC && src_of(S.b) >= 0 && kind_of(S.a) != GM_OPD_SCALAR) {
      // a OP b with neither in the accumulator: load a first
      const unsigned kind = UOP_EQ + (opcode - GM_OP_EQ);
      if (!push(UOP_LD, (unsigned)src_of(S.a), 0, zero, S.a.value, zero)) return -1;
      const unsigned x_has = has_ndv(S.a), y_has = kind_of(S.b) == GM_OPD_SCALAR ? 0u : has_ndv(S.b);
      if (!push(kind, (unsigned)src_of(S.b), x_has | (y_has << 1), S.a.value, S.b.value, S.fill)) return -1;
      fast = true;
    } else if ((opcode == GM_OP_DIGITIZE || opcode == GM_OP_LOOKUP || (opcode == GM_OP_MASK && (((S.bits >> 16) & 4u) != 0))) &&
               (kind_of(S.a) == GM_OPD_ACC || (src_of(S.a) > 0))) {
      // unary ops whose parameters fit one micro-op; an operand that is not the accumulator is loaded first
      if (kind_of(S.a) != GM_OPD_ACC && !push(UOP_LD, (unsigned)src_of(S.a), 0, zero, S.a.value, zero)) return -1;
      const unsigned flags = (S.bits >> 16) & 0xffu;
      bool ok;
      if (opcode == GM_OP_DIGITIZE) {
        const unsigned arg = has_ndv(S.a) | ((flags & 1u) << 1) | (((flags >> 1) & 1u) << 2) |
                             (((unsigned)S.aux1.u64 & 0xffu) << 8) | (((unsigned)S.aux0.u64 & 3u) << 16);
        ok = push(UOP_DIGITIZE, 0, arg, S.a.value, zero, S.fill);
      } else if (opcode == GM_OP_LOOKUP) {
        const unsigned arg = (flags & 1u) | (((unsigned)S.aux3.u64 & 0xfu) << 8) | (((unsigned)S.aux0.u64 & 3u) << 16);
        ok = push(UOP_LOOKUP, 0, arg, S.aux2, S.aux1, S.fill);
      } else {
        ok = push(UOP_MASKI, 0, has_ndv(S.a), S.a.value, S.aux0, S.fill);
      }
      if (!ok) return -1;
      fast = true;
    } else
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
        // acc / c with c a power of two is acc * (1 / c) bit for bit (1 / c is exact and both are the
        // correctly rounded value of the same real number, subnormal results included): a multiply
        // instead of the ~14-instruction correctly rounded division
        gm_scalar y_value = y->value;
        if (kind == UOP_DIV && kind_of(*y) == GM_OPD_SCALAR) {
          if (prog->reg_class == GM_CLASS_F32) {
            const float c = y->value.f32, r = 1.0f / c;
            int e;
            if (std::isnormal(c) && std::isnormal(r) && frexpf(fabsf(c), &e) == 0.5f) {
              kind = UOP_MUL;
              y_value.u64 = 0;
              y_value.f32 = r;
            }
          } else if (prog->reg_class == GM_CLASS_F64) {
            const double c = y->value.f64, r = 1.0 / c;
            int e;
            if (std::isnormal(c) && std::isnormal(r) && frexp(fabs(c), &e) == 0.5) {
              kind = UOP_MUL;
              y_value.f64 = r;
            }
          }
        }
        const unsigned x_has = kind_of(*x) == GM_OPD_SCALAR ? 0u : has_ndv(*x);
        const unsigned y_has = kind_of(*y) == GM_OPD_SCALAR ? 0u : has_ndv(*y);
        const unsigned arg = x_has | (y_has << 1) | (round_f32 << 3) | (wrap_i32 << 4);
        if (!push(kind, (unsigned)src_of(*y), arg, x->value, y_value, S.fill)) return -1;
        fast = true;
      }
    } else if (opcode == GM_OP_CLIP && fold_into[s] >= 0 && kind_of(S.a) != GM_OPD_NONE &&
               (kind_of(S.a) == GM_OPD_ACC || src_of(S.a) >= 0)) {
      const gm_step& M = prog->steps[fold_into[s]];  // the comparison: tile CMP literal
      if (kind_of(S.a) != GM_OPD_ACC && !push(UOP_LD, (unsigned)src_of(S.a), 0, zero, S.a.value, zero)) return -1;
      const unsigned cmp = (M.bits & 0xffu) - GM_OP_EQ;
      // a no-data cell of the tile takes the comparison's fill (False unless stated otherwise): dropped iff 0
      const bool fill_is_false = prog->reg_class <= GM_CLASS_F64
                                     ? (prog->reg_class == GM_CLASS_F32 ? M.fill.f32 == 0.0f : M.fill.f64 == 0.0)
                                     : (prog->reg_class == GM_CLASS_I32 ? M.fill.i32 == 0 : M.fill.i64 == 0);
      const unsigned arg = has_ndv(M.a) | ((fill_is_false ? 1u : 0u) << 1) | (cmp << 8);
      if (!push(UOP_CLIPCMP, (unsigned)src_of(M.a), arg, M.a.value, M.b.value, S.fill)) return -1;
      fast = true;
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
  *n_uops = n;
  return 0;
}

// LD x; OP y  ->  OP y with x named in bits 16-23 of the route (arithmetic, comparisons, Clip): one
// micro-op fetch + decode less.  Only for the kernels that implement it (Machine::FOLD_LD: the small
// LIGHT ones -- the canonical chain's: +0.8 % in A/B runs on one box).
void fold_leading_loads(VmCode* code, unsigned* n_uops) {
  const unsigned n = *n_uops;
  unsigned m = 0;
  for (unsigned i = 0; i < n; ++i) {
    const gm_uop& U = code->uops[i];
    const unsigned kind = U.route & 0xffu, src = (U.route >> 8) & 0xffu;
    if (kind == UOP_LD && src != UOP_SRC_SCALAR && i + 1 < n) {
      gm_uop& N = code->uops[i + 1];
      const unsigned nk = N.route & 0xffu;
      if (nk >= UOP_ADD && nk <= UOP_LE && ((N.route >> 16) & 0xffu) == 0) {
        N.route |= src << 16;
        continue;  // drop the LD
      }
    }
    if (m != i) code->uops[m] = code->uops[i];
    ++m;
  }
  *n_uops = m;
}

// ---- lowered programs, device resident, cached by content per device ------------------------
struct CodeEntry {
  gm_program prog;     // the key (compared in full; table pointers zeroed, contents in `tables`)
  std::vector<unsigned long long> tables;  // words of all tables, in order (part of the key)
  std::vector<unsigned long long> staged;  // device image of the tables: stays alive for the upload
  VmCode host;         // staging copy: stays alive for the asynchronous upload
  VmCode* dev = nullptr;
  unsigned n_uops = 0;
  unsigned n_uops_lowered = 0;  // before fold_leading_loads(): what the elements-per-lane rule was tuned on
  int flavor = VM_LIGHT;
  unsigned lds_table = 0, lds_words = 0;  // 1 + index / window words of the LOOKUP table staged in LDS
  int size = 0;        // kernel size class S/M/L
  bool native = false;
  // programs of ONE 1-byte input tile: the 256-entry table of the program's results (device), built on
  // first use by running the program itself on the tile 0..255
  mutable void* lut = nullptr;
  mutable std::once_flag lut_once;
  mutable int lut_rc = GM_OK;
  // programs that are nothing but LOOKUP(int32 tile) -> narrow integer tile (configs[2]): the dedicated
  // kernel of gm_tables.hip
  bool lookup_only = false;
  gm::Lookup32 lookup;
  // specialised kernels of this program (gm_spec.hip), resolved on first hot use: [strided][G - 1]
  int n_temps = 0;
  std::vector<size_t> table_offset;     // word offset of table t in `staged`
  mutable std::atomic<unsigned long long> elements_seen{0};
  mutable std::mutex spec_mu;
  mutable gmspec::Kernel spec[2][2];
  mutable gmspec::Ticket spec_ticket[2][2];
  mutable int spec_state[2][2] = {{0, 0}, {0, 0}};  // 0 = not asked for yet, 1 = being compiled (worker thread), 2 = settled
};

std::mutex g_code_mu;
std::unordered_map<unsigned long long, std::vector<std::unique_ptr<CodeEntry>>> g_code;  // (device, hash) buckets

unsigned long long hash_program(const gm_program& p, const std::vector<unsigned long long>& tables, int device) {
  const unsigned char* b = (const unsigned char*)&p;
  unsigned long long h = 1469598103934665603ull ^ (unsigned long long)device;
  for (size_t i = 0; i < sizeof(gm_program); ++i) h = (h ^ b[i]) * 1099511628211ull;
  for (unsigned long long w : tables) h = (h ^ w) * 1099511628211ull;
  return h;
}

int validate(const gm_program* prog) {
  if (prog->n_steps == 0 || prog->n_steps > GM_MAX_STEPS) return gm::fail(GM_ERR_INVALID, "n_steps %u", prog->n_steps);
  if (prog->n_inputs > GM_MAX_INPUTS) return gm::fail(GM_ERR_INVALID, "n_inputs %u", prog->n_inputs);
  if (prog->reg_class > GM_CLASS_I64) return gm::fail(GM_ERR_INVALID, "reg_class %u", prog->reg_class);
  if (gm::dtype_size(prog->out_dtype) == 0) return gm::fail(GM_ERR_INVALID, "out_dtype %u", prog->out_dtype);
  for (unsigned k = 0; k < prog->n_inputs; ++k)
    if (gm::dtype_size(prog->inputs[k].bits & 0xffu) == 0) return gm::fail(GM_ERR_INVALID, "input %u dtype %u", k, prog->inputs[k].bits & 0xffu);
  long long lookup_table = -1;
  for (unsigned s = 0; s < prog->n_steps; ++s) {
    const gm_step& S = prog->steps[s];
    const unsigned opcode = S.bits & 0xffu, save_temp = (S.bits >> 8) & 0xffu;
    if (opcode == GM_OP_NOP || opcode >= GM_OP_COUNT) return gm::fail(GM_ERR_INVALID, "step %u opcode %u", s, opcode);
    if (save_temp > GM_MAX_TEMPS) return gm::fail(GM_ERR_INVALID, "step %u save_temp %u", s, save_temp);
    int rc = validate_operand(S.a, prog->n_inputs, "a", s);
    if (rc) return rc;
    rc = validate_operand(S.b, prog->n_inputs, "b", s);
    if (rc) return rc;
    if ((opcode == GM_OP_DIV || opcode == GM_OP_NANIFNODATA || opcode == GM_OP_FMAX || opcode == GM_OP_NANTOFILL) &&
        prog->reg_class >= GM_CLASS_I32)
      return gm::fail(GM_ERR_INVALID, "step %u: opcode %u needs a float class", s, opcode);
    if (opcode == GM_OP_DIGITIZE || opcode == GM_OP_LOOKUP) {
      const unsigned long long t = S.aux0.u64;
      if (t >= prog->n_tables) return gm::fail(GM_ERR_INVALID, "step %u: table %llu of %u", s, t, prog->n_tables);
      const unsigned long long words = prog->tables[t].n_words;
      if (opcode == GM_OP_DIGITIZE) {
        if (S.aux1.u64 > GM_MAX_LINEAR_EDGES) return gm::fail(GM_ERR_UNSUPPORTED, "step %u: %llu edges (max %d inside a program)", s, (unsigned long long)S.aux1.u64, GM_MAX_LINEAR_EDGES);
        if (words < S.aux1.u64) return gm::fail(GM_ERR_INVALID, "step %u: table shorter than its edges", s);
        if (prog->reg_class <= GM_CLASS_F64 && (((S.bits >> 16) & 2u) != 0)) return gm::fail(GM_ERR_INVALID, "step %u: int64 edges in a float class", s);
      } else {
        const unsigned long long esize = (prog->reg_class == GM_CLASS_I64 || prog->reg_class == GM_CLASS_F64) ? 8 : 4;
        if (S.aux2.u64 == 0 || words < (S.aux2.u64 * esize + 7) / 8 + 2 * S.aux3.u64)
          return gm::fail(GM_ERR_INVALID, "step %u: table shorter than its window + exceptions", s);
        if (S.aux2.u64 * esize > 65536) return gm::fail(GM_ERR_UNSUPPORTED, "step %u: LOOKUP window larger than 64 KiB of LDS", s);
        if (lookup_table >= 0 && lookup_table != (long long)t)
          return gm::fail(GM_ERR_UNSUPPORTED, "step %u: a program stages one LOOKUP table in LDS (split it)", s);
        lookup_table = (long long)t;
      }
    }
  }
  return GM_OK;
}

// validate + lower + upload once per (device, program); later launches only fill the kernarg
constexpr size_t LUT8_MIN_ELEMS = 512;  // (the table itself is built with a 256-element run of the VM, which stays below this)

// the tile 0, 1, ..., 255 on the current device (uploaded once per device)
std::mutex g_ramp_mu;
std::unordered_map<int, void*> g_ramps;
int byte_ramp(hipStream_t stream, void** out) {
  std::mutex& mu = g_ramp_mu;
  std::unordered_map<int, void*>& ramps = g_ramps;
  static unsigned char host[256];
  int device = 0;
  GM_HIP(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mu);
  auto it = ramps.find(device);
  if (it == ramps.end()) {
    for (int i = 0; i < 256; ++i) host[i] = (unsigned char)i;
    void* dev = nullptr;
    const int rc = gm_malloc(&dev, 256);
    if (rc) return rc;
    GM_HIP(hipMemcpyAsync(dev, host, 256, hipMemcpyHostToDevice, stream));
    it = ramps.emplace(device, dev).first;
  }
  *out = it->second;
  return GM_OK;
}

// canonical key of a program (only the used part is meaningful: equal programs hash equal whatever the
// caller left in the unused slots) + the words of its tables
int canonical_key(const gm_program* prog, gm_program* key_out, std::vector<unsigned long long>* words) {
  gm_program& key = *key_out;
  memset(&key, 0, sizeof(key));
  key.n_steps = prog->n_steps;
  key.n_inputs = prog->n_inputs;
  key.reg_class = prog->reg_class;
  key.out_dtype = prog->out_dtype;
  if (prog->n_steps > GM_MAX_STEPS || prog->n_inputs > GM_MAX_INPUTS) return validate(prog);
  if (prog->n_tables > GM_MAX_TABLES) return gm::fail(GM_ERR_INVALID, "n_tables %u", prog->n_tables);
  memcpy(key.inputs, prog->inputs, sizeof(gm_input_desc) * prog->n_inputs);
  memcpy(key.steps, prog->steps, sizeof(gm_step) * prog->n_steps);
  key.n_tables = prog->n_tables;
  for (unsigned t = 0; t < prog->n_tables; ++t) {
    if (!prog->tables[t].data && prog->tables[t].n_words) return gm::fail(GM_ERR_INVALID, "table %u is null", t);
    if (prog->tables[t].n_words > (1ull << 24)) return gm::fail(GM_ERR_UNSUPPORTED, "table %u too large", t);
    key.tables[t].n_words = prog->tables[t].n_words;  // pointers stay null in the key
    const unsigned long long* w = (const unsigned long long*)prog->tables[t].data;
    words->insert(words->end(), w, w + prog->tables[t].n_words);
  }
  return GM_OK;
}

// host half of a CodeEntry: validate, lower, classify, stage the tables (no device needed)
int build_entry(const gm_program& key, const std::vector<unsigned long long>& words, CodeEntry* e) {
  int rc = validate(&key);
  if (rc) return rc;
  e->prog = key;
  e->tables = words;
  memset(&e->host, 0, sizeof(VmCode));
  memcpy(e->host.inputs, key.inputs, sizeof(key.inputs));
  memcpy(e->host.steps, key.steps, sizeof(key.steps));
  if (lower(&key, &e->host, &e->n_uops) != 0) return gm::fail(GM_ERR_INVALID, "program does not fit %d micro-ops", GM_MAX_UOPS);
  const int native_code = key.reg_class == GM_CLASS_F32 ? GM_F32 : (key.reg_class == GM_CLASS_I32 ? GM_I32 : -1);
  bool native = native_code >= 0 && (int)key.out_dtype == native_code;
  for (unsigned k = 0; k < key.n_inputs && native; ++k)
    native = ((int)(key.inputs[k].bits & 0xffu) == native_code) && !((key.inputs[k].bits >> 8) & 3u);
  e->native = native;
  unsigned max_tmp = 0;
  bool has_math = false, has_tables = false;
  for (unsigned s = 0; s < key.n_steps; ++s) {
    const gm_step& S = key.steps[s];
    const unsigned op = S.bits & 0xffu, save_temp = (S.bits >> 8) & 0xffu;
    if (op == GM_OP_POW || op == GM_OP_EXP || op == GM_OP_LOG || op == GM_OP_LOG10) has_math = true;
    if (op == GM_OP_DIGITIZE || op == GM_OP_LOOKUP) has_tables = true;
    if (save_temp > max_tmp) max_tmp = save_temp;
    for (const gm_operand* o : {&S.a, &S.b})
      if ((o->bits & 0xffu) == GM_OPD_TEMP && ((o->bits >> 8) & 0xffu) + 1 > max_tmp) max_tmp = ((o->bits >> 8) & 0xffu) + 1;
  }
  e->n_temps = (int)max_tmp;
  // (size 3, round 6: 9-16 inputs in one launch -- an element-wise sum of 16 resident sources went through intermediate tiles
  //  before: 384 us per 4096^2 request against 97 for its bytes at the copy ceiling)
  e->size = (key.n_inputs <= 2 && max_tmp <= 1) ? 0 : ((key.n_inputs <= 4 && max_tmp <= 2) ? 1 : (key.n_inputs <= 8 ? 2 : 3));
  if (has_math && has_tables)
    return gm::fail(GM_ERR_UNSUPPORTED, "pow/exp/log and table lookups in one program (split it)");
  e->flavor = has_math ? VM_MATH : (has_tables ? VM_TABLES : VM_LIGHT);
  e->n_uops_lowered = e->n_uops;
  if (e->flavor == VM_LIGHT && e->size == 0) fold_leading_loads(&e->host, &e->n_uops);
  for (unsigned s = 0; s < key.n_steps && !e->lds_table; ++s) {
    const gm_step& S = key.steps[s];
    const unsigned long long esize = (key.reg_class == GM_CLASS_I64 || key.reg_class == GM_CLASS_F64) ? 8 : 4;
    if ((S.bits & 0xffu) == GM_OP_LOOKUP) {  // (validate() checked that it fits 64 KiB of LDS)
      e->lds_table = (unsigned)S.aux0.u64 + 1;
      e->lds_words = (unsigned)((S.aux2.u64 * esize + 7) / 8);
    }
  }
  // device image of the tables: the words as given, except the edges of a DIGITIZE in the F32 class,
  // which become float32 thresholds making the same decisions as NumPy's float64 compares:
  //   edge <= x  <=>  ceil32(edge) <= x   (smallest float32 >= edge)
  //   edge <  x  <=>  floor32(edge) < x   (largest float32 <= edge)
  e->staged = words;
  std::vector<size_t>& offset = e->table_offset;
  offset.assign(key.n_tables + 1, 0);
  for (unsigned t = 0; t < key.n_tables; ++t) offset[t + 1] = offset[t] + key.tables[t].n_words;
  if (key.reg_class == GM_CLASS_F32) {
    for (unsigned s = 0; s < key.n_steps; ++s) {
      const gm_step& S = key.steps[s];
      if ((S.bits & 0xffu) != GM_OP_DIGITIZE) continue;
      const bool right = ((S.bits >> 16) & 1u) != 0;
      const size_t at = offset[S.aux0.u64];
      float* thr = (float*)&e->staged[at];
      for (unsigned long long i = 0; i < S.aux1.u64; ++i) {
        double edge;
        memcpy(&edge, &words[at + i], 8);
        float f = (float)edge;
        if (!right && (double)f < edge) f = nextafterf(f, INFINITY);
        if (right && (double)f > edge) f = nextafterf(f, -INFINITY);
        thr[i] = f;  // packs downwards: thr[i] never overtakes the words still to be read
      }
    }
  }
  return GM_OK;
}

gmspec::Request spec_request(const CodeEntry& e, bool strided, int G) {
  gmspec::Request r;
  r.host = &e.host;
  r.n_uops = e.n_uops;
  r.n_inputs = e.prog.n_inputs;
  r.reg_class = e.prog.reg_class;
  r.out_dtype = e.prog.out_dtype;
  r.flavor = e.flavor;
  r.n_temps = e.n_temps;
  r.native = e.native;
  r.strided = strided;
  r.G = G;
  r.tables = &e.staged;
  r.table_offset = e.table_offset;
  return r;
}

int get_code(const gm_program* prog, hipStream_t stream, const CodeEntry** out) {
  int device = 0;
  GM_HIP(hipGetDevice(&device));
  gm_program key;
  std::vector<unsigned long long> words;
  int rc = canonical_key(prog, &key, &words);
  if (rc) return rc;
  const unsigned long long h = hash_program(key, words, device);
  std::lock_guard<std::mutex> lk(g_code_mu);
  auto& bucket = g_code[h];
  for (auto& e : bucket)
    if (memcmp(&e->prog, &key, sizeof(key)) == 0 && e->tables == words) {
      *out = e.get();
      return GM_OK;
    }
  std::unique_ptr<CodeEntry> e(new CodeEntry);
  rc = build_entry(key, words, e.get());
  if (rc) return rc;
  const std::vector<size_t>& offset = e->table_offset;
  void* dev = nullptr;
  rc = gm_malloc(&dev, sizeof(VmCode) + e->staged.size() * 8 + 64);  // (+: DIGITIZE reads its edges four at a time)
  if (rc) return rc;
  e->dev = (VmCode*)dev;
  const unsigned long long tables_dev = (unsigned long long)(uintptr_t)dev + ((sizeof(VmCode) + 15) / 16) * 16;
  for (unsigned t = 0; t < key.n_tables; ++t) e->host.tables[t] = tables_dev + 8ull * offset[t];
  {
    const unsigned in_bits = key.inputs[0].bits;
    const unsigned od = key.out_dtype;
    const bool narrow_int_out = od == GM_U8 || od == GM_I8;  // (the kernel packs 1-byte results)
    if (key.reg_class == GM_CLASS_I32 && key.n_inputs == 1 && (in_bits & 0xffu) == GM_I32 && !((in_bits >> 8) & 1u) &&
        narrow_int_out && e->n_uops == 2 && (e->host.uops[0].route & 0xffu) == UOP_LD &&
        ((e->host.uops[0].route >> 8) & 0xffu) == UOP_SRC_INPUT && (e->host.uops[1].route & 0xffu) == UOP_LOOKUP) {
      const gm_uop& U = e->host.uops[1];
      const unsigned t = (U.arg >> 16) & 3u, n_exc = (U.arg >> 8) & 0xfu;
      const unsigned long long range = U.acc_ndv.u64;
      if (t < key.n_tables && n_exc <= 8 && range >= 1 && range * 4ull <= 61440ull) {
        gm::Lookup32& K = e->lookup;
        memset(&K, 0, sizeof(K));
        K.window = (const int*)(uintptr_t)e->host.tables[t];
        K.range = (unsigned)range;
        K.kmin = (int)U.b_val.i64;
        K.n_exc = (int)n_exc;
        K.select = (int)(U.arg & 1u);
        K.fill = U.fill.i32;
        const size_t exc_at = offset[t] + (size_t)((range * 4ull + 7ull) / 8ull);
        bool ok = exc_at + 2ull * n_exc <= e->staged.size();
        for (unsigned x = 0; x < n_exc && ok; ++x) {
          K.exc_key[x] = (int)(long long)e->staged[exc_at + 2 * x];
          K.exc_val[x] = (int)(unsigned)e->staged[exc_at + 2 * x + 1];
        }
        e->lookup_only = ok;
      }
    }
  }
  GM_HIP(hipMemcpyAsync(dev, &e->host, sizeof(VmCode), hipMemcpyHostToDevice, stream));
  if (!e->staged.empty())
    GM_HIP(hipMemcpyAsync((void*)(uintptr_t)tables_dev, e->staged.data(), e->staged.size() * 8, hipMemcpyHostToDevice, stream));
  *out = e.get();
  bucket.push_back(std::move(e));
  return GM_OK;
}

}  // namespace

extern "C" {

int gm_vm_set_tuning(const gm_vm_tuning* t) {
  if (!t) return gm::fail(GM_ERR_INVALID, "null tuning");
  gm_vm_tuning v = *t;
  if (v.vec != 0 && v.vec != 4 && v.vec != 8) v.vec = 0;
  if (v.max_batch < 0 || v.max_batch > GM_MAX_BATCH) v.max_batch = 0;
  g_tuning = v;
  return GM_OK;
}

int gm_program_cache_clear(void) {
  // (no launch may be in flight on another thread: entries are handed out as raw pointers)
  std::lock_guard<std::mutex> lk(g_code_mu);
  size_t n = 0;
  for (auto& bucket : g_code)
    for (auto& e : bucket.second) {
      if (e->dev) (void)gm_free(e->dev);
      if (e->lut) (void)gm_free(e->lut);
      ++n;
    }
  g_code.clear();
  (void)n;
  std::lock_guard<std::mutex> ramp_lock(g_ramp_mu);
  for (auto& kv : g_ramps) (void)gm_free(kv.second);
  g_ramps.clear();
  return GM_OK;
}

int gm_vm_get_tuning(gm_vm_tuning* t) {
  if (!t) return gm::fail(GM_ERR_INVALID, "null tuning");
  *t = g_tuning;
  return GM_OK;
}

int gm_elemwise_run_batch_2d(const gm_program* prog, int n_tiles, const void* const* inputs, const uint32_t* pitches,
                             void* const* outs, const uint32_t* out_pitches, size_t rows, size_t cols);

int gm_elemwise_run_batch(const gm_program* prog, int n_tiles, const void* const* inputs, void* const* outs, size_t n) {
  return gm_elemwise_run_batch_2d(prog, n_tiles, inputs, nullptr, outs, nullptr, 1, n);
}

// the longest time any caller has spent inside gm_elemwise_run* (what a thread that had to wait for a compilation,
// a lock or an upload would show): read and reset by gm_elemwise_call_stats
static std::atomic<unsigned long long> g_calls{0}, g_longest_ns{0};
static int run_batch_2d(const gm_program* prog, int n_tiles, const void* const* inputs, const uint32_t* pitches,
                        void* const* outs, const uint32_t* out_pitches, size_t rows, size_t cols);

int gm_elemwise_run_batch_2d(const gm_program* prog, int n_tiles, const void* const* inputs, const uint32_t* pitches,
                             void* const* outs, const uint32_t* out_pitches, size_t rows, size_t cols) {
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = run_batch_2d(prog, n_tiles, inputs, pitches, outs, out_pitches, rows, cols);
  const unsigned long long ns =
      (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  g_calls++;
  unsigned long long seen = g_longest_ns.load();
  while (ns > seen && !g_longest_ns.compare_exchange_weak(seen, ns)) {
  }
  return rc;
}

int gm_elemwise_call_stats(uint64_t* calls, double* longest_ms, int reset) {
  if (calls) *calls = g_calls.load();
  if (longest_ms) *longest_ms = (double)g_longest_ns.load() * 1e-6;
  if (reset) {
    g_calls.store(0);
    g_longest_ns.store(0);
  }
  return GM_OK;
}

static int run_batch_2d(const gm_program* prog, int n_tiles, const void* const* inputs, const uint32_t* pitches,
                        void* const* outs, const uint32_t* out_pitches, size_t rows, size_t cols) {
  const size_t n = rows * cols;
  if (!prog || !outs || (!inputs && prog->n_inputs)) return gm::fail(GM_ERR_INVALID, "null argument");
  if ((pitches || out_pitches) && cols > 0xffffffffull) return gm::fail(GM_ERR_UNSUPPORTED, "strided tiles wider than 2^32 elements");
  if (n_tiles < 0) return gm::fail(GM_ERR_INVALID, "n_tiles %d", n_tiles);
  hipStream_t stream = gm::current_stream();
  if (!stream) return GM_ERR_NO_DEVICE;
  const CodeEntry* code = nullptr;
  int rc = get_code(prog, stream, &code);
  if (rc) return rc;
  const unsigned n_in = code->prog.n_inputs;
  for (int t = 0; t < n_tiles; ++t) {
    if (!outs[t]) return gm::fail(GM_ERR_INVALID, "output of tile %d is null", t);
    for (unsigned k = 0; k < n_in; ++k)
      if (!inputs[(size_t)t * n_in + k]) return gm::fail(GM_ERR_INVALID, "input %u of tile %d is null", k, t);
  }
  if (n == 0 || n_tiles == 0) return GM_OK;

  // A program whose only input is a 1-byte tile (class rasters: Reclassify / Mask / Classify / arithmetic
  // on uint8) is a function of 256 values: tabulate it once with the VM itself, then apply the table
  // with the byte-LUT kernel (gm_tables.hip; 16 pixels per lane, table in LDS).  Bit-exact by
  // construction.  Small launches keep the VM; tiles that do not sit on 16 bytes take the kernel's row form.
  if (n_in == 1 && n >= LUT8_MIN_ELEMS && gm::dtype_size(code->prog.inputs[0].bits & 0xffu) == 1 &&
      !((code->prog.inputs[0].bits >> 8) & 1u) && n_tiles <= GM_MAX_BATCH * 64) {
    const bool addressable = rows < 0xffffffffull && cols < 0xffffffffull;  // (any alignment, any pitch: lut8_rows_kernel)
    if (addressable) {
      const int out_size = gm::dtype_size(code->prog.out_dtype);
      std::call_once(code->lut_once, [&]() {
        void* ramp = nullptr;
        code->lut_rc = byte_ramp(stream, &ramp);
        if (code->lut_rc) return;
        void* table = nullptr;
        code->lut_rc = gm_malloc(&table, 256 * (size_t)out_size);
        if (code->lut_rc) return;
        const void* in0 = ramp;
        // (256 elements: below LUT8_MIN_ELEMS, i.e. this runs on the VM)
        code->lut_rc = gm_elemwise_run_batch_2d(prog, 1, &in0, nullptr, &table, nullptr, 1, 256);
        if (!code->lut_rc) code->lut = table;
      });
      if (code->lut_rc) return code->lut_rc;
      for (int t0 = 0; t0 < n_tiles; t0 += 16) {
        const int nb = n_tiles - t0 < 16 ? n_tiles - t0 : 16;
        uint32_t ip[16], op[16];
        for (int t = 0; t < nb; ++t) {
          ip[t] = pitches && pitches[t0 + t] != cols ? pitches[t0 + t] : 0u;
          op[t] = out_pitches && out_pitches[t0 + t] != cols ? out_pitches[t0 + t] : 0u;
        }
        rc = gm::lut8_launch(nb, inputs + t0, ip, outs + t0, op, rows, cols, code->lut, out_size, stream);
        if (rc) return rc;
      }
      return GM_OK;
    }
  }

  // The whole program is LOOKUP(int32 tile) -> 1-byte tile: dedicated kernel (gm_tables.hip)
  if (code->lookup_only && n >= LUT8_MIN_ELEMS) {
    bool addressable = true;
    for (int t = 0; t < n_tiles && addressable; ++t) {
      const uint32_t ip = pitches ? pitches[t] : 0u, op = out_pitches ? out_pitches[t] : 0u;
      const bool strided = (ip && ip != cols) || (op && op != cols);
      if (((uintptr_t)inputs[t] % 16) || ((uintptr_t)outs[t] % 16)) addressable = false;
      if (strided && ((cols % 16) || (ip % 4) || (op % 16) || n > 0xffffffffull)) addressable = false;
    }
    if (addressable) {
      for (int t0 = 0; t0 < n_tiles; t0 += 16) {
        const int nb = n_tiles - t0 < 16 ? n_tiles - t0 : 16;
        uint32_t ip[16], op[16];
        for (int t = 0; t < nb; ++t) {
          ip[t] = pitches && pitches[t0 + t] != cols ? pitches[t0 + t] : 0u;
          op[t] = out_pitches && out_pitches[t0 + t] != cols ? out_pitches[t0 + t] : 0u;
        }
        rc = gm::lookup32_launch(nb, inputs + t0, ip, outs + t0, op, rows, cols, code->lookup,
                                 gm::dtype_size(code->prog.out_dtype), stream);
        if (rc) return rc;
      }
      return GM_OK;
    }
  }

  // elements per lane: 4 (G = 1) streams best when a lane moves >= 8 bytes per element; programs of
  // more than 3 micro-ops amortise their decode over 8 elements, and so do narrow streams (e.g.
  // int32 -> uint8 Mask, 5 B/px: 48 % of peak with 4 elements per lane, 66 % with 8)
  unsigned bytes_per_elem = (unsigned)gm::dtype_size(code->prog.out_dtype);
  for (unsigned k = 0; k < n_in; ++k)
    if (!((code->prog.inputs[k].bits >> 8) & 1u)) bytes_per_elem += (unsigned)gm::dtype_size(code->prog.inputs[k].bits & 0xffu);
  const bool wide_class = code->prog.reg_class == GM_CLASS_F64 || code->prog.reg_class == GM_CLASS_I64;
  // (the 64-bit classes pay 8 elements per lane with their occupancy: 132 VGPRs instead of 70)
  int G = g_tuning.vec == 8 ? 2 : (g_tuning.vec == 4 ? 1 : ((code->n_uops_lowered > 3 || (bytes_per_elem < 8 && !wide_class)) ? 2 : 1));
  if (code->flavor == VM_MATH || code->size >= 2) G = 1;
  // table ops cost the integer and 64-bit classes too many registers at 8 elements per lane
  // (102-136 VGPRs); the float32 class stays at 84
  if (code->flavor == VM_TABLES && code->prog.reg_class != GM_CLASS_F32 && g_tuning.vec == 0) G = 1;
  const int max_batch = g_tuning.max_batch > 0 ? g_tuning.max_batch : GM_MAX_BATCH;
  const int cus = gm::current_cu_count();
  // hot programs run as kernels of their own (gm_spec.hip); everything else -- and the scalar tails --
  // on the interpreter kernels
  const int spec_mode = gmspec::mode();
  bool want_spec = false;
  if (spec_mode == 2) {
    want_spec = true;
  } else if (spec_mode == 1) {
    const unsigned long long seen = code->elements_seen.fetch_add((unsigned long long)n * (unsigned long long)n_tiles) +
                                    (unsigned long long)n * (unsigned long long)n_tiles;
    want_spec = seen >= gmspec::hot_elements();
  }
  // a kernel of its own has no decode to amortise and few registers: the 4-byte classes always take 8
  // elements per lane there (at 4 waves per SIMD, gm_spec.hip), the 8-byte classes keep 4.  The interpreter
  // keeps the launch shape it was tuned for: G changes only for launches that do run a specialised kernel.
  const int G_interp = G;
  int G_spec = G;
  if (want_spec && g_tuning.vec == 0 && code->flavor != VM_MATH) G_spec = wide_class ? 1 : 2;
  // hot programs (mode 1) are compiled on the library's worker thread while the interpreter keeps serving them;
  // mode 2 (tests) and GM_SPEC_SYNC=1 compile on the calling thread
  static const bool spec_sync = getenv("GM_SPEC_SYNC") && atoi(getenv("GM_SPEC_SYNC")) != 0;
  const bool spec_async = spec_mode == 1 && !spec_sync;

  for (int t0 = 0; t0 < n_tiles; t0 += max_batch) {
    const int nb = n_tiles - t0 < max_batch ? n_tiles - t0 : max_batch;
    VmLaunch args;
    memset(&args, 0, sizeof(args));
    args.code = (unsigned long long)(uintptr_t)code->dev;
    args.n = n;
    args.n_uops = code->n_uops;
    args.n_inputs = n_in;
    args.out_dtype = code->prog.out_dtype;
    bool aligned = true, strided = false;
    static const bool unaligned_loads = !(getenv("GM_VM_UNALIGNED") && atoi(getenv("GM_VM_UNALIGNED")) == 0);
    for (int t = 0; t < nb; ++t) {
      args.out[t] = outs[t0 + t];
      if (((uintptr_t)args.out[t] % 16) != 0) aligned = false;
      const uint32_t opitch = out_pitches ? out_pitches[t0 + t] : 0u;
      if (opitch && opitch != cols) {
        if (opitch < cols) return gm::fail(GM_ERR_INVALID, "output pitch %u is smaller than the tile width %zu", opitch, cols);
        args.out_pitch[t] = opitch;
        strided = true;
        if (((size_t)opitch * (size_t)gm::dtype_size(code->prog.out_dtype)) % 16 != 0) aligned = false;
      }
      for (unsigned k = 0; k < n_in; ++k) {
        const void* p = inputs[(size_t)(t0 + t) * n_in + k];
        args.in[t][k] = p;
        const bool broadcast = ((code->prog.inputs[k].bits >> 8) & 1u) != 0;
        // Tile LOADS need not be 16-byte aligned: a group of 4 elements is one global_load of 4 x itemsize bytes, which
        // the memory pipeline takes at any element-aligned address (a window that starts at an odd column of a resident
        // raster -- every request a stencil Block grows by its radius -- used to send the whole tile down the
        // one-element-per-lane path: 2.7x the time of the vector kernel; round 4).  Stores stay aligned: the caller
        // allocates the result.  GM_VM_UNALIGNED=0 restores the old rule.
        const size_t item = (size_t)gm::dtype_size(code->prog.inputs[k].bits & 0xffu);
        const size_t need = unaligned_loads ? item : 16;
        if (!broadcast && ((uintptr_t)p % need) != 0) aligned = false;
        const uint32_t pitch = pitches ? pitches[(size_t)(t0 + t) * n_in + k] : 0u;
        if (pitch && pitch != cols && !broadcast) {
          if (pitch < cols) return gm::fail(GM_ERR_INVALID, "pitch %u of input %u is smaller than the tile width %zu", pitch, k, cols);
          args.pitch[t][k] = pitch;
          strided = true;
          if (!unaligned_loads && ((size_t)pitch * item) % 16 != 0) aligned = false;
        }
      }
    }
    args.flags = code->native ? VM_FLAG_NATIVE : 0u;
    if (strided) {
      args.flags |= VM_FLAG_STRIDED;
      args.cols = (unsigned)cols;
      {
        const unsigned long long d = cols / 4 ? cols / 4 : 1;
        unsigned l = 0;
        while ((1ull << l) < d) ++l;  // ceil(log2 d)
        args.div_magic = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1ull);
        args.div_sh1 = l < 1 ? l : 1;
        args.div_sh2 = l > 1 ? l - 1 : 0;
      }
      if (cols % 4 != 0 || n / 4 > 0xffffffffull) aligned = false;  // groups must not straddle rows
    }
    args.n_groups = aligned ? n / 4 : 0;
    args.tail_start = args.n_groups * 4;
    gmspec::Kernel spec;
    G = G_interp;
    if (want_spec && args.n_groups) {
      std::lock_guard<std::mutex> lock(code->spec_mu);
      const int si = strided ? 1 : 0, gi = G_spec - 1;
      int& state = code->spec_state[si][gi];
      bool pending = false;
      if (state == 0) {
        rc = gmspec::get_kernel(spec_request(*code, strided, G_spec), &code->spec[si][gi], spec_async,
                                &code->spec_ticket[si][gi], &pending);
        if (rc) return rc;
        state = pending ? 1 : 2;
      } else if (state == 1) {
        rc = gmspec::poll_kernel(code->spec_ticket[si][gi], &code->spec[si][gi], &pending);
        if (rc) return rc;
        if (!pending) state = 2;
      }
      spec = code->spec[si][gi];
      if (spec.function) G = G_spec;
    }
    const bool stage_lds = code->flavor == VM_TABLES && code->lds_table && (!spec.function || spec.stages_lds);
    const unsigned long long per_block = 256ull * G;
    unsigned long long blocks = (args.n_groups + per_block - 1) / per_block;
    args.spans = 1;
    if (stage_lds) {
      // a workgroup that stages W bytes of table streams >= 10 W bytes of tiles
      args.lds_table = code->lds_table;
      args.lds_words = code->lds_words;
      const unsigned long long span_bytes = per_block * 4ull * (bytes_per_elem ? bytes_per_elem : 1);
      unsigned long long spans = ((unsigned long long)code->lds_words * 8ull * 10ull + span_bytes - 1) / span_bytes;
      // ... and at least 8 spans while that leaves >= 4 workgroups per CU: a wave that handles a
      // single span lives mostly in its prologue (a chain of dependent scalar loads: kernarg -> code
      // -> micro-op -> table), measured +8 % on configs[2] (8192^2 int32 lookup), +5 % on configs[4]
      const unsigned long long fill_blocks = 4ull * (unsigned long long)(cus > 0 ? cus : 256);
      unsigned long long amortise = blocks / fill_blocks;
      if (amortise > 8) amortise = 8;
      if (spans < amortise) spans = amortise;
      if (spans > 64) spans = 64;
      args.spans = (unsigned)spans;
      blocks = (blocks + spans - 1) / spans;
    }
    if (blocks > 0x7fffffffull) return gm::fail(GM_ERR_UNSUPPORTED, "tile of %zu elements is too large for one launch", n);
    const unsigned long long tail = n - args.tail_start;
    unsigned long long tail_blocks = (tail + 255) / 256;
    const unsigned long long tail_cap = (unsigned long long)(cus > 0 ? cus : 256) * 8;
    if (tail_blocks > tail_cap) tail_blocks = tail_cap;
    dim3 grid((unsigned)blocks, (unsigned)nb), tail_grid((unsigned)tail_blocks, (unsigned)nb);
    if (spec.function && grid.x) {
      rc = gmspec::launch(spec, args, grid.x, grid.y, stage_lds ? (size_t)args.lds_words * 8 : 0, stream);
      if (rc) return rc;
      grid.x = 0;  // the interpreter only runs the scalar tail (if any)
      if (!tail_grid.x) continue;
      if (!stage_lds && code->lds_table) {  // the tail kernel stages its LOOKUP window itself
        args.lds_table = code->lds_table;
        args.lds_words = code->lds_words;
      }
    }
    switch (code->prog.reg_class) {
      case GM_CLASS_F32: rc = launch_class<float>(args, code->size, G, code->flavor, grid, tail_grid, stream); break;
      case GM_CLASS_F64: rc = launch_class<double>(args, code->size, G, code->flavor, grid, tail_grid, stream); break;
      case GM_CLASS_I32: rc = launch_class<int>(args, code->size, G, code->flavor, grid, tail_grid, stream); break;
      default: rc = launch_class<long long>(args, code->size, G, code->flavor, grid, tail_grid, stream); break;
    }
    if (rc) return rc;
  }
  return GM_OK;
}

/* The HIP source the specialiser generates for `prog` (no device needed): copies at most len - 1 bytes into
 * buf, *needed = strlen + 1.  group = 1 or 2 (4 or 8 elements per lane). */
int gm_spec_source(const gm_program* prog, int strided, int group, char* buf, size_t len, size_t* needed) {
  if (!prog) return gm::fail(GM_ERR_INVALID, "null program");
  gm_program key;
  std::vector<unsigned long long> words;
  int rc = canonical_key(prog, &key, &words);
  if (rc) return rc;
  CodeEntry e;
  rc = build_entry(key, words, &e);
  if (rc) return rc;
  const std::string src = gmspec::generate(spec_request(e, strided != 0, group == 2 ? 2 : 1), nullptr);
  if (needed) *needed = src.size() + 1;
  if (buf && len) {
    const size_t m = src.size() < len - 1 ? src.size() : len - 1;
    memcpy(buf, src.data(), m);
    buf[m] = 0;
  }
  return GM_OK;
}

/* Generate and compile (hiprtc, gfx950) the specialised kernel of `prog` without loading it: a build
 * check that needs no device.  *code_bytes = size of the code object. */
int gm_spec_compile_check(const gm_program* prog, int strided, int group, size_t* code_bytes) {
  if (!prog) return gm::fail(GM_ERR_INVALID, "null program");
  gm_program key;
  std::vector<unsigned long long> words;
  int rc = canonical_key(prog, &key, &words);
  if (rc) return rc;
  CodeEntry e;
  rc = build_entry(key, words, &e);
  if (rc) return rc;
  const std::string src = gmspec::generate(spec_request(e, strided != 0, group == 2 ? 2 : 1), nullptr);
  std::vector<char> code;
  std::string log;
  rc = gmspec::compile(src, &code, &log);
  if (rc) return rc;
  if (code_bytes) *code_bytes = code.size();
  return GM_OK;
}

int gm_elemwise_run(const gm_program* prog, const void* const* inputs, void* out, size_t n) {
  if (!out) return gm::fail(GM_ERR_INVALID, "null argument");
  void* outs[1] = {out};
  return gm_elemwise_run_batch(prog, 1, inputs, outs, n);
}

}  // extern "C"

// (gm_copy_2d / gm_copy_3d, the compaction of a window: csrc/gm_util.hip -- this file holds no device code)

// gm_elemwise_run_batch_2d over tiles of several bands whose bands are not back to back: input k of tile t is `bands` windows
// of rows x cols, band_pitches[t * n_inputs + k] elements apart (0: the same operand for every band -- a broadcast scalar);
// the result of tile t likewise with out_band_pitches[t].  An element-wise program treats every band alike, so the bands go
// as tiles of their own: bands x n_tiles windows in the launches of one call.
extern "C" int gm_elemwise_run_batch_bands(const gm_program* prog, int n_tiles, const void* const* inputs, const uint32_t* pitches,
                                           const uint64_t* band_pitches, void* const* outs, const uint32_t* out_pitches,
                                           const uint64_t* out_band_pitches, int bands, size_t rows, size_t cols) {
  if (!prog || !outs || !band_pitches || !out_band_pitches || (!inputs && prog->n_inputs)) return gm::fail(GM_ERR_INVALID, "null argument");
  if (n_tiles < 0 || bands < 0) return gm::fail(GM_ERR_INVALID, "gm_elemwise_run_batch_bands: %d tiles of %d bands", n_tiles, bands);
  const size_t n_in = prog->n_inputs, total = (size_t)n_tiles * (size_t)bands;
  if (total == 0) return GM_OK;
  if (total > 0x7fffffffull) return gm::fail(GM_ERR_UNSUPPORTED, "gm_elemwise_run_batch_bands: %zu windows", total);
  std::vector<const void*> in(total * n_in);
  std::vector<uint32_t> in_pitch(total * n_in);
  std::vector<void*> out(total);
  std::vector<uint32_t> out_pitch(total);
  const size_t out_item = gm::dtype_size(prog->out_dtype);
  for (int t = 0; t < n_tiles; ++t) {
    for (int b = 0; b < bands; ++b) {
      const size_t v = (size_t)t * bands + b;
      for (size_t k = 0; k < n_in; ++k) {
        const size_t at = (size_t)t * n_in + k;
        if (!inputs[at]) return gm::fail(GM_ERR_INVALID, "input %zu of tile %d is null", k, t);
        const size_t item = gm::dtype_size(prog->inputs[k].bits & 0xffu);
        in[v * n_in + k] = (const char*)inputs[at] + (size_t)b * band_pitches[at] * item;
        in_pitch[v * n_in + k] = pitches ? pitches[at] : 0u;
      }
      if (!outs[t]) return gm::fail(GM_ERR_INVALID, "output of tile %d is null", t);
      out[v] = (char*)outs[t] + (size_t)b * out_band_pitches[t] * out_item;
      out_pitch[v] = out_pitches ? out_pitches[t] : 0u;
    }
  }
  return gm_elemwise_run_batch_2d(prog, (int)total, in.data(), in_pitch.data(), out.data(), out_pitch.data(), rows, cols);
}

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
