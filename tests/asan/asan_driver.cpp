// Drives the host side of libgmhip under AddressSanitizer + UBSan (built by tests/asan/Makefile against the
// HIP test double): thousands of random VALID and INVALID programs through validation, lowering, the
// per-device program cache (> 512 distinct programs, > 256 of them with lookup tables), batched launches with
// strided / misaligned tiles, the specialiser's source generator, and the caching allocator from 8 threads.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "gmhip.h"

static std::mt19937_64 rng(12345);
static unsigned pick(unsigned n) { return (unsigned)(rng() % n); }

static gm_operand operand(unsigned kind, unsigned index, unsigned has_ndv, double value) {
  gm_operand o;
  memset(&o, 0, sizeof(o));
  o.bits = GM_OPERAND_BITS(kind, index, has_ndv, 0);
  o.value.f64 = value;
  return o;
}

// a random program: arithmetic chain with temps, optionally DIGITIZE + LOOKUP (tables), any class
static void random_program(gm_program* p, std::vector<uint64_t>* edges, std::vector<uint64_t>* table, bool valid) {
  memset(p, 0, sizeof(*p));
  p->reg_class = pick(4);
  const bool float_class = p->reg_class <= GM_CLASS_F64;
  p->n_inputs = 1 + pick(4);
  static const int dtypes[] = {GM_F32, GM_F64, GM_I32, GM_I64, GM_U8, GM_I16, GM_BOOL};
  for (unsigned k = 0; k < p->n_inputs; ++k) p->inputs[k].bits = GM_INPUT_BITS(dtypes[pick(7)], pick(8) == 0, 0);
  p->out_dtype = dtypes[pick(7)];
  const unsigned n = 1 + pick(GM_MAX_STEPS - 3);
  unsigned s = 0;
  for (; s < n; ++s) {
    gm_step& S = p->steps[s];
    static const int ops[] = {GM_OP_ADD, GM_OP_SUB, GM_OP_MUL, GM_OP_GT, GM_OP_LE, GM_OP_CLIP, GM_OP_MASK, GM_OP_COPY, GM_OP_ISDATA, GM_OP_DIV};
    int op = ops[pick(float_class ? 10 : 9)];
    const unsigned save = pick(4) == 0 ? 1 + pick(GM_MAX_TEMPS) : 0;
    S.bits = GM_STEP_BITS(op, save, op == GM_OP_MASK ? 4 : 0, 0, 0, float_class);
    S.a = s == 0 ? operand(GM_OPD_INPUT, pick(p->n_inputs), 1, 3.0) : operand(GM_OPD_ACC, 0, 1, 3.0);
    const unsigned bk = pick(3);
    S.b = bk == 0 ? operand(GM_OPD_SCALAR, 0, 0, 2.5) : (bk == 1 ? operand(GM_OPD_INPUT, pick(p->n_inputs), 1, 7.0)
                                                                  : operand(GM_OPD_TEMP, pick(GM_MAX_TEMPS), 0, 0.0));
    S.fill.f64 = 9.0;
  }
  if (pick(2) && s + 2 <= GM_MAX_STEPS) {  // table ops
    const unsigned nb = 1 + pick(12);
    edges->assign(nb + 4, 0);
    for (unsigned j = 0; j < nb; ++j) { double e = j * 1.5; memcpy(&(*edges)[j], &e, 8); }
    if (!float_class) for (unsigned j = 0; j < nb; ++j) (*edges)[j] = j * 2;
    gm_step& D = p->steps[s++];
    D.bits = GM_STEP_BITS(GM_OP_DIGITIZE, 0, float_class ? 0 : 2, 0, 0, 0);
    D.a = operand(GM_OPD_ACC, 0, 1, 3.0);
    D.aux0.u64 = 0;
    D.aux1.u64 = nb;
    D.fill.f64 = 255.0;
    const unsigned range = 1 + pick(300), n_exc = pick(4);
    const unsigned esize = (p->reg_class == GM_CLASS_F64 || p->reg_class == GM_CLASS_I64) ? 8 : 4;
    table->assign((range * esize + 7) / 8 + 2 * n_exc, 7);
    gm_step& L = p->steps[s++];
    L.bits = GM_STEP_BITS(GM_OP_LOOKUP, 0, pick(2), 0, 0, 0);
    L.a = operand(GM_OPD_ACC, 0, 0, 0.0);
    L.aux0.u64 = 1;
    L.aux1.i64 = -3;
    L.aux2.u64 = range;
    L.aux3.u64 = n_exc;
    p->n_tables = 2;
    p->tables[0].data = edges->data();
    p->tables[0].n_words = edges->size();
    p->tables[1].data = table->data();
    p->tables[1].n_words = table->size();
  }
  p->n_steps = s;
  if (!valid) {  // break one thing
    switch (pick(6)) {
      case 0: p->n_steps = GM_MAX_STEPS + 1 + pick(100); break;
      case 1: p->steps[pick(p->n_steps)].bits |= 0xffu; break;
      case 2: p->steps[pick(p->n_steps)].a = operand(GM_OPD_INPUT, 200, 0, 0); break;
      case 3: p->n_inputs = GM_MAX_INPUTS + 1; break;
      case 4: p->out_dtype = 99; break;
      default: p->n_tables = 9; break;
    }
  }
}

int main() {
  if (gm_init() != GM_OK) { printf("gm_init failed: %s\n", gm_last_error()); return 1; }
  int ok = 0, refused = 0, with_tables = 0;
  std::vector<char> src(1 << 20);
  for (int i = 0; i < 3000; ++i) {
    gm_program p;
    std::vector<uint64_t> edges, table;
    const bool valid = pick(5) != 0;
    random_program(&p, &edges, &table, valid);
    const size_t n = 1 + pick(5000);
    const int tiles = 1 + (int)pick(20);
    std::vector<void*> in((size_t)tiles * GM_MAX_INPUTS), out(tiles);
    std::vector<uint32_t> pitch((size_t)tiles * GM_MAX_INPUTS, 0), opitch(tiles, 0);
    for (auto& q : in) gm_malloc(&q, n * 8 + 64);
    for (auto& q : out) gm_malloc(&q, n * 8 + 64);
    const size_t cols = 1 + pick(64), rows = (n + cols - 1) / cols;
    if (pick(3) == 0) for (auto& w : pitch) w = (uint32_t)(cols + pick(3) * 4);
    std::vector<const void*> cin(in.begin(), in.end());
    if (pick(4) == 0) cin[0] = (const char*)cin[0] + 4;  // misaligned tile: scalar path
    const int rc = pick(2) ? gm_elemwise_run_batch(&p, tiles, cin.data(), out.data(), n)
                           : gm_elemwise_run_batch_2d(&p, tiles, cin.data(), pitch.data(), out.data(), opitch.data(), rows > 0 ? 1 : 1, n);
    if (rc == GM_OK) { ++ok; with_tables += p.n_tables ? 1 : 0; } else { ++refused; (void)gm_last_error(); }
    if (rc == GM_OK && i % 7 == 0) {
      size_t needed = 0;
      gm_spec_source(&p, pick(2), 1 + pick(2), src.data(), src.size(), &needed);
    }
    for (auto q : in) gm_free(q);
    for (auto q : out) gm_free(q);
    if (i % 500 == 499) gm_pool_trim();
  }
  // the allocator from several threads (the pool threads of a threaded dask scheduler)
  std::vector<std::thread> pool;
  for (int t = 0; t < 8; ++t)
    pool.emplace_back([t]() {
      std::mt19937 r(t);
      std::vector<void*> held;
      for (int i = 0; i < 20000; ++i) {
        if (held.size() < 64 && (r() & 1)) { void* q = nullptr; if (gm_malloc(&q, 1 + r() % (3 << 20)) == GM_OK) { memset(q, 1, 16); held.push_back(q); } }
        else if (!held.empty()) { gm_free(held.back()); held.pop_back(); }
        if (i % 5000 == 0) gm_pool_trim();
      }
      for (void* q : held) gm_free(q);
    });
  for (auto& th : pool) th.join();
  size_t in_use = 0, cached = 0;
  gm_program_cache_clear();
  gm_pool_stats(&in_use, &cached);
  gm_pool_trim();
  printf("ASAN_DRIVER ok programs=%d refused=%d with_tables=%d in_use_after=%zu\n", ok, refused, with_tables, in_use);
  return (ok > 512 && with_tables > 256 && refused > 100 && in_use == 0) ? 0 : 2;
}
