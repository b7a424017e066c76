// mpx_casadi.cpp -- the NLP oracles (nlp, nlp_f, nlp_g, nlp_grad_f, nlp_jac_g, nlp_hess_l, nlp_grad) with the calling convention of CasADi-generated C code, so
// that `ca.nlpsol("solver", "ipopt", "libmpx.so", opts)` can load this library in place of the SX
// graph that mpopt hands to nlpsol (mpopt.py:757).  SURVEY.md section 8(b), second surface.
//
// Convention (CasADi's public generated-code interface; casadi_int = long long, casadi_real = double):
//   int NAME(const double** arg, double** res, long long* iw, double* w, int mem);   0 = ok
//   NAME_n_in / _n_out / _name_in / _name_out / _sparsity_in / _sparsity_out / _work / _incref / _decref
//   and the optional _alloc_mem / _init_mem / _free_mem / _checkout / _release / _default_in (one stateless memory object)
//   sparsity = {nrow, ncol, colind[ncol+1], row[nnz]} (compressed column)
//   NULL arg[i] means zeros, NULL res[i] means "not requested".
// These entry points carry no user pointer, so they act on the process-wide *current* context chosen
// with mpx_set_current().  Values leave in compressed-column order (mpx_ccs_perm); hess_l is the upper
// triangle, as CasADi's "triu:hess:gamma:x:x".
//
// CasADi is not installable in the build image: the metadata and the numerical results are tested
// through ctypes (tests/test_host.py, tests/test_gpu_golden.py), the hand-off to nlpsol itself is not.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mpx.h"

namespace {
typedef long long cint;
struct Current {
  mpx_ctx* ctx = nullptr;
  mpx_sizes sz{};
  std::vector<cint> sp_x, sp_p, sp_g, sp_one, sp_jac, sp_hess;
  std::vector<int64_t> perm_j, perm_h;
  std::vector<double> zx, zp, zl, buf_j, buf_h, buf_g, buf_grad;
  bool pin = false;
  std::vector<std::pair<const void*, size_t>> pinned;  // caller arrays registered so far (mpx_current_pin_buffers)
  // Same-iterate cache.  An NLP solver asks for f, g (trial point) and then grad_f, jac_g, hess_l (accepted point) at the SAME x
  // through separate entry points, and each call pays the launch + completion floor of the device (13-19 us, more than the
  // kernels of a small problem).  The first call at a new (x, p) therefore evaluates f, g and grad_f (and jac_g when its values
  // are small enough that copying them later is cheaper than another call) in ONE fused pass into page-locked scratch of the
  // context; the sibling calls at that point are served from there by memcpy.  "Same point" = memcmp with the copy taken at
  // the first call (exact; ~3 us per 100 KB).  MPX_NO_COALESCE=1 switches it off (A/B).
  bool coalesce = true, cvalid = false, jac_small = false;
  int have = 0;  // MPX_F | MPX_G | MPX_GRAD | MPX_JAC present in the scratch for (cx, cp)
  std::vector<double> cx, cp;
  double cf = 0, *sg = nullptr, *sgrad = nullptr, *sjac = nullptr;
  long long n_fused = 0, n_served = 0;  // statistics (mpx_current_cache_stats)
  // Constants of a large Jacobian stay in the caller's array (opt-in, mpx_current_keep_jac_constants).  Three quarters of jac_g at
  // the metric's configuration are copies of table entries, the same at every iterate, and a single evaluation is bound by what it
  // writes over PCIe (0.96 MB).  When nlp_jac_g is handed the SAME res[1] it completed a full pass into, and the sampled constants
  // below still hold their values, only the (z, p)-dependent entries are rewritten (MPX_JAC_VARIABLE_ONLY | MPX_CCS_ORDER).
  // The CONTRACT is the caller's (CasADi's nlpsol writes that work-vector slice through nlp_jac_g only); the samples catch a buffer
  // that was cleared, reused or reallocated in between, not a caller that alters single constants.
  bool keep_const = false;
  const double* jc_ptr = nullptr;                         // array the last FULL pass went to
  std::vector<std::pair<int64_t, double>> jc_samples;    // (compressed-column position, value) of sampled constant entries
  std::vector<int64_t> const_pos;                         // compressed-column positions of all constant entries
  long long n_jac_var = 0, n_jac_full = 0;                // statistics (mpx_current_jac_stats)
} C;

bool jac_constants_intact(const double* jv) {
  if (C.jc_samples.empty()) return false;
  for (auto& s : C.jc_samples)
    if (memcmp(&jv[s.first], &s.second, 8) != 0) return false;
  return true;
}
void jac_take_samples(const double* jv) {
  C.jc_samples.clear();
  const size_t n = C.const_pos.size();
  if (n == 0) return;
  const size_t want = std::min<size_t>(n, 509);  // (a prime: positions spread over every column block)
  for (size_t i = 0; i < want; ++i) {
    const int64_t k = C.const_pos[(size_t)((double)i * (double)(n - 1) / (double)std::max<size_t>(want - 1, 1))];
    C.jc_samples.emplace_back(k, jv[k]);
  }
}

void release_scratch() {
  if (!C.ctx) return;
  for (double** q : {&C.sg, &C.sgrad, &C.sjac})
    if (*q) mpx_host_free(C.ctx, *q), *q = nullptr;
}

// Is (x, p) the point the cache holds?  Otherwise start a new cache entry for it.
bool at_cached_point(const double* x, const double* p) {
  const size_t nz = (size_t)C.sz.n_z, np_ = (size_t)C.sz.n_p;
  if (C.cvalid && memcmp(C.cx.data(), x, nz * 8) == 0 && (np_ == 0 || memcmp(C.cp.data(), p, np_ * 8) == 0)) return true;
  C.cx.assign(x, x + nz);
  C.cp.assign(p, p + np_);
  C.cvalid = true, C.have = 0;
  return false;
}

// Make the outputs `want` (bits of MPX_F | MPX_G | MPX_GRAD | MPX_JAC, the latter only when jac_small) available in the scratch
// for the point (x, p).  The first evaluation at a point fetches the whole cacheable set in one pass.  Returns 0 on success.
int ensure_cached(const double* x, const double* p, int want) {
  if (!C.sg) {  // page-locked scratch, once per selection (a structure-only context has none: mpx_eval reports that below)
    void *a = nullptr, *b = nullptr, *c = nullptr;
    if (mpx_host_alloc(C.ctx, (size_t)C.sz.n_g * 8 + 8, &a) || mpx_host_alloc(C.ctx, (size_t)C.sz.n_z * 8 + 8, &b) ||
        (C.jac_small && mpx_host_alloc(C.ctx, (size_t)C.sz.nnz_jac * 8 + 8, &c))) {
      if (a) mpx_host_free(C.ctx, a);
      if (b) mpx_host_free(C.ctx, b);
      return 1;
    }
    C.sg = (double*)a, C.sgrad = (double*)b, C.sjac = (double*)c;
  }
  at_cached_point(x, p);
  if ((want & ~C.have) == 0) {
    ++C.n_served;
    return 0;
  }
  int mask = (C.have == 0 ? (MPX_F | MPX_G | MPX_GRAD | (C.jac_small ? MPX_JAC : 0)) : want) & ~C.have;
  double f = 0;
  if (mpx_eval(C.ctx, mask | ((mask & MPX_JAC) ? MPX_CCS_ORDER : 0), 1, x, p, 0, 0, 0, &f, C.sg, C.sgrad, C.sjac, 0)) {
    C.cvalid = false;
    return 1;
  }
  if (mask & MPX_F) C.cf = f;
  C.have |= mask;
  ++C.n_fused;
  return 0;
}

void release_pins() {
  for (auto& e : C.pinned)
    if (e.second) mpx_host_unregister(C.ctx, const_cast<void*>(e.first));
  C.pinned.clear();
}

// Register a caller array once; a failed registration (e.g. memory that is already page-locked) is remembered
// with size 0 so that it is not retried on every call.
void pin(const void* p, size_t n_doubles) {
  if (!C.pin || !p || !n_doubles) return;
  for (auto& e : C.pinned)
    if (e.first == p) return;
  const bool ok = mpx_host_register(C.ctx, const_cast<void*>(p), n_doubles * sizeof(double)) == MPX_OK;
  C.pinned.emplace_back(p, ok ? n_doubles : 0);
}

std::vector<cint> dense_col(cint n) {
  std::vector<cint> s = {n, 1, 0, n};
  for (cint i = 0; i < n; ++i) s.push_back(i);
  return s;
}
std::vector<cint> ccs(cint nrow, cint ncol, const std::vector<int64_t>& colind, const std::vector<int32_t>& row, const std::vector<int64_t>& perm) {
  std::vector<cint> s = {nrow, ncol};
  for (auto v : colind) s.push_back(v);
  for (auto k : perm) s.push_back(row[k]);
  return s;
}
const double* in(const double** arg, int i, std::vector<double>& zeros) { return arg && arg[i] ? arg[i] : zeros.data(); }
}  // namespace

extern "C" int mpx_current_pin_buffers(int enable) {
  if (!C.ctx) return MPX_ERR_INVALID;
  if (!enable) release_pins();
  C.pin = enable != 0;
  return MPX_OK;
}

extern "C" int mpx_current_pin_stats(long long* registered, long long* failed) {
  long long ok = 0, bad = 0;
  for (auto& e : C.pinned) (e.second ? ok : bad)++;
  if (registered) *registered = ok;
  if (failed) *failed = bad;
  return MPX_OK;
}

extern "C" int mpx_current_keep_jac_constants(int enable) {
  if (!C.ctx) return MPX_ERR_INVALID;
  C.keep_const = enable != 0;
  C.jc_ptr = nullptr, C.jc_samples.clear();
  return MPX_OK;
}

extern "C" int mpx_current_jac_stats(long long* variable_only_passes, long long* full_passes) {
  if (variable_only_passes) *variable_only_passes = C.n_jac_var;
  if (full_passes) *full_passes = C.n_jac_full;
  return MPX_OK;
}

extern "C" int mpx_current_cache_stats(long long* fused_passes, long long* served_from_cache) {
  if (fused_passes) *fused_passes = C.n_fused;
  if (served_from_cache) *served_from_cache = C.n_served;
  return MPX_OK;
}

extern "C" int mpx_set_current(mpx_ctx* ctx) {
  if (C.ctx) {  // drop the registrations made for the previous selection (failed ones are skipped by the size)
    for (auto& e : C.pinned)
      if (e.second) mpx_host_unregister(C.ctx, const_cast<void*>(e.first));
    C.pinned.clear();
    release_scratch();
  }
  if (!ctx) {
    C = Current{};
    return MPX_OK;
  }
  Current n;
  n.ctx = ctx;
  int rc = mpx_get_sizes(ctx, &n.sz);
  if (rc) return rc;
  const mpx_sizes& z = n.sz;
  std::vector<int32_t> r(z.nnz_jac), c(z.nnz_jac), hr(z.nnz_hess), hc(z.nnz_hess);
  std::vector<int64_t> colind(z.n_z + 1);
  n.perm_j.resize(z.nnz_jac);
  n.perm_h.resize(z.nnz_hess);
  mpx_pattern_jac(ctx, r.data(), c.data());
  mpx_ccs_perm(ctx, MPX_JAC, n.perm_j.data(), colind.data());
  n.sp_jac = ccs(z.n_g, z.n_z, colind, r, n.perm_j);
  mpx_pattern_hess(ctx, hr.data(), hc.data());
  mpx_ccs_perm(ctx, MPX_HESS, n.perm_h.data(), colind.data());
  n.sp_hess = ccs(z.n_z, z.n_z, colind, hr, n.perm_h);
  n.sp_x = dense_col(z.n_z);
  n.sp_p = dense_col(z.n_p);
  n.sp_g = dense_col(z.n_g);
  n.sp_one = dense_col(1);
  n.zx.assign(z.n_z, 0.0);
  n.zp.assign(z.n_p, 0.0);
  n.zl.assign(z.n_g, 0.0);
  n.buf_j.resize(z.nnz_jac);
  n.buf_h.resize(z.nnz_hess);
  n.buf_g.resize(z.n_g);
  n.buf_grad.resize(z.n_z);
  {  // compressed-column positions of the constant entries (mpx_current_keep_jac_constants)
    std::vector<uint8_t> var(z.nnz_jac > 0 ? z.nnz_jac : 1);
    if (mpx_pattern_jac_variable(ctx, var.data()) == MPX_OK)
      for (int64_t k = 0; k < z.nnz_jac; ++k)
        if (!var[(size_t)n.perm_j[(size_t)k]]) n.const_pos.push_back(k);
  }
  n.coalesce = getenv("MPX_NO_COALESCE") == nullptr;
  n.jac_small = z.nnz_jac * 8 <= 65536;  // copying <= 64 KB costs ~2 us: cheaper than any second call
  C = std::move(n);
  return MPX_OK;
}

#define MPX_COMMON(NAME, NIN, NOUT)                                                     \
  extern "C" long long NAME##_n_in(void) { return NIN; }                                \
  extern "C" long long NAME##_n_out(void) { return NOUT; }                              \
  extern "C" void NAME##_incref(void) {}                                                \
  extern "C" void NAME##_decref(void) {}                                                \
  /* optional memory-object protocol of generated code: one stateless object */         \
  extern "C" int NAME##_alloc_mem(void) { return 0; }                                   \
  extern "C" int NAME##_init_mem(int) { return 0; }                                     \
  extern "C" void NAME##_free_mem(int) {}                                               \
  extern "C" int NAME##_checkout(void) { return 0; }                                    \
  extern "C" void NAME##_release(int) {}                                                \
  extern "C" double NAME##_default_in(long long) { return 0.0; }                        \
  extern "C" int NAME##_work(long long* a, long long* r, long long* iw, long long* w) { \
    if (a) *a = NIN;                                                                    \
    if (r) *r = NOUT;                                                                   \
    if (iw) *iw = 0;                                                                    \
    if (w) *w = 0;                                                                      \
    return 0;                                                                           \
  }

// ---- nlp_f : (x, p) -> (f) ---------------------------------------------------------------------
MPX_COMMON(nlp_f, 2, 1)
extern "C" const char* nlp_f_name_in(long long i) { return i == 0 ? "x" : (i == 1 ? "p" : 0); }
extern "C" const char* nlp_f_name_out(long long i) { return i == 0 ? "f" : 0; }
extern "C" const long long* nlp_f_sparsity_in(long long i) { return i == 0 ? C.sp_x.data() : (i == 1 ? C.sp_p.data() : 0); }
extern "C" const long long* nlp_f_sparsity_out(long long i) { return i == 0 ? C.sp_one.data() : 0; }
extern "C" int nlp_f(const double** arg, double** res, long long*, double*, int) {
  if (!C.ctx) return 1;
  double f;
  if (C.pin) pin(arg ? arg[0] : 0, C.sz.n_z);
  if (C.coalesce) {
    if (ensure_cached(in(arg, 0, C.zx), in(arg, 1, C.zp), MPX_F)) return 1;
    if (res && res[0]) res[0][0] = C.cf;
    return 0;
  }
  if (mpx_eval(C.ctx, MPX_F, 1, in(arg, 0, C.zx), in(arg, 1, C.zp), 0, 0, 0, &f, 0, 0, 0, 0)) return 1;
  if (res && res[0]) res[0][0] = f;
  return 0;
}

// ---- nlp_g : (x, p) -> (g) ---------------------------------------------------------------------
MPX_COMMON(nlp_g, 2, 1)
extern "C" const char* nlp_g_name_in(long long i) { return nlp_f_name_in(i); }
extern "C" const char* nlp_g_name_out(long long i) { return i == 0 ? "g" : 0; }
extern "C" const long long* nlp_g_sparsity_in(long long i) { return nlp_f_sparsity_in(i); }
extern "C" const long long* nlp_g_sparsity_out(long long i) { return i == 0 ? C.sp_g.data() : 0; }
extern "C" int nlp_g(const double** arg, double** res, long long*, double*, int) {
  if (!C.ctx) return 1;
  if (C.coalesce) {
    if (C.pin) pin(arg ? arg[0] : 0, C.sz.n_z);
    if (ensure_cached(in(arg, 0, C.zx), in(arg, 1, C.zp), MPX_G)) return 1;
    if (res && res[0]) memcpy(res[0], C.sg, (size_t)C.sz.n_g * 8);
    return 0;
  }
  double* g = res && res[0] ? res[0] : C.buf_g.data();
  if (C.pin) pin(arg ? arg[0] : 0, C.sz.n_z), pin(res ? res[0] : 0, C.sz.n_g);
  return mpx_eval(C.ctx, MPX_G, 1, in(arg, 0, C.zx), in(arg, 1, C.zp), 0, 0, 0, 0, g, 0, 0, 0) ? 1 : 0;
}

// ---- nlp_grad_f : (x, p) -> (f, grad_f_x) ------------------------------------------------------
MPX_COMMON(nlp_grad_f, 2, 2)
extern "C" const char* nlp_grad_f_name_in(long long i) { return nlp_f_name_in(i); }
extern "C" const char* nlp_grad_f_name_out(long long i) { return i == 0 ? "f" : (i == 1 ? "grad_f_x" : 0); }
extern "C" const long long* nlp_grad_f_sparsity_in(long long i) { return nlp_f_sparsity_in(i); }
extern "C" const long long* nlp_grad_f_sparsity_out(long long i) { return i == 0 ? C.sp_one.data() : (i == 1 ? C.sp_x.data() : 0); }
extern "C" int nlp_grad_f(const double** arg, double** res, long long*, double*, int) {
  if (!C.ctx) return 1;
  double f;
  if (C.coalesce) {
    if (C.pin) pin(arg ? arg[0] : 0, C.sz.n_z);
    if (ensure_cached(in(arg, 0, C.zx), in(arg, 1, C.zp), MPX_F | MPX_GRAD)) return 1;
    if (res && res[0]) res[0][0] = C.cf;
    if (res && res[1]) memcpy(res[1], C.sgrad, (size_t)C.sz.n_z * 8);
    return 0;
  }
  double* gr = res && res[1] ? res[1] : C.buf_grad.data();
  if (C.pin) pin(arg ? arg[0] : 0, C.sz.n_z), pin(res ? res[1] : 0, C.sz.n_z);
  if (mpx_eval(C.ctx, MPX_F | MPX_GRAD, 1, in(arg, 0, C.zx), in(arg, 1, C.zp), 0, 0, 0, &f, 0, gr, 0, 0)) return 1;
  if (res && res[0]) res[0][0] = f;
  return 0;
}

// ---- nlp_jac_g : (x, p) -> (g, jac_g_x) --------------------------------------------------------
MPX_COMMON(nlp_jac_g, 2, 2)
extern "C" const char* nlp_jac_g_name_in(long long i) { return nlp_f_name_in(i); }
extern "C" const char* nlp_jac_g_name_out(long long i) { return i == 0 ? "g" : (i == 1 ? "jac_g_x" : 0); }
extern "C" const long long* nlp_jac_g_sparsity_in(long long i) { return nlp_f_sparsity_in(i); }
extern "C" const long long* nlp_jac_g_sparsity_out(long long i) { return i == 0 ? C.sp_g.data() : (i == 1 ? C.sp_jac.data() : 0); }
extern "C" int nlp_jac_g(const double** arg, double** res, long long*, double*, int) {
  if (!C.ctx) return 1;
  const bool want_g = res && res[0], want_j = res && res[1];
  if (C.coalesce) {
    const double *x = in(arg, 0, C.zx), *pp = in(arg, 1, C.zp);
    if (C.pin) pin(arg ? arg[0] : 0, C.sz.n_z);
    if (C.jac_small) {  // small Jacobians are part of the fused first pass
      if (ensure_cached(x, pp, (want_g ? MPX_G : 0) | (want_j ? MPX_JAC : 0))) return 1;
      if (want_g) memcpy(res[0], C.sg, (size_t)C.sz.n_g * 8);
      if (want_j) memcpy(res[1], C.sjac, (size_t)C.sz.nnz_jac * 8);
      return 0;
    }
    // large Jacobians: the values go straight into the caller's array (a later copy of ~1 MB would cost what the call costs);
    // g is not recomputed when the caller does not ask for it (res[0] == NULL: CasADi's IPOPT interface) or the cache has it
    if (want_g && ensure_cached(x, pp, MPX_G)) return 1;
    if (want_g) memcpy(res[0], C.sg, (size_t)C.sz.n_g * 8);
    if (!want_j) return 0;
    if (C.pin) pin(res[1], C.sz.nnz_jac);
    const bool vo = C.keep_const && C.jc_ptr == res[1] && jac_constants_intact(res[1]);
    if (mpx_eval(C.ctx, MPX_JAC | MPX_CCS_ORDER | (vo ? MPX_JAC_VARIABLE_ONLY : 0), 1, x, pp, 0, 0, 0, 0, 0, 0, res[1], 0)) {
      C.jc_ptr = nullptr;
      return 1;
    }
    if (vo)
      ++C.n_jac_var;
    else {
      ++C.n_jac_full;
      if (C.keep_const) C.jc_ptr = res[1], jac_take_samples(res[1]);
    }
    return 0;
  }
  // (uncoalesced: what is not requested is not computed -- a NULL res[0] used to send g to a pageable spare buffer, which
  // also took the call off the zero-copy path)
  double* jv = want_j ? res[1] : C.buf_j.data();  // compressed-column order straight from the device
  if (C.pin) pin(arg ? arg[0] : 0, C.sz.n_z), pin(res ? res[0] : 0, C.sz.n_g), pin(res ? res[1] : 0, C.sz.nnz_jac);
  return mpx_eval(C.ctx, (want_g ? MPX_G : 0) | MPX_JAC | MPX_CCS_ORDER, 1, in(arg, 0, C.zx), in(arg, 1, C.zp), 0, 0, 0, 0, want_g ? res[0] : 0, 0, jv, 0) ? 1 : 0;
}

// ---- nlp_hess_l : (x, p, lam_f, lam_g) -> (triu_hess_gamma_x_x) ------------------------------------
// Names: CasADi 3.6.0 (the reference's pin, requirements.txt:4) asks an external oracle for each function BY NAME and then checks
// every input / output name against its request string with ':' replaced by '_' (External::factory).  IpoptInterface requests
// {"x","p","lam:f","lam:g"} -> {"triu:hess:gamma:x:x"}, so the one output must be called triu_hess_gamma_x_x -- like grad_f_x,
// jac_g_x, lam_f, lam_g, grad_gamma_x, grad_gamma_p elsewhere in this file.  tests/c_abi/nlpsol_like.c holds the request table and
// computes the expected names by that rule.
MPX_COMMON(nlp_hess_l, 4, 1)
extern "C" const char* nlp_hess_l_name_in(long long i) {
  static const char* n[] = {"x", "p", "lam_f", "lam_g"};
  return i >= 0 && i < 4 ? n[i] : 0;
}
// ... and the request is VERSION-specific: the reference admits casadi >= 3.5.5 (setup.py:29, requirements_dev.txt:5), whose solver
// interfaces ask for the same upper-triangular Hessian as "sym:hess:gamma:x:x" (the "triu:" attribute came with 3.6).  The name is a
// property of the process-wide hand-off like the current context: mpx_current_set_casadi_abi(305 | 306) picks it, and
// mpx_current_set_hess_l_output_name takes whatever string an importer's "Inconsistent output name. Expected: ..." asks for.
static char g_hess_l_out_name[64] = "triu_hess_gamma_x_x";
extern "C" int mpx_current_set_hess_l_output_name(const char* name) {
  if (!name || !*name || strlen(name) >= sizeof g_hess_l_out_name) return MPX_ERR_INVALID;
  for (const char* q = name; *q; ++q)
    if (!((*q >= 'a' && *q <= 'z') || (*q >= 'A' && *q <= 'Z') || (*q >= '0' && *q <= '9') || *q == '_')) return MPX_ERR_INVALID;
  strcpy(g_hess_l_out_name, name);
  return MPX_OK;
}
extern "C" int mpx_current_set_casadi_abi(int major_minor) {
  if (major_minor >= 306) return mpx_current_set_hess_l_output_name("triu_hess_gamma_x_x");
  if (major_minor >= 303) return mpx_current_set_hess_l_output_name("sym_hess_gamma_x_x");
  return MPX_ERR_UNSUPPORTED;
}
extern "C" const char* nlp_hess_l_name_out(long long i) { return i == 0 ? g_hess_l_out_name : 0; }
extern "C" const long long* nlp_hess_l_sparsity_in(long long i) {
  return i == 0 ? C.sp_x.data() : (i == 1 ? C.sp_p.data() : (i == 2 ? C.sp_one.data() : (i == 3 ? C.sp_g.data() : 0)));
}
extern "C" const long long* nlp_hess_l_sparsity_out(long long i) { return i == 0 ? C.sp_hess.data() : 0; }
extern "C" int nlp_hess_l(const double** arg, double** res, long long*, double*, int) {
  if (!C.ctx) return 1;
  const double sigma = arg && arg[2] ? arg[2][0] : 0.0;
  double* hv = res && res[0] ? res[0] : C.buf_h.data();
  if (C.pin) pin(arg ? arg[0] : 0, C.sz.n_z), pin(arg ? arg[3] : 0, C.sz.n_g), pin(res ? res[0] : 0, C.sz.nnz_hess);
  return mpx_eval(C.ctx, MPX_HESS | MPX_CCS_ORDER, 1, in(arg, 0, C.zx), in(arg, 1, C.zp), 0, in(arg, 3, C.zl), &sigma, 0, 0, 0, 0, hv) ? 1 : 0;
}

// ---- nlp : (x, p) -> (f, g) ----------------------------------------------------------------------
// The base oracle.  `ca.nlpsol(name, solver, "libmpx.so", opts)` is `nlpsol(name, solver, external("nlp", "libmpx.so"), opts)`:
// CasADi resolves THIS symbol first (it sizes the problem from its sparsities) and then asks the same library for nlp_f, nlp_g,
// nlp_grad_f, nlp_jac_g, nlp_hess_l and nlp_grad by name (OracleFunction::create_function on an external oracle).
MPX_COMMON(nlp, 2, 2)
extern "C" const char* nlp_name_in(long long i) { return nlp_f_name_in(i); }
extern "C" const char* nlp_name_out(long long i) { return i == 0 ? "f" : (i == 1 ? "g" : 0); }
extern "C" const long long* nlp_sparsity_in(long long i) { return nlp_f_sparsity_in(i); }
extern "C" const long long* nlp_sparsity_out(long long i) { return i == 0 ? C.sp_one.data() : (i == 1 ? C.sp_g.data() : 0); }
extern "C" int nlp(const double** arg, double** res, long long*, double*, int) {
  if (!C.ctx) return 1;
  const bool want_f = res && res[0], want_g = res && res[1];
  const double *x = in(arg, 0, C.zx), *pp = in(arg, 1, C.zp);
  if (C.pin) pin(arg ? arg[0] : 0, C.sz.n_z);
  if (C.coalesce) {
    if (ensure_cached(x, pp, (want_f ? MPX_F : 0) | (want_g ? MPX_G : 0))) return 1;
    if (want_f) res[0][0] = C.cf;
    if (want_g) memcpy(res[1], C.sg, (size_t)C.sz.n_g * 8);
    return 0;
  }
  double f = 0;
  if (!want_f && !want_g) return 0;
  if (mpx_eval(C.ctx, (want_f ? MPX_F : 0) | (want_g ? MPX_G : 0), 1, x, pp, 0, 0, 0, &f, want_g ? res[1] : 0, 0, 0, 0)) return 1;
  if (want_f) res[0][0] = f;
  return 0;
}

// ---- nlp_grad : (x, p, lam_f, lam_g) -> (f, g, grad_gamma_x, grad_gamma_p) -------------------------
// gamma = lam_f * f + lam_g^T g.  CasADi's Nlpsol calls it once after the last iterate (lam_f = 1, the final multipliers) for the
// outputs its options ask for -- by default only grad_gamma_p, whose negative is the lam_p of the result dict (calc_lam_p;
// calc_lam_x / calc_f / calc_g add the others).
MPX_COMMON(nlp_grad, 4, 4)
extern "C" const char* nlp_grad_name_in(long long i) { return nlp_hess_l_name_in(i); }
extern "C" const char* nlp_grad_name_out(long long i) {
  static const char* n[] = {"f", "g", "grad_gamma_x", "grad_gamma_p"};
  return i >= 0 && i < 4 ? n[i] : 0;
}
extern "C" const long long* nlp_grad_sparsity_in(long long i) { return nlp_hess_l_sparsity_in(i); }
extern "C" const long long* nlp_grad_sparsity_out(long long i) {
  return i == 0 ? C.sp_one.data() : (i == 1 ? C.sp_g.data() : (i == 2 ? C.sp_x.data() : (i == 3 ? C.sp_p.data() : 0)));
}
extern "C" int nlp_grad(const double** arg, double** res, long long* iw, double* w, int mem) {
  if (!C.ctx) return 1;
  if (res && (res[0] || res[1])) {
    double* r2[2] = {res[0], res[1]};
    if (nlp(arg, r2, iw, w, mem)) return 1;
  }
  double* gx = res ? res[2] : 0;
  double* gp = res && C.sz.n_p > 0 ? res[3] : 0;
  if (!gx && !gp) return 0;
  const double sigma = arg && arg[2] ? arg[2][0] : 0.0;
  return mpx_eval_grad_gamma(C.ctx, 1, in(arg, 0, C.zx), in(arg, 1, C.zp), 0, in(arg, 3, C.zl), &sigma, gx, gp) ? 1 : 0;
}
