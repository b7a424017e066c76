/* A CasADi-shaped caller of libmpx.so: what `ca.nlpsol("solver", "ipopt", "libmpx.so", opts)` does with an external NLP
 * library (reference hand-off: mpopt.py:757 creates the solver, mpopt.py:804 calls it), restated in plain C because CasADi is
 * not installable in the build image.  No CasADi source was available either: the protocol below is CasADi's documented
 * generated-code interface (the symbols a `CodeGenerator` emits and `external()` / `Importer` binds).
 *
 *   1. dlopen the library (no link-time dependency), dlsym EVERY companion symbol of the base oracle nlp ((x, p) -> (f, g):
 *      `nlpsol(name, solver, "lib.so")` is `nlpsol(name, solver, external("nlp", "lib.so"))`, so it is resolved FIRST and the
 *      problem is sized from ITS sparsities) and then of nlp_f, nlp_g, nlp_grad_f, nlp_jac_g, nlp_hess_l, nlp_grad: NAME, NAME_incref/_decref, NAME_n_in/_n_out, NAME_name_in/_name_out, NAME_sparsity_in/_sparsity_out,
 *      NAME_work and the optional NAME_alloc_mem/_init_mem/_free_mem/_checkout/_release/_default_in;
 *   2. read the sparsities (compressed column {nrow, ncol, colind[ncol+1], row[nnz]}; a dense pattern may stop after colind)
 *      and size ONE double arena `w` and ONE integer arena `iw` from them and from NAME_work, the way an nlpsol memory object
 *      holds x, p, lam, f, g, grad_f, jac_g, hess_l and the scratch of the embedded functions in one allocation;
 *      arg / res pointer arrays are carved per call out of shared arrays of max(sz_arg) / max(sz_res) entries;
 *   3. NAME_incref; mem = NAME_checkout(); the calls of an interior-point iteration in IPOPT's order --
 *          trial point:     nlp_f (x, p) -> f ;  nlp_g (x, p) -> g
 *          accepted point:  nlp_grad_f (x, p) -> (NULL, grad_f) ;  nlp_jac_g (x, p) -> (NULL, jac_g) ;
 *                           nlp_hess_l (x, p, lam_f, lam_g) -> hess
 *      with the SAME arena slices on every call (this is what lets mpx_current_pin_buffers(1) page-lock them once), a new x
 *      written into its slice per iterate, a rejected trial point now and then, and the NULL conventions (NULL arg = zeros,
 *      NULL res = not requested); NAME_release(mem); NAME_decref.
 *   3b. after the last iterate, what Nlpsol does to fill lam_p (calc_lam_p, on by default): ONE call of
 *          nlp_grad (x, p, lam_f = 1, lam_g) -> (NULL, NULL, NULL, grad_gamma_p)
 *      and, as with calc_f / calc_g / calc_lam_x, once more with all four outputs; the base oracle nlp at the same point.
 *   4. Results of every iterate go to a file; the test compares them with the goldens and with mpx_eval.
 *
 * The context itself is created through the context API of the same library (in a deployment Python's mp.mpopt does that and
 * calls mpx_set_current before handing the library path to nlpsol).
 *
 *   nlpsol_like /path/libmpx.so problem.bin [iterates.bin] out.bin        exit 0 = every step succeeded
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpx.h" /* types and constants only: every function is reached through dlsym */

typedef long long cint;
typedef int (*eval_t)(const double** arg, double** res, cint* iw, double* w, int mem);
typedef cint (*count_t)(void);
typedef const char* (*name_t)(cint);
typedef const cint* (*sp_t)(cint);
typedef int (*work_t)(cint*, cint*, cint*, cint*);
typedef void (*void_t)(void);
typedef int (*alloc_t)(void);
typedef int (*init_t)(int);
typedef void (*memv_t)(int);
typedef double (*defin_t)(cint);

typedef struct {
  const char* name;
  eval_t eval;
  count_t n_in, n_out;
  name_t name_in, name_out;
  sp_t sp_in, sp_out;
  work_t work;
  void_t incref, decref;
  alloc_t alloc_mem, checkout;
  init_t init_mem;
  memv_t free_mem, release;
  defin_t default_in;
  cint sz_arg, sz_res, sz_iw, sz_w;
  int mem;
} fn_t;

static void* lib;
static void* sym(const char* base, const char* suffix, int required) {
  char n[128];
  snprintf(n, sizeof n, "%s%s", base, suffix);
  void* p = dlsym(lib, n);
  if (!p && required) {
    fprintf(stderr, "missing symbol %s\n", n);
    exit(3);
  }
  return p;
}

static cint sp_nnz(const cint* sp) { return sp[2 + sp[1]]; } /* colind[ncol] */

static void* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  *n = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(*n ? *n : 1);
  if (fread(p, 1, *n, f) != *n) { free(p); p = NULL; }
  fclose(f);
  return p;
}

#define DIE(code, ...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return code; } while (0)

int main(int argc, char** argv) {
  if (argc < 4) return 64;
  const char* in_path = argc == 5 ? argv[3] : NULL;
  const char* out_path = argv[argc - 1];
  lib = dlopen(argv[1], RTLD_LAZY | RTLD_LOCAL);
  if (!lib) DIE(65, "dlopen: %s", dlerror());
  /* ---- context API (what Python does before the hand-off) ---- */
  int (*p_create)(const mpx_problem*, mpx_ctx**) = (int (*)(const mpx_problem*, mpx_ctx**))sym("mpx_create", "", 1);
  int (*p_destroy)(mpx_ctx*) = (int (*)(mpx_ctx*))sym("mpx_destroy", "", 1);
  int (*p_sizes)(const mpx_ctx*, mpx_sizes*) = (int (*)(const mpx_ctx*, mpx_sizes*))sym("mpx_get_sizes", "", 1);
  int (*p_set_current)(mpx_ctx*) = (int (*)(mpx_ctx*))sym("mpx_set_current", "", 1);
  int (*p_pin)(int) = (int (*)(int))sym("mpx_current_pin_buffers", "", 1);
  int (*p_cache)(long long*, long long*) = (int (*)(long long*, long long*))sym("mpx_current_cache_stats", "", 1);
  int (*p_pinstats)(long long*, long long*) = (int (*)(long long*, long long*))sym("mpx_current_pin_stats", "", 1);
  const char* (*p_err)(const mpx_ctx*) = (const char* (*)(const mpx_ctx*))sym("mpx_last_error", "", 1);
  size_t nb = 0;
  unsigned char* raw = slurp(argv[2], &nb);
  if (!raw) return 66;
  int64_t h[8], code_size;
  double tau[2];
  unsigned char* q = raw;
  memcpy(h, q, sizeof h); q += sizeof h;
  memcpy(tau, q, sizeof tau); q += sizeof tau;
  memcpy(&code_size, q, 8); q += 8;
  mpx_problem prob;
  memset(&prob, 0, sizeof prob);
  prob.version = MPX_VERSION;
  prob.n_phases = (int32_t)h[0]; prob.nx = (int32_t)h[1]; prob.nu = (int32_t)h[2]; prob.na = (int32_t)h[3];
  prob.n_segments = (int32_t)h[4]; prob.scheme = (int32_t)h[5]; prob.n_links = (int32_t)h[6];
  prob.tau0 = tau[0]; prob.tau1 = tau[1];
  prob.poly_orders = (const int32_t*)q; q += 4 * h[4];
  prob.links = (const int32_t*)q; q += 8 * h[6];
  prob.structure = (const int32_t*)q; prob.structure_len = h[7]; q += 4 * h[7];
  prob.code_object = code_size ? q : NULL;
  prob.code_object_size = (size_t)code_size;
  mpx_ctx* ctx = NULL;
  if (p_create(&prob, &ctx) != MPX_OK) DIE(2, "mpx_create: %s", p_err(NULL));
  mpx_sizes sz;
  if (p_sizes(ctx, &sz) != MPX_OK) return 2;
  if (p_set_current(ctx) != MPX_OK) return 2;

  /* ---- 1. bind the functions like an importer: the base oracle first, then the derived ones by name ---- */
#define NFN 7
#define I_NLP 5
#define I_GRAD 6
  /* ONE table: the seven requests exactly as CasADi 3.6.0 (the reference's pin, requirements.txt:4) states them --
   *   Nlpsol::init:         oracle "nlp"; create_function("nlp_grad", {"x","p","lam:f","lam:g"}, {"f","g","grad:gamma:x","grad:gamma:p"}, {{"gamma",{"f","g"}}})
   *   IpoptInterface::init: create_function("nlp_f", {"x","p"}, {"f"}), ("nlp_g", {"x","p"}, {"g"}), ("nlp_grad_f", {"x","p"}, {"f","grad:f:x"}),
   *                         ("nlp_jac_g", {"x","p"}, {"g","jac:g:x"}), ("nlp_hess_l", {"x","p","lam:f","lam:g"}, {"triu:hess:gamma:x:x"}, {{"gamma",{"f","g"}}})
   * For an external oracle External::factory looks the function up BY NAME in the library and then checks every input / output
   * name against the request string with ':' replaced by '_' ("Inconsistent input name. Expected: ..."); the expected symbol
   * names below are COMPUTED by that rule from the request strings, never written out. */
  /* NLPSOL_LIKE_CASADI_ABI=305: the table of CasADi 3.5.x (the reference's setup.py:29 admits casadi >= 3.5.5), which differs in ONE
   * request -- nlp_hess_l's output is "sym:hess:gamma:x:x" -- after telling the library so (mpx_current_set_casadi_abi). */
  const char* abi_env = getenv("NLPSOL_LIKE_CASADI_ABI");
  const int abi = abi_env ? atoi(abi_env) : 306;
  if (abi != 306) {
    int (*p_abi)(int) = (int (*)(int))sym("mpx_current_set_casadi_abi", "", 1);
    if (p_abi(abi) != MPX_OK) DIE(4, "mpx_current_set_casadi_abi(%d) refused", abi);
  }
  typedef struct { const char* name; const char* in[4]; const char* out[4]; } request_t;
  const request_t REQ[NFN] = {
      {"nlp_f", {"x", "p", 0, 0}, {"f", 0, 0, 0}},
      {"nlp_g", {"x", "p", 0, 0}, {"g", 0, 0, 0}},
      {"nlp_grad_f", {"x", "p", 0, 0}, {"f", "grad:f:x", 0, 0}},
      {"nlp_jac_g", {"x", "p", 0, 0}, {"g", "jac:g:x", 0, 0}},
      {"nlp_hess_l", {"x", "p", "lam:f", "lam:g"}, {abi >= 306 ? "triu:hess:gamma:x:x" : "sym:hess:gamma:x:x", 0, 0, 0}},
      {"nlp", {"x", "p", 0, 0}, {"f", "g", 0, 0}},
      {"nlp_grad", {"x", "p", "lam:f", "lam:g"}, {"f", "g", "grad:gamma:x", "grad:gamma:p"}}};
  const char* NAMES[NFN];
  cint NIN[NFN], NOUT[NFN];
  char IN_NAMES[NFN][4][40], OUT_NAMES[NFN][4][40];
  for (int k = 0; k < NFN; ++k) {
    NAMES[k] = REQ[k].name;
    NIN[k] = NOUT[k] = 0;
    for (int i = 0; i < 4; ++i) {
      const char* src[2] = {REQ[k].in[i], REQ[k].out[i]};
      char* dst[2] = {IN_NAMES[k][i], OUT_NAMES[k][i]};
      for (int d = 0; d < 2; ++d) {
        dst[d][0] = 0;
        if (!src[d]) continue;
        size_t n = strlen(src[d]);
        if (n >= sizeof IN_NAMES[0][0]) return 64;
        for (size_t c = 0; c <= n; ++c) dst[d][c] = src[d][c] == ':' ? '_' : src[d][c];
        ++*(d ? &NOUT[k] : &NIN[k]);
      }
    }
  }
  static const int ORDER[NFN] = {I_NLP, 0, 1, 2, 3, 4, I_GRAD};
  fn_t F[NFN];
  cint max_arg = 0, max_res = 0, max_iw = 0, max_w = 0;
  for (int kk = 0; kk < NFN; ++kk) {
    const int k = ORDER[kk];
    fn_t* f = &F[k];
    memset(f, 0, sizeof *f);
    f->name = NAMES[k];
    f->eval = (eval_t)sym(NAMES[k], "", 1);
    f->n_in = (count_t)sym(NAMES[k], "_n_in", 1); f->n_out = (count_t)sym(NAMES[k], "_n_out", 1);
    f->name_in = (name_t)sym(NAMES[k], "_name_in", 1); f->name_out = (name_t)sym(NAMES[k], "_name_out", 1);
    f->sp_in = (sp_t)sym(NAMES[k], "_sparsity_in", 1); f->sp_out = (sp_t)sym(NAMES[k], "_sparsity_out", 1);
    f->work = (work_t)sym(NAMES[k], "_work", 1);
    f->incref = (void_t)sym(NAMES[k], "_incref", 1); f->decref = (void_t)sym(NAMES[k], "_decref", 1);
    f->alloc_mem = (alloc_t)sym(NAMES[k], "_alloc_mem", 0); f->init_mem = (init_t)sym(NAMES[k], "_init_mem", 0);
    f->free_mem = (memv_t)sym(NAMES[k], "_free_mem", 0); f->checkout = (alloc_t)sym(NAMES[k], "_checkout", 0);
    f->release = (memv_t)sym(NAMES[k], "_release", 0); f->default_in = (defin_t)sym(NAMES[k], "_default_in", 0);
    if (!f->alloc_mem || !f->init_mem || !f->free_mem || !f->checkout || !f->release || !f->default_in)
      DIE(4, "%s: optional memory-object symbols are not all there", NAMES[k]);
    if (f->n_in() != NIN[k] || f->n_out() != NOUT[k]) DIE(4, "%s: n_in / n_out", NAMES[k]);
    for (cint i = 0; i < NIN[k]; ++i)
      if (!f->name_in(i) || strcmp(f->name_in(i), IN_NAMES[k][i]))
        DIE(4, "%s: Inconsistent input name. Expected: %s, got: %s", NAMES[k], IN_NAMES[k][i], f->name_in(i) ? f->name_in(i) : "(null)");
    for (cint i = 0; i < NOUT[k]; ++i)
      if (!f->name_out(i) || strcmp(f->name_out(i), OUT_NAMES[k][i]))
        DIE(4, "%s: Inconsistent output name. Expected: %s, got: %s", NAMES[k], OUT_NAMES[k][i], f->name_out(i) ? f->name_out(i) : "(null)");
    if (f->name_in(NIN[k]) || f->name_out(NOUT[k]) || f->sp_in(NIN[k]) || f->sp_out(NOUT[k])) DIE(4, "%s: out-of-range index must give NULL", NAMES[k]);
    if (f->work(&f->sz_arg, &f->sz_res, &f->sz_iw, &f->sz_w)) DIE(4, "%s_work failed", NAMES[k]);
    if (f->sz_arg < NIN[k] || f->sz_res < NOUT[k]) DIE(4, "%s_work: sz_arg / sz_res smaller than n_in / n_out", NAMES[k]);
    if (f->sz_arg > max_arg) max_arg = f->sz_arg;
    if (f->sz_res > max_res) max_res = f->sz_res;
    if (f->sz_iw > max_iw) max_iw = f->sz_iw;
    if (f->sz_w > max_w) max_w = f->sz_w;
    f->incref();
  }
  /* ---- 2. sparsities -> sizes; ONE arena ---- */
  const cint* sx = F[I_NLP].sp_in(0); const cint* spp = F[I_NLP].sp_in(1);  /* the base oracle sizes the problem */
  const cint* sg = F[I_NLP].sp_out(1); const cint* sj = F[3].sp_out(1); const cint* sh = F[4].sp_out(0);
  const cint n_x = sx[0], n_p = spp[0], n_g = sg[0], nnz_j = sp_nnz(sj), nnz_h = sp_nnz(sh);
  if (n_x != sz.n_z || n_p != sz.n_p || n_g != sz.n_g || nnz_j != sz.nnz_jac || nnz_h != sz.nnz_hess) DIE(4, "sparsities disagree with mpx_get_sizes");
  if (sx[1] != 1 || sp_nnz(sx) != n_x || sj[0] != n_g || sj[1] != n_x || sh[0] != n_x || sh[1] != n_x) DIE(4, "sparsity shapes");
  for (int k = 0; k < NFN; ++k) {  /* every function sees the same x / p patterns; grad_f is dense n_x, lam_g dense n_g */
    if (F[k].sp_in(0)[0] != n_x || F[k].sp_in(1)[0] != n_p) DIE(4, "%s: input patterns", NAMES[k]);
  }
  if (F[2].sp_out(1)[0] != n_x || F[4].sp_in(3)[0] != n_g || F[4].sp_in(2)[0] != 1 || F[0].sp_out(0)[0] != 1) DIE(4, "dense patterns");
  if (F[1].sp_out(0)[0] != n_g || F[I_NLP].sp_out(0)[0] != 1 || F[I_GRAD].sp_in(2)[0] != 1 || F[I_GRAD].sp_in(3)[0] != n_g || F[I_GRAD].sp_out(0)[0] != 1 ||
      F[I_GRAD].sp_out(1)[0] != n_g || F[I_GRAD].sp_out(2)[0] != n_x || F[I_GRAD].sp_out(3)[0] != n_p || sp_nnz(F[I_GRAD].sp_out(3)) != n_p)
    DIE(4, "patterns of nlp / nlp_grad");
  for (cint j = 0; j < n_x; ++j) {  /* column pointers monotone, rows sorted inside a column, hess upper triangular */
    if (sj[2 + j + 1] < sj[2 + j] || sh[2 + j + 1] < sh[2 + j]) DIE(4, "colind not monotone");
    for (cint e = sj[2 + j]; e + 1 < sj[2 + j + 1]; ++e)
      if (sj[2 + n_x + 1 + e] >= sj[2 + n_x + 1 + e + 1]) DIE(4, "jac rows not strictly increasing in column %lld", j);
    for (cint e = sh[2 + j]; e < sh[2 + j + 1]; ++e)
      if (sh[2 + n_x + 1 + e] > j) DIE(4, "hess entry below the diagonal");
  }
  /* arena layout (doubles): x | p | lam_f | lam_g | f | g | grad_f | jac | hess | grad_gamma_x | lam_p | function scratch */
  const cint o_x = 0, o_p = o_x + n_x, o_lf = o_p + n_p, o_lg = o_lf + 1, o_f = o_lg + n_g, o_g = o_f + 1, o_gr = o_g + n_g,
             o_j = o_gr + n_x, o_h = o_j + nnz_j, o_ggx = o_h + nnz_h, o_lp = o_ggx + n_x, o_w = o_lp + n_p, n_w = o_w + max_w;
  double* w = calloc((size_t)n_w + 1, sizeof(double));
  cint* iw = calloc((size_t)max_iw + 1, sizeof(cint));
  const double** arg = calloc((size_t)max_arg + 1, sizeof(double*));
  double** res = calloc((size_t)max_res + 1, sizeof(double*));
  if (!w || !iw || !arg || !res) return 70;
  for (int k = 0; k < NFN; ++k) {
    F[k].mem = F[k].checkout();
    if (F[k].mem < 0) DIE(5, "%s_checkout", NAMES[k]);
  }

  FILE* out = fopen(out_path, "wb");
  if (!out) return 66;
  int64_t K = 0;
  long long stats[6] = {0, 0, 0, 0, 0, 0}; /* fused passes, served calls, pins after iterate 1, pins at the end, failed pins, rejected points */
  int64_t head[7] = {n_x, n_p, n_g, nnz_j, nnz_h, 0, 0};
  if (!in_path) { /* structure only: the numerical entry point must fail loudly, not crash */
    arg[0] = w + o_x; arg[1] = w + o_p; res[0] = w + o_f;
    head[6] = F[0].eval(arg, res, iw, w + o_w, F[0].mem);
    fwrite(head, 8, 7, out);
    fwrite(sj, sizeof(cint), (size_t)(2 + n_x + 1 + nnz_j), out);
    fwrite(sh, sizeof(cint), (size_t)(2 + n_x + 1 + nnz_h), out);
  } else {
    size_t ni = 0;
    unsigned char* in = slurp(in_path, &ni);
    if (!in) return 67;
    memcpy(&K, in, 8); /* int64 K, then z[K][n_x], p[n_p], lam[K][n_g], sigma[K] */
    const double* Z = (const double*)(in + 8);
    const double* P = Z + K * n_x;
    const double* LAM = P + n_p;
    const double* SIG = LAM + K * n_g;
    if (p_pin(1) != MPX_OK) return 2;
    /* NLPSOL_LIKE_KEEP_JAC=1: the opt-in of a caller whose Jacobian array is written by nlp_jac_g only (mpx_current_keep_jac_constants)
     * -- and, to show the safety net, ONE iterate where this caller breaks that promise and clears the array between the calls */
    const int keep_jac = getenv("NLPSOL_LIKE_KEEP_JAC") != NULL;
    if (keep_jac) {
      int (*p_keep)(int) = (int (*)(int))sym("mpx_current_keep_jac_constants", "", 1);
      if (p_keep(1) != MPX_OK) return 2;
    }
    memcpy(w + o_p, P, 8 * (size_t)n_p);
    head[5] = K;
    fwrite(head, 8, 7, out);
    for (int64_t k = 0; k < K; ++k) {
      /* a rejected trial point before every third iterate: f and g only, at a point nobody asks derivatives for */
      if (k % 3 == 2) {
        for (cint i = 0; i < n_x; ++i) w[o_x + i] = 0.5 * (Z[k * n_x + i] + Z[(k - 1) * n_x + i]);
        arg[0] = w + o_x; arg[1] = w + o_p; res[0] = w + o_f;
        if (F[0].eval(arg, res, iw, w + o_w, F[0].mem)) DIE(6, "nlp_f (rejected point)");
        res[0] = w + o_g;
        if (F[1].eval(arg, res, iw, w + o_w, F[1].mem)) DIE(6, "nlp_g (rejected point)");
        ++stats[5];
      }
      memcpy(w + o_x, Z + k * n_x, 8 * (size_t)n_x); /* the solver writes the new iterate into ITS vector */
      memcpy(w + o_lg, LAM + k * n_g, 8 * (size_t)n_g);
      w[o_lf] = SIG[k];
      arg[0] = w + o_x; arg[1] = w + o_p;
      res[0] = w + o_f;
      if (F[0].eval(arg, res, iw, w + o_w, F[0].mem)) DIE(6, "nlp_f");
      const double f_from_f = w[o_f];
      res[0] = w + o_g;
      if (F[1].eval(arg, res, iw, w + o_w, F[1].mem)) DIE(6, "nlp_g");
      res[0] = NULL; res[1] = w + o_gr;
      if (F[2].eval(arg, res, iw, w + o_w, F[2].mem)) DIE(6, "nlp_grad_f");
      res[0] = NULL; res[1] = w + o_j;
      if (keep_jac && k == 4) memset(w + o_j, 0xFF, 8 * (size_t)nnz_j); /* (the scribble) */
      if (F[3].eval(arg, res, iw, w + o_w, F[3].mem)) DIE(6, "nlp_jac_g");
      arg[2] = w + o_lf; arg[3] = w + o_lg; res[0] = w + o_h;
      if (F[4].eval(arg, res, iw, w + o_w, F[4].mem)) DIE(6, "nlp_hess_l");
      fwrite(&f_from_f, 8, 1, out);
      fwrite(w + o_g, 8, (size_t)n_g, out);
      fwrite(w + o_gr, 8, (size_t)n_x, out);
      fwrite(w + o_j, 8, (size_t)nnz_j, out);
      fwrite(w + o_h, 8, (size_t)nnz_h, out);
      if (k == 0) p_pinstats(&stats[2], &stats[4]);
    }
    p_pinstats(&stats[3], &stats[4]); /* (before the extra arrays of the NULL-convention calls below) */
    /* NULL conventions on the last iterate: f through nlp_grad_f's first output equals nlp_f's; g through nlp_jac_g's first
       output equals nlp_g's; NULL lam_f and lam_g = zeros -> the Hessian of nothing is zero; all-NULL res = nothing to do */
    double f2 = -1;
    arg[0] = w + o_x; arg[1] = w + o_p; res[0] = &f2; res[1] = NULL;
    if (F[2].eval(arg, res, iw, w + o_w, F[2].mem) || f2 != w[o_f]) DIE(7, "nlp_grad_f(res = {f, NULL}): %g vs %g", f2, w[o_f]);
    double* g2 = malloc(8 * (size_t)n_g + 8);
    res[0] = g2; res[1] = NULL;
    if (F[3].eval(arg, res, iw, w + o_w, F[3].mem) || memcmp(g2, w + o_g, 8 * (size_t)n_g)) DIE(7, "nlp_jac_g(res = {g, NULL})");
    res[0] = NULL; res[1] = NULL;
    if (F[3].eval(arg, res, iw, w + o_w, F[3].mem) || F[1].eval(arg, res, iw, w + o_w, F[1].mem)) DIE(7, "all-NULL res");
    double* h0 = malloc(8 * (size_t)nnz_h + 8);
    arg[2] = NULL; arg[3] = NULL; res[0] = h0;
    if (F[4].eval(arg, res, iw, w + o_w, F[4].mem)) DIE(7, "nlp_hess_l with NULL multipliers");
    for (cint e = 0; e < nnz_h; ++e)
      if (h0[e] != 0.0) DIE(7, "Hessian with zero multipliers is not zero at %lld", e);
    /* 3b. after the last iterate: the base oracle at the same point, then nlp_grad the way Nlpsol calls it for lam_p (only the
       fourth output, lam_f = 1), then with all four outputs */
    double fg_f = -1;
    double* fg_g = malloc(8 * (size_t)n_g + 8);
    arg[0] = w + o_x; arg[1] = w + o_p; res[0] = &fg_f; res[1] = fg_g;
    if (F[I_NLP].eval(arg, res, iw, w + o_w, F[I_NLP].mem) || fg_f != w[o_f] || memcmp(fg_g, w + o_g, 8 * (size_t)n_g)) DIE(8, "nlp (x, p) -> (f, g)");
    const double one = 1.0;
    arg[2] = &one; arg[3] = w + o_lg; res[0] = NULL; res[1] = NULL; res[2] = NULL; res[3] = w + o_lp;
    if (F[I_GRAD].eval(arg, res, iw, w + o_w, F[I_GRAD].mem)) DIE(8, "nlp_grad (lam_p only)");
    fwrite(w + o_lp, 8, (size_t)n_p, out);
    double f3 = -1;
    arg[2] = w + o_lf; res[0] = &f3; res[1] = fg_g; res[2] = w + o_ggx; res[3] = w + o_lp;
    memset(fg_g, 0, 8 * (size_t)n_g);
    if (F[I_GRAD].eval(arg, res, iw, w + o_w, F[I_GRAD].mem) || f3 != w[o_f] || memcmp(fg_g, w + o_g, 8 * (size_t)n_g)) DIE(8, "nlp_grad (all outputs)");
    fwrite(w + o_ggx, 8, (size_t)n_x, out);
    fwrite(w + o_lp, 8, (size_t)n_p, out);
    res[0] = res[1] = res[2] = res[3] = NULL;
    if (F[I_GRAD].eval(arg, res, iw, w + o_w, F[I_GRAD].mem)) DIE(8, "nlp_grad with all-NULL res");
    free(fg_g);
    p_cache(&stats[0], &stats[1]);
    fwrite(stats, 8, 6, out);
    if (keep_jac) {
      long long nv = -1, nf = -1;
      int (*p_jst)(long long*, long long*) = (int (*)(long long*, long long*))sym("mpx_current_jac_stats", "", 1);
      p_jst(&nv, &nf);
      printf("jac_passes variable_only=%lld full=%lld\n", nv, nf);
    }
    if (p_pin(0) != MPX_OK) return 2; /* unregisters h0 too: only now may it be freed */
    free(g2); free(h0);
    free(in);
  }
  fclose(out);
  for (int k = 0; k < NFN; ++k) {
    F[k].release(F[k].mem);
    F[k].decref();
  }
  if (p_set_current(NULL) != MPX_OK) return 2;
  if (p_destroy(ctx) != MPX_OK) return 2;
  free(w); free(iw); free(arg); free(res); free(raw);
  dlclose(lib);
  return 0;
}
