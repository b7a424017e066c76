/* A plain-C caller of libmpx (include/mpx.h): no Python, no C++, no HIP headers.
 *
 * Reads a problem description written by tests/test_c_abi.py (the structure ints the tracer produced, the grid and,
 * optionally, the gfx950 code object), creates a context, queries sizes and patterns and -- when a code object and
 * inputs are present -- evaluates f, g, grad_f, jac_g, hess_l for a batch through mpx_eval, then writes everything to
 * an output file for the test to compare with the golden vectors.  Exit code 0 = every call succeeded.
 *
 *   capi_driver problem.bin [inputs.bin] outputs.bin
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpx.h"

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

#define CHECK(call)                                                                      \
  do {                                                                                   \
    int rc_ = (call);                                                                    \
    if (rc_ != MPX_OK) {                                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mpx_last_error(ctx));                \
      return 2;                                                                          \
    }                                                                                    \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) return 64;
  const char* prob_path = argv[1];
  const char* in_path = argc == 4 ? argv[2] : NULL;
  const char* out_path = argv[argc - 1];
  size_t nb = 0;
  unsigned char* raw = slurp(prob_path, &nb);
  if (!raw) return 65;
  /* header: 8 int64: n_phases nx nu na S scheme n_links structure_len ; 2 double: tau0 tau1 ; int64 code_size;
     then int32 orders[S], int32 links[2*n_links], int32 structure[structure_len], code bytes */
  int64_t h[8];
  double tau[2];
  int64_t code_size;
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
  prob.device = 0;

  mpx_ctx* ctx = NULL;
  int rc = mpx_create(&prob, &ctx);
  if (rc != MPX_OK) { fprintf(stderr, "mpx_create -> %d: %s\n", rc, mpx_last_error(NULL)); return 2; }
  mpx_sizes sz;
  CHECK(mpx_get_sizes(ctx, &sz));
  int32_t* jr = malloc(4 * (sz.nnz_jac + 1)); int32_t* jc = malloc(4 * (sz.nnz_jac + 1));
  int32_t* hr = malloc(4 * (sz.nnz_hess + 1)); int32_t* hc = malloc(4 * (sz.nnz_hess + 1));
  int64_t* perm = malloc(8 * (sz.nnz_jac + 1)); int64_t* colind = malloc(8 * (sz.n_z + 1));
  CHECK(mpx_pattern_jac(ctx, jr, jc));
  CHECK(mpx_pattern_hess(ctx, hr, hc));
  CHECK(mpx_ccs_perm(ctx, MPX_JAC, perm, colind));

  FILE* out = fopen(out_path, "wb");
  if (!out) return 66;
  int64_t head[6] = {sz.n_z, sz.n_p, sz.n_g, sz.nnz_jac, sz.nnz_hess, 0};
  int64_t B = 0;
  double *f = NULL, *g = NULL, *grad = NULL, *jv = NULL, *hv = NULL;
  if (in_path) {
    size_t ni = 0;
    unsigned char* in = slurp(in_path, &ni);
    if (!in) return 67;
    memcpy(&B, in, 8);                                    /* int64 batch, then z[B][n_z], p[n_p], lam[B][n_g], sigma[B] */
    const double* z = (const double*)(in + 8);
    const double* p = z + B * sz.n_z;
    const double* lam = p + sz.n_p;
    const double* sigma = lam + B * sz.n_g;
    f = malloc(8 * B); g = malloc(8 * B * sz.n_g); grad = malloc(8 * B * sz.n_z);
    jv = malloc(8 * B * (sz.nnz_jac + 1)); hv = malloc(8 * B * (sz.nnz_hess + 1));
    CHECK(mpx_eval(ctx, MPX_F | MPX_G | MPX_GRAD | MPX_JAC, B, z, p, 0, NULL, NULL, f, g, grad, jv, NULL));
    CHECK(mpx_eval(ctx, MPX_HESS, B, z, p, 0, lam, sigma, NULL, NULL, NULL, NULL, hv));
    free(in);
  }
  head[5] = B;
  fwrite(head, 8, 6, out);
  fwrite(jr, 4, sz.nnz_jac, out); fwrite(jc, 4, sz.nnz_jac, out);
  fwrite(hr, 4, sz.nnz_hess, out); fwrite(hc, 4, sz.nnz_hess, out);
  fwrite(colind, 8, sz.n_z + 1, out);
  if (B) {
    fwrite(f, 8, B, out); fwrite(g, 8, B * sz.n_g, out); fwrite(grad, 8, B * sz.n_z, out);
    fwrite(jv, 8, B * sz.nnz_jac, out); fwrite(hv, 8, B * sz.nnz_hess, out);
  }
  fclose(out);
  CHECK(mpx_destroy(ctx));
  free(raw);
  return 0;
}
