/* CPU ORACLE (C) -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar, single-threaded restatement of the reference's collocation hot path
 * (/root/reference/mpopt/mpopt.py) for the benchmark OCPs of BASELINE.json:
 *     f, g, grad_f, jac_g   = what CasADi's nlp_f / nlp_g / nlp_grad_f / nlp_jac_g compute for the
 *                             NLP built by mpopt.create_nlp (mpopt.py:574-639, 757);
 *     hess_l                = upper triangle of the Hessian of  sigma*f + lam_g^T g  (CasADi's nlp_hess_l,
 *                             derived inside ca.nlpsol at mpopt.py:757), orc_hess below.
 * Used (a) by tests/ as a second, independent checker of the HIP kernels and of the numpy oracle,
 * (b) by bench.py as the `cpu_baseline` ("kind": "port") timed on the GPU box's host cores.
 * Never linked into or called from the product (mpopt_amd/).
 *
 * Pinning: validated in tests/test_oracle.py against tests/golden/ (vectors produced by the
 * reference's own code, see tests/golden/make_golden.py) through the numpy oracle and directly.
 * CasADi itself is absent, so CasADi's evaluation order is unpinned (O(1e-16) differences).
 *
 * Arithmetic of the tables follows the reference's "numerical" back-end: Lagrange basis as
 * monomial-coefficient products (np.poly1d, mpopt.py:4006-4011), D by polyder + Horner
 * (mpopt.py:3842-3847), w by polyint (mpopt.py:3879-3880).  Node sets are an input (the caller
 * passes scipy's, as the reference does at mpopt.py:4220, 4246).
 *
 * The OCP functions and their first and second derivatives are hand-written below (independent of the
 * product's tracer); the chain rule through scaling, segment step h and node time t is done here
 * in the unscaled formulation (mpopt.py:175-206).  Structural masks are hand-written too; none of the
 * reference's six problems depends on t explicitly (there the (t0, tf) border of the Hessian comes from h only); the synthetic
 * `time_dependent` problem does, in dynamics, path row, running and terminal functions, with a parameter and non-unit scaling.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* LONG-DOUBLE ACCUMULATION MODE (-DORC_LONG_DOUBLE, built as liborc_ld.so by oracle/c_oracle.py): the SAME source with every
 * `double` -- arguments, tables, intermediates, sums -- an x87 80-bit long double (64-bit significand: 11 more bits than the
 * kernels and than this file's normal build).  Inputs are the same binary64 numbers converted exactly, so its results are the
 * formulas of this file evaluated (almost) without rounding.  The parity tests use it as the ARBITER: where a GPU value and
 * the double build of this oracle differ by a few 1e-11 in a cancelling entry, |gpu - ld| and |c - ld| say whose rounding it is
 * (tests/test_gpu_parity.py).  Literals such as 0.3 stay the binary64 constants of the double build (same problem). */
#ifdef ORC_LONG_DOUBLE
#define double long double
#define cos cosl
#define sin sinl
#define exp expl
#define fabs fabsl
#define sqrt sqrtl
#endif

#define MAXV 16 /* nx + nu + 1 + na */
#define MAXP 256 /* max degree + 1 (round 6: the library takes degrees up to 255; above degree 10 the tables come from orc_set_table, the monomial form below is only exact there) */

typedef struct {
  const char* name;
  int nx, nu, na, nc, ntc;
  /* dyn[nx], pc[nc], L ; derivatives w.r.t. v = (x[nx], u[nu], t, a[na]) row-major [.][nv] */
  void (*node)(const double* x, const double* u, double t, const double* a, double* dyn, double* pc, double* L);
  void (*node_d)(const double* x, const double* u, double t, const double* a, double* ddyn, double* dpc, double* dL);
  /* structural masks (1 = entry exists), same shapes as the derivative arrays */
  const unsigned char* m_dyn;
  const unsigned char* m_pc;
  /* terminal: w = (xf[nx], tf, x0[nx], t0, a[na]) */
  void (*term)(const double* xf, double tf, const double* x0, double t0, const double* a, double* M, double* tc);
  void (*term_d)(const double* xf, double tf, const double* x0, double t0, const double* a, double* dM, double* dtc);
  const unsigned char* m_M;
  const unsigned char* m_tc;
  /* second derivatives, weighted: Hpsi[nv][nv] += wL * d2L + sum_c wd[c] * d2dyn_c ; Hchi[nv][nv] += sum_j wc[j] * d2pc_j
   * (arrays arrive zeroed); masks: m1_L[nv] (dL), m2_psi / m2_chi [nv][nv] (union over the terms) */
  void (*node_dd)(const double* x, const double* u, double t, const double* a, const double* wd, const double* wc, double wL,
                  double* Hpsi, double* Hchi);
  const unsigned char* m1_L;
  const unsigned char* m2_psi;
  const unsigned char* m2_chi;
  /* Hw[ntv][ntv] += wM * d2M + sum_j wt[j] * d2tc_j ; mask m2_term[ntv][ntv] */
  void (*term_dd)(const double* xf, double tf, const double* x0, double t0, const double* a, double wM, const double* wt, double* Hw);
  const unsigned char* m2_term;
} ocp_fns;

/* ------------------------------------------------------------------ problems ---------------- */
/* moon lander: examples/singlephase/moon_lander.py:30-63 */
static void ml_node(const double* x, const double* u, double t, const double* a, double* d, double* pc, double* L) {
  d[0] = x[1];
  d[1] = u[0] - 1.5;
  *L = u[0];
}
static void ml_node_d(const double* x, const double* u, double t, const double* a, double* dd, double* dpc, double* dL) {
  /* nv = 4: x0 x1 u t */
  memset(dd, 0, 8 * sizeof(double));
  dd[0 * 4 + 1] = 1.0;
  dd[1 * 4 + 2] = 1.0;
  dL[0] = 0, dL[1] = 0, dL[2] = 1.0, dL[3] = 0;
}
static const unsigned char ml_mdyn[8] = {0, 1, 0, 0, 0, 0, 1, 0};
static void ml_term(const double* xf, double tf, const double* x0, double t0, const double* a, double* M, double* tc) {
  *M = 0;
  tc[0] = xf[0];
  tc[1] = xf[1];
}
static void ml_term_d(const double* xf, double tf, const double* x0, double t0, const double* a, double* dM, double* dtc) {
  /* ntv = 6: xf0 xf1 tf x00 x01 t0 */
  memset(dM, 0, 6 * sizeof(double));
  memset(dtc, 0, 12 * sizeof(double));
  dtc[0] = 1.0;
  dtc[6 + 1] = 1.0;
}
static const unsigned char ml_mM[6] = {0, 0, 0, 0, 0, 0};
static const unsigned char ml_mtc[12] = {1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0};

/* Van der Pol: tests/test_mpopt.py:206-227 */
static void vdp_node(const double* x, const double* u, double t, const double* a, double* d, double* pc, double* L) {
  d[0] = (1 - x[1] * x[1]) * x[0] - x[1] + u[0];
  d[1] = x[0];
  *L = x[0] * x[0] + x[1] * x[1] + u[0] * u[0];
}
static void vdp_node_d(const double* x, const double* u, double t, const double* a, double* dd, double* dpc, double* dL) {
  memset(dd, 0, 8 * sizeof(double));
  dd[0] = 1 - x[1] * x[1];
  dd[1] = -2 * x[1] * x[0] - 1;
  dd[2] = 1;
  dd[4] = 1;
  dL[0] = 2 * x[0], dL[1] = 2 * x[1], dL[2] = 2 * u[0], dL[3] = 0;
}
static const unsigned char vdp_mdyn[8] = {1, 1, 1, 0, 1, 0, 0, 0};
static void none_term(const double* xf, double tf, const double* x0, double t0, const double* a, double* M, double* tc) { *M = 0; }
static void none_term_d(const double* xf, double tf, const double* x0, double t0, const double* a, double* dM, double* dtc) {
  memset(dM, 0, MAXV * 2 * sizeof(double));
}
static const unsigned char zeros_mask[4 * MAXV * MAXV] = {0};

/* Van der Pol with parameter and path row: examples/singlephase/dae_vdp.py:28-60; nv = 5: x0 x1 u t a0 */
static void dvdp_node(const double* x, const double* u, double t, const double* a, double* d, double* pc, double* L) {
  vdp_node(x, u, t, a, d, pc, L);
  pc[0] = a[0] - x[1];
}
static void dvdp_node_d(const double* x, const double* u, double t, const double* a, double* dd, double* dpc, double* dL) {
  memset(dd, 0, 10 * sizeof(double));
  dd[0] = 1 - x[1] * x[1];
  dd[1] = -2 * x[1] * x[0] - 1;
  dd[2] = 1;
  dd[5] = 1;
  memset(dpc, 0, 5 * sizeof(double));
  dpc[1] = -1;
  dpc[4] = 1;
  dL[0] = 2 * x[0], dL[1] = 2 * x[1], dL[2] = 2 * u[0], dL[3] = 0, dL[4] = 0;
}
static const unsigned char dvdp_mdyn[10] = {1, 1, 1, 0, 0, 1, 0, 0, 0, 0};
static const unsigned char dvdp_mpc[5] = {0, 1, 0, 0, 1};

/* hypersensitive: examples/singlephase/hyper_sensitive.py:31-41; nv = 3: x u t */
static void hs_node(const double* x, const double* u, double t, const double* a, double* d, double* pc, double* L) {
  d[0] = -x[0] * x[0] * x[0] + u[0];
  *L = 0.5 * (x[0] * x[0] + u[0] * u[0]);
}
static void hs_node_d(const double* x, const double* u, double t, const double* a, double* dd, double* dpc, double* dL) {
  dd[0] = -3 * x[0] * x[0];
  dd[1] = 1;
  dd[2] = 0;
  dL[0] = x[0], dL[1] = u[0], dL[2] = 0;
}
static const unsigned char hs_mdyn[3] = {1, 1, 0};
static void hs_term(const double* xf, double tf, const double* x0, double t0, const double* a, double* M, double* tc) {
  *M = 0;
  tc[0] = xf[0] - 1.0;
}
static void hs_term_d(const double* xf, double tf, const double* x0, double t0, const double* a, double* dM, double* dtc) {
  memset(dM, 0, 4 * sizeof(double));
  memset(dtc, 0, 4 * sizeof(double));
  dtc[0] = 1.0;
}
static const unsigned char hs_mtc[4] = {1, 0, 0, 0};

/* two-phase Schwartz: tests/test_mpopt.py:165-202; nv = 4 */
static void sw_node0(const double* x, const double* u, double t, const double* a, double* d, double* pc, double* L) {
  d[0] = x[1];
  d[1] = u[0] - 0.1 * (1.0 + 2.0 * x[0] * x[0]) * x[1];
  pc[0] = 1.0 - 9.0 * (x[0] - 1) * (x[0] - 1) - (x[1] - 0.4) * (x[1] - 0.4) / (0.3 * 0.3);
  *L = 0;
}
static void sw_node0_d(const double* x, const double* u, double t, const double* a, double* dd, double* dpc, double* dL) {
  memset(dd, 0, 8 * sizeof(double));
  dd[1] = 1;
  dd[4] = -0.1 * 4.0 * x[0] * x[1];
  dd[5] = -0.1 * (1.0 + 2.0 * x[0] * x[0]);
  dd[6] = 1;
  dpc[0] = -18.0 * (x[0] - 1);
  dpc[1] = -2.0 * (x[1] - 0.4) / (0.3 * 0.3);
  dpc[2] = 0, dpc[3] = 0;
  memset(dL, 0, 4 * sizeof(double));
}
static void sw_node1(const double* x, const double* u, double t, const double* a, double* d, double* pc, double* L) {
  d[0] = x[1];
  d[1] = u[0] - 0.1 * (1.0 + 2.0 * x[0] * x[0]) * x[1];
  *L = 0;
}
static void sw_node1_d(const double* x, const double* u, double t, const double* a, double* dd, double* dpc, double* dL) {
  double dummy[4];
  sw_node0_d(x, u, t, a, dd, dummy, dL);
}
static const unsigned char sw_mdyn[8] = {0, 1, 0, 0, 1, 1, 1, 0};
static const unsigned char sw_mpc[4] = {1, 1, 0, 0};
static void sw_term1(const double* xf, double tf, const double* x0, double t0, const double* a, double* M, double* tc) {
  *M = 5 * (xf[0] * xf[0] + xf[1] * xf[1]);
}
static void sw_term1_d(const double* xf, double tf, const double* x0, double t0, const double* a, double* dM, double* dtc) {
  memset(dM, 0, 6 * sizeof(double));
  dM[0] = 10 * xf[0];
  dM[1] = 10 * xf[1];
}
static const unsigned char sw_mM1[6] = {1, 1, 0, 0, 0, 0};


/* ---- second derivatives (weighted sums, see ocp_fns) ---- */
static void zero_node_dd(const double* x, const double* u, double t, const double* a, const double* wd, const double* wc, double wL,
                         double* Hp, double* Hc) {}
static void zero_term_dd(const double* xf, double tf, const double* x0, double t0, const double* a, double wM, const double* wt, double* Hw) {}
/* moon lander: dyn and L are linear; L = u */
static const unsigned char ml_m1L[4] = {0, 0, 1, 0};
/* Van der Pol (nv = 4: x0 x1 u t): dyn0 = (1 - x1^2) x0 - x1 + u ; L = x0^2 + x1^2 + u^2 */
static void vdp_node_dd(const double* x, const double* u, double t, const double* a, const double* wd, const double* wc, double wL,
                        double* Hp, double* Hc) {
  const int nv = 4;
  Hp[0 * nv + 1] += wd[0] * (-2 * x[1]);
  Hp[1 * nv + 0] += wd[0] * (-2 * x[1]);
  Hp[1 * nv + 1] += wd[0] * (-2 * x[0]);
  Hp[0 * nv + 0] += wL * 2, Hp[1 * nv + 1] += wL * 2, Hp[2 * nv + 2] += wL * 2;
}
static const unsigned char vdp_m1L[4] = {1, 1, 1, 0};
static const unsigned char vdp_m2psi[16] = {1, 1, 0, 0, 1, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0};
/* dae_vdp (nv = 5: x0 x1 u t a0): same dyn and L, path row linear */
static void dvdp_node_dd(const double* x, const double* u, double t, const double* a, const double* wd, const double* wc, double wL,
                         double* Hp, double* Hc) {
  const int nv = 5;
  Hp[0 * nv + 1] += wd[0] * (-2 * x[1]);
  Hp[1 * nv + 0] += wd[0] * (-2 * x[1]);
  Hp[1 * nv + 1] += wd[0] * (-2 * x[0]);
  Hp[0 * nv + 0] += wL * 2, Hp[1 * nv + 1] += wL * 2, Hp[2 * nv + 2] += wL * 2;
}
static const unsigned char dvdp_m1L[5] = {1, 1, 1, 0, 0};
static const unsigned char dvdp_m2psi[25] = {1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
/* hypersensitive (nv = 3: x u t): dyn = -x^3 + u ; L = (x^2 + u^2)/2 */
static void hs_node_dd(const double* x, const double* u, double t, const double* a, const double* wd, const double* wc, double wL,
                       double* Hp, double* Hc) {
  const int nv = 3;
  Hp[0 * nv + 0] += wd[0] * (-6 * x[0]) + wL;
  Hp[1 * nv + 1] += wL;
}
static const unsigned char hs_m1L[3] = {1, 1, 0};
static const unsigned char hs_m2psi[9] = {1, 0, 0, 0, 1, 0, 0, 0, 0};
/* Schwartz (nv = 4: x0 x1 u t): dyn1 = u - 0.1 (1 + 2 x0^2) x1 ; phase 0 path row 1 - 9 (x0-1)^2 - (x1-0.4)^2/0.09 */
static void sw_node0_dd(const double* x, const double* u, double t, const double* a, const double* wd, const double* wc, double wL,
                        double* Hp, double* Hc) {
  const int nv = 4;
  Hp[0 * nv + 0] += wd[1] * (-0.1 * 4.0 * x[1]);
  Hp[0 * nv + 1] += wd[1] * (-0.1 * 4.0 * x[0]);
  Hp[1 * nv + 0] += wd[1] * (-0.1 * 4.0 * x[0]);
  Hc[0 * nv + 0] += wc[0] * (-18.0);
  Hc[1 * nv + 1] += wc[0] * (-2.0 / (0.3 * 0.3));
}
static void sw_node1_dd(const double* x, const double* u, double t, const double* a, const double* wd, const double* wc, double wL,
                        double* Hp, double* Hc) {
  const int nv = 4;
  Hp[0 * nv + 0] += wd[1] * (-0.1 * 4.0 * x[1]);
  Hp[0 * nv + 1] += wd[1] * (-0.1 * 4.0 * x[0]);
  Hp[1 * nv + 0] += wd[1] * (-0.1 * 4.0 * x[0]);
}
static const unsigned char sw_m2psi[16] = {1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static const unsigned char sw_m2chi[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
/* Schwartz phase 1 Mayer term 5 (xf0^2 + xf1^2); ntv = 6: xf0 xf1 tf x00 x01 t0 */
static void sw_term1_dd(const double* xf, double tf, const double* x0, double t0, const double* a, double wM, const double* wt, double* Hw) {
  Hw[0 * 6 + 0] += wM * 10, Hw[1 * 6 + 1] += wM * 10;
}
static const unsigned char sw_m2term1[36] = {1, 0, 0, 0, 0, 0, 0, 1};

/* Synthetic, explicitly time-dependent (tests/problems.py: time_dependent): nx = 2, nu = 1, na = 1, nv = 5: x0 x1 u t a0.
 *   dyn0 = x1 cos(0.3 t) + a0 u          dyn1 = -x0 x1 + u exp(-0.1 t) - 0.3 t x0
 *   path = x0 t - 3 - a0 x1              L    = u^2 + 0.1 x0 t + a0^2 x1^2
 *   M    = 0.3 xf0 x01 + 0.2 tf a0 + 0.05 (tf - t0)^2          tc = xf1 tf - x00 a0       (ntv = 7: xf0 xf1 tf x00 x01 t0 a0)
 * Every d/dt, d2/dt2 and d2/dt dv term of the chain rule through t(t0, tf) is non-zero here, which none of the reference's
 * benchmark problems exercises. */
static void td_node(const double* x, const double* u, double t, const double* a, double* d, double* pc, double* L) {
  d[0] = x[1] * cos(0.3 * t) + a[0] * u[0];
  d[1] = -x[0] * x[1] + u[0] * exp(-0.1 * t) - 0.3 * t * x[0];
  pc[0] = x[0] * t - 3.0 - a[0] * x[1];
  *L = u[0] * u[0] + 0.1 * x[0] * t + a[0] * a[0] * x[1] * x[1];
}
static void td_node_d(const double* x, const double* u, double t, const double* a, double* dd, double* dpc, double* dL) {
  const double c = cos(0.3 * t), sn = sin(0.3 * t), e = exp(-0.1 * t);
  memset(dd, 0, 10 * sizeof(double));
  dd[0 * 5 + 1] = c, dd[0 * 5 + 2] = a[0], dd[0 * 5 + 3] = -0.3 * x[1] * sn, dd[0 * 5 + 4] = u[0];
  dd[1 * 5 + 0] = -x[1] - 0.3 * t, dd[1 * 5 + 1] = -x[0], dd[1 * 5 + 2] = e, dd[1 * 5 + 3] = -0.1 * u[0] * e - 0.3 * x[0];
  dpc[0] = t, dpc[1] = -a[0], dpc[2] = 0, dpc[3] = x[0], dpc[4] = -x[1];
  dL[0] = 0.1 * t, dL[1] = 2 * a[0] * a[0] * x[1], dL[2] = 2 * u[0], dL[3] = 0.1 * x[0], dL[4] = 2 * a[0] * x[1] * x[1];
}
static const unsigned char td_mdyn[10] = {0, 1, 1, 1, 1, 1, 1, 1, 1, 0};
static const unsigned char td_mpc[5] = {1, 1, 0, 1, 1};
static void td_term(const double* xf, double tf, const double* x0, double t0, const double* a, double* M, double* tc) {
  *M = 0.3 * xf[0] * x0[1] + 0.2 * tf * a[0] + 0.05 * (tf - t0) * (tf - t0);
  tc[0] = xf[1] * tf - x0[0] * a[0];
}
static void td_term_d(const double* xf, double tf, const double* x0, double t0, const double* a, double* dM, double* dtc) {
  dM[0] = 0.3 * x0[1], dM[1] = 0, dM[2] = 0.2 * a[0] + 0.1 * (tf - t0), dM[3] = 0, dM[4] = 0.3 * xf[0], dM[5] = -0.1 * (tf - t0), dM[6] = 0.2 * tf;
  dtc[0] = 0, dtc[1] = tf, dtc[2] = xf[1], dtc[3] = -a[0], dtc[4] = 0, dtc[5] = 0, dtc[6] = -x0[0];
}
static const unsigned char td_mM[7] = {1, 0, 1, 0, 1, 1, 1};
static const unsigned char td_mtc[7] = {0, 1, 1, 1, 0, 0, 1};
static void td_node_dd(const double* x, const double* u, double t, const double* a, const double* wd, const double* wc, double wL,
                       double* Hp, double* Hc) {
  const int nv = 5;
  const double c = cos(0.3 * t), sn = sin(0.3 * t), e = exp(-0.1 * t);
  /* dyn0 */
  Hp[1 * nv + 3] += wd[0] * (-0.3 * sn), Hp[3 * nv + 1] += wd[0] * (-0.3 * sn);
  Hp[3 * nv + 3] += wd[0] * (-0.09 * x[1] * c);
  Hp[2 * nv + 4] += wd[0], Hp[4 * nv + 2] += wd[0];
  /* dyn1 */
  Hp[0 * nv + 1] += -wd[1], Hp[1 * nv + 0] += -wd[1];
  Hp[0 * nv + 3] += wd[1] * (-0.3), Hp[3 * nv + 0] += wd[1] * (-0.3);
  Hp[2 * nv + 3] += wd[1] * (-0.1 * e), Hp[3 * nv + 2] += wd[1] * (-0.1 * e);
  Hp[3 * nv + 3] += wd[1] * (0.01 * u[0] * e);
  /* L */
  Hp[2 * nv + 2] += wL * 2;
  Hp[0 * nv + 3] += wL * 0.1, Hp[3 * nv + 0] += wL * 0.1;
  Hp[1 * nv + 1] += wL * 2 * a[0] * a[0];
  Hp[1 * nv + 4] += wL * 4 * a[0] * x[1], Hp[4 * nv + 1] += wL * 4 * a[0] * x[1];
  Hp[4 * nv + 4] += wL * 2 * x[1] * x[1];
  /* path row */
  Hc[0 * nv + 3] += wc[0], Hc[3 * nv + 0] += wc[0];
  Hc[1 * nv + 4] += -wc[0], Hc[4 * nv + 1] += -wc[0];
}
static const unsigned char td_m1L[5] = {1, 1, 1, 1, 1};
static const unsigned char td_m2psi[25] = {0, 1, 0, 1, 0, /**/ 1, 1, 0, 1, 1, /**/ 0, 0, 1, 1, 1, /**/ 1, 1, 1, 1, 0, /**/ 0, 1, 1, 0, 1};
static const unsigned char td_m2chi[25] = {0, 0, 0, 1, 0, /**/ 0, 0, 0, 0, 1, /**/ 0, 0, 0, 0, 0, /**/ 1, 0, 0, 0, 0, /**/ 0, 1, 0, 0, 0};
static void td_term_dd(const double* xf, double tf, const double* x0, double t0, const double* a, double wM, const double* wt, double* Hw) {
  const int n = 7;
  Hw[0 * n + 4] += wM * 0.3, Hw[4 * n + 0] += wM * 0.3;
  Hw[2 * n + 6] += wM * 0.2, Hw[6 * n + 2] += wM * 0.2;
  Hw[2 * n + 2] += wM * 0.1, Hw[5 * n + 5] += wM * 0.1;
  Hw[2 * n + 5] += -wM * 0.1, Hw[5 * n + 2] += -wM * 0.1;
  Hw[1 * n + 2] += wt[0], Hw[2 * n + 1] += wt[0];
  Hw[3 * n + 6] += -wt[0], Hw[6 * n + 3] += -wt[0];
}
static const unsigned char td_m2term[49] = {0, 0, 0, 0, 1, 0, 0, /**/ 0, 0, 1, 0, 0, 0, 0, /**/ 0, 1, 1, 0, 0, 1, 1, /**/ 0, 0, 0, 0, 0, 0, 1,
                                            /**/ 1, 0, 0, 0, 0, 0, 0, /**/ 0, 0, 1, 0, 0, 1, 0, /**/ 0, 0, 1, 1, 0, 0, 0};

static const ocp_fns PROBLEMS[] = {
    {"moon_lander", 2, 1, 0, 0, 2, ml_node, ml_node_d, ml_mdyn, zeros_mask, ml_term, ml_term_d, ml_mM, ml_mtc,
     zero_node_dd, ml_m1L, zeros_mask, zeros_mask, zero_term_dd, zeros_mask},
    {"van_der_pol", 2, 1, 0, 0, 0, vdp_node, vdp_node_d, vdp_mdyn, zeros_mask, none_term, none_term_d, zeros_mask, zeros_mask,
     vdp_node_dd, vdp_m1L, vdp_m2psi, zeros_mask, zero_term_dd, zeros_mask},
    {"dae_vdp", 2, 1, 1, 1, 0, dvdp_node, dvdp_node_d, dvdp_mdyn, dvdp_mpc, none_term, none_term_d, zeros_mask, zeros_mask,
     dvdp_node_dd, dvdp_m1L, dvdp_m2psi, zeros_mask, zero_term_dd, zeros_mask},
    {"hyper_sensitive", 1, 1, 0, 0, 1, hs_node, hs_node_d, hs_mdyn, zeros_mask, hs_term, hs_term_d, zeros_mask, hs_mtc,
     hs_node_dd, hs_m1L, hs_m2psi, zeros_mask, zero_term_dd, zeros_mask},
    {"schwartz_phase0", 2, 1, 0, 1, 0, sw_node0, sw_node0_d, sw_mdyn, sw_mpc, none_term, none_term_d, zeros_mask, zeros_mask,
     sw_node0_dd, zeros_mask, sw_m2psi, sw_m2chi, zero_term_dd, zeros_mask},
    {"schwartz_phase1", 2, 1, 0, 0, 0, sw_node1, sw_node1_d, sw_mdyn, zeros_mask, sw_term1, sw_term1_d, sw_mM1, zeros_mask,
     sw_node1_dd, zeros_mask, sw_m2psi, zeros_mask, sw_term1_dd, sw_m2term1},
    {"time_dependent", 2, 1, 1, 1, 1, td_node, td_node_d, td_mdyn, td_mpc, td_term, td_term_d, td_mM, td_mtc,
     td_node_dd, td_m1L, td_m2psi, td_m2chi, td_term_dd, td_m2term},
};
#define N_PROBLEMS ((int)(sizeof(PROBLEMS) / sizeof(PROBLEMS[0])))

/* ------------------------------------------------------------------ tables ------------------ */
/* monomial coefficients, highest power first (numpy poly1d convention) */
static void lagrange_coeffs(const double* x, int n, int j, double* c /* n */) {
  double tmp[MAXP];
  int len = 1;
  c[0] = 1.0;
  for (int i = 0; i < n; ++i) {
    if (i == j) continue;
    double den = x[j] - x[i];
    /* c <- c * [1, -x_i] / den          (mpopt.py:4010) */
    tmp[0] = c[0];
    for (int k = 1; k < len; ++k) tmp[k] = c[k] - x[i] * c[k - 1];
    tmp[len] = -x[i] * c[len - 1];
    ++len;
    for (int k = 0; k < len; ++k) c[k] = tmp[k] / den;
  }
}
static double polyval(const double* c, int len, double t) {
  double v = 0;
  for (int k = 0; k < len; ++k) v = v * t + c[k];
  return v;
}

typedef struct {
  int deg;
  double tau[MAXP], D[MAXP * MAXP], w[MAXP], Cmid[MAXP * MAXP];
} deg_table;

static void build_table(deg_table* T, int deg, const double* taus, double tau0, double tau1) {
  int n = deg + 1;
  T->deg = deg;
  memcpy(T->tau, taus, n * sizeof(double));
  for (int j = 0; j < n; ++j) {
    double c[MAXP], d[MAXP], I[MAXP + 1];
    lagrange_coeffs(taus, n, j, c);
    for (int k = 0; k < n - 1; ++k) d[k] = c[k] * (double)(n - 1 - k); /* np.polyder */
    for (int i = 0; i < n; ++i) T->D[i * n + j] = n > 1 ? polyval(d, n - 1, taus[i]) : 0.0;
    for (int k = 0; k < n; ++k) I[k] = c[k] / (double)(n - k); /* np.polyint */
    I[n] = 0.0;
    T->w[j] = polyval(I, n + 1, tau1) - polyval(I, n + 1, tau0);
    for (int m = 0; m < deg; ++m) T->Cmid[m * n + j] = polyval(c, n, (taus[m] + taus[m + 1]) / 2.0); /* mpopt.py:350-352 */
  }
}

/* ------------------------------------------------------------------ NLP --------------------- */
typedef struct {
  int n_ph, S, N, nx, nu, na;
  int* orders;
  int* start;
  int* seg;
  int* pt;
  deg_table* tabs;
  int n_tabs;
  double tau0, tau1;
  double *sx, *su, *sa, st;
  const ocp_fns* fn[8];
  int midu[8];
  double* compW;
  int64_t n_zp, n_z, n_g, nnz;
  int64_t off_F[8], off_C[8], off_mU[8], off_TC[8], off_ev;
} orc;

static const deg_table* tab(const orc* o, int deg) {
  for (int k = 0; k < o->n_tabs; ++k)
    if (o->tabs[k].deg == deg) return &o->tabs[k];
  return 0;
}

/* names: one per phase; degs/taus: distinct degrees and their node sets (concatenated) */
orc* orc_create(int n_ph, const char** names, int S, const int* orders, int n_degs, const int* degs, const double* taus,
                double tau0, double tau1, const double* sx, const double* su, const double* sa, double st, const int* midu) {
  orc* o = (orc*)calloc(1, sizeof(orc));
  o->n_ph = n_ph;
  o->S = S;
  for (int p = 0; p < n_ph; ++p) {
    o->fn[p] = 0;
    for (int k = 0; k < N_PROBLEMS; ++k)
      if (!strcmp(PROBLEMS[k].name, names[p])) o->fn[p] = &PROBLEMS[k];
    if (!o->fn[p]) {
      free(o);
      return 0;
    }
    o->midu[p] = midu[p];
  }
  o->nx = o->fn[0]->nx, o->nu = o->fn[0]->nu, o->na = o->fn[0]->na;
  o->orders = (int*)malloc(S * sizeof(int));
  o->start = (int*)malloc((S + 1) * sizeof(int));
  memcpy(o->orders, orders, S * sizeof(int));
  o->start[0] = 0;
  for (int s = 0; s < S; ++s) o->start[s + 1] = o->start[s] + orders[s];
  o->N = o->start[S] + 1;
  o->tau0 = tau0, o->tau1 = tau1;
  o->tabs = (deg_table*)calloc(n_degs, sizeof(deg_table));
  o->n_tabs = n_degs;
  const double* t = taus;
  for (int k = 0; k < n_degs; ++k) {
    build_table(&o->tabs[k], degs[k], t, tau0, tau1);
    t += degs[k] + 1;
  }
  o->sx = (double*)malloc(o->nx * sizeof(double)), o->su = (double*)malloc((o->nu + 1) * sizeof(double));
  o->sa = (double*)malloc((o->na + 1) * sizeof(double));
  memcpy(o->sx, sx, o->nx * sizeof(double));
  memcpy(o->su, su, o->nu * sizeof(double));
  memcpy(o->sa, sa, o->na * sizeof(double));
  o->st = st;
  /* node -> (segment, point), mpopt.py:189-195, 208 */
  o->seg = (int*)malloc(o->N * sizeof(int)), o->pt = (int*)malloc(o->N * sizeof(int));
  for (int i = 0, s = 0, k = 0; i < o->N; ++i) {
    if (k > orders[s]) s++, k = 1;
    o->seg[i] = s, o->pt[i] = k++;
  }
  /* composite weights, mpopt.py:4060-4062 */
  o->compW = (double*)malloc(o->N * sizeof(double));
  o->compW[0] = tab(o, orders[0])->w[0];
  for (int s = 0; s < S; ++s)
    for (int k = 1; k <= orders[s]; ++k) o->compW[o->start[s] + k] = tab(o, orders[s])->w[k];
  int N = o->N;
  o->n_zp = (int64_t)N * (o->nx + o->nu) + 2 + o->na;
  o->n_z = o->n_zp * n_ph;
  int64_t r = 0, nnz = 0;
  int nv = o->nx + o->nu + 1 + o->na, ntv = 2 * o->nx + 2 + o->na;
  for (int p = 0; p < n_ph; ++p) {
    const ocp_fns* f = o->fn[p];
    o->off_F[p] = r, r += (int64_t)o->nx * N;
    o->off_C[p] = r, r += (int64_t)f->nc * N;
    o->off_mU[p] = r, r += o->midu[p] ? (int64_t)o->nu * (N - 1) : 0;
    o->off_TC[p] = r, r += f->ntc;
    for (int i = 0; i < N; ++i) {
      int pdeg = orders[o->seg[i]];
      for (int a = 0; a < o->nx; ++a) {
        nnz += pdeg + 1;
        for (int v = 0; v < nv; ++v) {
          if (v == a) continue;
          int m = f->m_dyn[a * nv + v];
          if (v == o->nx + o->nu) nnz += 2; /* t0, tf always: h multiplies dyn */
          else nnz += m;
        }
      }
      for (int j = 0; j < f->nc; ++j)
        for (int v = 0; v < nv; ++v) nnz += (v == o->nx + o->nu ? 2 : 1) * f->m_pc[j * nv + v];
      if (o->midu[p] && i > 0) nnz += (int64_t)o->nu * (pdeg + 1);
    }
    for (int j = 0; j < f->ntc; ++j)
      for (int v = 0; v < ntv; ++v) nnz += f->m_tc[j * ntv + v];
  }
  o->off_ev = r;
  if (n_ph > 1) r += (int64_t)(n_ph - 1) * (o->nx + o->nu + 1), nnz += 2 * (int64_t)(n_ph - 1) * (o->nx + o->nu + 1);
  o->n_g = r;
  o->nnz = nnz;
  return o;
}

/* Replace the tables of one degree (used above degree 10, where monomial-coefficient arithmetic
 * is no longer accurate: the caller passes 50-digit tables from oracle/mpopt_oracle.py). */
int orc_set_table(orc* o, int deg, const double* D, const double* w, const double* Cmid) {
  for (int k = 0; k < o->n_tabs; ++k)
    if (o->tabs[k].deg == deg) {
      int n = deg + 1;
      memcpy(o->tabs[k].D, D, n * n * sizeof(double));
      memcpy(o->tabs[k].w, w, n * sizeof(double));
      memcpy(o->tabs[k].Cmid, Cmid, deg * n * sizeof(double));
      o->compW[0] = tab(o, o->orders[0])->w[0];
      for (int s = 0; s < o->S; ++s)
        for (int q = 1; q <= o->orders[s]; ++q) o->compW[o->start[s] + q] = tab(o, o->orders[s])->w[q];
      return 0;
    }
  return -1;
}

/* The tables of one degree as this oracle holds them (the long-double build is given the double build's tables, so that both
 * evaluate the same problem). */
int orc_get_table(const orc* o, int deg, double* D, double* w, double* Cmid) {
  const deg_table* T = tab(o, deg);
  if (!T) return -1;
  int n = deg + 1;
  memcpy(D, T->D, n * n * sizeof(double));
  memcpy(w, T->w, n * sizeof(double));
  memcpy(Cmid, T->Cmid, deg * n * sizeof(double));
  return 0;
}

void orc_destroy(orc* o) {
  if (!o) return;
  free(o->orders), free(o->start), free(o->seg), free(o->pt), free(o->tabs), free(o->sx), free(o->su), free(o->sa), free(o->compW);
  free(o);
}
int64_t orc_n_z(const orc* o) { return o->n_z; }
int64_t orc_n_g(const orc* o) { return o->n_g; }
int64_t orc_nnz(const orc* o) { return o->nnz; }

/* One evaluation of f, g, grad_f (dense) and jac_g (COO triplets, row-major emission order).
 * rows/cols may be NULL (values only, the timed configuration).  Returns the number of triplets. */
int64_t orc_eval(const orc* o, const double* z, const double* p, double* f_out, double* g, double* grad, int32_t* rows,
                 int32_t* cols, double* vals) {
  const int N = o->N, nx = o->nx, nu = o->nu, na = o->na, nv = nx + nu + 1 + na, ntv = 2 * nx + 2 + na;
  int64_t q = 0;
  double ftot = 0;
  if (grad) memset(grad, 0, o->n_z * sizeof(double));
#define EMIT(r, c, v)                 \
  do {                                \
    if (rows) rows[q] = (int32_t)(r); \
    if (cols) cols[q] = (int32_t)(c); \
    if (vals) vals[q] = (v);          \
    ++q;                              \
  } while (0)
  for (int ph = 0; ph < o->n_ph; ++ph) {
    const ocp_fns* F = o->fn[ph];
    const double* zp = z + ph * o->n_zp;
    const double* X = zp;                      /* X[a*N+i] */
    const double* U = zp + (int64_t)nx * N;    /* U[b*N+i] */
    const int64_t zt = (int64_t)(nx + nu) * N; /* t0, tf, A */
    const double t0 = zp[zt] / o->st, tf = zp[zt + 1] / o->st; /* mpopt.py:175-176 */
    const int64_t zb = ph * o->n_zp;
    double a[MAXV];
    for (int c = 0; c < na; ++c) a[c] = zp[zt + 2 + c] / o->sa[c];
    const double* w = p + ph * o->S;
    const double dtau = o->tau1 - o->tau0;
    double t_seg0 = t0, h = (tf - t0) / dtau * w[0]; /* mpopt.py:180-184 */
    double dt0 = 0, dtf = 0, da[MAXV] = {0};
    for (int i = 0, s = 0; i < N; ++i) {
      if (o->seg[i] != s) { /* mpopt.py:190-195 */
        s = o->seg[i];
        t_seg0 += h * dtau;
        h = (tf - t0) / dtau * w[s];
      }
      const int pdeg = o->orders[s], k = o->pt[i], st = o->start[s];
      const deg_table* T = tab(o, pdeg);
      const double tk = T->tau[k] - o->tau0;
      const double t = t_seg0 + h * tk; /* mpopt.py:198 */
      /* d t / d(t0,tf), d h / d(t0,tf) in the unscaled times */
      const double theta = (t - t0) / (tf - t0);
      const double kap = w[s] / dtau;
      double x[MAXV], u[MAXV], dyn[MAXV], pc[MAXV], L, ddyn[MAXV * MAXV], dpc[MAXV * MAXV], dL[MAXV];
      for (int c = 0; c < nx; ++c) x[c] = X[(int64_t)c * N + i] / o->sx[c]; /* mpopt.py:196 */
      for (int c = 0; c < nu; ++c) u[c] = U[(int64_t)c * N + i] / o->su[c]; /* mpopt.py:197 */
      F->node(x, u, t, a, dyn, pc, &L);
      if (grad || vals) F->node_d(x, u, t, a, ddyn, dpc, dL);
      for (int c = 0; c < nx; ++c) { /* defect rows, mpopt.py:201, 227-232 */
        const int64_t row = o->off_F[ph] + (int64_t)c * N + i;
        double acc = 0;
        for (int j = 0; j <= pdeg; ++j) acc += T->D[k * (pdeg + 1) + j] * X[(int64_t)c * N + st + j];
        const double fi = h * o->sx[c] * dyn[c];
        if (g) g[row] = acc - fi;
        if (!vals) continue;
        for (int j = 0; j <= pdeg; ++j) {
          double v = T->D[k * (pdeg + 1) + j];
          if (j == k) v -= h * o->sx[c] * ddyn[c * nv + c] / o->sx[c];
          EMIT(row, zb + (int64_t)c * N + st + j, v);
        }
        for (int v = 0; v < nv; ++v) {
          if (v == c) continue;
          const double d = ddyn[c * nv + v];
          if (v < nx) {
            if (F->m_dyn[c * nv + v]) EMIT(row, zb + (int64_t)v * N + i, -h * o->sx[c] * d / o->sx[v]);
          } else if (v < nx + nu) {
            if (F->m_dyn[c * nv + v]) EMIT(row, zb + (int64_t)v * N + i, -h * o->sx[c] * d / o->su[v - nx]);
          } else if (v == nx + nu) {
            /* f = kap (tf - t0) Sx dyn(.., t0 + (tf-t0) theta, ..) */
            EMIT(row, zb + zt, -(-kap * o->sx[c] * dyn[c] + h * o->sx[c] * d * (1 - theta)) / o->st);
            EMIT(row, zb + zt + 1, -(kap * o->sx[c] * dyn[c] + h * o->sx[c] * d * theta) / o->st);
          } else if (F->m_dyn[c * nv + v]) {
            EMIT(row, zb + zt + 2 + (v - nx - nu - 1), -h * o->sx[c] * d / o->sa[v - nx - nu - 1]);
          }
        }
      }
      for (int j = 0; j < F->nc; ++j) { /* path rows, mpopt.py:204, 255 */
        const int64_t row = o->off_C[ph] + (int64_t)j * N + i;
        if (g) g[row] = pc[j];
        if (!vals) continue;
        for (int v = 0; v < nv; ++v) {
          if (!F->m_pc[j * nv + v]) continue;
          const double d = dpc[j * nv + v];
          if (v < nx) EMIT(row, zb + (int64_t)v * N + i, d / o->sx[v]);
          else if (v < nx + nu) EMIT(row, zb + (int64_t)v * N + i, d / o->su[v - nx]);
          else if (v == nx + nu) {
            EMIT(row, zb + zt, d * (1 - theta) / o->st);
            EMIT(row, zb + zt + 1, d * theta / o->st);
          } else EMIT(row, zb + zt + 2 + (v - nx - nu - 1), d / o->sa[v - nx - nu - 1]);
        }
      }
      /* running cost, mpopt.py:206, 455 */
      const double W = o->compW[i];
      ftot += W * h * L;
      if (grad) {
        for (int v = 0; v < nx; ++v) grad[zb + (int64_t)v * N + i] += W * h * dL[v] / o->sx[v];
        for (int v = 0; v < nu; ++v) grad[zb + (int64_t)(nx + v) * N + i] += W * h * dL[nx + v] / o->su[v];
        dt0 += W * (-kap * L + h * dL[nx + nu] * (1 - theta)) / o->st;
        dtf += W * (kap * L + h * dL[nx + nu] * theta) / o->st;
        for (int c = 0; c < na; ++c) da[c] += W * h * dL[nx + nu + 1 + c] / o->sa[c];
      }
    }
    if (o->midu[ph]) { /* control at mid-points, mpopt.py:350-369 */
      for (int b = 0; b < nu; ++b)
        for (int s = 0; s < o->S; ++s) {
          const int pdeg = o->orders[s], st = o->start[s];
          const deg_table* T = tab(o, pdeg);
          for (int m = 0; m < pdeg; ++m) {
            const int64_t row = o->off_mU[ph] + (int64_t)b * (N - 1) + st + m;
            double acc = 0;
            for (int j = 0; j <= pdeg; ++j) {
              acc += T->Cmid[m * (pdeg + 1) + j] * U[(int64_t)b * N + st + j];
              EMIT(row, zb + (int64_t)(nx + b) * N + st + j, T->Cmid[m * (pdeg + 1) + j]);
            }
            if (g) g[row] = acc;
          }
        }
    }
    /* terminal cost and constraints, mpopt.py:277-298 */
    double xf[MAXV], x0[MAXV], M, tc[MAXV], dM[2 * MAXV + 2], dtc[MAXV * (2 * MAXV + 2)];
    for (int c = 0; c < nx; ++c) xf[c] = X[(int64_t)c * N + N - 1] / o->sx[c], x0[c] = X[(int64_t)c * N] / o->sx[c];
    F->term(xf, tf, x0, t0, a, &M, tc);
    if (grad || vals) F->term_d(xf, tf, x0, t0, a, dM, dtc);
    ftot += M;
    for (int j = 0; j < F->ntc; ++j) {
      const int64_t row = o->off_TC[ph] + j;
      if (g) g[row] = tc[j];
      if (!vals) continue;
      for (int v = 0; v < ntv; ++v) {
        if (!F->m_tc[j * ntv + v]) continue;
        const double d = dtc[j * ntv + v];
        if (v < nx) EMIT(row, zb + (int64_t)v * N + N - 1, d / o->sx[v]);
        else if (v == nx) EMIT(row, zb + zt + 1, d / o->st);
        else if (v < 2 * nx + 1) EMIT(row, zb + (int64_t)(v - nx - 1) * N, d / o->sx[v - nx - 1]);
        else if (v == 2 * nx + 1) EMIT(row, zb + zt, d / o->st);
        else EMIT(row, zb + zt + 2 + (v - 2 * nx - 2), d / o->sa[v - 2 * nx - 2]);
      }
    }
    if (grad) {
      grad[zb + zt] += dt0, grad[zb + zt + 1] += dtf;
      for (int c = 0; c < na; ++c) grad[zb + zt + 2 + c] += da[c];
      for (int v = 0; v < ntv; ++v) {
        if (!F->m_M[v]) continue;
        const double d = dM[v];
        if (v < nx) grad[zb + (int64_t)v * N + N - 1] += d / o->sx[v];
        else if (v == nx) grad[zb + zt + 1] += d / o->st;
        else if (v < 2 * nx + 1) grad[zb + (int64_t)(v - nx - 1) * N] += d / o->sx[v - nx - 1];
        else if (v == 2 * nx + 1) grad[zb + zt] += d / o->st;
        else grad[zb + zt + 2 + (v - 2 * nx - 2)] += d / o->sa[v - 2 * nx - 2];
      }
    }
  }
  if (o->n_ph > 1) { /* events with the default consecutive links, mpopt.py:484-519 */
    int64_t row = o->off_ev;
    const int64_t zt = (int64_t)(nx + nu) * N;
    for (int blk = 0; blk < 3; ++blk)
      for (int l = 0; l + 1 < o->n_ph; ++l) {
        const int64_t zi = l * o->n_zp, zj = (l + 1) * o->n_zp;
        int cnt = blk == 0 ? nx : (blk == 1 ? nu : 1);
        for (int c = 0; c < cnt; ++c) {
          int64_t cj, ci;
          if (blk == 2) cj = zj + zt, ci = zi + zt + 1;
          else {
            int comp = blk == 0 ? c : nx + c;
            cj = zj + (int64_t)comp * N, ci = zi + (int64_t)comp * N + N - 1;
          }
          if (g) g[row] = z[cj] - z[ci];
          EMIT(row, cj, 1.0);
          EMIT(row, ci, -1.0);
          ++row;
        }
      }
  }
  *f_out = ftot;
  return q;
#undef EMIT
}


/* Upper bound on the number of triplets orc_hess emits. */
int64_t orc_hess_capacity(const orc* o) {
  const int ny = o->nx + o->nu + 2 + o->na, ntv = 2 * o->nx + 2 + o->na;
  return (int64_t)o->n_ph * ((int64_t)o->N * ny * (ny + 1) / 2 + (int64_t)ntv * (ntv + 1) / 2 + (int64_t)(2 + o->na) * (3 + o->na) / 2);
}

/* hess_l: upper triangle (row <= col) of the Hessian of  sigma*f + lam_g^T g  with respect to z, as COO triplets
 * (CasADi's nlp_hess_l, derived by ca.nlpsol at mpopt.py:757 from the dict built at mpopt.py:631).  A position may be
 * emitted more than once (a node entry and a terminal entry of the last node): duplicates add up.  Only structural
 * entries are emitted (masks of the problem).  Per node, with v = (x, u, t, a) the unscaled arguments, y = (X_i, U_i,
 * t0_var, tf_var, A) the NLP variables they come from (v linear in y, mpopt.py:175-177, 196-198) and
 *     phi_i = h * psi(v) + chi(v),   psi = sigma W_i L - sum_c lam_F[c] Sx_c dyn_c,   chi = sum_j lam_C[j] path_j,
 * h = (tf - t0) kap linear in y:   d2 phi = Jv^T (h psi'' + chi'') Jv + dh (Jv^T psi')^T + (Jv^T psi') dh^T.
 * Defect D.X terms, mid-point control rows and phase events are linear in z: no contribution.  Returns the count. */
int64_t orc_hess(const orc* o, const double* z, const double* p, double sigma, const double* lam, int32_t* rows, int32_t* cols,
                 double* vals) {
  const int N = o->N, nx = o->nx, nu = o->nu, na = o->na, nv = nx + nu + 1 + na, ntv = 2 * nx + 2 + na, ny = nx + nu + 2 + na;
  const int it = nx + nu; /* index of t in v; y: [0,nx) X, [nx,nx+nu) U, it -> t0, it+1 -> tf, it+2.. A */
  int64_t q = 0;
  for (int ph = 0; ph < o->n_ph; ++ph) {
    const ocp_fns* F = o->fn[ph];
    const double* zp = z + ph * o->n_zp;
    const double* X = zp;
    const double* U = zp + (int64_t)nx * N;
    const int64_t zt = (int64_t)(nx + nu) * N, zb = ph * o->n_zp;
    const double t0 = zp[zt] / o->st, tf = zp[zt + 1] / o->st;
    double a[MAXV];
    for (int c = 0; c < na; ++c) a[c] = zp[zt + 2 + c] / o->sa[c];
    const double* w = p + ph * o->S;
    const double dtau = o->tau1 - o->tau0;
    double t_seg0 = t0, h = (tf - t0) / dtau * w[0];
    /* (t0, tf, A) corner: summed over the nodes, emitted once */
    double corner[MAXV * MAXV] = {0};
    unsigned char scorner[MAXV * MAXV] = {0};
    const int ncn = 2 + na;
    /* structure of one node block (the same for every node) */
    unsigned char S2[MAXV * MAXV] = {0}, S1[MAXV] = {0};
    for (int r = 0; r < nv; ++r) {
      S1[r] = F->m1_L[r];
      for (int c = 0; c < nx; ++c) S1[r] |= F->m_dyn[c * nv + r];
      for (int s = 0; s < nv; ++s) S2[r * nv + s] = F->m2_psi[r * nv + s] | F->m2_chi[r * nv + s];
    }
    /* node-independent pieces: scale factors, v -> y map (t feeds t0 and tf), structure of the y-block */
    double jf[MAXV];
    for (int r = 0; r < nx; ++r) jf[r] = 1.0 / o->sx[r];
    for (int r = 0; r < nu; ++r) jf[nx + r] = 1.0 / o->su[r];
    for (int r = 0; r < na; ++r) jf[it + 1 + r] = 1.0 / o->sa[r];
    int ymap[MAXV][2], ycnt[MAXV];
    for (int r = 0; r < nv; ++r) {
      if (r < it) ymap[r][0] = r, ycnt[r] = 1;
      else if (r == it) ymap[r][0] = it, ymap[r][1] = it + 1, ycnt[r] = 2;
      else ymap[r][0] = r + 1, ycnt[r] = 1;
    }
    unsigned char Sy[MAXV * MAXV] = {0}, sgy[MAXV] = {0};
    for (int r = 0; r < nv; ++r)
      for (int m = 0; m < ycnt[r]; ++m) sgy[ymap[r][m]] |= S1[r];
    for (int r = 0; r < nv; ++r)
      for (int c = 0; c < nv; ++c)
        for (int m = 0; m < ycnt[r]; ++m)
          for (int n = 0; n < ycnt[c]; ++n) Sy[ymap[r][m] * ny + ymap[c][n]] |= S2[r * nv + c];
    for (int m = 0; m < ny; ++m)
      for (int n = 0; n < ny; ++n) Sy[m * ny + n] |= ((m == it || m == it + 1) && sgy[n]) | ((n == it || n == it + 1) && sgy[m]);
    const deg_table* T = 0;
    for (int i = 0, s = 0; i < N; ++i) {
      if (o->seg[i] != s) {
        s = o->seg[i];
        t_seg0 += h * dtau;
        h = (tf - t0) / dtau * w[s];
      }
      const int pdeg = o->orders[s], k = o->pt[i];
      if (!T || T->deg != pdeg) T = tab(o, pdeg);
      const double t = t_seg0 + h * (T->tau[k] - o->tau0);
      const double theta = (t - t0) / (tf - t0), kap = w[s] / dtau, W = o->compW[i];
      double x[MAXV], u[MAXV], dyn[MAXV], pc[MAXV], L, ddyn[MAXV * MAXV], dpc[MAXV * MAXV], dL[MAXV];
      for (int c = 0; c < nx; ++c) x[c] = X[(int64_t)c * N + i] / o->sx[c];
      for (int c = 0; c < nu; ++c) u[c] = U[(int64_t)c * N + i] / o->su[c];
      F->node(x, u, t, a, dyn, pc, &L);
      F->node_d(x, u, t, a, ddyn, dpc, dL);
      double wd[MAXV], wc[MAXV];
      for (int c = 0; c < nx; ++c) wd[c] = -lam[o->off_F[ph] + (int64_t)c * N + i] * o->sx[c];
      for (int j = 0; j < F->nc; ++j) wc[j] = lam[o->off_C[ph] + (int64_t)j * N + i];
      const double wL = sigma * W;
      double Hp[MAXV * MAXV], Hc[MAXV * MAXV], g1[MAXV], Hy[MAXV * MAXV], gy[MAXV], yfac[MAXV][2];
      memset(Hp, 0, nv * nv * sizeof(double));
      memset(Hc, 0, nv * nv * sizeof(double));
      memset(Hy, 0, ny * ny * sizeof(double));
      memset(gy, 0, ny * sizeof(double));
      F->node_dd(x, u, t, a, wd, wc, wL, Hp, Hc);
      for (int r = 0; r < nv; ++r) {
        double v = F->m1_L[r] ? wL * dL[r] : 0.0;
        for (int c = 0; c < nx; ++c)
          if (F->m_dyn[c * nv + r]) v += wd[c] * ddyn[c * nv + r];
        g1[r] = v; /* psi' */
        yfac[r][0] = jf[r];
      }
      yfac[it][0] = (1 - theta) / o->st, yfac[it][1] = theta / o->st;
      for (int r = 0; r < nv; ++r)
        for (int m = 0; m < ycnt[r]; ++m) gy[ymap[r][m]] += g1[r] * yfac[r][m]; /* Jv^T psi' */
      for (int r = 0; r < nv; ++r)
        for (int c = 0; c < nv; ++c) {
          if (!S2[r * nv + c]) continue;
          const double v2 = h * Hp[r * nv + c] + Hc[r * nv + c];
          for (int m = 0; m < ycnt[r]; ++m)
            for (int n = 0; n < ycnt[c]; ++n) Hy[ymap[r][m] * ny + ymap[c][n]] += v2 * yfac[r][m] * yfac[c][n];
        }
      const double dh0 = -kap / o->st, dh1 = kap / o->st; /* dh / d(t0_var, tf_var) */
      for (int n = 0; n < ny; ++n) {
        Hy[it * ny + n] += dh0 * gy[n], Hy[(it + 1) * ny + n] += dh1 * gy[n];
        Hy[n * ny + it] += gy[n] * dh0, Hy[n * ny + it + 1] += gy[n] * dh1;
      }
      /* emit the upper triangle; pairs inside (t0, tf, A) go to the corner */
      for (int m = 0; m < ny; ++m)
        for (int n = m; n < ny; ++n) {
          if (!Sy[m * ny + n]) continue;
          if (m >= it) {
            corner[(m - it) * ncn + (n - it)] += Hy[m * ny + n];
            scorner[(m - it) * ncn + (n - it)] = 1;
            continue;
          }
          const int64_t gm = zb + (int64_t)m * N + i;
          const int64_t gn = n < it ? zb + (int64_t)n * N + i : zb + zt + (n - it);
          rows[q] = (int32_t)gm, cols[q] = (int32_t)gn, vals[q++] = Hy[m * ny + n];
        }
    }
    for (int m = 0; m < ncn; ++m)
      for (int n = m; n < ncn; ++n)
        if (scorner[m * ncn + n]) rows[q] = (int32_t)(zb + zt + m), cols[q] = (int32_t)(zb + zt + n), vals[q++] = corner[m * ncn + n];
    /* terminal cost / constraints: w = (xf, tf, x0, t0, a), each a scaled NLP variable (mpopt.py:277-298) */
    double xf[MAXV], x0[MAXV], wt[MAXV], Hw[4 * MAXV * MAXV] = {0};
    for (int c = 0; c < nx; ++c) xf[c] = X[(int64_t)c * N + N - 1] / o->sx[c], x0[c] = X[(int64_t)c * N] / o->sx[c];
    for (int j = 0; j < F->ntc; ++j) wt[j] = lam[o->off_TC[ph] + j];
    F->term_dd(xf, tf, x0, t0, a, sigma, wt, Hw);
    int64_t gidx[2 * MAXV + 2];
    double gfac[2 * MAXV + 2];
    for (int v = 0; v < ntv; ++v) {
      if (v < nx) gidx[v] = zb + (int64_t)v * N + N - 1, gfac[v] = 1.0 / o->sx[v];
      else if (v == nx) gidx[v] = zb + zt + 1, gfac[v] = 1.0 / o->st;
      else if (v < 2 * nx + 1) gidx[v] = zb + (int64_t)(v - nx - 1) * N, gfac[v] = 1.0 / o->sx[v - nx - 1];
      else if (v == 2 * nx + 1) gidx[v] = zb + zt, gfac[v] = 1.0 / o->st;
      else gidx[v] = zb + zt + 2 + (v - 2 * nx - 2), gfac[v] = 1.0 / o->sa[v - 2 * nx - 2];
    }
    for (int m = 0; m < ntv; ++m)
      for (int n = m; n < ntv; ++n) {
        if (!(F->m2_term[m * ntv + n] | F->m2_term[n * ntv + m])) continue;
        const double v = Hw[m * ntv + n] * gfac[m] * gfac[n];
        int64_t r = gidx[m], c = gidx[n];
        if (r > c) { int64_t tmp = r; r = c; c = tmp; }
        rows[q] = (int32_t)r, cols[q] = (int32_t)c, vals[q++] = v;
      }
  }
  return q;
}

/* All-host-cores variant of the timed loop: OpenMP over evaluation points, private output buffers per
 * thread (values are discarded; the single-core loop above is the like-for-like stand-in for CasADi's
 * serial SX interpreter, this one is the "whole CPU" number).  Returns the thread count used. */
#ifdef _OPENMP
#include <omp.h>
#endif
int orc_eval_many_omp(const orc* o, int64_t n_points, int reps, const double* Z, const double* p, double* f) {
  int nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel
  {
#pragma omp single
    nthreads = omp_get_num_threads();
    double* g = (double*)malloc(o->n_g * sizeof(double));
    double* grad = (double*)malloc(o->n_z * sizeof(double));
    double* vals = (double*)malloc(o->nnz * sizeof(double));
    for (int r = 0; r < reps; ++r) {
#pragma omp for schedule(static)
      for (int64_t b = 0; b < n_points; ++b) orc_eval(o, Z + b * o->n_z, p, f + b, g, grad, 0, 0, vals);
    }
    free(g), free(grad), free(vals);
  }
#else
  double* g = (double*)malloc(o->n_g * sizeof(double));
  double* grad = (double*)malloc(o->n_z * sizeof(double));
  double* vals = (double*)malloc(o->nnz * sizeof(double));
  for (int r = 0; r < reps; ++r)
    for (int64_t b = 0; b < n_points; ++b) orc_eval(o, Z + b * o->n_z, p, f + b, g, grad, 0, 0, vals);
  free(g), free(grad), free(vals);
#endif
  return nthreads;
}

/* nlp_grad (the sixth oracle ca.nlpsol derives at mpopt.py:757): gradient of gamma = sigma*f + lam_g^T g with respect to z
 * (ggx[n_z] = sigma grad_f + J^T lam_g, accumulated over the hand-derived triplets of orc_eval) and with respect to the
 * parameters p (ggp[n_ph*S]).  A width w_s enters node i of segment s through h = (tf - t0)/dtau * w_s (mpopt.py:184) and
 * through t = t_seg0 + h (tau_k - tau0), t_seg0 = t0 + sum_{r<s} (tf - t0) w_r (mpopt.py:192-198):
 *     d gamma_i / d w_s = d gamma_i/dh * (tf - t0)/dtau + d gamma_i/dt * (tf - t0)/dtau * (tau_k - tau0)     (own segment)
 *     d gamma_i / d w_r = d gamma_i/dt * (tf - t0)                                                          (every r < s)
 * with gamma_i = -sum_c lam_F[c,i] h Sx_c dyn_c + sum_j lam_C[j,i] pc_j + sigma W_i h L in the unscaled arguments. */
int orc_grad_gamma(const orc* o, const double* z, const double* p, double sigma, const double* lam, double* ggx, double* ggp) {
  const int N = o->N, nx = o->nx, nu = o->nu, na = o->na, nv = nx + nu + 1 + na;
  if (ggx) {
    double f, *grad = (double*)malloc(o->n_z * sizeof(double)), *vals = (double*)malloc((o->nnz + 1) * sizeof(double));
    int32_t *rows = (int32_t*)malloc((o->nnz + 1) * sizeof(int32_t)), *cols = (int32_t*)malloc((o->nnz + 1) * sizeof(int32_t));
    const int64_t n = orc_eval(o, z, p, &f, 0, grad, rows, cols, vals);
    for (int64_t c = 0; c < o->n_z; ++c) ggx[c] = sigma * grad[c];
    for (int64_t e = 0; e < n; ++e) ggx[cols[e]] += lam[rows[e]] * vals[e];
    free(grad), free(vals), free(rows), free(cols);
  }
  if (!ggp) return 0;
  double* later = (double*)malloc(o->S * sizeof(double)); /* sum over the nodes of segment s of d gamma_i/dt * (tf - t0) */
  for (int ph = 0; ph < o->n_ph; ++ph) {
    const ocp_fns* F = o->fn[ph];
    const double* zp = z + ph * o->n_zp;
    const double *X = zp, *U = zp + (int64_t)nx * N;
    const int64_t zt = (int64_t)(nx + nu) * N;
    const double t0 = zp[zt] / o->st, tf = zp[zt + 1] / o->st;
    double a[MAXV];
    for (int c = 0; c < na; ++c) a[c] = zp[zt + 2 + c] / o->sa[c];
    const double* w = p + ph * o->S;
    double* out = ggp + ph * o->S;
    const double dtau = o->tau1 - o->tau0;
    for (int s = 0; s < o->S; ++s) out[s] = 0, later[s] = 0;
    double t_seg0 = t0, h = (tf - t0) / dtau * w[0];
    for (int i = 0, s = 0; i < N; ++i) {
      if (o->seg[i] != s) {
        s = o->seg[i];
        t_seg0 += h * dtau;
        h = (tf - t0) / dtau * w[s];
      }
      const deg_table* T = tab(o, o->orders[s]);
      const double tk = T->tau[o->pt[i]] - o->tau0, t = t_seg0 + h * tk;
      double x[MAXV], u[MAXV], dyn[MAXV], pc[MAXV], L, ddyn[MAXV * MAXV], dpc[MAXV * MAXV], dL[MAXV];
      for (int c = 0; c < nx; ++c) x[c] = X[(int64_t)c * N + i] / o->sx[c];
      for (int c = 0; c < nu; ++c) u[c] = U[(int64_t)c * N + i] / o->su[c];
      F->node(x, u, t, a, dyn, pc, &L);
      F->node_d(x, u, t, a, ddyn, dpc, dL);
      const int vt = nx + nu; /* index of t among the arguments */
      double gh = sigma * o->compW[i] * L, gt = sigma * o->compW[i] * h * dL[vt];
      for (int c = 0; c < nx; ++c) {
        const double lf = lam[o->off_F[ph] + (int64_t)c * N + i];
        gh -= lf * o->sx[c] * dyn[c];
        gt -= lf * h * o->sx[c] * ddyn[c * nv + vt];
      }
      for (int j = 0; j < F->nc; ++j) gt += lam[o->off_C[ph] + (int64_t)j * N + i] * dpc[j * nv + vt];
      out[s] += gh * (tf - t0) / dtau + gt * (tf - t0) / dtau * tk;
      later[s] += gt * (tf - t0);
    }
    double run = 0;
    for (int s = o->S - 1; s >= 0; --s) {
      out[s] += run;
      run += later[s];
    }
  }
  free(later);
  return 0;
}

/* Timed loop for bench.py: `reps` evaluations of f+g+grad_f+jac_g (values only) on `n_points`
 * different points; returns nothing, the caller clocks it. */
void orc_eval_many(const orc* o, int64_t n_points, int reps, const double* Z, const double* p, double* f, double* g, double* grad,
                   double* vals) {
  for (int r = 0; r < reps; ++r)
    for (int64_t b = 0; b < n_points; ++b) orc_eval(o, Z + b * o->n_z, p, f + b, g, grad, 0, 0, vals);
}

/* Timed loop for the "oracle time per IPOPT iteration" figure: per pass and point, the reference's recorded call mix
 * (docs/source/notebooks/moon_lander.ipynb:192-198): n_g calls of nlp_g, then nlp_grad_f, nlp_jac_g, nlp_hess_l once each,
 * as separate functions like CasADi's.  The caller clocks it. */
void orc_ipopt_mix(const orc* o, int64_t n_points, int reps, int n_g_calls, const double* Z, const double* p, double sigma,
                   const double* lam, double* g, double* grad, double* vals, int32_t* hr, int32_t* hc, double* hv) {
  double f;
  for (int r = 0; r < reps; ++r)
    for (int64_t b = 0; b < n_points; ++b) {
      const double* z = Z + b * o->n_z;
      for (int k = 0; k < n_g_calls; ++k) orc_eval(o, z, p, &f, g, 0, 0, 0, 0);
      orc_eval(o, z, p, &f, 0, grad, 0, 0, 0);
      orc_eval(o, z, p, &f, g, 0, 0, 0, vals);
      orc_hess(o, z, p, sigma, lam, hr, hc, hv);
    }
}

/* One oracle function at a time, as CasADi exposes them: which = 0 nlp_g, 1 nlp_grad_f (f + grad_f), 2 nlp_jac_g (g + values),
 * 3 nlp_hess_l, 4 nlp_f.  `reps` passes over `n_points` points; the caller clocks it. */
void orc_eval_fn(const orc* o, int which, int64_t n_points, int reps, const double* Z, const double* p, double sigma, const double* lam,
                 double* g, double* grad, double* vals, int32_t* hr, int32_t* hc, double* hv) {
  double f;
  for (int r = 0; r < reps; ++r)
    for (int64_t b = 0; b < n_points; ++b) {
      const double* z = Z + b * o->n_z;
      switch (which) {
        case 0: orc_eval(o, z, p, &f, g, 0, 0, 0, 0); break;
        case 1: orc_eval(o, z, p, &f, 0, grad, 0, 0, 0); break;
        case 2: orc_eval(o, z, p, &f, g, 0, 0, 0, vals); break;
        case 3: orc_hess(o, z, p, sigma, lam, hr, hc, hv); break;
        default: orc_eval(o, z, p, &f, 0, 0, 0, 0, 0); break;
      }
    }
}
