// mpx_colloc.cpp -- collocation node sets, differentiation / interpolation matrices and
// quadrature weights (host side, tiny, once per distinct degree).
//
// Restates the *semantics* of the reference's CollocationRoots (mpopt.py:4134-4276) and of
// Collocation with D_MATRIX_METHOD="numerical" (mpopt.py:3815-3905), not its arithmetic:
// the reference multiplies out np.poly1d coefficient vectors (mpopt.py:4006-4011) and calls
// scipy's Golub-Welsch (mpopt.py:4220, 4246); here nodes come from a deflated Newton iteration
// on the Jacobi three-term recurrence in extended precision, D from barycentric weights,
// off-node matrices from exact re-interpolation and weights from Gauss-Legendre quadrature of
// the Lagrange basis -- all of which stay accurate at degrees where coefficient products lose
// digits (SURVEY.md section 7 "High degree").
#include <algorithm>
#include <cmath>
#include <vector>

#include "mpx.h"

namespace {
typedef long double real;

// value and derivative of the Jacobi polynomial P_n^{(a,b)} at x
void jacobi_eval(int n, real a, real b, real x, real& p, real& dp) {
  if (n == 0) {
    p = 1;
    dp = 0;
    return;
  }
  real p0 = 1, p1 = (a - b) / 2 + (a + b + 2) * x / 2;
  for (int k = 2; k <= n; ++k) {
    real k2 = 2 * (real)k + a + b;
    real c1 = 2 * k * (k + a + b) * (k2 - 2);
    real c2 = (k2 - 1) * (k2 * (k2 - 2) * x + a * a - b * b);
    real c3 = 2 * (k + a - 1) * (k + b - 1) * k2;
    real pn = (c2 * p1 - c3 * p0) / c1;
    p0 = p1;
    p1 = pn;
  }
  p = p1;
  // (2n+a+b)(1-x^2) P_n' = n[a-b-(2n+a+b)x] P_n + 2(n+a)(n+b) P_{n-1}
  real n2 = 2 * (real)n + a + b;
  real den = n2 * (1 - x * x);
  dp = (n * (a - b - n2 * x) * p1 + 2 * (n + a) * (n + b) * p0) / den;
}

// roots of P_n^{(a,b)} in ascending order
std::vector<real> jacobi_roots(int n, real a, real b) {
  std::vector<real> r(n);
  const real pi = acosl(-1.0L);
  for (int k = 0; k < n; ++k) {
    // asymptotic initial guess, descending in k -> we fill from the right
    real th = pi * (4 * (real)(k + 1) - 1 + 2 * a) / (4 * (real)n + 2 * (a + b + 1));
    real x = cosl(th);
    for (int it = 0; it < 100; ++it) {
      real p, dp;
      jacobi_eval(n, a, b, x, p, dp);
      real s = 0;
      for (int j = 0; j < k; ++j) s += 1 / (x - r[j]);
      real dx = p / (dp - p * s);
      x -= dx;
      if (fabsl(dx) < 1e-19L * (1 + fabsl(x))) break;
    }
    r[k] = x;
  }
  for (int k = 0; k < n; ++k) {  // polish without deflation
    for (int it = 0; it < 3; ++it) {
      real p, dp;
      jacobi_eval(n, a, b, r[k], p, dp);
      r[k] -= p / dp;
    }
  }
  std::sort(r.begin(), r.end());
  return r;
}

real lagrange(const double* x, int n, int j, real t) {
  real v = 1;
  for (int m = 0; m < n; ++m)
    if (m != j) v *= (t - (real)x[m]) / ((real)x[j] - (real)x[m]);
  return v;
}

// first-derivative matrix at the nodes via barycentric weights
void diff_at_nodes(const double* x, int n, std::vector<real>& D) {
  std::vector<real> lam(n, 1);
  for (int j = 0; j < n; ++j)
    for (int m = 0; m < n; ++m)
      if (m != j) lam[j] /= ((real)x[j] - (real)x[m]);
  D.assign((size_t)n * n, 0);
  for (int i = 0; i < n; ++i) {
    real s = 0;
    for (int j = 0; j < n; ++j) {
      if (i == j) continue;
      real v = (lam[j] / lam[i]) / ((real)x[i] - (real)x[j]);
      D[(size_t)i * n + j] = v;
      s += v;
    }
    D[(size_t)i * n + i] = -s;
  }
}
}  // namespace

extern "C" int mpx_colloc_n_nodes(int scheme, int deg) {
  if (deg < 0) return MPX_ERR_INVALID;
  switch (scheme) {
    case MPX_SCHEME_LGR:
    case MPX_SCHEME_LGL:
      return deg == 0 ? 1 : deg + 1;
    case MPX_SCHEME_CGL:
      return deg + 1;
    case MPX_SCHEME_LG:
      return deg < 2 ? MPX_ERR_INVALID : deg;  // the reference raises for deg < 2 (leggauss(0))
    case MPX_SCHEME_EQUI:
      return deg > 1 ? deg : 2;
    default:
      return MPX_ERR_INVALID;
  }
}

extern "C" int mpx_colloc_roots(int scheme, int deg, double tmin, double tmax, double* out) {
  int n = mpx_colloc_n_nodes(scheme, deg);
  if (n < 0 || !out) return MPX_ERR_INVALID;
  auto map = [&](double r) { return tmin + (tmax - tmin) / 2 * (r + 1); };  // mpopt.py:4224
  switch (scheme) {
    case MPX_SCHEME_LGR:
    case MPX_SCHEME_LGL: {
      if (deg == 0) {
        out[0] = 0.0;  // mpopt.py:4229
        return MPX_OK;
      }
      if (deg == 1) {
        out[0] = tmin;  // mpopt.py:4227 (unmapped end points)
        out[1] = tmax;
        return MPX_OK;
      }
      std::vector<real> r = jacobi_roots(deg - 1, 1, scheme == MPX_SCHEME_LGR ? 0 : 1);
      out[0] = map(-1.0);
      for (int k = 0; k < deg - 1; ++k) out[k + 1] = map((double)r[k]);
      out[deg] = map(1.0);
      return MPX_OK;
    }
    case MPX_SCHEME_CGL: {
      for (int j = 0; j <= deg; ++j) {  // np.cos(np.pi*j/deg)[::-1]  (mpopt.py:4271)
        int jj = deg - j;
        double c = deg == 0 ? cos(M_PI * 0.0) : cos(M_PI * (double)jj / (double)deg);
        out[j] = map(c);
      }
      return MPX_OK;
    }
    case MPX_SCHEME_LG: {
      std::vector<real> r = jacobi_roots(deg - 1, 0, 0);
      out[0] = map(-1.0);
      for (int k = 0; k < deg - 1; ++k) out[k + 1] = map((double)r[k]);
      return MPX_OK;
    }
    case MPX_SCHEME_EQUI: {
      if (deg > 1) {
        for (int k = 0; k < deg; ++k) out[k] = tmin + (tmax - tmin) * (double)k / (double)(deg - 1);
        out[deg - 1] = tmax;
      } else {
        out[0] = tmin;
        out[1] = tmax;
      }
      return MPX_OK;
    }
  }
  return MPX_ERR_INVALID;
}

extern "C" int mpx_colloc_interp_matrix(const double* x, int n, const double* taus, int nt, double* C) {
  if (!x || !taus || !C || n < 1 || nt < 0) return MPX_ERR_INVALID;
  for (int i = 0; i < nt; ++i)
    for (int j = 0; j < n; ++j) C[(size_t)i * n + j] = (double)lagrange(x, n, j, (real)taus[i]);
  return MPX_OK;
}

extern "C" int mpx_colloc_diff_matrix(const double* x, int n, const double* taus, int nt, int order, double* D) {
  if (!x || !D || n < 1 || (order != 1 && order != 2)) return MPX_ERR_INVALID;
  std::vector<real> D1;
  diff_at_nodes(x, n, D1);
  std::vector<real> Dk = D1;
  if (order == 2) {  // l_j'' at the nodes = (D1*D1)[i][j]: l_j' has degree < n and is re-interpolated exactly
    Dk.assign((size_t)n * n, 0);
    for (int i = 0; i < n; ++i)
      for (int m = 0; m < n; ++m) {
        real a = D1[(size_t)i * n + m];
        for (int j = 0; j < n; ++j) Dk[(size_t)i * n + j] += a * D1[(size_t)m * n + j];
      }
  }
  if (!taus) {
    for (size_t k = 0; k < (size_t)n * n; ++k) D[k] = (double)Dk[k];
    return MPX_OK;
  }
  // off-node: l_j^(k)(tau) = sum_m l_m(tau) * l_j^(k)(x_m)
  for (int i = 0; i < nt; ++i) {
    std::vector<real> c(n);
    for (int m = 0; m < n; ++m) c[m] = lagrange(x, n, m, (real)taus[i]);
    for (int j = 0; j < n; ++j) {
      real s = 0;
      for (int m = 0; m < n; ++m) s += c[m] * Dk[(size_t)m * n + j];
      D[(size_t)i * n + j] = (double)s;
    }
  }
  return MPX_OK;
}

extern "C" int mpx_colloc_quad_weights(const double* x, int n, double a, double b, double* w) {
  if (!x || !w || n < 1) return MPX_ERR_INVALID;
  int nq = n / 2 + 2;  // exact for degree 2*nq-1 >= n-1
  std::vector<real> gx = jacobi_roots(nq, 0, 0), gw(nq);
  for (int k = 0; k < nq; ++k) {
    real p, dp;
    jacobi_eval(nq, 0, 0, gx[k], p, dp);
    gw[k] = 2 / ((1 - gx[k] * gx[k]) * dp * dp);
  }
  real half = ((real)b - (real)a) / 2, mid = ((real)b + (real)a) / 2;
  for (int j = 0; j < n; ++j) {
    real s = 0;
    for (int k = 0; k < nq; ++k) s += gw[k] * lagrange(x, n, j, mid + half * gx[k]);
    w[j] = (double)(s * half);
  }
  return MPX_OK;
}
