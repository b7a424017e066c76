// Probe of v_mfma_f64_16x16x4f64 on gfx950: operand / result layout and the order of its four accumulations (is
// D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,C))))?), what mpx_light_* relies on.   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D, int chain) {  // A[16][4*chain], B[4*chain][16], D[16][16]
  const int l = threadIdx.x, n = l % 16, q = l / 16;
  d4 acc = {0, 0, 0, 0};
  for (int ks = 0; ks < chain; ++ks) {
    const double a = A[(l % 16) * (4 * chain) + 4 * ks + q];  // A[i = l%16][k = 4ks + l/16]
    const double b = B[(4 * ks + q) * 16 + n];                // B[k = 4ks + l/16][j = l%16]
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];  // raw: [lane][register]; main() finds the (i, j) every slot holds
}
int main() {
  const int chain = 8, K = 4 * chain;
  double hA[16 * K], hB[K * 16], hD[256], ref_fwd[256], ref_tree[256];
  srand(1);
  for (auto& v : hA) v = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 40) - 20.0);
  for (auto& v : hB) v = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 40) - 20.0);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int k = 0; k < K; ++k) s = fma(hA[i * K + k], hB[k * 16 + j], s);
      ref_fwd[i * 16 + j] = s;
      double t = 0;  // per instruction: pairwise inside, then added
      for (int ks = 0; ks < chain; ++ks) {
        double p = 0;
        for (int k = 4 * ks; k < 4 * ks + 4; ++k) p = fma(hA[i * K + k], hB[k * 16 + j], p);
        t += p;
      }
      ref_tree[i * 16 + j] = t;
    }
  double *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA), hipMalloc(&dB, sizeof hB), hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice), hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, chain);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int same_fwd = 0, same_tree = 0, close = 0;
  double worst = 0;
  int fi[256], fj[256];
  for (int e = 0; e < 256; ++e) {  // which entry of the product does slot (lane, register) hold?
    fi[e] = fj[e] = -1;
    for (int m = 0; m < 256; ++m)
      if (fabs(hD[e] - ref_fwd[m]) <= 1e-12 * fabs(ref_fwd[m])) fi[e] = m / 16, fj[e] = m % 16, same_fwd += hD[e] == ref_fwd[m], same_tree += hD[e] == ref_tree[m];
    close += fi[e] >= 0;
  }
  int f1 = 1, f2 = 1;  // candidate formulas
  for (int e = 0; e < 256; ++e) {
    const int l = e / 4, r = e % 4;
    f1 = f1 && fi[e] == 4 * (l / 16) + r && fj[e] == l % 16;
    f2 = f2 && fi[e] == (l / 16) + 4 * r && fj[e] == l % 16;
  }
  printf("layout: D[4*(lane/16) + r][lane%%16]: %s;  D[(lane/16) + 4*r][lane%%16]: %s\n", f1 ? "yes" : "no", f2 ? "yes" : "no");
  for (int l = 0; l < 64; l += 15) printf("  lane %2d: (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, fi[4*l], fj[4*l], fi[4*l+1], fj[4*l+1], fi[4*l+2], fj[4*l+2], fi[4*l+3], fj[4*l+3]);
  printf("mfma_f64_16x16x4: layout ok (rel < 1e-12) for %d / 256 entries, worst rel %.2e; bit-equal to the sequential fma chain: %d / 256; to the per-instruction partial sums: %d / 256\n",
         close, worst, same_fwd, same_tree);
  return close == 256 ? 0 : 1;
}
