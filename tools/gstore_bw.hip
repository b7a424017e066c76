// Scratch: do the 8-byte-per-lane g / grad_f stores of the node kernel cost more than their share of the bytes?  Write-only model
// of the headline kernel's store mix (config 2): per tile (250 lanes) and evaluation point 12 pair stores of the Jacobian block
// (16 B per lane, tile-major) + 6 row stores (g: 3 rows, grad_f: 3 rows; row r of point b at rows + b * RS + r * N + node), one point
// per workgroup, XCD-blocked mapping.  Row-store variants: 0 = 8 B per lane (today), 1 = even lanes store 16 B (own + right
// neighbour), 2 = 16 B per lane with lanes [0,128) on row r and [128,256) on row r+1 (needs an LDS transpose in the real kernel),
// 3 = no row stores at all (how much they cost in total).
// hipcc --offload-arch=gfx950 -O3 -o gstore_bw gstore_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256) void mix(double* jac, double* rows, int n, int tiles, long jstride, long rstride, int N) {
  const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, tot = gridDim.x * gridDim.y;
  const unsigned x = lin % 8, q = tot / 8, r = tot % 8;
  const unsigned item = x * q + (x < r ? x : r) + lin / 8;
  const int t = item % gridDim.x, b = item / gridDim.x, l = threadIdx.x;
  double v = 1.0 + l + b;
  d2* jb = (d2*)(jac + (long)b * jstride + (long)t * 24 * n);
  if (l < n) {
#pragma unroll
    for (int p = 0; p < 12; ++p) jb[(long)p * n + l] = d2{v + p, v - p};
  }
  double* rb = rows + (long)b * rstride + (long)t * n;
  if (V == 0) {
    if (l < n) {
#pragma unroll
      for (int k = 0; k < 6; ++k) rb[(long)k * N + l] = v + k;
    }
  } else if (V == 1) {
    if (l < n && !(l & 1)) {
#pragma unroll
      for (int k = 0; k < 6; ++k) *(d2*)&rb[(long)k * N + l] = d2{v + k, v + k + 1};
    }
  } else if (V == 2) {
    const int half = l >> 7, ll = (l & 127) * 2;
    if (ll < n) {
#pragma unroll
      for (int k = 0; k < 6; k += 2) *(d2*)&rb[(long)(k + half) * N + ll] = d2{v + k, v + k + 1};
    }
  }
}

int main() {
  const int n = 250, tiles = 20, B = 4096, N = n * tiles + 2;  // N even: 16-byte alignment of every row start holds for t * n + even l
  const long jstride = (long)tiles * 24 * n, rstride = (long)6 * N;
  double *jac, *rows;
  CHK(hipMalloc(&jac, jstride * B * 8)); CHK(hipMalloc(&rows, rstride * B * 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 g(tiles, B);
  const double jb = (double)jstride * B * 8, rbts = (double)tiles * n * 6 * B * 8;
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-44s %8.1f us  %7.1f GB/s\n", name, best * 100, bytes / (best / 10 * 1e-3) / 1e9);
  };
  for (int round = 0; round < 2; ++round) {
    run("jac pairs only", [&] { mix<3><<<g, 256>>>(jac, rows, n, tiles, jstride, rstride, N); }, jb);
    run("jac + rows 8 B per lane (today)", [&] { mix<0><<<g, 256>>>(jac, rows, n, tiles, jstride, rstride, N); }, jb + rbts);
    run("jac + rows 16 B on even lanes", [&] { mix<1><<<g, 256>>>(jac, rows, n, tiles, jstride, rstride, N); }, jb + rbts);
    run("jac + rows 16 B per lane, two rows per store", [&] { mix<2><<<g, 256>>>(jac, rows, n, tiles, jstride, rstride, N); }, jb + rbts);
  }
  return 0;
}
