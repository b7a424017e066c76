// Probe of global_load_lds_dwordx4 on gfx950: (a) 8-byte-aligned (not 16) global sources, (b) EXEC-masked lanes leave their LDS
// slots untouched, (c) destination = uniform base + lane * 16.  hipcc --offload-arch=gfx950 -O3 tools/glds_probe.hip -o tools/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
__global__ void probe(const double* __restrict__ src, double* __restrict__ out, int shift, int active_pairs) {
  __shared__ double buf[256];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) buf[i] = -1.0;
  __syncthreads();
  if (l < active_pairs) __builtin_amdgcn_global_load_lds((gbl_void*)(src + shift + 2 * l), (lds_void*)&buf[0], 16, 0, 0);
  if (l < active_pairs) __builtin_amdgcn_global_load_lds((gbl_void*)(src + shift + 128 + 2 * l), (lds_void*)&buf[128], 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = l; i < 256; i += 64) out[i] = buf[i];
}
int main() {
  std::vector<double> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i;
  double *d, *o;
  hipMalloc(&d, 8192), hipMalloc(&o, 2048);
  hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
  for (int shift : {0, 1, 3}) for (int ap : {64, 37}) {
    probe<<<1, 64>>>(d, o, shift, ap);
    std::vector<double> r(256);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) {
      const int j = i & 127;
      const double want = j < 2 * ap ? shift + i : -1.0;
      if (r[i] != want) { if (bad < 4) printf("  shift %d ap %d: [%d] = %g want %g\n", shift, ap, i, r[i], want); ++bad; }
    }
    printf("shift %d active pairs %d: %s (%d wrong)\n", shift, ap, bad ? "MISMATCH" : "ok", bad);
  }
  return 0;
}
