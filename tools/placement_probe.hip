// Scratch: does write bandwidth depend on WHICH allocation the buffer lives in?  Repeated hipMalloc/hipFree of a 4.7 GB
// buffer in one process, same kernel each time (64 KB contiguous chunk per workgroup, XCD-blocked) + hipMemsetAsync.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void chunk_store(double* out, long n_chunk_d2) {
  const unsigned lin = blockIdx.x, tot = gridDim.x, x = lin % 8, q = tot / 8, r = tot % 8;
  const unsigned item = x * q + (x < r ? x : r) + lin / 8;
  double2* o = (double2*)out + (long)item * n_chunk_d2;
  for (long i = threadIdx.x; i < n_chunk_d2; i += 256) o[i] = make_double2(1.0, 2.0);
}
int main(int argc, char** argv) {
  const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 4.7) * (1ull << 30);
  const long d2 = 64 * 1024 / 16, chunks = bytes / (64 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<void*> kept;
  for (int k = 0; k < 12; ++k) {
    double* out; CHK(hipMalloc(&out, bytes));
    for (int i = 0; i < 3; ++i) chunk_store<<<(unsigned)chunks, 256>>>(out, d2);
    CHK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) chunk_store<<<(unsigned)chunks, 256>>>(out, d2);
    hipEventRecord(e1); CHK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipMemsetAsync(out, 0, bytes);
    hipEventRecord(e1); CHK(hipEventSynchronize(e1));
    float ms2; hipEventElapsedTime(&ms2, e0, e1);
    printf("alloc %2d ptr %p  chunk store %6.2f TB/s   memset %6.2f TB/s\n", k, (void*)out, (double)chunks * 65536 * 10 / (ms * 1e-3) / 1e12, (double)bytes * 10 / (ms2 * 1e-3) / 1e12);
    if (k % 3 == 1) kept.push_back(out); else CHK(hipFree(out));
  }
  return 0;
}
