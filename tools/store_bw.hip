// Scratch microbenchmark: write-only streaming patterns resembling the jac_g scatter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// V=0: 8B/lane plain, V=1: 8B nontemporal, V=2: 16B/lane plain, V=3: 16B nontemporal
// MAP=1: XCD-blocked (linear workgroup id mod 8 = XCD; every XCD walks a contiguous range of (chunk, tile) items)
template <int V, int SLOTS, int MAP = 0>
__global__ __launch_bounds__(256) void tile_store(double* out, int n, int tiles, int B, int bpb, long stride) {
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (MAP == 1) {
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, tot = gridDim.x * gridDim.y;
    const unsigned x = lin % 8, q = tot / 8, r = tot % 8;
    const unsigned item = x * q + (x < r ? x : r) + lin / 8;
    bx = item % gridDim.x, by = item / gridDim.x;
  }
  const int t = bx, l = threadIdx.x;
  const int b0 = by * bpb, b1 = min(B, b0 + bpb);
  double v = 1.0 + l;
  for (int b = b0; b < b1; ++b) {
    double* jb = out + (long)b * stride + (long)t * SLOTS * n;
    if (V < 2) {
      if (l < n) {
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
          if (V == 0) jb[(long)q * n + l] = v + q; else __builtin_nontemporal_store(v + q, &jb[(long)q * n + l]);
        }
      }
    } else {
      // two slots per instruction: lanes [0,128) slot q, lanes [128,256) slot q+1, 16B each
      const int half = l >> 7, ll = (l & 127) * 2;
      if (ll < n) {
#pragma unroll
        for (int q = 0; q < SLOTS; q += 2) {
          typedef double d2 __attribute__((ext_vector_type(2)));
          d2 w = {v + q, v + q + 1};
          d2* p = (d2*)&jb[(long)(q + half) * n + ll];
          if (V == 2) *p = w; else __builtin_nontemporal_store(w, p);
        }
      }
    }
    v += 1e-9;
  }
}

// every workgroup streams one contiguous chunk; MAP=1: the chunks of an XCD are contiguous too
template <int MAP>
__global__ __launch_bounds__(256) void chunk_store(double* out, long n_chunk_d2) {
  unsigned item = blockIdx.x;
  if (MAP == 1) {
    const unsigned lin = blockIdx.x, tot = gridDim.x, x = lin % 8, q = tot / 8, r = tot % 8;
    item = x * q + (x < r ? x : r) + lin / 8;
  }
  double2* o = (double2*)out + (long)item * n_chunk_d2;
  for (long i = threadIdx.x; i < n_chunk_d2; i += 256) o[i] = make_double2(1.0, 2.0);
}

template <int W>
__global__ void stream_store(double* out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long step = (long)gridDim.x * blockDim.x;
  if (W == 1) for (; i < n; i += step) out[i] = 1.0;
  else { double2* o = (double2*)out; for (; i < n / 2; i += step) o[i] = make_double2(1.0, 2.0); }
}
__global__ void stream_copy(const double2* in, double2* out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x; long step = (long)gridDim.x * blockDim.x;
  for (; i < n; i += step) out[i] = in[i];
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 250, SL = 46, tiles = 20, B = 4096;
  long stride = (long)tiles * SL * n + (argc > 2 ? atoi(argv[2]) : 0);  // doubles per b (+ optional misalignment)
  printf("n=%d stride=%ld doubles\n", n, stride);
  long total = stride * B;
  double* out; CHK(hipMalloc(&out, total * 8 + 64));
  double* in; CHK(hipMalloc(&in, total * 8 + 64)); CHK(hipMemset(in, 0, total * 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 3; ++i) launch();
    CHK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); CHK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f us  %7.1f GB/s\n", name, ms * 100, bytes / (ms / 10 * 1e-3) / 1e9);
  };
  double bytes = (double)total * 8;
  for (int bpb : {4, 8}) {
    dim3 g(tiles, (B + bpb - 1) / bpb);
    printf("bpb=%d grid=%d x %d\n", bpb, g.x, g.y);
    run("tile 8B plain", [&] { tile_store<0, SL><<<g, 256>>>(out, n, tiles, B, bpb, stride); }, bytes);
    run("tile 8B nontemporal", [&] { tile_store<1, SL><<<g, 256>>>(out, n, tiles, B, bpb, stride); }, bytes);
    run("tile 16B plain", [&] { tile_store<2, SL><<<g, 256>>>(out, n, tiles, B, bpb, stride); }, bytes);
    run("tile 16B nontemporal", [&] { tile_store<3, SL><<<g, 256>>>(out, n, tiles, B, bpb, stride); }, bytes);
    run("tile 16B plain, XCD-blocked", [&] { tile_store<2, SL, 1><<<g, 256>>>(out, n, tiles, B, bpb, stride); }, bytes);
    run("tile 8B plain, XCD-blocked", [&] { tile_store<0, SL, 1><<<g, 256>>>(out, n, tiles, B, bpb, stride); }, bytes);
  }
  for (long kb : {64, 256, 1024, 4096}) {
    long d2 = kb * 1024 / 16, chunks = total * 8 / (kb * 1024);
    char nm[64];
    snprintf(nm, 64, "chunk %ld KB natural", kb);
    run(nm, [&] { chunk_store<0><<<(unsigned)chunks, 256>>>(out, d2); }, (double)chunks * kb * 1024);
    snprintf(nm, 64, "chunk %ld KB XCD-blocked", kb);
    run(nm, [&] { chunk_store<1><<<(unsigned)chunks, 256>>>(out, d2); }, (double)chunks * kb * 1024);
  }
  run("stream 8B grid-stride", [&] { stream_store<1><<<2048, 256>>>(out, total); }, bytes);
  run("stream 16B grid-stride", [&] { stream_store<2><<<2048, 256>>>(out, total); }, bytes);
  run("copy 16B (r+w bytes)", [&] { stream_copy<<<2048, 256>>>((double2*)in, (double2*)out, total / 2); }, 2 * bytes);
  run("hipMemsetAsync", [&] { CHK(hipMemsetAsync(out, 0, total * 8)); }, bytes);
  return 0;
}
