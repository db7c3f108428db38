// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the
// sampler's row pass and likelihood pass use (round-2 VERDICT, weak #5: the blanket x2 of the guide is stated
// for 16 B/lane streams only; k_transpose read 40 MB and reported 29 MB raw).
//
// Every kernel streams a buffer of KNOWN size that is far beyond the 256 MiB Infinity Cache (default 2 GiB),
// once, with one access width:
//   rd_u32     4 B / lane   (the row pass's label words: 4 row labels per thread)
//   rd_f32x4  16 B / lane   (float32 shadow of the split column)
//   rd_f64x2  16 B / lane   ({sum_trees, r} pairs)
//   rd_f64x4  32 B / lane   (float64 split column: two double2 per thread)
//   rd_u8      1 B / lane   (lone-FINAL pass, one label byte per thread)
//   rd_tile   256-B segments at an 800-B row stride (k_transpose's read of a row-major n x 100 matrix)
//   wr_u32 / wr_f64x2 / wr_f64   stores of 4 / 16 / 8 B per lane
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); the program prints the bytes
// each kernel moved; tools/microbench/fetch_calib.py divides.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/fetch_calib.hip -o tools/microbench/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void rd(const T* __restrict__ p, size_t count, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    const T v = p[i];  // every byte of v is consumed, so the load keeps its width
    if constexpr (sizeof(T) >= 4) {
      const unsigned* u = (const unsigned*)&v;
#pragma unroll
      for (int k = 0; k < (int)(sizeof(T) / 4); ++k) acc += u[k];
    } else {
      acc += (unsigned long long)*(const unsigned char*)&v;
    }
  }
  if (acc == 0x123456789abcull) *sink = acc;  // never true for a zero buffer: keeps the loads alive
}
struct f64x4 { double2 a, b; };

__global__ __launch_bounds__(256) void rd_tile(const double* __restrict__ X, long long n, int ld, unsigned long long* sink) {
  // 32 x 32 tiles like k_transpose: a tile row is 32 doubles = 256 B, rows are ld doubles apart
  unsigned long long acc = 0;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 8 rows per step
  const long long tiles_r = n / 32;
  const int tiles_c = ld / 32;
  for (long long t = blockIdx.x; t < tiles_r * tiles_c; t += gridDim.x) {
    const long long r0 = (t / tiles_c) * 32;
    const int c0 = (int)(t % tiles_c) * 32;
    for (int k = 0; k < 32; k += 8) acc += (unsigned long long)X[(size_t)(r0 + k + ty) * ld + c0 + tx];
  }
  if (acc == 0x123456789abcull) *sink = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void wr(T* __restrict__ p, size_t count, T v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) p[i] = v;
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 2048) << 20;
  void* buf = nullptr;
  unsigned long long* sink = nullptr;
  HC(hipMalloc(&buf, bytes));
  HC(hipMalloc((void**)&sink, 8));
  HC(hipMemset(buf, 0, bytes));
  HC(hipDeviceSynchronize());
  const dim3 grid(256 * 8), block(256);
  hipLaunchKernelGGL(rd<unsigned>, grid, block, 0, 0, (const unsigned*)buf, bytes / 4, sink);
  hipLaunchKernelGGL(rd<float4>, grid, block, 0, 0, (const float4*)buf, bytes / 16, sink);
  hipLaunchKernelGGL(rd<double2>, grid, block, 0, 0, (const double2*)buf, bytes / 16, sink);
  hipLaunchKernelGGL(rd<f64x4>, grid, block, 0, 0, (const f64x4*)buf, bytes / 32, sink);
  hipLaunchKernelGGL(rd<unsigned char>, grid, block, 0, 0, (const unsigned char*)buf, bytes / 4, sink);  // a quarter: 1 B loads are slow
  const int ld = 100;  // 96 of 100 columns are read (3 tile columns): 768 of every 800 B
  const long long n = (long long)(bytes / 8 / ld) / 32 * 32;
  hipLaunchKernelGGL(rd_tile, grid, block, 0, 0, (const double*)buf, n, ld, sink);
  HC(hipDeviceSynchronize());
  hipLaunchKernelGGL(wr<unsigned>, grid, block, 0, 0, (unsigned*)buf, bytes / 4, 0u);
  hipLaunchKernelGGL(wr<double2>, grid, block, 0, 0, (double2*)buf, bytes / 16, make_double2(0.0, 0.0));
  hipLaunchKernelGGL(wr<double>, grid, block, 0, 0, (double*)buf, bytes / 8, 0.0);
  HC(hipDeviceSynchronize());
  printf("{\"rd<unsigned int>\": %zu, \"rd<HIP_vector_type<float, 4u> >\": %zu, \"rd<HIP_vector_type<double, 2u> >\": %zu, "
         "\"rd<f64x4>\": %zu, \"rd<unsigned char>\": %zu, \"rd_tile\": %zu, \"rd_tile_lines_bytes\": %zu, "
         "\"wr<unsigned int>\": %zu, \"wr<HIP_vector_type<double, 2u> >\": %zu, \"wr<double>\": %zu}\n",
         bytes, bytes, bytes, bytes, bytes / 4, (size_t)n * 96 * 8, (size_t)n * 800, bytes, bytes, bytes);
  HC(hipFree(buf));
  HC(hipFree(sink));
  return 0;
}
