// pmc_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE / TCC_EA0_* counters in the
// access patterns of the rows either side of the sweep (round 4).  The guide's gfx950 rule
// (FETCH_SIZE = 1/2 of a wide coalesced stream) was calibrated for 16 B/lane streaming reads only; Step's list walk, the
// delta scatter and the compaction gather one 8-byte word per lane from scattered cache lines, and store the same way.
// Three kernels, every launch over a fresh slice of a 2 GiB arena (nothing is served from the 256 MiB Infinity Cache):
//   calib_stream_kernel   64 MiB read with 16 B/lane non-temporal loads, 16 MiB written with 16 B/lane stores
//   calib_gather_kernel   2^20 lanes, each ONE 8-byte load from a line of its own (128 B apart), 8 MiB written densely
//   calib_scatter_kernel  2^20 lanes, each ONE 8-byte store into a line of its own (128 B apart), 8 MiB read densely
// Usage: pmc_calib [launches]   (run under `rocprofv3 --pmc <counter> --kernel-trace`; tools/pmc_legs.py reads the CSVs)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                           \
    }                                                                                    \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void calib_stream_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n_rd16,
                                                           uint64_t n_wr16) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  u32x4 acc = {0, 0, 0, 0};
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_rd16; i += stride) acc ^= __builtin_nontemporal_load(src + i);
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_wr16; i += stride) dst[i] = acc;
}

// line(i) = (i * odd) mod n_lines: a permutation of the lines, so every lane has a line to itself
__global__ __launch_bounds__(256) void calib_gather_kernel(const uint64_t* __restrict__ arena, uint64_t n_lines, uint64_t* __restrict__ dense,
                                                           uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t line = (i * 0x9E3779B1ull) & (n_lines - 1);
  dense[i] = arena[line * 16 + (i & 15)];
}

__global__ __launch_bounds__(256) void calib_scatter_kernel(uint64_t* __restrict__ arena, uint64_t n_lines, const uint64_t* __restrict__ dense,
                                                            uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t line = (i * 0x9E3779B1ull) & (n_lines - 1);
  arena[line * 16 + (i & 15)] = dense[i];
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 12;
  CK(hipSetDevice(0));
  const uint64_t arena_bytes = 2ull << 30;
  uint8_t* arena;
  uint64_t* dense;
  CK(hipMalloc((void**)&arena, arena_bytes));
  CK(hipMalloc((void**)&dense, 64u << 20));
  CK(hipMemset(arena, 1, arena_bytes));
  CK(hipMemset(dense, 2, 64u << 20));
  CK(hipDeviceSynchronize());
  const uint64_t slice = 128ull << 20;           // per launch: 2^20 lines of 128 B, or 64 + 16 MiB of stream
  const uint64_t n_slices = arena_bytes / slice;  // 16: a slice is revisited after 2 GiB of other traffic
  for (int k = 0; k < launches; ++k) {
    uint8_t* base = arena + (uint64_t)(k % n_slices) * slice;
    hipLaunchKernelGGL(calib_stream_kernel, dim3(2048), dim3(256), 0, 0, (const u32x4*)base, (u32x4*)(base + (64u << 20)), (uint64_t)(64u << 20) / 16,
                       (uint64_t)(16u << 20) / 16);
    hipLaunchKernelGGL(calib_gather_kernel, dim3(4096), dim3(256), 0, 0, (const uint64_t*)base, (uint64_t)1 << 20, dense + (uint64_t)(k & 3) * (1u << 20),
                       (uint64_t)1 << 20);
    hipLaunchKernelGGL(calib_scatter_kernel, dim3(4096), dim3(256), 0, 0, (uint64_t*)base, (uint64_t)1 << 20, dense + (uint64_t)(4 + (k & 3)) * (1u << 20),
                       (uint64_t)1 << 20);
    CK(hipDeviceSynchronize());
  }
  printf("{\"launches\":%d,\"stream_read_bytes\":%u,\"stream_write_bytes\":%u,\"gather_lines\":%u,\"gather_dense_write_bytes\":%u,"
         "\"scatter_lines\":%u,\"scatter_dense_read_bytes\":%u}\n",
         launches, 64u << 20, 16u << 20, 1u << 20, 8u << 20, 1u << 20, 8u << 20);
  return 0;
}
