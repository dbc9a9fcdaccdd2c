// pcie_duplex_probe.hip -- what the link under a codec call can do (round 4): how fast workgroups pull
// page-locked host memory in (by request depth and grid size), how fast they push results out, whether the two directions
// overlap -- inside ONE kernel (disjoint workgroup ranges), across TWO streams (the round-1 finding "a kernel that writes
// host memory holds back the other queues' next kernel", re-probed on ROCm 7.2), and with the runtime's SDMA copies on
// the inbound side.  One JSON object per line.  Sizes: 1 MiB, 3.1 MB (a 64K-frame decode's inbound), 4.9 MB (its
// outbound), 6.4 MB (a 64K-message encode's inbound).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("{\"error\":\"%s -> %s\"}\n", #x, hipGetErrorString(e_));            \
      return 3;                                                                   \
    }                                                                             \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256;

// inbound: U independent 16-byte loads per lane in flight before the first store
template <int U>
__global__ __launch_bounds__(kBlock) void rd_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n16) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
    for (int k = 0; k < U; ++k) dst[i + k * stride] = v[k];
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

// the fused-tile shape: a workgroup pulls ITS contiguous chunk into LDS (one row of 16 B per lane per step, all steps
// issued before the first wait), then leaves 16 bytes per lane in device memory
template <int ROWS>
__global__ __launch_bounds__(kBlock) void rd_tile_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n16) {
  __shared__ u32x4 tile[ROWS * kBlock];
  const uint64_t base = (uint64_t)blockIdx.x * ROWS * kBlock;
  u32x4 v[ROWS];
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    const uint64_t i = base + (uint64_t)k * kBlock + threadIdx.x;
    v[k] = i < n16 ? __builtin_nontemporal_load(src + i) : u32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int k = 0; k < ROWS; ++k) tile[k * kBlock + threadIdx.x] = v[k];
  __syncthreads();
  u32x4 acc = tile[threadIdx.x];
#pragma unroll
  for (int k = 1; k < ROWS; ++k) acc ^= tile[k * kBlock + (threadIdx.x ^ k)];
  dst[(uint64_t)blockIdx.x * kBlock + threadIdx.x] = acc;
}

// outbound: non-temporal 16-byte stores into host memory
__global__ __launch_bounds__(kBlock) void wr_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n16) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) __builtin_nontemporal_store(src[i], dst + i);
}

// both directions in ONE launch: the first rb workgroups pull, the others push
template <int U>
__global__ __launch_bounds__(kBlock) void duplex_kernel(const u32x4* __restrict__ hsrc, u32x4* __restrict__ ddst, uint64_t n_in,
                                                        const u32x4* __restrict__ dsrc, u32x4* __restrict__ hdst, uint64_t n_out,
                                                        uint32_t rb) {
  if (blockIdx.x < rb) {
    const uint64_t stride = (uint64_t)rb * kBlock;
    uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + (U - 1) * stride < n_in; i += U * stride) {
      u32x4 v[U];
#pragma unroll
      for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(hsrc + i + k * stride);
#pragma unroll
      for (int k = 0; k < U; ++k) ddst[i + k * stride] = v[k];
    }
    for (; i < n_in; i += stride) ddst[i] = hsrc[i];
  } else {
    const uint64_t stride = (uint64_t)(gridDim.x - rb) * kBlock;
    for (uint64_t i = (uint64_t)(blockIdx.x - rb) * kBlock + threadIdx.x; i < n_out; i += stride)
      __builtin_nontemporal_store(dsrc[i], hdst + i);
  }
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t cap = 8u << 20;
  u32x4 *hin, *hout, *din, *dout, *hin_d, *hout_d;
  CK(hipHostMalloc((void**)&hin, cap, hipHostMallocMapped));
  CK(hipHostMalloc((void**)&hout, cap, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&hin_d, hin, 0));
  CK(hipHostGetDevicePointer((void**)&hout_d, hout, 0));
  CK(hipMalloc((void**)&din, cap));
  CK(hipMalloc((void**)&dout, cap));
  memset(hin, 0x5a, cap);
  memset(hout, 0, cap);
  CK(hipMemset(dout, 0x33, cap));
  CK(hipDeviceSynchronize());
  const int reps = 20;
  const size_t sizes[] = {1u << 20, 3100000, 4900000, 6400000};

  auto timed = [&](auto&& launch, hipStream_t s) -> float {  // us per repetition, events on the launching stream
    for (int r = 0; r < 3; ++r) launch();
    (void)hipStreamSynchronize(s);
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
  };

  for (size_t bytes : sizes) {
    const uint64_t n16 = bytes / 16;
    // 1. inbound by workgroups: grid size x loads in flight
    for (unsigned blocks : {48u, 96u, 192u, 384u, 768u}) {
      const float u1 = timed([&] { hipLaunchKernelGGL(rd_kernel<1>, dim3(blocks), dim3(kBlock), 0, s1, hin_d, din, n16); }, s1);
      const float u2 = timed([&] { hipLaunchKernelGGL(rd_kernel<2>, dim3(blocks), dim3(kBlock), 0, s1, hin_d, din, n16); }, s1);
      const float u4 = timed([&] { hipLaunchKernelGGL(rd_kernel<4>, dim3(blocks), dim3(kBlock), 0, s1, hin_d, din, n16); }, s1);
      const float u8 = timed([&] { hipLaunchKernelGGL(rd_kernel<8>, dim3(blocks), dim3(kBlock), 0, s1, hin_d, din, n16); }, s1);
      printf("{\"probe\":\"rd_kernel\",\"bytes\":%zu,\"blocks\":%u,\"us_U1\":%.1f,\"us_U2\":%.1f,\"us_U4\":%.1f,\"us_U8\":%.1f,"
             "\"GBps_best\":%.1f}\n",
             bytes, blocks, u1, u2, u4, u8, bytes / (1e3 * std::min(std::min(u1, u2), std::min(u4, u8))));
    }
    {  // the fused-tile shape: one 4 / 8 / 16 KB chunk per workgroup
      const float t1 = timed([&] { hipLaunchKernelGGL(rd_tile_kernel<1>, dim3((unsigned)((n16 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s1, hin_d, din, n16); }, s1);
      const float t2 = timed([&] { hipLaunchKernelGGL(rd_tile_kernel<2>, dim3((unsigned)((n16 + 2 * kBlock - 1) / (2 * kBlock))), dim3(kBlock), 0, s1, hin_d, din, n16); }, s1);
      const float t4 = timed([&] { hipLaunchKernelGGL(rd_tile_kernel<4>, dim3((unsigned)((n16 + 4 * kBlock - 1) / (4 * kBlock))), dim3(kBlock), 0, s1, hin_d, din, n16); }, s1);
      printf("{\"probe\":\"rd_tile_kernel\",\"bytes\":%zu,\"us_4KB_tiles\":%.1f,\"us_8KB_tiles\":%.1f,\"us_16KB_tiles\":%.1f,\"GBps_best\":%.1f}\n",
             bytes, t1, t2, t4, bytes / (1e3 * std::min(t1, std::min(t2, t4))));
    }
    // 2. outbound by workgroups
    for (unsigned blocks : {48u, 96u, 192u, 384u}) {
      const float w = timed([&] { hipLaunchKernelGGL(wr_kernel, dim3(blocks), dim3(kBlock), 0, s1, dout, hout_d, n16); }, s1);
      printf("{\"probe\":\"wr_kernel\",\"bytes\":%zu,\"blocks\":%u,\"us\":%.1f,\"GBps\":%.1f}\n", bytes, blocks, w, bytes / (1e3 * w));
    }
    // 3. the runtime's copies (SDMA), alone
    {
      const float h2d = timed([&] { (void)hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s1); }, s1);
      const float d2h = timed([&] { (void)hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s1); }, s1);
      printf("{\"probe\":\"sdma\",\"bytes\":%zu,\"h2d_us\":%.1f,\"h2d_GBps\":%.1f,\"d2h_us\":%.1f,\"d2h_GBps\":%.1f}\n", bytes, h2d,
             bytes / (1e3 * h2d), d2h, bytes / (1e3 * d2h));
    }
    // 4. both directions in ONE kernel: `bytes` in and `bytes` out
    for (unsigned rb : {96u, 192u, 384u}) {
      for (unsigned wb : {96u, 192u}) {
        const float d = timed([&] { hipLaunchKernelGGL(duplex_kernel<4>, dim3(rb + wb), dim3(kBlock), 0, s1, hin_d, din, n16, dout, hout_d, n16, rb); }, s1);
        printf("{\"probe\":\"duplex_one_kernel\",\"bytes_each_way\":%zu,\"read_blocks\":%u,\"write_blocks\":%u,\"us\":%.1f,\"GBps_each_way\":%.1f}\n",
               bytes, rb, wb, d, bytes / (1e3 * d));
      }
    }
    // 5. two streams (wall clock around both): read kernel on s1 | write kernel on s2; SDMA H2D on s1 | write kernel on s2;
    //    SDMA both ways
    auto wall = [&](auto&& a, auto&& b) -> double {
      for (int r = 0; r < 3; ++r) { a(); b(); }
      (void)hipStreamSynchronize(s1);
      (void)hipStreamSynchronize(s2);
      const double t0 = now_us();
      for (int r = 0; r < reps; ++r) { a(); b(); }
      (void)hipStreamSynchronize(s1);
      (void)hipStreamSynchronize(s2);
      return (now_us() - t0) / reps;
    };
    const double kk = wall([&] { hipLaunchKernelGGL(rd_kernel<4>, dim3(192), dim3(kBlock), 0, s1, hin_d, din, n16); },
                           [&] { hipLaunchKernelGGL(wr_kernel, dim3(96), dim3(kBlock), 0, s2, dout, hout_d, n16); });
    const double sk = wall([&] { (void)hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s1); },
                           [&] { hipLaunchKernelGGL(wr_kernel, dim3(96), dim3(kBlock), 0, s2, dout, hout_d, n16); });
    const double ss = wall([&] { (void)hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s1); },
                           [&] { (void)hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s2); });
    const double ks = wall([&] { hipLaunchKernelGGL(rd_kernel<4>, dim3(192), dim3(kBlock), 0, s1, hin_d, din, n16); },
                           [&] { (void)hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s2); });
    printf("{\"probe\":\"two_streams\",\"bytes_each_way\":%zu,\"rdkernel_wrkernel_us\":%.1f,\"sdmaH2D_wrkernel_us\":%.1f,"
           "\"sdmaH2D_sdmaD2H_us\":%.1f,\"rdkernel_sdmaD2H_us\":%.1f}\n",
           bytes, kk, sk, ss, ks);
    fflush(stdout);
  }
  // 6. a small compute kernel on s2 while a host-writing kernel runs on s1: does it start? (device-only kernel timed alone / beside)
  {
    const uint64_t n16 = 4900000 / 16;
    auto small = [&] { hipLaunchKernelGGL(rd_kernel<1>, dim3(64), dim3(kBlock), 0, s2, (const u32x4*)dout, din, (uint64_t)65536); };
    const float alone = timed(small, s2);
    // beside: keep s1 busy with host writes for the whole timed region
    for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(wr_kernel, dim3(96), dim3(kBlock), 0, s1, dout, hout_d, n16);
    const float beside = timed(small, s2);
    (void)hipStreamSynchronize(s1);
    printf("{\"probe\":\"device_kernel_beside_host_writer\",\"alone_us\":%.2f,\"beside_us\":%.2f}\n", alone, beside);
  }
  return 0;
}
