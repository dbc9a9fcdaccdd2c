// single_launch_ab.hip -- round 4: north_star's literal shape, ONE 1M x 5 launch at a time, reads at
// 0.58 of the HBM peak (11.2 us per launch) where a dispatch of >= 27 batches reaches 0.65.  What a launch loses is its
// boundary: the ramp of 1,024 workgroups and the drain of the last ones.  This tool times K = 33 resident 1M x 5 batches
// (2.2 GB, 8x the Infinity Cache) swept one launch per batch through
//   same      K launches on one stream (the shipped raftq_sweep_many_async)
//   tiles     the same with 512- / 2,048-group tiles, and with 128- / 512-thread workgroups
//   streams   consecutive launches alternated over 2 / 4 streams, so that launch k+1's ramp overlaps launch k's drain
//   graph     the K launches captured once into a hipGraph (one stream; and forked over 2 / 4 streams) and replayed
// Measurement tool, not part of libraftq.so.  One JSON object per line; medians of 7 repetitions of 10 steps.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "raftq_kernels.hpp"

using namespace raftqk;

#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                           \
    }                                                                                    \
  } while (0)

__global__ void fill_kernel(uint64_t* p, uint64_t n, uint64_t seed, uint64_t mask) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    p[i] = (z ^ (z >> 31)) & mask;
  }
}

constexpr int N = 5;
constexpr uint64_t G = 1ull << 20;
constexpr int K = 33;

struct Member {
  SweepArgs a;
};

template <int GPL, int POLICY, int BLOCK>
static void launch_one(const SweepArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((sweep_kernel<N, GPL, true, false, true, POLICY, true, BLOCK>), dim3((unsigned)(G / ((uint64_t)BLOCK * GPL))), dim3(BLOCK), 0, s, a);
}

typedef void (*LaunchFn)(const SweepArgs&, hipStream_t);

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  const int reps = argc > 1 ? atoi(argv[1]) : 7;
  const int steps = 10;
  std::vector<Member> m(K);
  for (int k = 0; k < K; ++k) {
    SweepArgs& a = m[k].a;
    uint64_t *match, *c0, *c1, *bits;
    uint8_t *votes, *outcome;
    uint4* partials;
    CK(hipMalloc((void**)&match, N * G * 8));
    CK(hipMalloc((void**)&c0, G * 8));
    CK(hipMalloc((void**)&c1, G * 8));
    CK(hipMalloc((void**)&votes, G * 2));
    CK(hipMalloc((void**)&outcome, G / 4));
    CK(hipMalloc((void**)&bits, G / 8));
    CK(hipMalloc((void**)&partials, (G / 128) * sizeof(uint4)));  // enough for the smallest tile's wave count
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, match, N * G, 0x1000 + k, (1ull << 40) - 1);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, c0, G, 0x2000 + k, (1ull << 39) - 1);
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, (uint64_t*)votes, G / 4, 0x3000 + k, 0x0155015501550155ull);
    a.match = match;
    a.committed = c0;
    a.committed_out = c1;
    a.first_idx = nullptr;
    a.votes = votes;
    a.outcome = outcome;
    a.changed_bits = nullptr;
    a.partials = partials;
    a.ld = G;
  }
  CK(hipDeviceSynchronize());
  const int kMaxStreams = 4;
  hipStream_t st[kMaxStreams];
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1, fork, join[kMaxStreams];
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  for (auto& j : join) CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));

  const double rd_bytes = (8.0 * N + 8 + 2) * G;

  // one step = K launches through `fn`, spread over `ns` streams round-robin; stream 0 forks / joins the others by events
  auto step_streams = [&](LaunchFn fn, int ns) {
    if (ns > 1) {
      CK(hipEventRecord(fork, st[0]));
      for (int i = 1; i < ns; ++i) CK(hipStreamWaitEvent(st[i], fork, 0));
    }
    for (int k = 0; k < K; ++k) fn(m[k].a, st[k % ns]);
    if (ns > 1)
      for (int i = 1; i < ns; ++i) {
        CK(hipEventRecord(join[i], st[i]));
        CK(hipStreamWaitEvent(st[0], join[i], 0));
      }
  };
  auto median_us = [&](auto&& one_step) -> double {
    std::vector<double> v;
    for (int r = 0; r < reps; ++r) {
      one_step();
      one_step();
      CK(hipStreamSynchronize(st[0]));
      CK(hipEventRecord(e0, st[0]));
      for (int s = 0; s < steps; ++s) one_step();
      CK(hipEventRecord(e1, st[0]));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      v.push_back(ms * 1e3 / (steps * K));
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
  };
  auto report = [&](const char* name, double us) {
    printf("{\"variant\":\"%s\",\"us_per_launch\":%.3f,\"read_frac_of_8TBps\":%.4f}\n", name, us, rd_bytes / (us * 1e-6) / 8e12);
    fflush(stdout);
  };

  struct V {
    const char* name;
    LaunchFn fn;
  };
  const V tiles[] = {
      {"same_stream gpl4 block256 nt_ld (shipped shape, 1024 wgs)", launch_one<4, kLdNT, 256>},
      {"same_stream gpl4 block256 nt_ld_st", launch_one<4, kLdNT | kStNT, 256>},
      {"same_stream gpl4 block256 cached", launch_one<4, 0, 256>},
      {"same_stream gpl2 block256 nt_ld (512-group tiles, 2048 wgs)", launch_one<2, kLdNT, 256>},
      {"same_stream gpl8 block256 nt_ld (2048-group tiles, 512 wgs)", launch_one<8, kLdNT, 256>},
      {"same_stream gpl4 block128 nt_ld (512-group tiles, 2048 wgs of 2 waves)", launch_one<4, kLdNT, 128>},
      {"same_stream gpl2 block128 nt_ld (256-group tiles, 4096 wgs)", launch_one<2, kLdNT, 128>},
      {"same_stream gpl2 block512 nt_ld (1024-group tiles, 1024 wgs of 8 waves)", launch_one<2, kLdNT, 512>},
  };
  for (const V& v : tiles) report(v.name, median_us([&] { step_streams(v.fn, 1); }));
  for (int ns : {2, 4}) {
    char name[128];
    snprintf(name, sizeof name, "%d_streams gpl4 block256 nt_ld", ns);
    report(name, median_us([&] { step_streams(launch_one<4, kLdNT, 256>, ns); }));
    snprintf(name, sizeof name, "%d_streams gpl2 block256 nt_ld", ns);
    report(name, median_us([&] { step_streams(launch_one<2, kLdNT, 256>, ns); }));
  }
  // graphs: the K launches of one step captured once, replayed per step
  for (int ns : {1, 2, 4}) {
    for (int gpl : {4, 2}) {
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
      step_streams(gpl == 4 ? (LaunchFn)launch_one<4, kLdNT, 256> : (LaunchFn)launch_one<2, kLdNT, 256>, ns);
      CK(hipStreamEndCapture(st[0], &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      char name[128];
      snprintf(name, sizeof name, "graph_%d_stream%s gpl%d block256 nt_ld", ns, ns > 1 ? "s" : "", gpl);
      report(name, median_us([&] { CK(hipGraphLaunch(ge, st[0])); }));
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
    }
  }
  // the K-deep set dispatch beside it, same data (what the headline uses)
  {
    SweepArgs* tab;
    std::vector<SweepArgs> host(K);
    for (int k = 0; k < K; ++k) host[k] = m[k].a;
    CK(hipMalloc((void**)&tab, K * sizeof(SweepArgs)));
    CK(hipMemcpy(tab, host.data(), K * sizeof(SweepArgs), hipMemcpyHostToDevice));
    report("set_dispatch gpl8 block256 nt_ld_st (one launch for all 33)", median_us([&] {
             hipLaunchKernelGGL((sweep_set_kernel<N, 8, true, false, true, kLdNT | kStNT, true>), dim3((unsigned)(G / 2048), K), dim3(256), 0, st[0],
                                (const SweepArgs*)tab, 0u);
           }));
  }
  return 0;
}
