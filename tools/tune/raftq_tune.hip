// raftq_tune.hip -- within-process A/B of the sweep-kernel variants on MI355X.
// Not part of libraftq.so: a measurement tool.  For each (N, variant, GPL, NT)
// it times R launches with HIP events, once re-sweeping a single resident set
// (1M groups fit the 256 MiB Infinity Cache -> an L3 number) and once rotating
// through K independent sets totalling > 1 GiB (an HBM number); SURVEY.md F9.
// A plain 16 B/lane copy of the same byte volume is timed beside them as the
// chip's attainable streaming rate for this footprint.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
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

__global__ void fill_kernel(uint64_t* p, uint64_t n, uint64_t seed, uint64_t mask, uint64_t add) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    p[i] = (z & mask) + add;
  }
}

__global__ void fill_votes_kernel(uint8_t* p, uint64_t n, uint64_t seed) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 29;
    const unsigned u = (unsigned)(z % 10);
    p[i] = u < 3 ? 0 : (u < 8 ? 1 : 2);
  }
}

// streaming copy reference: in_bytes read, out_bytes written, 16 B per lane
template <bool NT>
__global__ __launch_bounds__(256) void copy_ref_kernel(const u64x2* __restrict__ in, uint64_t n_in,
                                                        u64x2* __restrict__ out, uint64_t n_out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  u64x2 acc = {0, 0};
  for (uint64_t k = i; k < n_in; k += stride) {
    const u64x2 v = NT ? __builtin_nontemporal_load(in + k) : in[k];
    acc ^= v;
    if (k < n_out) {
      if (NT) __builtin_nontemporal_store(v, out + k); else out[k] = v;
    }
  }
  if (acc.x == 0x123456789abcdefull && acc.y == 1) out[0] = acc;  // keep the loads alive
}

struct Set {
  uint8_t* arena;
  size_t arena_bytes;
  uint64_t *match, *committed, *committed_out, *first_idx, *changed;
  uint8_t *votes, *outcome, *copy_out;
  uint4* partials;
};

// one arena per set: match | committed | committed_out | first_idx | votes |
// outcome | changed | partials | copy_out (scratch for the copy reference)
static Set make_set(int N, uint64_t ld, uint64_t seed) {
  Set s;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_match = carve((size_t)N * ld * 8), o_c = carve(ld * 8), o_co = carve(ld * 8), o_f = carve(ld * 8);
  const size_t o_v = carve((size_t)N * ld), o_o = carve(ld), o_ch = carve(ld / 8);
  const size_t o_p = carve(ld / 512 * 4 * sizeof(uint4)), o_copy = carve(ld * 16);
  s.arena_bytes = off;
  CK(hipMalloc(&s.arena, off));
  s.match = (uint64_t*)(s.arena + o_match);
  s.committed = (uint64_t*)(s.arena + o_c);
  s.committed_out = (uint64_t*)(s.arena + o_co);
  s.first_idx = (uint64_t*)(s.arena + o_f);
  s.votes = s.arena + o_v;
  s.outcome = s.arena + o_o;
  s.changed = (uint64_t*)(s.arena + o_ch);
  s.partials = (uint4*)(s.arena + o_p);
  s.copy_out = s.arena + o_copy;
  const uint64_t base = 1ull << 30;
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, s.match, (uint64_t)N * ld, seed, 2047ull, base);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, s.committed, ld, seed + 1, 1023ull, base + 512);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, s.first_idx, ld, seed + 2, 2047ull, base);
  hipLaunchKernelGGL(fill_votes_kernel, dim3(2048), dim3(256), 0, 0, s.votes, (uint64_t)N * ld, seed + 3);
  CK(hipGetLastError());
  return s;
}

static void free_set(Set& s) { (void)hipFree(s.arena); }

static SweepArgs args_of(const Set& s, uint64_t ld) {
  SweepArgs a;
  a.match = s.match; a.committed = s.committed; a.committed_out = s.committed_out;
  a.first_idx = s.first_idx; a.votes = s.votes; a.outcome = s.outcome;
  a.changed_bits = nullptr; a.partials = s.partials; a.ld = ld;
  return a;
}

typedef void (*launch_fn)(const SweepArgs&, uint64_t G, hipStream_t);

template <int N, int GPL, bool GATED, bool VOTES, int POLICY, int BLOCK>
static void launch_reg(const SweepArgs& a, uint64_t G, hipStream_t st) {
  hipLaunchKernelGGL((sweep_kernel<N, GPL, true, GATED, VOTES, POLICY, true, BLOCK>),
                     dim3((unsigned)(G / (BLOCK * GPL))), dim3(BLOCK), 0, st, a);
}
template <int N, int GPL, bool GATED, bool VOTES>
static void launch_lds(const SweepArgs& a, uint64_t G, hipStream_t st) {
  constexpr size_t lds = (size_t)4 * (GPL / 2) * (N + 1 + (GATED ? 1 : 0)) * 1024;
  hipLaunchKernelGGL((sweep_lds_kernel<N, GPL, GATED, VOTES, true>), dim3((unsigned)(G / (256 * GPL))), dim3(256),
                     lds, st, a);
}

struct Variant {
  const char* name;
  int N, GPL, policy, block, gated, votes;
  launch_fn fn;
};

#define REG(N, GPL, G_, V_, POL, BLK) \
  { "reg", N, GPL, POL, BLK, G_, V_, launch_reg<N, GPL, G_, V_, POL, BLK> }
#define LDS(N, GPL, G_, V_) \
  { "lds", N, GPL, 0, 256, G_, V_, launch_lds<N, GPL, G_, V_> }

// policy: 1 = NT loads, 2 = NT stores, 4 = ablation: no stores
static const Variant kVariants[] = {
    // config 3: 1M x 5 commit + votes (the headline): policy x tile x block
    REG(5, 4, false, true, 0, 256), REG(5, 4, false, true, 1, 256), REG(5, 4, false, true, 2, 256),
    REG(5, 4, false, true, 3, 256), REG(5, 2, false, true, 3, 256), REG(5, 8, false, true, 3, 256),
    REG(5, 4, false, true, 3, 128), REG(5, 4, false, true, 3, 512), REG(5, 4, false, true, 3, 1024),
    REG(5, 2, false, true, 3, 512), REG(5, 2, false, true, 3, 1024),
    REG(5, 4, false, true, 7, 256), REG(5, 4, false, true, 4, 256),  // no-store ablations
    LDS(5, 4, false, true),
    // config 2: 1M x 3 commit only
    REG(3, 4, false, false, 0, 256), REG(3, 4, false, false, 3, 256), REG(3, 4, false, false, 1, 256),
    REG(3, 8, false, false, 3, 256),
    REG(3, 4, false, false, 3, 512), LDS(3, 4, false, false),
    // config 4 shard: 2M x 7 commit + votes
    REG(7, 4, false, true, 0, 256), REG(7, 4, false, true, 3, 256), REG(7, 4, false, true, 1, 256),
    REG(7, 2, false, true, 3, 256),
    REG(7, 4, false, true, 3, 512), LDS(7, 4, false, true),
    // config 5: 1M x 5 gated
    REG(5, 4, true, false, 3, 256), REG(5, 4, true, false, 1, 256), LDS(5, 4, true, false),
    // N = 9 upper bound of the network
    REG(9, 4, false, true, 3, 256), REG(9, 4, false, true, 1, 256), REG(9, 2, false, true, 3, 256),
    LDS(9, 2, false, true),
};

static double bytes_per_group(const Variant& v) {
  double b = 8.0 * v.N + 8 + 8;            // match + committed in + committed out
  if (v.gated) b += 8;                      // first_idx (cur_term is folded on the host)
  if (v.votes) b += v.N + 1;                // vote bytes + outcome
  return b;
}

int main(int argc, char** argv) {
  uint64_t G1 = 1ull << 20;
  int reps = 200;
  if (argc > 1) reps = atoi(argv[1]);
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  printf("{\"device\":\"%s\",\"cus\":%d,\"clock_mhz\":%d}\n", prop.name, prop.multiProcessorCount,
         prop.clockRate / 1000);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  const char* only = getenv("RAFTQ_TUNE_ONLY_N");
  const int only_n = only ? atoi(only) : 0;
  int curN = -1;
  uint64_t curG = 0;
  std::vector<Set> sets;
  for (const Variant& v : kVariants) {
    if (only_n && v.N != only_n) continue;
    const uint64_t G = v.N == 7 ? 2 * G1 : G1;
    const uint64_t ld = (G + 2047) / 2048 * 2048;
    if (v.N != curN || G != curG) {
      for (auto& s : sets) free_set(s);
      sets.clear();
      const double set_bytes = ld * (8.0 * v.N + 24 + v.N + 1);
      const int K = (int)(1.5 * 1024 * 1024 * 1024 / set_bytes) + 1;
      for (int k = 0; k < K; ++k) sets.push_back(make_set(v.N, ld, 1000 * v.N + k));
      CK(hipDeviceSynchronize());
      curN = v.N;
      curG = G;
      // copy reference for this footprint
      const double bpg = 8.0 * v.N + 16 + v.N + 1;
      const uint64_t n_in = (uint64_t)(G * (bpg - 9) / 16), n_out = (uint64_t)(G * 9 / 16);
      for (int nt = 0; nt < 2; ++nt) {
        for (int rot = 0; rot < 2; ++rot) {
          for (int w = 0; w < 5; ++w) {
            const Set& s = sets[rot ? w % sets.size() : 0];
            if (nt) hipLaunchKernelGGL(copy_ref_kernel<true>, dim3(2048), dim3(256), 0, st, (const u64x2*)s.arena, n_in, (u64x2*)s.copy_out, n_out);
            else hipLaunchKernelGGL(copy_ref_kernel<false>, dim3(2048), dim3(256), 0, st, (const u64x2*)s.arena, n_in, (u64x2*)s.copy_out, n_out);
          }
          CK(hipEventRecord(e0, st));
          for (int r = 0; r < reps; ++r) {
            const Set& s = sets[rot ? r % sets.size() : 0];
            if (nt) hipLaunchKernelGGL(copy_ref_kernel<true>, dim3(2048), dim3(256), 0, st, (const u64x2*)s.arena, n_in, (u64x2*)s.copy_out, n_out);
            else hipLaunchKernelGGL(copy_ref_kernel<false>, dim3(2048), dim3(256), 0, st, (const u64x2*)s.arena, n_in, (u64x2*)s.copy_out, n_out);
          }
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          const double us = 1e3 * ms / reps;
          printf("{\"kernel\":\"copy_ref\",\"N\":%d,\"G\":%llu,\"NT\":%d,\"rotate\":%d,\"K\":%zu,\"us\":%.3f,\"GBps\":%.1f}\n",
                 v.N, (unsigned long long)G, nt, rot, sets.size(), us, (n_in + n_out) * 16.0 / us / 1e3);
        }
      }
    }
    for (int rot = 0; rot < 2; ++rot) {
      for (int w = 0; w < 10; ++w) v.fn(args_of(sets[rot ? w % sets.size() : 0], ld), G, st);
      CK(hipGetLastError());
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) v.fn(args_of(sets[rot ? r % sets.size() : 0], ld), G, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = 1e3 * ms / reps;
      const double bpg = bytes_per_group(v);
      printf("{\"kernel\":\"%s\",\"N\":%d,\"GPL\":%d,\"policy\":%d,\"block\":%d,\"gated\":%d,\"votes\":%d,\"G\":%llu,\"rotate\":%d,\"K\":%zu,"
             "\"us\":%.3f,\"GBps\":%.1f,\"Gdec_per_s\":%.2f}\n",
             v.name, v.N, v.GPL, v.policy, v.block, v.gated, v.votes, (unsigned long long)G, rot, sets.size(), us,
             G * bpg / us / 1e3, G / us / 1e3);
      fflush(stdout);
    }
  }
  for (auto& s : sets) free_set(s);
  sets.clear();

  // ---- row-stride stagger: rows exactly 2^k bytes apart put a wave's N+1 row loads on the same
  // HBM channel at the same time; pad the row stride by `pad` groups (u64 rows: 8*pad B, vote rows: pad B)
  if (!only_n || only_n == 5) {
    const uint64_t G = 1ull << 20;
    const uint64_t pads[] = {0, 32, 64, 160, 288, 544, 2048, 2080, 6144, 10240 + 32, 34816};
    for (uint64_t pad : pads) {
      const uint64_t ld = G + pad;
      const int K = 26;
      for (int k = 0; k < K; ++k) sets.push_back(make_set(5, ld, 555 + k));
      CK(hipDeviceSynchronize());
      for (int rep = 0; rep < 2; ++rep) {
        for (int w = 0; w < 10; ++w) launch_reg<5, 4, false, true, 3, 256>(args_of(sets[w % sets.size()], ld), G, st);
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) launch_reg<5, 4, false, true, 3, 256>(args_of(sets[r % sets.size()], ld), G, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / reps;
        printf("{\"kernel\":\"stagger\",\"pad_groups\":%llu,\"rep\":%d,\"us\":%.3f,\"GBps\":%.1f}\n",
               (unsigned long long)pad, rep, us, G * 62.0 / us / 1e3);
      }
      fflush(stdout);
      for (auto& s : sets) free_set(s);
      sets.clear();
    }
  }

  // ---- G-curve: fixed per-launch overhead vs per-byte slope (N=5, commit+votes, NT, GPL=4) ----
  if (!only_n || only_n == 5) {
    for (int lg = 18; lg <= 24; ++lg) {
      const uint64_t G = 1ull << lg, ld = G;
      const double set_bytes = ld * (8.0 * 5 + 24 + 5 + 1 + 16);
      const int K = (int)(2.0 * 1024 * 1024 * 1024 / set_bytes) + 2;
      for (int k = 0; k < K; ++k) sets.push_back(make_set(5, ld, 777 + k));
      CK(hipDeviceSynchronize());
      const int r2 = lg >= 22 ? reps / 4 + 1 : reps;
      for (int w = 0; w < 10; ++w) launch_reg<5, 4, false, true, 3, 256>(args_of(sets[w % sets.size()], ld), G, st);
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < r2; ++r) launch_reg<5, 4, false, true, 3, 256>(args_of(sets[r % sets.size()], ld), G, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = 1e3 * ms / r2;
      printf("{\"kernel\":\"gcurve\",\"N\":5,\"G\":%llu,\"K\":%zu,\"us\":%.3f,\"GBps\":%.1f,\"Gdec_per_s\":%.2f}\n",
             (unsigned long long)G, sets.size(), us, G * 62.0 / us / 1e3, G / us / 1e3);
      fflush(stdout);
      if (lg == 20) {
        // independent batches on 2 and 4 streams: lets one launch's drain overlap the next one's ramp
        for (int ns = 2; ns <= 4; ns += 2) {
          hipStream_t ss[4];
          for (int i = 0; i < ns; ++i) CK(hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
          CK(hipDeviceSynchronize());
          hipEvent_t a0, a1;
          CK(hipEventCreate(&a0));
          CK(hipEventCreate(&a1));
          CK(hipEventRecord(a0, ss[0]));
          for (int i = 1; i < ns; ++i) CK(hipStreamWaitEvent(ss[i], a0, 0));
          for (int r = 0; r < reps; ++r)
            launch_reg<5, 4, false, true, 3, 256>(args_of(sets[r % sets.size()], ld), G, ss[r % ns]);
          for (int i = 1; i < ns; ++i) {
            hipEvent_t j;
            CK(hipEventCreate(&j));
            CK(hipEventRecord(j, ss[i]));
            CK(hipStreamWaitEvent(ss[0], j, 0));
          }
          CK(hipEventRecord(a1, ss[0]));
          CK(hipEventSynchronize(a1));
          CK(hipEventElapsedTime(&ms, a0, a1));
          const double us2 = 1e3 * ms / reps;
          printf("{\"kernel\":\"multistream\",\"streams\":%d,\"N\":5,\"G\":%llu,\"us_per_sweep\":%.3f,\"GBps\":%.1f}\n", ns,
                 (unsigned long long)G, us2, G * 62.0 / us2 / 1e3);
        }
      }
      for (auto& s : sets) free_set(s);
      sets.clear();
    }
  }
  return 0;
}
