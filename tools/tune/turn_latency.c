/* The batching turn as a C caller (what cgo sees) makes it: raftq_stage_packed -> fill -> raftq_cycle_packed -> read the list in
 * place -- timed around the two calls of a turn, the producer's fill outside the clock.  bench.py's pipeline leg times the same
 * turn through the Python mirror, whose interpreter work (building ctypes arguments, ~6 us) is not the library's.
 * Workload = bench.py pipeline_measure's: 1M groups x 5 peers, per turn 21,845 groups (four rotating subsets) each acked by
 * a quorum of 3 peers at a value above everything seen -> 65,535 records in, 21,845 advances out.
 *   gcc -std=c99 -O2 -Iinclude tools/tune/turn_latency.c -Lraftsql_amd -lraftq -Wl,-rpath,$PWD/raftsql_amd -o tools/tune/turn_latency
 *   usage: turn_latency [turns]        -> one JSON line */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "raftq.h"

static double now_us(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

#define CHECK(call)                                                                        \
  do {                                                                                     \
    int rc_ = (call);                                                                      \
    if (rc_ != RAFTQ_OK) {                                                                 \
      fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, raftq_last_error(h));                 \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

int main(int argc, char** argv) {
  enum { N = 5, Q = 3, SUBSETS = 4 };
  const uint64_t G = 1u << 20, n_groups = 65536 / Q, nd = n_groups * Q;
  const int turns = argc > 1 ? atoi(argv[1]) : 300, warm = 20;
  raftq_t* h = NULL;
  if (raftq_create(0, G, N, &h) != RAFTQ_OK) {
    fprintf(stderr, "raftq_create: %s\n", raftq_last_error(NULL));
    return 1;
  }
  uint64_t* match = (uint64_t*)malloc(sizeof(uint64_t) * N * G);
  uint64_t* committed = (uint64_t*)malloc(sizeof(uint64_t) * G);
  uint32_t* subset = (uint32_t*)malloc(sizeof(uint32_t) * SUBSETS * n_groups);
  uint32_t* perm = (uint32_t*)malloc(sizeof(uint32_t) * G);
  if (!match || !committed || !subset || !perm) return 1;
  for (uint64_t g = 0; g < G; ++g) {
    committed[g] = 1000 + g % 977;
    for (int p = 0; p < N; ++p) match[(uint64_t)p * G + g] = committed[g]; /* (nothing to advance before the first ack) */
  }
  uint64_t x = 88172645463325252ull; /* xorshift: four random subsets of distinct groups */
  for (int v = 0; v < SUBSETS; ++v) {
    for (uint64_t g = 0; g < G; ++g) perm[g] = (uint32_t)g;
    for (uint64_t i = 0; i < n_groups; ++i) {
      x ^= x << 13, x ^= x >> 7, x ^= x << 17;
      const uint64_t j = i + x % (G - i);
      const uint32_t t = perm[i];
      perm[i] = perm[j], perm[j] = t;
      subset[(uint64_t)v * n_groups + i] = perm[i];
    }
  }
  CHECK(raftq_load_match(h, match, committed));
  double t_sum[2] = {0, 0};
  uint64_t adv_sum[2] = {0, 0};
  uint64_t c = 0;
  for (int form = 0; form < 2; ++form) { /* 0: the contiguous 16-byte list, 1: RAFTQ_CYCLE_SEGMENTED */
    const unsigned flags = RAFTQ_SWEEP_COMMIT | RAFTQ_CYCLE_TRUSTED | (form ? RAFTQ_CYCLE_SEGMENTED : 0u);
    for (int k = 0; k < turns + warm; ++k, ++c) {
      raftq_delta16_t* d = NULL;
      raftq_vote_delta_t* vd = NULL;
      CHECK(raftq_stage_packed(h, nd, 0, &d, &vd));
      const uint32_t* s = subset + (c % SUBSETS) * n_groups;
      for (uint64_t i = 0; i < n_groups; ++i) /* the producer (rafthttp's handlers), outside the clock */
        for (uint32_t p = 0; p < Q; ++p) {
          raftq_delta16_t r;
          r.match = 3000 + 16 * (c + 1);
          r.group = s[i];
          r.peer = p;
          d[i * Q + p] = r;
        }
      uint64_t n_adv = 0, seen = 0;
      const double t0 = now_us();
      CHECK(raftq_cycle_packed(h, d, nd, NULL, 0, flags, NULL, G, &n_adv, NULL));
      if (form) {
        const raftq_advance16_t* recs;
        const uint32_t* counts;
        uint32_t n_seg;
        uint64_t stride;
        CHECK(raftq_last_advance_segments(h, &recs, &counts, &n_seg, &stride));
        const double t1 = now_us();
        if (k >= warm) t_sum[form] += t1 - t0;
        for (uint32_t sg = 0; sg < n_seg; ++sg) seen += counts[sg];
      } else {
        const raftq_advance16_t* list;
        CHECK(raftq_last_advances_packed(h, &list, &seen));
        const double t1 = now_us();
        if (k >= warm) t_sum[form] += t1 - t0;
      }
      if (seen != n_adv || n_adv != n_groups) {
        fprintf(stderr, "turn %llu: %llu advanced, %llu listed, %llu expected\n", (unsigned long long)c, (unsigned long long)n_adv,
                (unsigned long long)seen, (unsigned long long)n_groups);
        return 1;
      }
      if (k >= warm) adv_sum[form] += n_adv;
    }
  }
  printf("{\"what\": \"the batching turn timed in a C caller around raftq_cycle_packed + the list accessor (RAFTQ_CYCLE_TRUSTED, zero-copy "
         "staging), %d turns each\", \"groups\": %llu, \"peers\": %d, \"deltas_per_turn\": %llu, \"advanced_per_turn\": %.1f, "
         "\"us_per_turn_contiguous_list\": %.2f, \"us_per_turn_segmented_list\": %.2f}\n",
         turns, (unsigned long long)G, N, (unsigned long long)nd, (double)adv_sum[1] / turns, t_sum[0] / turns, t_sum[1] / turns);
  raftq_destroy(h);
  return 0;
}
