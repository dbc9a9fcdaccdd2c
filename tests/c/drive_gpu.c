/* A C host (what cgo sees) driving the whole boundary on a GPU: sweep, sparse ingest + advance list, batched Step,
 * stream frames and a WAL segment, sweep sets and the packed batching turn -- no Python, no C++ in the caller.  Built with gcc -std=c99 and run by
 * tests/test_abi_gpu.py on the GPU box.  Expected values are worked out by hand below (3 groups x 3 peers). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raftq.h"
#include "raftq_step.h"
#include "raftq_wire.h"

#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      printf("FAILED line %d: %s (%s)\n", __LINE__, #cond, raftq_last_error(h)); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main(void) {
  raftq_t* h = NULL;
  enum { G = 3, N = 3 };
  /* peer-major match[N][G]; quorum 2 -> the 2nd largest per group: {5, 7, 2} */
  uint64_t match[N * G] = {5, 9, 2, /* peer 0 */ 5, 7, 1, /* peer 1 */ 1, 3, 2 /* peer 2 */};
  uint64_t committed[G] = {3, 7, 4};
  uint64_t out[G];
  uint8_t votes[N * G] = {1, 1, 1, /* peer 0 */ 1, 2, 0, /* peer 1 */ 0, 2, 0 /* peer 2 */};
  uint8_t outcome[G];
  raftq_counts_t counts;
  uint64_t n_changed = 0;
  CHECK(raftq_create(0, G, N, &h) == RAFTQ_OK);
  CHECK(raftq_load_match(h, match, committed) == RAFTQ_OK);
  CHECK(raftq_load_votes(h, votes) == RAFTQ_OK);
  CHECK(raftq_commit_advance(h, 0, out, &n_changed) == RAFTQ_OK);
  CHECK(out[0] == 5 && out[1] == 7 && out[2] == 4 && n_changed == 1); /* stale peers never move a commit back */
  CHECK(raftq_vote_tally(h, outcome, &counts) == RAFTQ_OK);
  CHECK(outcome[0] == 1 && outcome[1] == 2 && outcome[2] == 0 && counts.n_won == 1 && counts.n_lost == 1);

  /* an ack raises peer 2 of group 2 to 9: quorum index becomes 2 -> no; raise peer 1 too -> 2nd largest 9 */
  {
    raftq_delta_t d[2];
    raftq_advance_t adv[G];
    uint64_t n_adv = 0;
    memset(d, 0, sizeof d);
    d[0].group = 2, d[0].peer = 2, d[0].match = 9;
    d[1].group = 2, d[1].peer = 1, d[1].match = 9;
    CHECK(raftq_cycle(h, d, 2, NULL, 0, RAFTQ_SWEEP_COMMIT, adv, G, &n_adv, NULL) == RAFTQ_OK);
    CHECK(n_adv == 1 && adv[0].group == 2 && adv[0].old_commit == 4 && adv[0].new_commit == 9);
    d[0].group = 7; /* out of range: refused, nothing applied */
    CHECK(raftq_apply_deltas(h, d, 2) == RAFTQ_EINVAL);
  }

  /* batched Step: group 0 campaigns (MsgHup), wins with one more vote, commits its empty entry on one ack */
  {
    raftq_msg_t m[3];
    raftq_step_out_t o[3];
    raftq_step_counts_t sc;
    uint64_t zero[G] = {0, 0, 0};
    CHECK(raftq_load_match(h, NULL, zero) == RAFTQ_OK);
    memset(m, 0, sizeof m);
    m[0].group = 0, m[0].type = RAFTQ_MSG_HUP;
    m[1].group = 0, m[1].type = RAFTQ_MSG_VOTE_RESP, m[1].term = 1, m[1].from = 1;
    m[2].group = 0, m[2].type = RAFTQ_MSG_APP_RESP, m[2].term = 1, m[2].from = 2, m[2].index = 1;
    CHECK(raftq_set_self(h, 0) == RAFTQ_OK);
    CHECK(raftq_step_batch(h, m, 3, o, &sc) == RAFTQ_OK);
    CHECK(sc.n_msgs == 3 && sc.n_groups_touched == 1);
    CHECK(o[0].type == RAFTQ_OUT_CAMPAIGN && o[0].term == 1 && o[0].role == RAFTQ_ROLE_CANDIDATE);
    CHECK(o[1].type == RAFTQ_OUT_BECAME_LEADER && o[1].index == 1 && o[1].role == RAFTQ_ROLE_LEADER);
    CHECK(o[2].type == RAFTQ_OUT_PROGRESS && o[2].commit == 1 && (o[2].flags & RAFTQ_OUTF_COMMITTED));
  }

  /* the formats either side: three messages -> stream frames -> back; two WAL records with their CRC chain */
  {
    raftq_wire_msg_t wm[3], back[3];
    raftq_wire_ent_t we[1], eback[2];
    raftq_wire_counts_t wc;
    unsigned char stream[512], wal[512];
    uint64_t off[4], woff[4], nf = 0, used = 0;
    raftq_wal_rec_t r[3], rb[3];
    raftq_wal_counts_t lc;
    memset(wm, 0, sizeof wm);
    memset(we, 0, sizeof we);
    wm[0].type = RAFTQ_MSG_APP, wm[0].group = 2, wm[0].term = 7, wm[0].from = 0, wm[0].to = 1, wm[0].index = 4, wm[0].log_term = 7;
    wm[0].n_ents = 1, wm[0].ent_first = 0;
    we[0].term = 7, we[0].index = 5, we[0].data_len = 5, we[0].data_off = 0;
    wm[1].type = RAFTQ_MSG_HEARTBEAT, wm[1].group = 1, wm[1].term = 7, wm[1].to = 2, wm[1].commit = 3;
    wm[2].type = RAFTQ_MSG_VOTE_RESP, wm[2].term = 8, wm[2].from = 2, wm[2].to = 0, wm[2].reject = 1;
    CHECK(raftq_wire_encode(h, wm, 3, we, 1, "hello", 5, stream, sizeof stream, off, &wc) == RAFTQ_OK);
    CHECK(wc.bytes == off[3] && off[0] == 0);
    CHECK(raftq_wire_scan_frames(stream, wc.bytes, 1, woff, 3, &nf, &used) == RAFTQ_OK && nf == 3 && used == wc.bytes);
    CHECK(memcmp(off, woff, sizeof off) == 0);
    CHECK(raftq_wire_decode(h, stream, wc.bytes, off, 3, back, eback, 2, &wc) == RAFTQ_OK);
    CHECK(wc.n_malformed == 0 && wc.n_ents == 1 && back[0].n_ents == 1 && back[0].group == 2 && back[0].to == 1);
    CHECK(back[1].commit == 3 && back[2].reject == 1 && back[2].term == 8 && back[2].from == 2);
    CHECK(eback[0].data_len == 5 && memcmp(stream + eback[0].data_off, "hello", 5) == 0 && eback[0].index == 5);
    memset(r, 0, sizeof r);
    r[0].kind = RAFTQ_WAL_CRC;
    r[1].kind = RAFTQ_WAL_ENTRY, r[1].group = 2, r[1].term = 7, r[1].index = 5, r[1].data_len = 5;
    r[2].kind = RAFTQ_WAL_STATE, r[2].group = 2, r[2].term = 7, r[2].vote = 1, r[2].index = 5;
    CHECK(raftq_wal_encode(h, r, 3, "hello", 5, 0, wal, sizeof wal, woff, &lc) == RAFTQ_OK);
    CHECK(raftq_wal_decode(h, wal, lc.bytes, woff, 3, 0, rb, &lc) == RAFTQ_OK && lc.n_valid == 3);
    CHECK(rb[1].kind == RAFTQ_WAL_ENTRY && rb[1].index == 5 && rb[2].vote == 1 && rb[2].index == 5);
    wal[woff[1] + 14] ^= 1; /* a flipped bit in the entry: its CRC no longer matches */
    CHECK(raftq_wal_decode(h, wal, lc.bytes, woff, 3, 0, rb, &lc) == RAFTQ_OK && lc.n_valid == 1);
    CHECK((rb[1].flags & (RAFTQ_WAL_F_BADCRC | RAFTQ_WAL_F_MALFORMED)) != 0);
  }
  /* the round-2 inbound forms of Step, on the leader group 0 just elected (term 1, last index 1, committed 1):
   *   packed 40-byte records staged in place: a heartbeat response, then a higher-term vote request the leader grants
   *   (its log (1, 1) is not ahead of the candidate's (index 1, term 1)) -> steps down, term 2, votes for peer 1;
   *   frames staged in place (raftq_step_stage_wire): a stale MsgAppResp of term 1 -> ignored by the follower.
   * Three batches may be in flight: both are submitted before the first collect. */
  {
    raftq_msg40_t* pm = NULL;
    raftq_wire_msg_t wm[1];
    raftq_wire_counts_t wc;
    unsigned char tmp[128];
    uint64_t toff[2], *soff = NULL;
    void* sstream = NULL;
    const raftq_step_out_t* res = NULL;
    uint64_t nres = 0;
    raftq_step_counts_t sc;
    CHECK(raftq_step_stage_packed(h, 2, &pm) == RAFTQ_OK && pm != NULL);
    memset(pm, 0, 2 * sizeof *pm);
    pm[0].group = 0, pm[0].type = RAFTQ_MSG_HEARTBEAT_RESP, pm[0].term = 1, pm[0].from = 2;
    pm[1].group = 0, pm[1].type = RAFTQ_MSG_VOTE, pm[1].term = 2, pm[1].from = 1, pm[1].index = 1, pm[1].aux = 1; /* aux = LogTerm */
    CHECK(raftq_step_submit_packed(h, pm, 2) == RAFTQ_OK);
    memset(wm, 0, sizeof wm);
    wm[0].type = RAFTQ_MSG_APP_RESP, wm[0].group = 0, wm[0].term = 1, wm[0].from = 2, wm[0].to = 0, wm[0].index = 1;
    CHECK(raftq_commit_advance(h, 0, NULL, NULL) == RAFTQ_ESTATE); /* state calls wait for the collects */
    CHECK(raftq_step_collect(h, NULL, &sc) == RAFTQ_OK && sc.n_msgs == 2 && sc.n_groups_touched == 1);
    CHECK(raftq_step_results(h, &res, &nres) == RAFTQ_OK && nres == 2);
    CHECK(res[0].type == RAFTQ_OUT_PROGRESS && res[0].role == RAFTQ_ROLE_LEADER && res[0].term == 1);
    CHECK(res[1].type == RAFTQ_OUT_VOTE_RESP && res[1].reject == 0 && res[1].term == 2 && res[1].vote == 2 &&
          res[1].role == RAFTQ_ROLE_FOLLOWER && (res[1].flags & RAFTQ_OUTF_STEPPED_DOWN) && (res[1].flags & RAFTQ_OUTF_HARDSTATE));
    CHECK(raftq_wire_encode(h, wm, 1, NULL, 0, NULL, 0, tmp, sizeof tmp, toff, &wc) == RAFTQ_OK);
    CHECK(raftq_step_stage_wire(h, 4, 256, &soff, &sstream) == RAFTQ_OK && soff != NULL && sstream != NULL);
    soff[0] = toff[0], soff[1] = toff[1];
    memcpy(sstream, tmp, (size_t)wc.bytes);
    CHECK(raftq_step_submit_wire(h, sstream, wc.bytes, soff, 1) == RAFTQ_OK);
    CHECK(raftq_step_collect(h, NULL, &sc) == RAFTQ_OK && sc.n_msgs == 1);
    CHECK(raftq_step_results(h, &res, &nres) == RAFTQ_OK && nres == 1);
    CHECK(res[0].type == RAFTQ_OUT_NONE && res[0].term == 2 && res[0].role == RAFTQ_ROLE_FOLLOWER);
  }
  /* sweep sets: two handles of one shape, ONE dispatch; then a batching turn in the packed 16-byte records written
   * in place into the handle's ack buffer (device memory behind a large BAR).  Hand-derived, 2 groups x 3 peers:
   *   a: match {4,8 | 4,2 | 1,8}  committed {1,5}  -> 2nd largest {4,8}: both advance
   *   b: match {6,3 | 2,3 | 6,9}  committed {6,1}  -> 2nd largest {6,3}: group 1 advances
   *   votes a: group 0 {1,1,0} won, group 1 {1,2,2} lost;  b: all pending */
  {
    raftq_t *a = NULL, *b = NULL, *both[2];
    raftq_set_t* set = NULL;
    uint64_t ma[6] = {4, 8, 4, 2, 1, 8}, ca[2] = {1, 5}, mb[6] = {6, 3, 2, 3, 6, 9}, cb[2] = {6, 1}, got[2];
    uint8_t va[6] = {1, 1, 1, 2, 0, 2}, vb[6] = {1, 1, 0, 0, 0, 0}, oc[2];
    raftq_counts_t per[2], tot;
    raftq_delta16_t* acks = NULL;
    const raftq_advance16_t* list = NULL;
    uint64_t n_adv = 0, n_listed = 0;
    CHECK(raftq_create(0, 2, 3, &a) == RAFTQ_OK && raftq_create(0, 2, 3, &b) == RAFTQ_OK);
    CHECK(raftq_load_match(a, ma, ca) == RAFTQ_OK && raftq_load_votes(a, va) == RAFTQ_OK);
    CHECK(raftq_load_match(b, mb, cb) == RAFTQ_OK && raftq_load_votes(b, vb) == RAFTQ_OK);
    both[0] = a, both[1] = b;
    CHECK(raftq_set_create(both, 2, &set) == RAFTQ_OK && raftq_set_size(set) == 2);
    CHECK(raftq_set_stream(a, NULL) == RAFTQ_ESTATE); /* the set owns its members' stream */
    CHECK(raftq_set_sweep_async(set, RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_VOTES) == RAFTQ_OK);
    CHECK(raftq_set_wait(set, per, &tot) == RAFTQ_OK);
    CHECK(per[0].n_changed == 2 && per[0].n_won == 1 && per[0].n_lost == 1 && per[1].n_changed == 1 && per[1].n_won == 0);
    CHECK(tot.n_changed == 3 && tot.n_won == 1 && tot.n_lost == 1);
    CHECK(raftq_read_committed(a, got) == RAFTQ_OK && got[0] == 4 && got[1] == 8);
    CHECK(raftq_read_committed(b, got) == RAFTQ_OK && got[0] == 6 && got[1] == 3);
    CHECK(raftq_read_outcome(a, oc) == RAFTQ_OK && oc[0] == 1 && oc[1] == 2);
    CHECK(raftq_set_mode(set, RAFTQ_SET_PERSISTENT, 2) == RAFTQ_OK); /* same answers from the resident walk */
    CHECK(raftq_set_sweep_async(set, RAFTQ_SWEEP_COMMIT) == RAFTQ_OK && raftq_set_wait(set, NULL, &tot) == RAFTQ_OK && tot.n_changed == 0);
    /* a turn on member b between set sweeps: peers 0 and 1 of group 0 acknowledge 7 -> 2nd largest 7 > 6 */
    CHECK(raftq_stage_packed(b, 2, 0, &acks, NULL) == RAFTQ_OK && acks != NULL);
    acks[0].group = 0, acks[0].peer = 0, acks[0].match = 7;
    acks[1].group = 0, acks[1].peer = 1, acks[1].match = 7;
    CHECK(raftq_cycle_packed(b, acks, 2, NULL, 0, RAFTQ_SWEEP_COMMIT | RAFTQ_CYCLE_TRUSTED, NULL, 2, &n_adv, NULL) == RAFTQ_OK);
    CHECK(n_adv == 1 && raftq_last_advances_packed(b, &list, &n_listed) == RAFTQ_OK && n_listed == 1);
    CHECK(list[0].group == 0 && list[0].new_commit == 7 && list[0].advanced_by == 1);
    acks[1].peer = 3; /* not a peer: without TRUSTED the whole turn is refused and nothing moves */
    acks[0].match = 50;
    CHECK(raftq_cycle_packed(b, acks, 2, NULL, 0, RAFTQ_SWEEP_COMMIT, NULL, 2, &n_adv, NULL) == RAFTQ_EINVAL && n_adv == 0);
    CHECK(raftq_read_committed(b, got) == RAFTQ_OK && got[0] == 7 && got[1] == 3);
    raftq_set_destroy(set); /* members get streams of their own back */
    CHECK(raftq_commit_advance(a, 0, got, &n_adv) == RAFTQ_OK && n_adv == 0 && got[0] == 4);
    raftq_destroy(a);
    raftq_destroy(b);
  }
  raftq_destroy(h);
  printf("C-HOST-GPU-OK\n");
  return 0;
}
