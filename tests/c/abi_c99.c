/* The drop-in boundary must be consumable by a plain C compiler (cgo compiles the preamble as C):
 * every public header, C99, pedantic.  Linked against libraftq.so and run by tests/test_abi.py. */
#include <stdio.h>
#include <string.h>

#include "raftq.h"
#include "raftq_node.h"
#include "raftq_pipe.h"
#include "raftq_step.h"
#include "raftq_wire.h"

int main(void) {
  raftq_t* h = NULL;
  raftq_node_t* n = NULL;
  raftq_msg_t m;
  raftq_step_out_t o;
  int ndev = -1;
  memset(&m, 0, sizeof m);
  memset(&o, 0, sizeof o);
  if (sizeof m != 64 || sizeof o != 64 || sizeof(raftq_log_delta_t) != 32 || sizeof(raftq_delta_t) != 24 ||
      sizeof(raftq_vote_delta_t) != 16 || sizeof(raftq_advance_t) != 24) {
    printf("struct sizes differ from the ABI\n");
    return 1;
  }
  if (raftq_abi_version() != RAFTQ_ABI_VERSION) return 2;
  if (raftq_quorum(5) != 3) return 3;
  if (raftq_create(0, 0, 3, &h) != RAFTQ_EINVAL || h != NULL) return 4;
  if (raftq_node_create(0, 8, 3, 3, &n) != RAFTQ_EINVAL || n != NULL) return 5; /* self_peer out of range */
  if (raftq_step_batch(NULL, &m, 1, &o, NULL) != RAFTQ_EINVAL) return 6;
  if (sizeof(raftq_wire_msg_t) != 64 || sizeof(raftq_wire_ent_t) != 32 || sizeof(raftq_wal_rec_t) != 48) return 7;
  {
    /* the one host-only codec entry point: the length-word walk over two whole frames and a torn tail */
    const unsigned char buf[8 + 2 + 8 + 0 + 8] = {0, 0, 0, 0, 0, 0, 0, 2, 0xaa, 0xbb, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 9};
    uint64_t off[4], nf = 0, used = 0;
    if (raftq_wire_scan_frames(buf, sizeof buf, 1, off, 3, &nf, &used) != RAFTQ_OK) return 8;
    if (nf != 2 || used != 18 || off[0] != 0 || off[1] != 10 || off[2] != 18) return 9;
  }
  if (raftq_wal_decode(NULL, NULL, 0, NULL, 0, 0, NULL, NULL) != RAFTQ_EINVAL) return 10;
  (void)raftq_device_count(&ndev);
  printf("C99-ABI-OK devices=%d err=\"%s\"\n", ndev, raftq_last_error(NULL));
  return 0;
}
