#!/bin/bash
# the completion word of a batching turn / a Tick + lists three ways, one box, back to back --
#   kernel  a one-thread kernel behind the last kernel (shipped since round 3)
#   packet  the runtime's stream write-value packet (hipStreamWriteValue64) behind the last kernel
#   arrive  no extra launch: every workgroup fences its stores system-wide, counts itself in, the LAST one to arrive raises the word
# Each mode: 2 x 8,000 turns under RAFTQ_CYCLE_CHECK=1 (what the host reads when the word lands must be what it reads after a full
# synchronisation; a mismatch fails the run), then the same turns timed without the check (tools/tune/turn_latency.c: a C caller),
# then the Tick + lists call (tools/probe/tick_lists_probe.py).     bash tools/probe/flag_ab.sh > profiles/r06/flag_ab.txt
for mode in kernel packet arrive; do
  echo "== RAFTQ_CYCLE_FLAG=$mode"
  echo -n "checked (16,000 turns): "
  if RAFTQ_CYCLE_FLAG=$mode RAFTQ_CYCLE_CHECK=1 timeout 600 tools/tune/turn_latency 8000 > /tmp/flag_chk.json 2> /tmp/flag_chk.err; then echo "ok (no turn's list differed from what a full synchronisation showed)"; else echo "FAILED rc=$? $(tail -2 /tmp/flag_chk.err)"; fi
  for rep in 1 2 3; do
    echo -n "timed: "; RAFTQ_CYCLE_FLAG=$mode timeout 300 tools/tune/turn_latency 3000 2> /tmp/flag_t.err | grep -o '"us_per_turn_contiguous_list.*' || tail -1 /tmp/flag_t.err
  done
  echo -n "tick + lists, checked: "; RAFTQ_CYCLE_FLAG=$mode RAFTQ_CYCLE_CHECK=1 timeout 300 python tools/probe/tick_lists_probe.py 1500 2>&1 | tail -1
  echo -n "tick + lists, timed:   "; RAFTQ_CYCLE_FLAG=$mode timeout 300 python tools/probe/tick_lists_probe.py 3000 2>&1 | tail -1
done
