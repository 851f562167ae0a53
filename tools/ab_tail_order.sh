#!/bin/bash
# ab_tail_order.sh -- round 6: the marker launch's tail with its ticket add as an agent-scope ACQ_REL (the
# hand-off formal) against the library of the commit before (relaxed ticket; pic1dp_amd/lib/v_prev.so built from a worktree
# of that commit with PIC1DP_LIB_OUT), alternating fresh processes: the 8-way share through every launch of the RCCL step
# with a one-rank communicator (tools/nrank_rehearsal.py part A).
for r in 1 2; do
  for v in new prev; do
    if [ $v = new ]; then unset PIC1DP_LIB; else export PIC1DP_LIB=$(cd "$(dirname "$0")/.." && pwd)/pic1dp_amd/lib/v_prev.so; fi
    echo "== run $r $v"
    python tools/nrank_rehearsal.py --no-two-ranks --steps 300 2>&1 | grep -E "^   1-rank RCCL, charge packed|^   1-rank RCCL, tail again|^   plain|^0\."
  done
done
