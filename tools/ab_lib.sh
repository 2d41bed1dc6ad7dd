#!/bin/bash
# tools/ab_lib.sh LIB_A LIB_B [bench args...] -- A/B of two builds of librc_hip.so on one box, interleaved:
# three rounds of (A, B), frame_ms of each.  The library in the tree is put back at the end.
A=$1; B=$2; shift 2
ORIG=cudaraytracing_amd/librc_hip.so
cp $ORIG /tmp/librc_orig.so
for round in 1 2 3; do
  for lib in $A $B; do
    cp $lib $ORIG
    python bench.py --steps 200 --warmup 20 --no-cpu --no-extras "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib', d['config']['workload'], d['config']['mode'][:3], d['roofline']['kernel'], round(d['frame_ms'],4))"
  done
done
cp /tmp/librc_orig.so $ORIG
