#!/bin/bash
# tools/ab_lib.sh LIB... -- A/B of alternative builds of librc_hip.so on the default workload (same box, interleaved).
ORIG=cudaraytracing_amd/librc_hip.so
cp $ORIG /tmp/librc_orig.so
for round in 1 2 3; do
  for lib in /tmp/librc_orig.so "$@"; do
    cp $lib $ORIG
    for v in tiled8; do
      python bench.py --steps 200 --warmup 20 --variant $v --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib', d['config']['variant'], round(d['frame_ms'],4))"
    done
  done
done
cp /tmp/librc_orig.so $ORIG
