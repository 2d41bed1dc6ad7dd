#!/bin/bash
# tools/ab_ext.sh LIB... -- A/B of alternative builds of librc_hip.so on the EXT 4K workload.
ORIG=cudaraytracing_amd/librc_hip.so
cp $ORIG /tmp/librc_orig.so
for round in 1 2 3; do
  for lib in /tmp/librc_orig.so "$@"; do
    cp $lib $ORIG
    python bench.py --steps 60 --warmup 6 --shading ext --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib', 'ext', round(d['frame_ms'],4))"
  done
done
cp /tmp/librc_orig.so $ORIG
