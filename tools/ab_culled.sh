#!/bin/bash
# tools/ab_culled.sh LIB... -- A/B of alternative builds of librc_hip.so with the culled variant (4K and 8K workloads).
ORIG=cudaraytracing_amd/librc_hip.so
cp $ORIG /tmp/librc_orig.so
for round in 1 2; do
  for lib in /tmp/librc_orig.so "$@"; do
    cp $lib $ORIG
    for c in c3_3840x2160_n1024 c4_7680x4320_n4096 c2_1920x1080_n64; do
      python bench.py --config $c --variant culled --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib', '$c', round(d['frame_ms'],4))"
    done
  done
done
cp /tmp/librc_orig.so $ORIG
