#!/bin/bash
# tools/ab_env.sh VAR [bench args...] -- A/B of a default-on tuning switch of librc_hip.so (VAR=0 turns it
# off) on one box, interleaved: three rounds of (on, off), frame_ms of each.
VAR=$1; shift
for round in 1 2 3; do
  for val in 1 0; do
    env $VAR=$val python bench.py --steps 200 --warmup 20 --no-cpu --no-extras "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$VAR=$val', d['config']['workload'], d['config']['mode'][:3], d['roofline']['kernel'], round(d['frame_ms'],4))"
  done
done
