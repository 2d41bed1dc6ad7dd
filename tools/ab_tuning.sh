#!/bin/bash
# tools/ab_env.sh BITS [bench args...] -- A/B of rc_opts.tuning bits (RC_TUNE_* in include/rc_abi.h) on one
# box, interleaved: three rounds of (default, --tuning BITS), frame_ms of each.
BITS=$1; shift
for round in 1 2 3; do
  for val in 0 $BITS; do
    python bench.py --steps 200 --warmup 20 --no-cpu --no-extras --tuning $val "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('tuning=$val', d['config']['workload'], d['config']['mode'][:3], d['roofline']['kernel'], round(d['frame_ms'],4))"
  done
done
