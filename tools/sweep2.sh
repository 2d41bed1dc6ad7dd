#!/bin/bash
# tools/sweep2.sh -- pixels-per-lane sweep over the benchmark configurations.
for c in c1_640x480_n8 ref_1280x768_n512 c2_1920x1080_n64 c3_3840x2160_n1024 c4_7680x4320_n4096; do for v in tiled2 tiled4 tiled8; do
  python bench.py --steps 60 --warmup 10 --variant $v --config $c --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$c $v', round(d['value']), round(d['frame_ms'],4))"
done; done
