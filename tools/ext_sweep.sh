#!/bin/bash
# tools/ext_sweep.sh -- EXT frame time with the dominant-axis shadow filter forced on (tuning 1) and off
# (tuning 2) over frame sizes and sphere counts; the launcher threshold RC_SHADOW_FAST_MIN comes from this.
for cfg in "3840 2160 256" "3840 2160 512" "1920 1080 512" "1920 1080 1024" "1920 1080 2048" "1280 768 1024" "1280 768 2048" "7680 4320 4096" "3840 2160 2048"; do
  for t in 1 2; do timeout -k 5 120 python tools/ext_sweep.py $cfg ext $t || exit 1; done
done
