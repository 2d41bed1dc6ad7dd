#!/bin/bash
# tools/profile_all.sh ROUND -- the round's counter set: tools/profile.sh for every workload whose counters the
# documents quote, then the summaries (gpurun_out/pmc_<ROUND>_*.txt) and the merged profiles/counters.json.
R=${1:-r3}
set -e
run() { tag=$1; shift; bash tools/profile.sh ${R}_$tag "$@"; python tools/pmc_summary.py gpurun_out/prof_${R}_$tag --update profiles/counters.json --round $R > gpurun_out/${R}_pmc_$(echo "$1|$2|$3" | tr '|' '_').txt; }
run c2_ref c2_1920x1080_n64 ref tiled
run c2_ext c2_1920x1080_n64 ext tiled
run c1_ref c1_640x480_n8 ref tiled
run c1_ext c1_640x480_n8 ext tiled
run c3_ref c3_3840x2160_n1024 ref tiled
run c3_ext c3_3840x2160_n1024 ext tiled
run c3_culled c3_3840x2160_n1024 ref culled
run c4_ref c4_7680x4320_n4096 ref tiled
cp profiles/counters.json gpurun_out/${R}_counters.json
