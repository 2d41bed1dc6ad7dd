#!/bin/bash
# tools/store_profiles.sh ROUND TRACE_DIR -- copies what tools/bench_all.sh, tools/profile_all.sh, tools/count_tests.py and the
# rocprofv3 trace of the default bench command left under gpurun_out/ into profiles/ under the names the documents cite.
R=${1:-r3}; T=${2:-gpurun_out/${R}_trace_default}
for f in gpurun_out/${R}_bench/*.json; do cp $f profiles/${R}_$(basename $f); done
cp gpurun_out/${R}_bench_default_under_rocprof.json profiles/
cp $T/*/*_kernel_stats.csv profiles/${R}_kernel_trace_stats_bench_default.csv
rm -rf /tmp/ptrace; mkdir -p /tmp/ptrace/trace; cp -r $T/* /tmp/ptrace/trace/; cp gpurun_out/${R}_bench_default_under_rocprof.json /tmp/ptrace/trace.log
echo "c3_3840x2160_n1024|ref|tiled" > /tmp/ptrace/workload.txt
python tools/pmc_summary.py /tmp/ptrace --round $R | sed "s#/tmp/ptrace#$T#" > profiles/${R}_trace_summary_bench_default.txt
cp gpurun_out/${R}_pmc_*.txt profiles/
cp gpurun_out/${R}_counters.json profiles/counters.json
cp gpurun_out/executed_tests.json profiles/executed_tests.json
grep "timed launches\|bench line" profiles/${R}_trace_summary_bench_default.txt | cut -c1-200
