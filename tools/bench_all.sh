# tools/bench_all.sh [ROUND] -- every bench line the documents quote, one file each under gpurun_out/<ROUND>_bench/
set -e
ROUND=${1:-r3}
mkdir -p gpurun_out/${ROUND}_bench
run() { name=$1; shift; timeout -k 10 280 python bench.py "$@" > gpurun_out/${ROUND}_bench/$name.json 2> gpurun_out/${ROUND}_bench/$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/${ROUND}_bench/$name.json")); r=d["roofline"]; c=d.get("cpu_baseline") or {}
print("$name", round(d["frame_ms"],4), "ms", round(d["value"]), "Mrays/s frac", round(r["frac"],3), "cpu", round(c.get("value",0),2), c.get("cores"))
PY
}
run bench_default
run bench_c5_seq --sequence --steps 300
run bench_c5_seq_ext --sequence --steps 300 --shading ext
run bench_c4_7680x4320_n4096 --config c4_7680x4320_n4096 --steps 50
run bench_c2_1920x1080_n64 --config c2_1920x1080_n64
run bench_c2_ext --config c2_1920x1080_n64 --shading ext
run bench_c1_640x480_n8 --config c1_640x480_n8
run bench_c1_ext --config c1_640x480_n8 --shading ext
run bench_c1_ext_1thread --config c1_640x480_n8 --shading ext --cpu-threads 1
run bench_c3_ext --shading ext
run bench_ref_1280x768_n512 --config ref_1280x768_n512
run bench_literal --variant literal --steps 30
run bench_readback_rgba8 --readback rgba8 --no-cpu
run bench_bands_1rank_c --mode bands --gather-impl c --no-cpu
run bench_c4_ext --config c4_7680x4320_n4096 --shading ext --steps 30 --no-cpu
run bench_graph_c1 --config c1_640x480_n8 --graph --sequence --steps 300 --no-cpu
