"""Markdown table of the bench lines under profiles/<ROUND>_bench_*.json (the measurement table of the design
document is pasted from this):    python tools/bench_table.py [ROUND]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r3"
counters = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))


def L(name):
    return json.load(open(os.path.join(ROOT, "profiles", f"{R}_bench_{name}.json")))


def num(x):
    return f"{x:,.0f}".replace(",", " ")


def row(label, name, key=None, cpu_note=""):
    x = L(name)
    c = counters.get(key, {}) if key else {}
    busy = f"{c['valu_busy']:.2f}" if c.get("valu_busy") else "—"
    insts = (f"{c['insts_valu'] / 1e6:.0f} M" if c.get("insts_valu", 0) >= 1e8 else f"{c['insts_valu'] / 1e6:.3g} M") if c.get("insts_valu") else "—"
    cb = x.get("cpu_baseline")
    cpu = (f"{cb['value']:.3g}" + cpu_note) if cb else "—"
    return f"| {label} | {x['frame_ms']:.4g} | {num(x['value'])} | {x['roofline']['frac']:.2f} | {busy} | {insts} | {cpu} |"


print("| workload | frame ms | Mrays/s | `roofline.frac` (executed work) | VALU busy | VALU instructions per launch | CPU oracle, 16 threads (Mrays/s) |")
print("|---|---|---|---|---|---|---|")
print(row("3840×2160 × 1024 (headline)", "default", "c3_3840x2160_n1024|ref|tiled", " (784 centre rows)"))
print(row("3840×2160 × 1024, camera path (`--sequence`, 300 frames)", "c5_seq"))
print(row("7680×4320 × 4096", "c4_7680x4320_n4096", "c4_7680x4320_n4096|ref|tiled", " (96 centre rows)"))
print(row("1920×1080 × 64", "c2_1920x1080_n64", "c2_1920x1080_n64|ref|tiled"))
print(row("1280×768 × 512 (the reference's own window)", "ref_1280x768_n512"))
print(row("640×480 × 8", "c1_640x480_n8", "c1_640x480_n8|ref|tiled"))
print(row("EXT 1920×1080 × 64 + plane (BASELINE `configs[1]`)", "c2_ext", "c2_1920x1080_n64|ext|tiled"))
one = L("c1_ext_1thread")["cpu_baseline"]["value"]
print(row("EXT 640×480 × 8 + plane (BASELINE `configs[0]`)", "c1_ext", "c1_640x480_n8|ext|tiled", f" ({one:.2g} on one thread)"))
print(row("EXT 3840×2160 × 1024 + plane", "c3_ext", "c3_3840x2160_n1024|ext|tiled"))
print(row("EXT 3840×2160 × 1024, camera + light path (300 frames; shadow loop priced with hit pixels × spheres: ≤)", "c5_seq_ext"))
print(row("EXT 7680×4320 × 4096 + plane", "c4_ext"))
print(row("literal kernel, 3840×2160 × 1024", "literal"))
cv = L("default")["culled_variant"]
cc = counters.get("c3_3840x2160_n1024|ref|culled", {})
print(f"| culled variant, 3840×2160 × 1024 (opt-in, not `value`) | {cv['frame_ms']:.4f} | {num(cv['value'])} | — (no roofline block: "
      f"instruction-bound, not one test per pair) | {cc.get('valu_busy', 0):.2f} | {cc.get('insts_valu', 0) / 1e6:.3g} M | — |")
b1 = L("bands_1rank_c")
print(f"| `--mode bands` on one rank, C gather, 2 pieces + device copy | {b1['frame_ms']:.4g} | {num(b1['value'])} | — | — | — | — |")
g1 = L("graph_c1")
print(f"| 640×480 × 8 camera path replayed as one HIP graph of 300 frames | {g1['graph_replay']['wall_ms_per_frame']:.4f} wall per frame "
      f"({g1['frame_ms']:.4f} launched one by one) | {num(g1['graph_replay']['value'])} | — | — | — | — |")
