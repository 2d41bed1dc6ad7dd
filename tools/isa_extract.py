"""Inner loops of the draw kernels as the compiler emits them for gfx950, with an instruction histogram: the
evidence behind the per-test instruction counts the kernels' comments and the bench line quote.

    python tools/isa_extract.py [out_dir]        (default: profiles/, files r3_isa_*.txt)

Compiles csrc/rc_kernels.hip to assembly with the library's own flags (cudaraytracing_amd/build.py), cuts each
kernel into basic blocks at its labels and prints the block richest in v_pk_fma_f32 -- the unrolled body of the
test loop -- for: the 8-pixel REF kernel (primary loop), the 4-pixel EXT kernel with the dominant-axis shadow
filter (its shadow loop is the second-richest block after the primary one) and the 4-pixel REF kernel."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudaraytracing_amd import build  # noqa: E402

out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "k.s")
    subprocess.run([build.hipcc_path()] + build.HIPCC_FLAGS + ["-S", "--cuda-device-only", "-o", asm,
                                                                os.path.join(build.CSRC, "rc_kernels.hip")], check=True)
    text = open(asm).read()


def kernel(sym):
    m = re.search(r"^%s:.*?\.end_amdhsa_kernel" % re.escape(sym), text, re.S | re.M)
    return m.group(0)


def blocks(body):
    out, cur, name = [], [], "entry"
    for ln in body.split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", ln):
            out.append((name, cur))
            name, cur = ln.split(":")[0], []
        elif re.match(r"^\s+[a-z_0-9]+", ln) and not ln.strip().startswith((".", ";")):
            cur.append(ln.rstrip())
    out.append((name, cur))
    return out


def report(sym, title, rank, fname, tests_per_lane, no_mul=False):
    body = kernel(sym)
    bl = blocks(body)
    if no_mul:          # the dominant-axis form has no product outside its FMAs (the three-product form has)
        bl = [b for b in bl if not any("v_pk_mul_f32" in i for i in b[1])]
    bl = sorted(bl, key=lambda b: -sum("v_pk_fma_f32" in i for i in b[1]))
    name, ins = bl[rank]
    for k, i in enumerate(ins):                 # the loop body ends at its first branch; what follows in the
        if i.split()[0].startswith("s_cbranch"):  # same label-to-label stretch is the fall-through of the rare path
            ins = ins[:k + 1]
            break
    hist = collections.Counter(i.split()[0] for i in ins)
    valu = sum(c for op, c in hist.items() if op.startswith("v_"))
    meta = {k: re.search(r"\.%s:\s+(\d+)" % k, text[text.index(".name:           " + sym):][:2000]) for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size")}
    with open(os.path.join(out_dir, fname), "w") as f:
        f.write(f"# {title}\n# kernel {sym}, basic block {name}: {len(ins)} instructions, {valu} VALU\n")
        f.write("# " + ", ".join(f"{k} {v.group(1)}" for k, v in meta.items() if v) + "\n")
        f.write(f"# (pixel, sphere) tests per lane in this block: {tests_per_lane} -> {valu / tests_per_lane:.3f} VALU instructions per test\n")
        f.write("# histogram: " + ", ".join(f"{op} {c}" for op, c in hist.most_common()) + "\n\n")
        f.write("\n".join(ins) + "\n")
    print(fname, name, len(ins), "instructions,", valu, "VALU,", f"{valu / tests_per_lane:.3f} per test")


report("_Z13rc_draw_tiledILi8ELb0ELb0ELb0ELb0EEv10RcKernArgs",
       "primary test loop, 8 pixels per lane (3840x2160 x 1024 and larger): one block of 32 spheres", 0,
       "r3_isa_ref_p8_primary_block.txt", 32 * 8)
report("_Z13rc_draw_tiledILi4ELb0ELb0ELb0ELb0EEv10RcKernArgs",
       "primary test loop, 4 pixels per lane (1920x1080): one block of 32 spheres", 0,
       "r3_isa_ref_p4_primary_block.txt", 32 * 4)
report("_Z13rc_draw_tiledILi4ELb1ELb0ELb0ELb1EEv10RcKernArgs",
       "EXT shadow loop, dominant-axis filter, 4 pixels per lane: four groups of 4 spheres (the loop is unrolled by 4 of 8)", 1,
       "r3_isa_ext_p4_shadow_axis_loop.txt", 16 * 4, no_mul=True)
