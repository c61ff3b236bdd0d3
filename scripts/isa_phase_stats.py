#!/usr/bin/env python3
"""dev tool (CPU only): per-phase instruction mix of the default stage-1 kernel's gfx950 ISA — how much of each phase's
VALU work is SGPR-spill traffic (v_writelane / v_readlane into VGPR lanes)?  VERDICT r2 #8 / weak #10: "78 SGPR spills
cost nothing" should be a number.

The tile loop has three workgroup barriers: T (ticket known), A (tile counted), B (held tile resolved).  Regions:
  top     loop top -> barrier T          token + ticket
  count   barrier T -> barrier A         LDS-DMA loads, classification, masks  (straight-line: static = per tile)
  resolve barrier A -> barrier B         publish, look-back issue, speculative scatter, resolve
  emit    barrier B -> loop back-edge    flush / emit_span, hand-over of the held tile
  finish  after the loop                 last-workgroup bookkeeping
Static counts; the count phase is fully unrolled straight-line code, so its static count IS its per-tile dynamic count.
The scatter loops of `resolve` / `emit` run once per set bit: lane moves sit outside them (reported separately).
usage: python scripts/isa_phase_stats.py [mangled-kernel-substring]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1] if len(sys.argv) > 1 else "stage1_kernelILb1ELi0ELi0ELb0EE"
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "csv-simd_amd", "csrc"), "-S", "--cuda-device-only",
                    os.path.join(ROOT, "csv-simd_amd", "csrc", "stage1_kernels.hip"), "-o", out], check=True,
                   stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN7csvsimd13") and want in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
barriers = [i for i, l in enumerate(body) if re.match(r"\s+s_barrier", l)]
assert len(barriers) == 3, barriers
bT, bA, bB = barriers
# loop header: target of the LAST backward branch that jumps from behind barrier B to before barrier T
back = [(i, labels[m.group(1)]) for i, l in enumerate(body)
        if (m := re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)) and m.group(1) in labels and labels[m.group(1)] < bT < bB < i]
loop_end, loop_top = max(back)
regions = [("top (token + ticket)", loop_top, bT), ("count", bT, bA), ("resolve (+ speculative scatter)", bA, bB),
           ("emit (+ hand-over)", bB, loop_end), ("finish (after the loop)", loop_end, len(body))]
def kind(op):
    if op in ("v_readlane_b32", "v_writelane_b32"): return "lane_move"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    return "other"
# innermost data-dependent loops (one trip per set bit): blocks that branch back to themselves
def inner_loop_lines(a, b):
    inner = set()
    for i in range(a, b):
        m = re.match(r"\s+s_cbranch\w*\s+(\.LBB\d+_\d+)", body[i])
        if m and m.group(1) in labels and a <= labels[m.group(1)] < i and i - labels[m.group(1)] < 40:
            inner.update(range(labels[m.group(1)], i + 1))
    return inner
print(f"kernel {body[0].split(':')[0]}: {len(body)} lines; loop top line {loop_top}, barriers T/A/B at {bT}/{bA}/{bB}, back-edge at {loop_end}")
print(f"{'region':34s} {'instr':>6s} {'VALU':>6s} {'lane moves':>10s} {'= % of VALU+moves':>18s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s}  lane moves inside per-bit loops")
for name, a, b in regions:
    c = collections.Counter()
    inner = inner_loop_lines(a, b)
    moves_inner = 0
    for i in range(a, b):
        m = re.match(r"\s+([a-z][a-z0-9_]+)", body[i])
        if not m: continue
        k = kind(m.group(1))
        c[k] += 1
        if k == "lane_move" and i in inner: moves_inner += 1
    tot = sum(c.values())
    v = c["valu"] + c["lane_move"]
    print(f"{name:34s} {tot:6d} {c['valu']:6d} {c['lane_move']:10d} {100.0 * c['lane_move'] / max(v, 1):17.1f}% {c['salu']:6d} {c['lds']:5d} {c['vmem']:5d}  {moves_inner}")
