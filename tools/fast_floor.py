#!/usr/bin/env python3
"""tools/fast_floor.py -- issue floor of k_fast<608, 256> from its gfx950 ISA.

Step 1 (here, no GPU): compile csrc/orb_kernels.hip to device assembly with the library's flags, cut out k_fast<608, 256>, split it
into basic blocks, find the loops (backward branches) and histogram every block's instructions by ISSUE CLASS, with the costs
tools/ubench.hip measured on MI355X at the occupancy this kernel runs at (profiles/r02_ubench_instruction_classes.txt):
    fast  2.26 cycles per wave64 instruction per SIMD: VOP2 forms of v_add/sub/subrev_u32, v_and/or/xor_b32, v_not_b32, v_lshrrev_b32, v_ashrrev_i32,
          v_mov_b32, v_add/sub/mul_f32, v_fmac_f32, v_add/sub_u16, v_min/max_i16/u16
    slow  4.5 cycles: every other vector instruction (VOP3 encodings of the above, compares, v_cndmask, v_bcnt, v_mbcnt, v_mul_*, v_perm, SDWA / DPP forms,
          v_readlane / v_readfirstlane, conversions, v_lshlrev_b32, 32-bit min / max ...)
    LDS   one ds_* instruction occupies the CU's LDS pipe for >= 2 cycles (64 lanes, 32 banks); scalar and vector-memory instructions are listed too
Blocks are attributed to the kernel's phases by their content (printed, so a reader can check): 1 staging (global_load -> ds_write_b128),
2a compass pre-test (5 ds_read_u8 + 16-bit min / max + v_cmp_gt_i16 + ds_write_b16), 2b one-sided score (16 ds_read_u8 + the min / max
network; inlined at four call sites), 3 NMS (9 ds_read_u8 + ds_or), 4 compaction / emit.
Step 2 (tools/fast_floor.sh on the GPU box): SQ_INSTS_VALU / SQ_INSTS_LDS per launch of the full kernel and of the ablation builds
(tools/fast_ablate.sh: s1 = staging only, s2a = + compass, s2 = + scoring, s3 = + NMS) -> DYNAMIC instruction counts per phase (differences).
Step 3 (--pmc <json>): floor_ms = sum over phases of dynamic VALU count x (static class mix of the phase's blocks) x class cost / (1024 SIMDs x clock);
LDS floor likewise per CU; written to profiles/r04_fast_floor.txt.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
        "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fmac_f32", "v_add_u16", "v_sub_u16", "v_min_i16", "v_max_i16", "v_min_u16", "v_max_u16"}
C_FAST, C_SLOW, CLOCK, SIMDS, CUS = 2.26, 4.5, 2.43e9, 1024, 256


def classify(ins):
    m = ins.split()[0]
    if m.startswith("v_"):
        base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", m)
        enc_fast = m.endswith("_e32") or base == m      # the assembler prints _e32 / _e64 suffixes; bare = VOP2 / VOP1
        if base in FAST and enc_fast and "sdwa" not in ins and "dpp" not in ins:
            return "fast"
        return "slow"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if m.startswith("s_"):
        return "salu"
    return "other"


def assembly():
    out = "/tmp/orb_kernels_r04.s"
    src = os.path.join(ROOT, "visual-slam_amd", "csrc", "orb_kernels.hip")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                               "--offload-device-only", "-w", src, "-o", out], cwd=os.path.dirname(src))
    return open(out).read().split("\n")


def kernel_blocks(lines, name="_Z6k_fastILi608ELi256EE"):
    start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().split(":")[0].startswith(name) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"label": "entry", "ins": [], "targets": []}
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) and not t.startswith(".LBB"):
            continue
        if t.startswith(".LBB") and t.split()[0].endswith(":"):
            blocks.append(cur)
            cur = {"label": t.split(":")[0], "ins": [], "targets": []}
            continue
        ins = t.split(";")[0].strip()
        if not ins:
            continue
        cur["ins"].append(ins)
        if ins.startswith(("s_cbranch", "s_branch")):
            cur["targets"].append(ins.split()[-1])
    blocks.append(cur)
    return blocks


def content_of(h, mn):
    """what a block is, by content: 'stage' (global loads + LDS writes), '2a' (compass: masked 16-bit stack pushes + 16-bit min / max), '2b' (the
    score network: >= 40 16-bit min / max), 'nms' (ds_or of the keep bitmap / mulhi + byte reads), '' (control, glue)"""
    minmax = sum(v for k, v in mn.items() if k.startswith(("v_min_i16", "v_max_i16")))
    wr = sum(v for k, v in mn.items() if k.startswith("ds_write"))
    rd8 = sum(v for k, v in mn.items() if k.startswith("ds_read_u8"))
    if minmax >= 40:
        return "2b"
    if mn["ds_write_b16"] and minmax:
        return "2a"
    if h["vmem"] and wr and not mn["global_store_dword"]:
        return "stage"
    if mn["ds_or_b32"] or (rd8 >= 4 and any(k.startswith("v_mul_hi_u32") for k in mn)):
        return "nms"
    return ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", default="", help="JSON of tools/fast_floor.sh: per build SQ_INSTS_VALU / SQ_INSTS_LDS / duration per launch")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    blocks = kernel_blocks(assembly())
    index = {b["label"]: i for i, b in enumerate(blocks)}
    loops = []
    for j, b in enumerate(blocks):
        for t in b["targets"]:
            if t in index and index[t] <= j:
                loops.append((index[t], j))
    depth = [sum(1 for a, z in loops if a <= i <= z) for i in range(len(blocks))]
    rows, per_phase = [], collections.defaultdict(lambda: collections.Counter())
    for i, b in enumerate(blocks):
        h, mn = collections.Counter(), collections.Counter()
        for ins in b["ins"]:
            h[classify(ins)] += 1
            mn[ins.split()[0]] += 1
        rows.append([i, b["label"], depth[i], len(b["ins"]), h, content_of(h, mn), mn])
    # the kernel is one straight sequence of phases: blocks up to the first compass block belong to staging, from there to the last
    # score block to phase 2 (a score block is 2b, everything else there is the compass loop and its control), up to the last NMS block
    # to phase 3, the rest to the compaction / emit
    first2a = min(r[0] for r in rows if r[5] == "2a")
    last2b = max(r[0] for r in rows if r[5] == "2b")
    lastnms = max(r[0] for r in rows if r[5] == "nms")
    for r in rows:
        i, c = r[0], r[5]
        r[5] = "1 staging" if i < first2a else ("2b score" if c == "2b" else "2a compass") if i <= last2b else "3 nms" if i <= lastnms else "4 emit"
    out = []
    out.append("k_fast<608, 256>: %d basic blocks, %d instructions, %d loops (backward branches)" % (len(blocks), sum(r[3] for r in rows), len(loops)))
    out.append("%-4s %-12s %5s %5s | %5s %5s %5s %5s %5s | %s" % ("#", "block", "depth", "ins", "fast", "slow", "lds", "vmem", "salu", "phase"))
    for i, lab, d, n, h, ph, mn in rows:
        if n >= 12:
            out.append("%-4d %-12s %5d %5d | %5d %5d %5d %5d %5d | %s" % (i, lab, d, n, h["fast"], h["slow"], h["lds"], h["vmem"], h["salu"], ph))
        if d >= 1 or ph == "2b score":   # (the four inlined copies of the score network: two sit in loops, two are the wave-uniform tails)
            per_phase[ph].update(h)
    out.append("")
    out.append("static class mix of the blocks inside loops, per phase (the mix of the instructions a phase executes over and over):")
    mix = {}
    for ph in sorted(per_phase):
        h = per_phase[ph]
        v = h["fast"] + h["slow"]
        mix[ph] = (h["fast"] / max(v, 1), h["slow"] / max(v, 1), h["lds"] / max(v, 1))
        out.append("  %-12s vector %5d = fast %5d (%.0f %%) + slow %5d (%.0f %%)   lds %4d (%.2f per vector instruction)   salu %4d   mean cost %.2f cycles per vector instruction"
                   % (ph, v, h["fast"], 100 * mix[ph][0], h["slow"], 100 * mix[ph][1], h["lds"], mix[ph][2], h["salu"], mix[ph][0] * C_FAST + mix[ph][1] * C_SLOW))
    if args.pmc:
        p = json.load(open(args.pmc))
        order = [("1 staging", None, "s1"), ("2a compass", "s1", "s2a"), ("2b score", "s2a", "s2"), ("3 nms", "s2", "s3"), ("4 emit", "s3", "full")]
        out.append("")
        out.append("dynamic counts per launch (256 frames; SQ_INSTS_VALU / SQ_INSTS_LDS of the ablation builds, differences = phases) and the issue floor:")
        out.append("  %-12s %12s %12s %8s | %10s %10s" % ("phase", "VALU", "LDS", "build ms", "VALU floor", "LDS floor"))
        tot_v = tot_l = 0.0
        for ph, lo, hi in order:
            v = p[hi]["valu"] - (p[lo]["valu"] if lo else 0.0)
            l = p[hi]["lds"] - (p[lo]["lds"] if lo else 0.0)
            f, s_, _ = mix.get(ph, (0.5, 0.5, 0))
            fv = v * (f * C_FAST + s_ * C_SLOW) / SIMDS / CLOCK * 1e3
            fl = l * 2.0 / CUS / CLOCK * 1e3
            tot_v += fv; tot_l += fl
            out.append("  %-12s %12.4g %12.4g %8.3f | %10.3f %10.3f" % (ph, v, l, p[hi]["ms"] - (p[lo]["ms"] if lo else 0.0), fv, fl))
        out.append("  %-12s %12.4g %12.4g %8.3f | %10.3f %10.3f" % ("total", p["full"]["valu"], p["full"]["lds"], p["full"]["ms"], tot_v, tot_l))
        out.append("")
        out.append("floor (vector issue, the binding pipe; the LDS pipe runs beside it) = %.3f ms per 256 frames at %.2f GHz; measured %.3f ms = %.0f %% of the floor's rate"
                   % (tot_v, CLOCK / 1e9, p["full"]["ms"], 100 * tot_v / p["full"]["ms"]))
        out.append("if the two pipes did not overlap at all: %.3f ms" % (tot_v + tot_l))
    text = "\n".join(out)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
