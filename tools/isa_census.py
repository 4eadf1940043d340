#!/usr/bin/env python3
"""Instruction census of one kernel in an AMDGPU assembly listing (`make asm`).

  python tools/isa_census.py libsbn_amd/csrc/kernels_walk3.s 'gradient_walk_lut_kernelILb0ELb0ELi4ELb0E'

Per basic block (label to label) and in total: instructions by class -- matrix (v_mfma),
f64 vector arithmetic, other vector ALU (32-bit / moves / conversions), cross-lane
(DPP / readlane / permute), scalar ALU, branches, waits, LDS, vector memory, scalar memory.
Static counts: combine with trip counts (or the SQ_INSTS_* counters) for dynamic ones."""
import re
import sys
from collections import Counter, OrderedDict

CLASSES = ["mfma", "valu_f64", "valu_other", "xlane", "salu", "branch", "wait", "lds", "vmem", "smem", "other"]


def classify(op, rest):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "ds_bpermute", "ds_permute", "ds_swizzle")):
        return "xlane"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store"):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call", "s_endpgm")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        if "dpp" in rest or op.endswith("_dpp"):
            return "xlane"
        if "_f64" in op or op.startswith(("v_ldexp_f64", "v_frexp")):
            return "valu_f64"
        return "valu_other"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    verbose = len(sys.argv) > 3
    lines = open(path).read().splitlines()
    start = None
    for i, ln in enumerate(lines):
        if re.match(r"^[A-Za-z_][A-Za-z0-9_$.]*:", ln) and pat in ln.split(":")[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = Counter()
    ops = Counter()
    for ln in lines[start + 1:]:
        if ln.startswith(".Lfunc_end") or "\t.section" in ln and ".rodata" in ln:
            break
        m = re.match(r"^(\.LBB[0-9_]+):", ln) or re.match(r"^; %(bb\.[0-9]+):", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = Counter()
            continue
        m = re.match(r"^\t([a-z][a-z0-9_]+)\s*(.*)$", ln)
        if not m or m.group(1).startswith(("s_code_end",)):
            continue
        op, rest = m.group(1), m.group(2)
        c = classify(op, rest)
        blocks[cur][c] += 1
        ops[op] += 1
    total = Counter()
    print(f"{'block':14s}" + "".join(f"{c:>11s}" for c in CLASSES) + f"{'sum':>8s}")
    for name, cnt in blocks.items():
        total.update(cnt)
        if sum(cnt.values()) >= (1 if verbose else 12):
            print(f"{name:14s}" + "".join(f"{cnt[c]:11d}" for c in CLASSES) + f"{sum(cnt.values()):8d}")
    print(f"{'TOTAL':14s}" + "".join(f"{total[c]:11d}" for c in CLASSES) + f"{sum(total.values()):8d}")
    print("\nmost frequent opcodes:")
    for op, k in ops.most_common(45):
        print(f"  {k:5d}  {op}")


if __name__ == "__main__":
    main()
