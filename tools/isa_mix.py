#!/usr/bin/env python3
"""Instruction mix of one kernel from hipcc -S output, priced at the issue costs measured on gfx950
(tools/valu_rate.hip, profiles/r02_valu_issue_rates.txt): what a kernel's OWN scheme costs in vector-issue
cycles, beside the fold it contains.

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S csrc/fwx_fused.hip -o build/isa/fwx_fused.s
    python3 tools/isa_mix.py build/isa/fwx_fused.s 'fused_main_argILi3ELi8ELi1E'

Static counts per class and, for loops, per basic block (label): the dynamic count of the fold is
(trips x block count), everything else runs about once per tile -- the caller supplies trip counts."""
import collections
import re
import sys

# issue cycles per wave64 instruction at a saturated SIMD (profiles/r02_valu_issue_rates.txt)
COST = {"v_mul_f32": 2.15, "v_max3_f32": 4.1, "v_max_f32": 4.1, "v_cmp": 4.1, "v_cndmask": 4.1, "valu_other": 4.1,
        "v_mul_f64": 4.35, "v_max_f64": 4.35, "salu": 1.0, "ds": 0.0, "vmem": 0.0, "other": 0.0}


def classify(op):
    if op.startswith("v_mul_f32"):
        return "v_mul_f32"
    if op.startswith("v_max3_f32"):
        return "v_max3_f32"
    if op.startswith("v_max_f32"):
        return "v_max_f32"
    if op.startswith("v_mul_f64"):
        return "v_mul_f64"
    if op.startswith("v_max_f64"):
        return "v_max_f64"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return op.split()[0]
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(pat), l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = collections.Counter()
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                cur = m.group(1)
                blocks[cur] = collections.Counter()
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            cur = m.group(1)
            blocks[cur] = collections.Counter()
            continue
        blocks[cur][classify(t.split()[0] if t.split() else "")] += 1
    total = collections.Counter()
    for b in blocks.values():
        total.update(b)
    print("kernel %s: %d instructions" % (pat, sum(total.values())))
    for k, v in sorted(total.items(), key=lambda kv: -kv[1]):
        print("  %-14s %6d" % (k, v))
    print("largest blocks (label: instructions, mul / max3 / ds):")
    for name, b in sorted(blocks.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
        cyc = sum(COST.get(k, 0.0) * v for k, v in b.items())
        print("  %-12s %5d  mul %4d max3 %4d ds %3d vmem %3d cmp %3d cnd %3d valu_other %4d salu %4d  ~%.0f issue cycles"
              % (name, sum(b.values()), b["v_mul_f32"] + b["v_mul_f64"], b["v_max3_f32"] + b["v_max_f64"], b["ds"],
                 b["vmem"], b["v_cmp"], b["v_cndmask"], b["valu_other"], b["salu"], cyc))


if __name__ == "__main__":
    main()
