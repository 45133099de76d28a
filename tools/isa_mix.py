#!/usr/bin/env python3
"""isa_mix.py <file.s> <kernel-name-substring> [--blocks]

Static VALU issue-cost mix of one kernel of an `-S` listing, priced with the classes measured by
tools/valu_issue_bench.hip on MI355X (profiles/r02_valu_issue_costs.json), full occupancy:

  class A  2 cycles per wave-instruction per SIMD: v_mul/add/sub/fma/fmac_f32, v_mov_b32, v_add/sub_u32,
           v_and/or/xor_b32, v_lshrrev_b32 - while every source is a VGPR, an inline constant or a literal
  class B  4 cycles: the same opcodes with an SGPR (or VCC/EXEC) source, every v_pk_*, v_cmp*, v_cndmask, v_max/min/med3,
           conversions, integer multiplies, three-operand integer ops, v_lshlrev_b32, f64 arithmetic, lane ops
  class C  8 cycles: v_rcp/rsq/sqrt/exp/log/sin/cos_f32
A class-A instruction of one wave issues beside a class-B instruction of another (a mul/max pair costs 4.1, a
cmp/mul/cndmask/add quadruple 8.2), so a stream's floor is max(4*B + 8*C, 2*(A + B) + 8*C) SIMD-cycles.
"""
import re
import sys

A_OPS = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_add_u32",
         "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32"}
C_OPS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
SGPR_SRC = re.compile(r"(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|scc)(?![\w])")


def classify(op, operands):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in C_OPS:
        return "C"
    if base in A_OPS:
        srcs = operands.split(",")[1:]  # the first operand is the destination
        if any(SGPR_SRC.search(s) for s in srcs):
            return "B"
        return "A"
    return "B"


def category(op):
    """The SQ_INSTS_VALU_* counter an opcode is tallied under (gfx950 PMC: ADD_F32 / MUL_F32 / FMA_F32 / TRANS_F32 / ADD|MUL|FMA_F64
    / INT32 / INT64 / CVT), or "other" (selects, compares, min/max, moves, lane ops: INSTS_VALU minus the categories)."""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in C_OPS:
        return "trans_f32"
    if base.endswith("_f64"):
        return "f64"
    if base.startswith("v_cvt"):
        return "cvt"
    if base in ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_add_f32"):
        return "add_f32"
    if base in ("v_mul_f32", "v_pk_mul_f32", "v_mul_legacy_f32"):
        return "mul_f32"
    if base in ("v_fma_f32", "v_fmac_f32", "v_pk_fma_f32", "v_mad_f32", "v_mac_f32"):
        return "fma_f32"
    if base.endswith("_b64") or base.endswith("_u64") or base.endswith("_i64") or base == "v_mad_u64_u32":
        return "int64"
    if re.search(r"_(u32|i32|b32|u16|i16|u24|i24)(_|$)", base) and not base.startswith(("v_cmp", "v_cndmask", "v_mov", "v_readlane",
                                                                                       "v_writelane", "v_readfirstlane", "v_mbcnt")):
        return "int32"
    return "other"


def mix_by_category(path, name):
    """Static instruction counts and additive issue cost of the kernel's VALU instructions per PMC category."""
    in_kernel = False
    cats = {}
    for line in open(path):
        line = line.split(";")[0].rstrip()
        if not in_kernel:
            if re.match(r"^[\w.$]+:", line) and name in line and not line.startswith(".L"):
                in_kernel = True
            continue
        if line.strip().startswith("s_endpgm"):
            break
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", line)
        if m and m.group(1).startswith("v_"):
            c = cats.setdefault(category(m.group(1)), {"count": 0, "cycles": 0})
            c["count"] += 1
            c["cycles"] += {"A": 2, "B": 4, "C": 8}[classify(m.group(1), m.group(2))]
    return {k: dict(v, avg_cost=v["cycles"] / v["count"]) for k, v in cats.items()} if in_kernel else None


def mix(path, name):
    """Static class counts of the first kernel of `path` whose symbol contains `name`: {"A":…, "B":…, "C":…, "valu":…,
    "avg_cost": additive cycles per VALU instruction} - or None when the kernel is not in the listing."""
    in_kernel = False
    tot = {"A": 0, "B": 0, "C": 0}
    for line in open(path):
        line = line.split(";")[0].rstrip()
        if not in_kernel:
            if re.match(r"^[\w.$]+:", line) and name in line and not line.startswith(".L"):
                in_kernel = True
            continue
        if line.strip().startswith("s_endpgm"):
            break
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", line)
        if m and m.group(1).startswith("v_"):
            tot[classify(m.group(1), m.group(2))] += 1
    n = sum(tot.values())
    if not in_kernel or n == 0:
        return None
    return dict(tot, valu=n, avg_cost=(2 * tot["A"] + 4 * tot["B"] + 8 * tot["C"]) / n)


def main():
    path, name = sys.argv[1], sys.argv[2]
    per_block = "--blocks" in sys.argv
    in_kernel = False
    block = "entry"
    tot = {"A": 0, "B": 0, "C": 0}
    blocks = {}
    ops = {}
    salu = smem = vmem = lds = 0
    for line in open(path):
        line = line.split(";")[0].rstrip()
        if not in_kernel:
            if re.match(r"^[\w.$]+:", line) and name in line and not line.startswith(".L"):
                in_kernel = True
            continue
        if line.strip().startswith("s_endpgm"):
            break
        m = re.match(r"^(\.LBB[\w]+):", line)
        if m:
            block = m.group(1)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", line)
        if not m:
            continue
        op, operands = m.group(1), m.group(2)
        if op.startswith("v_"):
            c = classify(op, operands)
            tot[c] += 1
            b = blocks.setdefault(block, {"A": 0, "B": 0, "C": 0})
            b[c] += 1
            key = re.sub(r"_(e32|e64)$", "", op) + ("" if c != "B" or re.sub(r"_(e32|e64)$", "", op) not in A_OPS else " (sgpr src)")
            ops[key] = ops.get(key, 0) + 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            smem += 1
        elif op.startswith("s_"):
            salu += 1
        elif op.startswith("ds_"):
            lds += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            vmem += 1
    n = sum(tot.values())
    print("kernel *%s*: VALU %d  (A %d, B %d, C %d)  SALU %d  SMEM %d  VMEM %d  LDS %d" %
          (name, n, tot["A"], tot["B"], tot["C"], salu, smem, vmem, lds))
    add = 2 * tot["A"] + 4 * tot["B"] + 8 * tot["C"]
    floor = max(4 * tot["B"], 2 * (tot["A"] + tot["B"])) + 8 * tot["C"]
    print("static cost: additive %.2f cycles/inst, co-issue floor %.2f cycles/inst" % (add / n, floor / n))
    for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:40]:
        print("  %5d  %s" % (v, k))
    if per_block:
        for b, c in blocks.items():
            if sum(c.values()) >= 8:
                print("  block %-12s A %4d  B %4d  C %3d" % (b, c["A"], c["B"], c["C"]))


if __name__ == "__main__":
    main()
