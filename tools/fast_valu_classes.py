#!/usr/bin/env python3
"""Static VALU class mix of k_fast_rows' hot loops, from the compiler's ISA (round 5).

gfx950 issues two classes of vector instructions at different rates (tools/micro/valu_peak.hip -> profiles/r05_valu_issue_rates.txt): plain 32-bit
integer ALU operations (add / sub / and / or / xor / mov / shift right — whatever their encoding: v_add_u32_e64 is as fast as v_add_u32_e32) issue at up
to ~1.8x the rate of everything else (min / max, 24-bit multiplies, v_perm, v_dot*, packed 16-bit, DPP moves, three-operand ops, v_lshlrev, v_cmp).
This tool compiles kernels_fast.hip with -save-temps, takes the basic blocks of k_fast_rows<6, 40, false> (the 32-frame instance) and prints, for the
largest straight-line blocks — the scan block loop, the corner pass, the bit loop of the list expansion, the NMS loop — how many instructions of each
class they hold.  Combined with the per-phase DYNAMIC instruction counts (tools/fast_instr_breakdown.sh run: SQ_INSTS_VALU of builds cut short after a
phase) and the class rates at the kernel's occupancy this gives the mix-weighted issue ceiling of a phase:  1 / (f_fast / R_fast + f_slow / R_slow).

usage: python3 tools/fast_valu_classes.py [--rates FAST SLOW]      (rates in VALU instructions per busy CU cycle at the kernel's occupancy)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_not_b32",
        "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_bitop3_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}


def classify(op):
    base = re.sub(r"_(e32|e64|sdwa)$", "", op)
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return "slow"
    return "fast" if base in FAST else "slow"


def main():
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                               "-fno-fast-math", "-Wno-unused-value", "-save-temps=obj", "-c", os.path.join(ROOT, "hyslam_amd", "csrc", "kernels_fast.hip"),
                               "-o", os.path.join(td, "k.o")], stderr=subprocess.DEVNULL)
        asm = open(os.path.join(td, "kernels_fast-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    m = re.search(r"^_Z11k_fast_rowsILi6ELi40ELb0E\w*:.*?^\.Lfunc_end\d+:", asm, re.S | re.M)
    body = m.group(0)
    blocks, cur, name = [], [], "entry"
    for line in body.splitlines():
        lm = re.match(r"^(\.LBB\d+_\d+):", line)
        if lm:
            blocks.append((name, cur))
            cur, name = [], lm.group(1)
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.append(t.split()[0])
    blocks.append((name, cur))
    tot = {"fast": 0, "slow": 0}
    rows = []
    for name, ops in blocks:
        v = [o for o in ops if o.startswith("v_")]
        if not v:
            continue
        c = {"fast": 0, "slow": 0}
        for o in v:
            c[classify(o)] += 1
        tot["fast"] += c["fast"]
        tot["slow"] += c["slow"]
        tag = ""
        if any(o.startswith("v_pk_minimum3") for o in v):
            tag = "corner pass (score network on f16 pairs)"
        elif sum(o.startswith("v_alignbyte") for o in v) >= 8:
            tag = "scan A block (8 rows x 4 px per lane)"
        elif any(o.startswith("v_ffbl") for o in v) and len(v) < 16:
            tag = "list expansion bit loop"
        elif sum(o.startswith("v_max") for o in v) >= 3 and any(o.startswith("ds_read_u8") for o in ops):
            tag = "NMS (3x3, strict)"
        rows.append((len(v), name, c, len([o for o in ops if o.startswith("s_")]), len([o for o in ops if o.startswith("ds_")]), tag))
    rows.sort(reverse=True)
    print("k_fast_rows<6, 40, false>: %d VALU instructions in the ISA, %.0f %% of them plain-ALU class (static)"
          % (tot["fast"] + tot["slow"], 100.0 * tot["fast"] / max(1, tot["fast"] + tot["slow"])))
    print("%-10s %6s %6s %6s %6s %5s %5s  %s" % ("block", "VALU", "fast", "slow", "%fast", "SALU", "LDS", "what"))
    rf = rs = None
    if "--rates" in sys.argv:
        i = sys.argv.index("--rates")
        rf, rs = float(sys.argv[i + 1]), float(sys.argv[i + 2])
    for n, name, c, ns, nl, tag in rows[:14]:
        ff = c["fast"] / n
        extra = ""
        if rf:
            extra = "  mix-weighted ceiling %.2f VALU / busy CU cycle" % (1.0 / (ff / rf + (1 - ff) / rs))
        print("%-10s %6d %6d %6d %5.0f%% %5d %5d  %s%s" % (name, n, c["fast"], c["slow"], 100 * ff, ns, nl, tag, extra))


if __name__ == "__main__":
    main()
