#!/usr/bin/env python3
"""Issue-cost model of one kernel's instruction stream, from the measured costs of tools/ubench/issue_cost.hip
(profiles/r02_issue_cost.txt: nanoseconds per wave-instruction per SIMD at 6 waves per SIMD, ~1 ns = one VOP2).

usage: isa_cost.py <kernel name substring> [--asm file.s] [--min 60] [extra hipcc flags ...]

For every basic block (label to label) with at least --min instructions it prints the VALU issue time the block costs a
SIMD per execution, the time its LDS instructions keep the CU's LDS pipe busy (x4: four SIMDs share one pipe), and the
instruction mix by cost class.  The fast band of the iteration kernel is the one big straight-line block: its numbers
divided by the rows it processes (4) are the per-pixel-row costs quoted in DESIGN.md.
"""
import collections
import os
import re
import subprocess
import sys

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "inverse_compositional_algorithm_amd")

VOP1ISH = ("v_cvt_", "v_floor_", "v_fract_", "v_ceil_", "v_trunc_", "v_rndne_", "v_max_f32", "v_min_f32", "v_lshlrev_",
           "v_lshrrev_", "v_ashrrev_", "v_not_", "v_bfrev_", "v_max_u32", "v_min_u32", "v_max_i32", "v_min_i32")
VOP3INT = ("v_lshl_add_", "v_add3_", "v_and_or_", "v_bfe_", "v_bfi_", "v_perm_", "v_alignbit_", "v_alignbyte_", "v_med3_",
           "v_min3_", "v_max3_", "v_lshl_or_", "v_add_lshl_", "v_xad_", "v_or3_", "v_sad_")
SLOWINT = ("v_mad_", "v_mul_i32_i24", "v_mul_u32_u24", "v_mul_lo_", "v_mul_hi_", "v_fma_mix", "v_mad_mix")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
FMA3 = ("v_fma_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mad_f32", "v_fma_f64", "v_div_")


def classify(line):
    parts = line.split(None, 1)
    op = parts[0]
    args = parts[1] if len(parts) > 1 else ""
    if not op.startswith("v_"):
        return None, 0.0
    ops = [a.strip() for a in args.split(",")]
    srcs = ops[1:] if ops else []
    sgpr = any(re.match(r"^-?\|?(s\d+|s\[|vcc|exec|m0)", a) for a in srcs)
    if "dpp" in line or "sdwa" in line or "row_shr" in line or "wave_shr" in line:
        return "dpp/sdwa", 2.93
    if op.startswith("v_pk_"):
        return "packed", 2.27
    if op.startswith(TRANS):
        return "transcendental", 2.66
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return ("cmp->sgpr", 3.1) if (op.endswith("_e64") or re.match(r"^s\[", ops[0])) else ("cmp->vcc", 1.97)
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "readlane", 3.0
    if op.startswith(SLOWINT):
        return "mad/mul24", 2.9
    if op.startswith(VOP3INT):
        return "vop3 3-op int", 2.35
    if op.startswith("v_cndmask"):
        return ("cndmask sgpr", 2.41) if op.endswith("_e64") else ("cndmask vcc", 1.1)
    if op.startswith(FMA3) or op.endswith("_e64"):
        return ("vop3+sgpr", 2.53) if sgpr else ("vop3 fma", 1.6)
    if op.startswith(VOP1ISH):
        return ("vop1+sgpr", 2.0) if sgpr else ("vop1/max/shift", 1.34)
    if op.endswith("_f64") or "_f64_" in op or op.endswith("_b64") and not op.startswith("v_mov"):
        return "fp64", 4.0
    return ("vop2+sgpr", 1.5) if sgpr else ("vop2", 1.08)


def lds_cost(op):
    if op.startswith(("ds_read2_b32", "ds_read_b128", "ds_read2_b64", "ds_write_b128", "ds_write2")):
        return 2.39
    if op.startswith("ds_write_b64"):
        return 1.5
    return 0.96


def main():
    args = sys.argv[1:]
    pat = args[0] if args else "ica_iterate_kernelILi5ELi1ELi1ELi3E"
    asm, minlen, extra = None, 60, []
    i = 1
    while i < len(args):
        if args[i] == "--asm":
            asm = args[i + 1]; i += 2
        elif args[i] == "--min":
            minlen = int(args[i + 1]); i += 2
        else:
            extra.append(args[i]); i += 1
    if asm is None:
        os.makedirs("/tmp/asm", exist_ok=True)
        asm = "/tmp/asm/ica.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", asm,
                        os.path.join(HERE, "csrc", "ica_api.hip")] + extra, check=True, capture_output=True)
    s = open(asm).read()
    m = re.search(r"^(\S*%s\S*):" % re.escape(pat), s, re.M)
    if not m:
        sys.exit("kernel not found: " + pat)
    body = s[m.start():s.index(".Lfunc_end", m.start())]
    meta = re.search(r"\.amdhsa_kernel %s(.*?)\.end_amdhsa_kernel" % re.escape(m.group(1)), s, re.S).group(1)
    vg = re.search(r"amdhsa_next_free_vgpr (\d+)", meta).group(1)
    sg = re.search(r"amdhsa_next_free_sgpr (\d+)", meta).group(1)
    ldsb = re.search(r"amdhsa_group_segment_fixed_size (\d+)", meta).group(1)
    scr = re.search(r"amdhsa_private_segment_fixed_size (\d+)", meta).group(1)
    print(f"{m.group(1)}: vgpr {vg} sgpr {sg} lds {ldsb} B scratch {scr} B")
    blocks, cur, name = [], [], "entry"
    for raw in body.splitlines():
        l = raw.strip()
        if not l or l.startswith((";", ".")) and not re.match(r"^\.LBB\S+:", l):
            continue
        mm = re.match(r"^(\.LBB\S+):", l)
        if mm:
            blocks.append((name, cur)); cur, name = [], mm.group(1)
            continue
        if l.endswith(":"):
            continue
        cur.append(l.split(";")[0].strip())
    blocks.append((name, cur))
    for name, ins in blocks:
        if len(ins) < minlen:
            continue
        valu, lds, nl = 0.0, 0.0, 0
        mix, cnt = collections.Counter(), collections.Counter()
        for l in ins:
            op = l.split()[0]
            if op.startswith("ds_"):
                lds += lds_cost(op); nl += 1
                cnt[op.split("_")[0] + "_" + "_".join(op.split("_")[1:3])] += 1
                continue
            k, c = classify(l)
            if k:
                valu += c; mix[k] += c; cnt[k] += 1
        nv = sum(cnt[k] for k in mix)
        print(f"block {name:<12} instrs {len(ins):4d}  VALU {nv:4d} = {valu:7.1f} ns/SIMD   LDS {nl:3d} = {lds:6.1f} ns/CU (x4 = {4*lds:6.1f})"
              f"  waitcnt {sum(1 for l in ins if l.startswith('s_waitcnt'))} branch {sum(1 for l in ins if l.startswith(('s_cbranch', 's_branch')))}")
        print("      " + "  ".join(f"{k}: {cnt[k]} ({mix[k]:.0f})" for k, _ in mix.most_common()))
        print("      " + "  ".join(f"{k}: {v}" for k, v in cnt.items() if k.startswith("ds_")))


if __name__ == "__main__":
    main()
