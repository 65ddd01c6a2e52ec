#!/usr/bin/env python3
"""Static instruction mix of the loops of a gfx950 kernel (from `hipcc -S --cuda-device-only`).

    python profiles/tools/isa_loops.py k.s <substring of the mangled kernel name>

For every backward branch it prints the instruction classes between the target label and the
branch (the loop body), which is what bounds a VALU-limited marching kernel."""
import re, sys, collections

def classify(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64", "v_max_f64", "v_min_f64", "v_pk_")): return "valu_f64"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_", "v_trig", "v_ldexp_f64", "v_frexp", "v_floor_f64", "v_fract_f64", "v_rndne_f64", "v_cvt_")): return "valu_f64_slow"
    if op.startswith("v_cmp") : return "valu_cmp"
    if op.startswith("v_cndmask"): return "valu_cndmask"
    if op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle", "v_readlane", "v_writelane", "v_permlane", "v_mov_b32_dpp")) or "dpp" in op: return "xlane"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load")): return "vmem_load"
    if op.startswith(("global_store", "flat_store", "buffer_store", "scratch_store")): return "vmem_store"
    if op.startswith("global_atomic"): return "atomic"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("s_load"): return "smem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and key in l.split(":")[0] and ":" in l:
            start = i; break
    if start is None: raise SystemExit("kernel not found")
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"): break
        body.append(l)
    labels, ins = {}, []
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m: labels[m.group(1)] = len(ins)
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = len(ins); continue
        op = t.split()[0]
        ins.append((op, t))
    tot = collections.Counter(classify(o) for o, _ in ins)
    print(f"{lines[start].split(chr(58))[0]}: {len(ins)} instructions", dict(tot))
    for idx, (op, t) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= idx:
                seg = ins[labels[tgt]:idx + 1]
                c = collections.Counter(classify(o) for o, _ in seg)
                valu = sum(v for k, v in c.items() if k.startswith("valu") or k == "xlane")
                print(f"  loop {tgt}: {len(seg)} instr, VALU-class {valu}", dict(c))

main()
