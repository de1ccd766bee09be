#!/usr/bin/env python3
"""Basic-block instruction census of one kernel in a hipcc -S listing (tuning aid).
usage: asm_blocks.py file.s kernel_symbol_substring"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
blocks, cur = [], {"label": "entry", "ins": []}
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = {"label": m.group(1), "ins": []}
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur["ins"].append(t.split(";")[0].strip())
blocks.append(cur)
tot = {}
for b in blocks:
    c = {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "smem": 0, "f64": 0, "trans": 0, "br": []}
    for ins in b["ins"]:
        op = ins.split()[0]
        if op.startswith("v_"):
            c["valu"] += 1
            if "f64" in op: c["f64"] += 1
            if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_div_scale", "v_div_fmas", "v_div_fixup")): c["trans"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer"): c["smem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
            if op.startswith(("s_cbranch", "s_branch")): c["br"].append(op.replace("s_cbranch_", "").replace("s_branch", "jmp") + ">" + ins.split()[-1].replace(".LBB", ""))
    print(f'{b["label"]:12s} n={len(b["ins"]):4d} valu={c["valu"]:3d} f64={c["f64"]:3d} tr={c["trans"]:2d} salu={c["salu"]:3d} vmem={c["vmem"]:2d} lds={c["lds"]:2d} smem={c["smem"]:2d}  {" ".join(c["br"])}')
    for k in ("valu", "salu", "vmem", "lds", "smem"):
        tot[k] = tot.get(k, 0) + c[k]
print(tot)
