#!/usr/bin/env python
"""VGPR liveness over the assembly of one kernel (tools/kstat.sh KEEP=...): where is the register-pressure peak and which
values are alive there.  usage: asm_liveness.py kernel.s [top_n]"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
ins = []  # (lineno, text)
labels = {}
for ln, t in enumerate(src):
    t2 = t.split(";")[0].strip()
    if not t2:
        continue
    m = re.match(r"^(\.LBB\w+):", t2)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if t2.startswith(".") or t2.endswith(":"):
        continue
    ins.append((ln + 1, t2))


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


n = len(ins)
defs, uses, succ = [set() for _ in range(n)], [set() for _ in range(n)], [[] for _ in range(n)]
for i, (ln, t) in enumerate(ins):
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest else []
    is_store = op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "global_atomic", "ds_add", "ds_max")) and "rtn" not in op and not (op.startswith("global_atomic") and "glc" in t or "sc0" in t and op.startswith("global_atomic"))
    nd = 0
    if op.startswith("v_") or op.startswith(("global_load", "scratch_load", "ds_read", "buffer_load", "ds_bpermute", "ds_swizzle")) or (op.startswith("global_atomic") and not is_store) or op.startswith("ds_add_rtn") or op.startswith("ds_append") or op.startswith("ds_consume"):
        nd = 1
        if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            nd = 0
    if is_store:
        nd = 0
    for k, o in enumerate(ops):
        r = regs(o)
        if k < nd:
            defs[i].update(r)
        else:
            uses[i].update(r)
    # partial writes (v_fmac, v_pk_fma with dst as acc handled as use when repeated); v_fmac reads dst
    if op.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_writelane", "v_cndmask_b32_dpp", "v_mov_b32_dpp", "v_add_f32_dpp")) and nd:
        uses[i].update(regs(ops[0]))
    if op in ("s_branch",):
        succ[i] = [labels[ops[0]]] if ops[0] in labels else []
    elif op.startswith("s_cbranch"):
        succ[i] = ([labels[ops[0]]] if ops[0] in labels else []) + ([i + 1] if i + 1 < n else [])
    elif op in ("s_endpgm",):
        succ[i] = []
    else:
        succ[i] = [i + 1] if i + 1 < n else []
live_in = [set() for _ in range(n)]
changed = True
while changed:
    changed = False
    for i in range(n - 1, -1, -1):
        out = set()
        for s in succ[i]:
            out |= live_in[s]
        # exec-masked writes do not kill (conservative: treat defs inside as kill anyway)
        li = uses[i] | (out - defs[i])
        if li != live_in[i]:
            live_in[i] = li
            changed = True
press = [len(live_in[i]) for i in range(n)]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 1
order = sorted(range(n), key=lambda i: -press[i])
print("max live VGPRs:", press[order[0]], "at line", ins[order[0]][0], ins[order[0]][1])
# pressure profile (every 40 instructions)
prof = []
for i in range(0, n, 40):
    prof.append(f"{ins[i][0]}:{max(press[i:i+40])}")
print("profile (line:max live over next 40 instrs):", " ".join(prof))
i = order[0]
lastdef = {}
for j in range(0, i):
    for r in defs[j]:
        lastdef[r] = j
print("live at the peak, by last textual definition:")
bydef = {}
for r in sorted(live_in[i]):
    j = lastdef.get(r, -1)
    bydef.setdefault(j, []).append(r)
for j in sorted(bydef):
    print(f"  line {ins[j][0] if j >= 0 else 0:5d}: v{bydef[j]}  <- {ins[j][1] if j >= 0 else 'kernel entry'}")
