"""Static instruction counts between the MPM_MARK comments of a kernel's assembly (tools/kstat.sh -DMPM_ASM_MARKS with KEEP=...)."""
import collections
import re
import sys

L = open(sys.argv[1]).read().split('\n')
cur, regs, order = None, collections.defaultdict(collections.Counter), []
for l in L:
    m = re.search(r'MPM_MARK (\w+)', l)
    if m:
        cur = m.group(1)
        if cur not in order:
            order.append(cur)
        continue
    if cur is None:
        continue
    m = re.match(r'\s+([a-z]+)_(\w+)', l)
    if not m:
        continue
    op = m.group(0).strip()
    kind = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'scratch_', 'buffer_')) else 'other'
    regs[cur][kind] += 1
    if op.startswith('v_pk'):
        regs[cur]['pk'] += 1
    if op.startswith('v_mov'):
        regs[cur]['mov'] += 1
    if re.match(r'v_(cmp|cndmask)', op):
        regs[cur]['cmp/sel'] += 1
    if re.match(r'v_(add|sub|mul|lshl|lshr|ashr|and|or|xor|bfe|mad|add3|bitop|max|min).*(u32|i32|b32|u24|i24|u64|b64)', op):
        regs[cur]['int'] += 1
for r in order:
    c = regs[r]
    print(f"{r:14s} valu {c['valu']:4d} (pk {c['pk']:3d}, mov {c['mov']:3d}, cmp/sel {c['cmp/sel']:3d}, int {c['int']:3d})  salu {c['salu']:4d}  lds {c['lds']:3d}  vmem {c['vmem']:3d}")
