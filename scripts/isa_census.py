"""Development aid: per-phase instruction census of a straight-line MFMA kernel from `hipcc -S` output.
usage: isa_census.py file.s mangled_kernel_name [mfma_per_gap_histogram]
Splits the kernel body at its labels / branches into basic blocks, prints for each block the instruction classes, and
a histogram of non-MFMA issue slots per MFMA gap (how clumpy the filler work is)."""
import re, sys, collections
txt = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^%s:(.*?)^\.Lfunc_end' % re.escape(name), txt, re.S | re.M)
body = m.group(1).split('\n')

def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'ds_read'
    if op.startswith('ds_'): return 'ds_other'
    if op.startswith('global_load_lds'): return 'dma'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_nop'): return 's_nop'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_accvgpr'): return 'accvgpr'
    if op.startswith('v_'): return 'valu'
    return 'other'

blocks = []   # (label, [ops])
cur = ('entry', [])
for l in body:
    x = l.strip()
    mm = re.match(r'^(\.LBB\d+_\d+):', x)
    if mm:
        blocks.append(cur)
        cur = (mm.group(1), [])
        continue
    if not x or x.startswith(('.', ';')) or x.endswith(':'): continue
    cur[1].append(x)
blocks.append(cur)
keys = ['mfma', 'valu', 'accvgpr', 'ds_read', 'ds_other', 'dma', 'vmem', 'salu', 's_nop', 'waitcnt', 'barrier']
print('%-12s %6s ' % ('block', 'total') + ' '.join('%8s' % k for k in keys) + '  branch')
tot = collections.Counter()
for lab, ops in blocks:
    if not ops: continue
    c = collections.Counter(cls(o.split()[0]) for o in ops)
    br = [o for o in ops if re.match(r's_(cbranch|branch)', o)]
    print('%-12s %6d ' % (lab, len(ops)) + ' '.join('%8d' % c[k] for k in keys) + '  ' + '; '.join(b.split()[0] + ' ' + b.split()[-1] for b in br))
    tot.update(c)
print('%-12s %6d ' % ('sum(static)', sum(tot.values())) + ' '.join('%8d' % tot[k] for k in keys))
if len(sys.argv) > 3:
    want = sys.argv[3]
    for lab, ops in blocks:
        if lab != want: continue
        vc = collections.Counter(o.split()[0] for o in ops if cls(o.split()[0]) in ('valu', 'accvgpr', 'salu'))
        for k, v in vc.most_common(40): print('   %-28s %d' % (k, v))
        gaps = collections.Counter(); n = 0
        for o in ops:
            if cls(o.split()[0]) == 'mfma': gaps[n] += 1; n = 0
            else: n += 1
        print('   fillers-per-gap histogram:', sorted(gaps.items()))
