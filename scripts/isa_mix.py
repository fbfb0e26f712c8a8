"""Development aid: instruction mix of a kernel's hottest loop (largest backward-branch span) from `hipcc -S` output.
usage: isa_mix.py file.s mangled_kernel_name"""
import re, sys, collections
txt = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^%s:(.*?)^\s*s_endpgm' % re.escape(name), txt, re.S | re.M)
body = m.group(1).split('\n')
labels = {}
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm: labels[mm.group(1)] = i
best = (0, 0)
for i, l in enumerate(body):
    mm = re.search(r's_(?:cbranch_\w+|branch) (\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i and i - labels[mm.group(1)] > best[1] - best[0]:
        best = (labels[mm.group(1)], i)
def mix(lo, hi):
    c = collections.Counter()
    for x in body[lo:hi]:
        x = x.strip()
        if not x or x.startswith(('.', ';')) or x.endswith(':'): continue
        op = x.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('ds_read') or op.startswith('ds_load'): c['ds_read'] += 1
        elif op.startswith('ds_'): c['ds_other'] += 1
        elif op.startswith('global_load_lds'): c['dma'] += 1
        elif op.startswith('global_') or op.startswith('buffer_') or op.startswith('scratch_'): c['vmem:' + op] += 1
        elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif op.startswith('s_barrier'): c['barrier'] += 1
        elif op.startswith('s_nop'): c['s_nop'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith('v_accvgpr'): c['accvgpr'] += 1
        elif op.startswith('v_'): c['valu:' + op] += 1
        else: c[op] += 1
    return c
for tag, (lo, hi) in (("whole kernel", (0, len(body))), ("hot loop", best)):
    c = mix(lo, hi)
    tot = sum(c.values())
    nm = c['mfma']
    print(f"== {tag}: {tot} instructions, {nm} mfma, non-mfma non-salu = {tot - nm - c['salu'] - c['s_nop']}")
    for k, v in sorted(c.items(), key=lambda x: -x[1])[:24]:
        print(f"   {k:28s} {v}")
