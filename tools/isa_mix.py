#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S dump: tools/isa_mix.py file.s <kernel-substring> [min_mfma]"""
import collections, re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
start = next(i for i, l in enumerate(s) if l.startswith('_ZN') and key in l and l.rstrip().endswith(('E', ':')) or (key in l and re.match(r'^_Z\S+:', l)))
end = next(i for i in range(start, len(s)) if '.amdhsa_kernel' in s[i])
body = s[start:end]
labels = [i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)] + [len(body)]
def kind(b):
    if b.startswith('v_mfma'): return 'mfma'
    if b.startswith('v_'): return 'valu'
    if b.startswith('s_'): return 'salu'
    if b.startswith('ds_'): return 'ds'
    if b.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    return 'other'
for a, b in zip(labels[:-1], labels[1:]):
    blk = [x.strip().split()[0] for x in body[a + 1:b] if x.strip() and not x.strip().startswith((';', '.'))]
    c = collections.Counter(kind(x) for x in blk)
    if c['mfma'] >= minm:
        print(body[a], len(blk), dict(c))
        print('   valu:', collections.Counter(x for x in blk if kind(x) == 'valu').most_common(30))
        print('   salu:', collections.Counter(x for x in blk if kind(x) == 'salu').most_common(12))
        print('   ds/vmem:', collections.Counter(x for x in blk if kind(x) in ('ds', 'vmem')).most_common(12))
