#!/usr/bin/env python3
"""Count instructions per basic block of one kernel in a hipcc -S listing.
usage: count_isa.py file.s kernel_substring [min_block_size]"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]; minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 100
start = next(i for i, l in enumerate(s) if key in l and l.rstrip().endswith(':') or (key in l and ': ' in l and l.startswith('_Z')))
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
blocks = []; cur = ['entry', {}]; blocks.append(cur)
for l in s[start + 1:end]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        cur = [l.split(':')[0], {}]; blocks.append(cur)
    else:
        t = l.strip().split(' ')[0].split('\t')[0] if l.strip() else ''
        if t and not t.startswith(';') and not t.startswith('.'):
            cur[1][t] = cur[1].get(t, 0) + 1
for b in blocks:
    tot = sum(b[1].values())
    if tot >= minsz:
        f64 = sum(v for k, v in b[1].items() if 'f64' in k)
        trans = sum(v for k, v in b[1].items() if k.startswith(('v_rsq', 'v_rcp', 'v_sqrt', 'v_div')))
        v32 = sum(v for k, v in b[1].items() if k.startswith('v_') and 'f64' not in k)
        ds = sum(v for k, v in b[1].items() if k.startswith('ds_'))
        print('%s: total %d, f64 %d (trans %d), other VALU %d, ds %d' % (b[0], tot, f64, trans, v32, ds))
        print('   ', sorted(b[1].items(), key=lambda x: -x[1])[:30])
