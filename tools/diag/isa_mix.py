"""Instruction mix of the MFMA-carrying basic blocks of a kernel in a hipcc -S listing: python isa_mix.py file.s mangled-prefix..."""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
for kn in sys.argv[2:]:
    starts = [i for i, l in enumerate(lines) if l.startswith(kn + ':')]
    for start in starts[:1]:
        end = start
        while not lines[end].strip().startswith('s_endpgm'): end += 1
        body = lines[start:end + 1]
        print(lines[start][:70], 'saveexec', sum('s_and_saveexec' in l for l in body), 'scratch', sum('scratch_' in l for l in body))
        blocks = []; cur = []; name = 'entry'
        for l in body:
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m: blocks.append((name, cur)); name = m.group(1); cur = []
            else: cur.append(l)
        blocks.append((name, cur))
        for name, b in blocks:
            ins = [x.strip().split()[0] for x in b if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
            if any('mfma' in x for x in ins) or any(x.startswith('ds_write') for x in ins):
                c = Counter('mfma' if 'mfma' in x else 'valu' if x.startswith('v_') else 'lds' if x.startswith('ds_') else 'vmem' if x.startswith(('buffer_', 'global_')) else 'salu' if x.startswith('s_') else x for x in ins)
                vc = Counter(x for x in ins if x.startswith('v_') and 'mfma' not in x)
                print('  ', name, len(ins), dict(c), vc.most_common(8))
