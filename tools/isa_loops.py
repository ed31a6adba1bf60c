"""Instruction mix of the loops of each kernel in a hipcc -S listing:  python tools/isa_loops.py file.s [name-filter]"""
import re, sys, collections
s = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(i, re.match(r'(_Z\w+):', l).group(1)) for i, l in enumerate(s) if re.match(r'_Z\w+:', l)]
starts.append((len(s), None))
for (a0, name), (a1, _) in zip(starts, starts[1:]):
    if flt not in name:
        continue
    lines = s[a0:a1]
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    print(name[:70])
    for a, b in loops:
        c = collections.Counter()
        n = 0
        for l in lines[a:b + 1]:
            if not l.startswith('\t'):
                continue
            t = l.strip()
            if t.startswith(('.', ';')):
                continue
            op = t.split()[0]
            n += 1
            if op.startswith('v_pk_'): c['pk'] += 1
            elif op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): c['lane'] += 1
            elif 'dpp' in t: c['dpp'] += 1
            elif op.startswith('v_mov') or op.startswith('v_accvgpr'): c['mov'] += 1
            elif op.startswith('v_'): c['valu'] += 1
            elif op.startswith('ds_'): c['ds'] += 1
            elif op.startswith(('buffer_', 'global_')): c['vmem'] += 1
            elif op.startswith('scratch_'): c['scratch'] += 1
            elif op.startswith('s_nop'): c['nop'] += 1
            elif op.startswith('s_waitcnt'): c['wait'] += 1
            elif op.startswith('s_'): c['salu'] += 1
        if n > 60:
            print('   loop at +%d..+%d  n=%d  %s' % (a, b, n, dict(c)))
