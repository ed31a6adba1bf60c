"""Counts the software integer divisions (v_rcp_iflag_f32 marks one) per kernel in `hipcc -S --cuda-device-only` listings:
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only lm_net_amd/csrc/na.hip -o /tmp/na.s
    python tools/isa_divisions.py /tmp/na.s [/tmp/rows.s ...]
A division by a run-time value costs ~20 (unsigned 32-bit), ~35 (signed) or > 100 (64-bit) VALU instructions; inside a per-item loop
that was a quarter of the instructions of the neighborhood-attention backward."""
import collections, re, sys
for path in sys.argv[1:]:
    cur, cnt, tot = None, collections.Counter(), collections.Counter()
    for line in open(path):
        m = re.match(r'^(_Z[\w$.]+):', line)
        if m:
            cur = m.group(1)
            continue
        if cur and line.startswith('\t') and not line.startswith('\t.'):
            tot[cur] += 1
            if 'v_rcp_iflag_f32' in line:
                cnt[cur] += 1
    rows = sorted(((c, k, tot[k]) for k, c in cnt.items()), reverse=True)
    print("%s: %d kernels with run-time integer divisions" % (path, len(rows)))
    for c, k, t in rows[:20]:
        print("   %3d divisions / %5d instructions  %s" % (c, t, k[:110]))
