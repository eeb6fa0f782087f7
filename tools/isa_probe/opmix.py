"""Instruction mix of one kernel / function of a device assembly listing: opmix.py dev.s <mangled-or-demangled substring> [top N]"""
import re, sys, subprocess
from collections import Counter
txt = open(sys.argv[1]).read(); pat = sys.argv[2]; top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
names = [m.group(1) for m in re.finditer(r'\.type\s+(\S+),@function', txt)]
dem = dict(zip(names, subprocess.run(['/usr/bin/c++filt'], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.split("\n")))
for n in names:
    if pat not in n and pat not in dem[n]: continue
    a = txt.index("\n" + n + ":"); e = txt.index(".Lfunc_end", a)
    ops = Counter()
    for l in txt[a:e].split("\n"):
        m = re.match(r'\s+([a-z][a-z0-9_]+)\s', l)
        if m: ops[m.group(1)] += 1
    tot = sum(ops.values())
    cls = Counter()
    for o, c in ops.items():
        k = 'valu_f64' if re.search(r'_f64|f64_', o) and o.startswith('v_') else ('valu' if o.startswith('v_') else ('salu' if o.startswith('s_') and not o.startswith('s_load') and not o.startswith('s_waitcnt') else ('smem' if o.startswith('s_load') else ('vmem' if re.match(r'(global|buffer|scratch|flat)_', o) else ('lds' if o.startswith('ds_') else 'other')))))
        cls[k] += c
    print(dem[n][:100], 'total', tot, dict(cls))
    print('  ', ', '.join('%s %d' % (o, c) for o, c in ops.most_common(top)))
