"""Register / scratch / code-size figures of the out-of-line device FUNCTIONS of a device assembly listing (hipcc --cuda-device-only -S).  usage: fstats.py dev.s [substring ...]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
names = [m.group(1) for m in re.finditer(r'\.type\s+(\S+),@function', txt)]
dem = dict(zip(names, subprocess.run(['/usr/bin/c++filt'], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.split("\n")))
for n in names:
    a = txt.find("\n" + n + ":")
    e = txt.find(".Lfunc_end", a)
    seg = txt[e:e + 3000]
    g = lambda k: (re.search(r'; ' + k + r':? =? ?(\d+)', seg) or [None, '?'])[1]
    d = dem[n]
    if pats and not any(p in d for p in pats): continue
    print('%-120s vgpr %4s sgpr %4s scratch %5s code %6s' % (d[:120], g('NumVgprs'), g('NumSgprs'), g('ScratchSize'), g('codeLenInByte')))
