"""Static instruction mix of one kernel instantiation from a hipcc -S listing.
Usage: python scripts/asm_mix.py file.s <substring of the mangled name> [--dump out.s]"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'\n(_ZN\S*%s\S*): *;[^\n]*\n(.*?)\n\.Lfunc_end' % re.escape(key), s, re.S)
name, body = m.group(1), m.group(2)
if '--dump' in sys.argv:
    open(sys.argv[sys.argv.index('--dump') + 1], 'w').write(body)
lines = [l.strip() for l in body.splitlines()]
lines = [l for l in lines if l and not l.startswith(('.', ';', '/')) and not l.endswith(':')]
ops = collections.Counter(l.split()[0] for l in lines)
tot = collections.Counter()
for k, v in ops.items():
    c = 'mfma' if k.startswith('v_mfma') else 'valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else \
        'lds' if k.startswith('ds_') else 'vmem'
    tot[c] += v
print(name, len(lines), dict(tot))
print(ops.most_common(50))
