#!/usr/bin/env python3
"""Static instruction counts of the timed loops in a -save-temps gfx950 .s file (kernels whose name contains
the given substring): VALU / SALU per loop iteration divided by `per`, and the opcode histogram."""
import collections, re, subprocess, sys
path, sub, per = sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
s = open(path).read()
for f in re.split(r'\n(?=_Z\w+:)', s):
    m = re.match(r'(_Z\w+):', f)
    if not m or sub not in m.group(1):
        continue
    body = f.split('s_endpgm')[0]
    loops = re.findall(r'(\.LBB\d+_\d+):(.*?)s_cbranch_\w+ \1', body, re.S)
    loop = max((l[1] for l in loops), key=len) if loops else body
    c = collections.Counter(re.findall(r'^\s+([vs]_\w+|ds_\w+|scratch_\w+|buffer_\w+|global_\w+)', loop, re.M))
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    salu = sum(v for k, v in c.items() if k.startswith('s_'))
    dn = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'pm::|\(unsigned int\*.*', '', dn)
    top = ', '.join(f'{k}:{v/per:.1f}' for k, v in c.most_common(10))
    print(f'{dn[:64]:64s} VALU={valu/per:7.1f} SALU={salu/per:6.1f} | {top}')
