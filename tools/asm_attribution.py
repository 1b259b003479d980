#!/usr/bin/env python3
"""Static instruction attribution of one kernel: tools/asm_attribution.py <asm with .loc lines> <kernel symbol substring>
(build the listing with: hipcc ... -gline-tables-only -S --cuda-device-only -o x.s dmt_hip.hip)."""
import bisect, collections, re, sys
from pathlib import Path
asm, sym = sys.argv[1], sys.argv[2]
src_dir = Path(__file__).resolve().parent.parent / "cuda-optix-pathtracing_amd" / "csrc"
lines = open(asm).read().split('\n')
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ':'))
cnt, cur, total = collections.Counter(), ('?', 0), 0
for l in lines[start:]:
    if 's_endpgm' in l: break
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(int(m.group(1)), '?'), int(m.group(2))); continue
    t = l.strip()
    if not t or t.startswith(('.', ';', '/')) or t.endswith(':'): continue
    cnt[cur] += 1; total += 1
def func_ranges(path):
    out = []
    for i, l in enumerate(open(path).read().split('\n'), 1):
        if (l.startswith('DMT_DEV') or l.startswith('__global__')) and '(' in l and not l.strip().endswith(';'):
            name = re.findall(r'([A-Za-z_][A-Za-z0-9_]*)\s*\(', l)
            if name: out.append((i, name[0]))
    return out
rng = {f.name: func_ranges(f) for f in src_dir.glob('*.h*')}
agg = collections.Counter()
for (f, ln), c in cnt.items():
    name = f
    if f in rng and rng[f]:
        k = bisect.bisect_right([a for a, _ in rng[f]], ln) - 1
        name = f + ':' + (rng[f][k][1] if k >= 0 else '?')
    agg[name] += c
print('total instructions', total)
for n, c in agg.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30): print(f'{c:6d}  {n}')
