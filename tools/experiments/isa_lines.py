"""Attribute the static instructions of one kernel to source lines (compile with -gline-tables-only -S first).
usage: isa_lines.py file.s mangled_kernel_prefix [pattern]   -> instructions matching `pattern` (default: all VALU) per line"""
import re, sys, collections
path, kernel = sys.argv[1], sys.argv[2]
pat = re.compile(sys.argv[3] if len(sys.argv) > 3 else r"\s+v_")
files = {}
inside = False
cur = None
cnt = collections.Counter()
for l in open(path):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
    if l.startswith(kernel) and ":" in l: inside = True
    if not inside: continue
    if "s_endpgm" in l: break
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    if pat.match(l): cnt[cur] += 1
print("total", sum(cnt.values()))
for k, v in sorted(cnt.items(), key=lambda x: -x[1])[:60]: print(v, k)
