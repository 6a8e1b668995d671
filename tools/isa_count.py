"""Instruction histogram of one kernel from a -save-temps gfx950 assembly file (development aid).
usage: python tools/isa_count.py file.s kernel_name_substring"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(_Z\S*' + re.escape(name) + r'\S*):.*?\n(.*?)\.Lfunc_end', s, re.S | re.M)
body = m.group(2)
ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
c = Counter(ins)
print(m.group(1), 'static instructions:', len(ins))
print(c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30))
tail = s[m.end():m.end() + 4000]
for k in ('NumVgprs', 'NumAgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize'):
    mm = re.search(r'; ' + k + r': (\d+)', tail)
    if mm:
        print(k, mm.group(1))
