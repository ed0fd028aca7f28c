#!/usr/bin/env python3
"""Per-kernel register / scratch summary of a `hipcc -save-temps=obj` device assembly file (the *-gfx950.s):
   python tools/isa_summary.py <file.s> [name filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+): *; @\1\n(.*?)^\.Lfunc_end\d+:\n(.*?)(?=^\s*\.(?:text|section|globl|protected))', s, re.S | re.M):
    name, body, tail = m.group(1), m.group(2), m.group(3)
    if flt not in name:
        continue
    g = lambda k: (re.search(r'; %s: (\d+)' % k, tail) or [None, '?'])[1]
    vm0 = len(re.findall(r'vmcnt\(0\)', body))
    short = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', name)[:70]
    print(f"{short:72s} V {g('NumVgprs'):>4} A {g('NumAgprs'):>4} scratch {g('ScratchSize'):>5} occ {g('Occupancy')} scratch_ops {len(re.findall(r'scratch_(?:load|store)', body)):4d} "
          f"mfma {len(re.findall(r'v_mfma', body)):4d} swizzle {len(re.findall('ds_swizzle', body)):3d} vmcnt0 {vm0:3d}")
