#!/usr/bin/env python3
"""Instruction mix between consecutive MFMAs of one kernel in a hipcc -S listing (checks that staging code sits in the MFMA gaps).
usage: isa_gaps.py file.s mangled_kernel_name"""
import sys
from collections import Counter
T = open(sys.argv[1]).read().split('\n')
name = sys.argv[2] + ':'
start = [i for i, l in enumerate(T) if l.startswith(name)][0]
end = [i for i, l in enumerate(T) if i > start and l.strip().startswith('s_endpgm')][0]
L = T[start:end]
idx = [i for i, l in enumerate(L) if 'v_mfma' in l]
print(sys.argv[2], len(idx), 'MFMAs')
prev = idx[0]
for n, i in enumerate(idx[1:], 1):
    gap = [l.split()[0] for l in L[prev + 1:i] if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
    lab = [l.split(':')[0] for l in L[prev + 1:i] if l.startswith('.LBB')]
    if gap:
        print(n, len(gap), dict(Counter(gap).most_common(7)), lab[:3])
    prev = i
