#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (KB counters -> MB per launch).
gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> the read side is doubled
(MI355X_MICROARCH.md, HBM section; checked here on pil_resize_h, which reads 24.9 MB per launch)."""
import collections
import csv
import glob
import sys


def agg(path):
    a = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0]
        a[k][0] += float(r['Counter_Value'])
        a[k][1] += 1
    return a


af = agg(glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0])
aw = agg(glob.glob(sys.argv[2] + '/*/*_counter_collection.csv')[0])
print('%-46s %8s %16s %16s %14s' % ('kernel', 'launches', 'fetch MB/launch', 'fetch x2 (gfx950)', 'write MB/launch'))
for k in sorted(af, key=lambda k: -af[k][0]):
    n = af[k][1]
    print('%-46s %8d %16.2f %16.2f %14.2f' % (k[:46], n, af[k][0] / n / 1024, 2 * af[k][0] / n / 1024,
                                                aw[k][0] / max(aw[k][1], 1) / 1024))

# machine-readable copy for bench.py's roofline.traffic (bytes per launch, read side already doubled)
if len(sys.argv) > 3:
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from apse_uav_amd import _lib
    out = {"__build__": _lib.load().apse_version().decode()}          # bench.py drops the figures when the library changed
    out.update({k.replace('void ', ''): {"launches": af[k][1], "fetch_bytes": 2 * af[k][0] / af[k][1] * 1024,
                                        "write_bytes": aw[k][0] / max(aw[k][1], 1) * 1024} for k in af})
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
