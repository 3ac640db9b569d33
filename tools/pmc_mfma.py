#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from a rocprofv3 --pmc pass with
SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
(MI355X_MICROARCH.md: MFMA_BUSY counts SIMD cycles, the SQ wave counters quad-cycles, GRBM_GUI_ACTIVE the sum over 8 XCDs)."""
import collections
import csv
import glob
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0])))
a = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    a[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in a.items():
    if '--all' not in sys.argv and not any(t in k for t in ('conv_igemm', 'conv_glds16', 'conv1x1_stream', 'conv_skinny16', 'stem_s2d_pool16')):
        continue          # default: the convolution kernels; --all: every kernel of the run (the tail between the GEMMs)
    m = {c: sum(x) / len(x) for c, x in v.items()}
    cyc = m['GRBM_GUI_ACTIVE'] / 8.0
    simd_cycles = cyc * 1024
    wc = m['SQ_WAVE_CYCLES'] * 4
    print('%s  launches=%d  cycles/XCD=%.0f' % (k, len(v['GRBM_GUI_ACTIVE']), cyc))
    print('   MFMA busy %.1f %% of SIMD cycles; waves/SIMD %.2f; wave time: parked %.1f %%, issue-stalled %.1f %%, issuing %.1f %%' % (
        100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles, wc / simd_cycles, 100 * m['SQ_WAIT_ANY'] * 4 / wc,
        100 * m['SQ_WAIT_INST_ANY'] * 4 / wc, 100 * m['SQ_ACTIVE_INST_ANY'] * 4 / wc))
