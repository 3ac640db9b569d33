#!/usr/bin/env python3
"""Per-layer convolution efficiency from a rocprofv3 --kernel-trace CSV of bench.py (last step).
usage: layer_report.py <kernel_trace.csv> [detections per FRAME] [batch]      (M, FLOPs and TFLOP/s are those of the whole batch)"""
import collections
import csv
import sys

path = sys.argv[1]
ndet = int(sys.argv[2]) if len(sys.argv) > 2 else 8
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ndet *= BATCH                      # the packed detection list of a step holds every image's detections
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'pil_resize_h' in r['Kernel_Name'].split('(')[0]]
fr = rows[idx[-2]:idx[-1]]
t0 = int(fr[0]['Start_Timestamp'])
span = (int(fr[-1]['End_Timestamp']) - t0) / 1e3
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in fr) / 1e3
print('step (batch %d): %d kernels, span %.1f us, busy %.1f us' % (BATCH, len(fr), span, busy))
layers = []


def conv(name, M, Cout, K, per_image=True):
    M = M * BATCH if per_image else M
    layers.append((name, M, Cout, K, 2.0 * M * Cout * K))


FUSED_BNECK = any('bottleneck64_fused16' in r['Kernel_Name'] for r in fr)      # 16-bit modes: a res2 bottleneck is one launch


H, W = 768, 1344
conv('stem', (H // 2) * (W // 2), 64, 147)
h, w, cin = H // 4, W // 4, 64
for si, nb in enumerate((3, 4, 23, 3)):
    mid = 64 * 2 ** si
    cout = 4 * mid
    for bi in range(nb):
        st = 2 if (bi == 0 and si > 0) else 1
        oh, ow = h // st, w // st
        if bi == 0:
            conv('res%d.sc' % (si + 2), oh * ow, cout, cin)
        if FUSED_BNECK and mid == 64 and st == 1:
            # one launch: conv1 + conv2 + conv3; the K column shows the three K's summed, FLOPs are the three layers' algorithmic FLOPs
            m = oh * ow * BATCH
            layers.append(('res%d.blk' % (si + 2), m, cout, cin + mid * 9 + mid, 2.0 * m * (mid * cin + mid * mid * 9 + cout * mid)))
        else:
            conv('res%d.c1' % (si + 2), oh * ow, mid, cin)
            conv('res%d.c2' % (si + 2), oh * ow, mid, mid * 9)
            conv('res%d.c3' % (si + 2), oh * ow, cout, mid)
        h, w, cin = oh, ow, cout
dims = {2: (192, 336, 256), 3: (96, 168, 512), 4: (48, 84, 1024), 5: (24, 42, 2048)}
for l in (5, 4, 3, 2):
    hh, ww, c = dims[l]
    conv('lat%d' % l, hh * ww, 256, c)
    conv('out%d' % l, hh * ww, 256, 256 * 9)
rpn_levels = [(192, 336), (96, 168), (48, 84), (24, 42), (12, 21)]
for l, (hh, ww) in enumerate(rpn_levels):
    conv('rpn_t%d' % (l + 2), hh * ww, 256, 2304)
conv('rpn_head', sum(hh * ww for hh, ww in rpn_levels), 15, 256)        # one launch over the rows of all levels
conv('fc1', 1000, 1024, 12544)
conv('fc2', 1000, 1024, 1024)
conv('pred', 1000, 21, 1024)
for i in range(4):
    conv('mask%d' % i, ndet * 196, 256, 2304, False)
conv('deconv', ndet * 196, 1024, 256, False)
conv('mlogit', ndet * 784, 4, 256, False)
conv('assoc', ndet, 128, 25600, False)
# attribute split-K reduce kernels to the preceding conv
items = []
for r in fr:
    kn = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if any(t in kn.split('(')[0] for t in ('conv_igemm', 'conv1x1_stream', 'conv_glds16', 'stem_s2d_pool16', 'conv_skinny16', 'assoc_fc_slices', 'bottleneck64_fused16')):
        # the __bf16 template argument defeats rocprofv3's demangler: fall back to the raw name
        if 'stem_s2d_pool16' in kn:
            lab = 'stem+pool fused'
        elif 'assoc_fc_slices' in kn:
            lab = 'K slices + finish'
        elif 'bottleneck64_fused16' in kn:
            lab = 'bneck' + (kn.split('<')[1].split('>')[0] if '<' in kn else '')
        elif 'conv_skinny16' in kn:
            lab = 'skinny' + (kn.split('<')[1].split('>')[0] if '<' in kn else '')
        elif 'conv1x1_stream' in kn:
            lab = 'stream' + (kn.split('<')[1].split('>')[0] if '<' in kn else '')
        elif 'conv_glds16' in kn:
            lab = 'glds' + (kn.split('<')[1].split('>')[0] if '<' in kn else '')
        else:
            lab = kn.split('<')[1].split('>')[0] if '<' in kn else kn.split('conv_igemm')[1][:28]
        items.append([lab[-22:], d, r['Grid_Size_X'], r['Grid_Size_Y']])
    elif ('conv_splitk_reduce' in kn.split('(')[0] or 'assoc_fc_finish' in kn.split('(')[0]) and items:
        items[-1][1] += d
assert len(items) == len(layers), (len(items), len(layers))
agg = collections.OrderedDict()
tot = 0
for (name, M, Cout, K, fl), (cfg, d, gx, gy) in zip(layers, items):
    a = agg.setdefault((name, cfg, M, Cout, K, gx, gy), [0, 0, 0])
    a[0] += d
    a[1] += fl
    a[2] += 1
    tot += d
print('conv total %.1f us' % tot)
for k, (d, fl, n) in agg.items():
    print('%-8s <%s> M=%6d N=%5d K=%5d grid=%sx%s n=%2d %8.1f us %6.1f TF' % (k[0], k[1], k[2], k[3], k[4], k[5], k[6], n, d, fl / d / 1e6))
others = collections.Counter()
for r in fr:
    kn = r['Kernel_Name']
    if not any(t in kn.split('(')[0] for t in ('conv_igemm', 'conv1x1_stream', 'conv_glds16', 'conv_splitk_reduce', 'stem_s2d_pool16', 'conv_skinny16', 'assoc_fc_', 'bottleneck64_fused16')):
        others[kn.split('(')[0].replace('void ', '')[:40]] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('non-conv kernels:')
for k, v in others.most_common(14):
    print('  %-42s %8.1f us' % (k, v))

if '--timeline' in sys.argv:
    # every launch of the step in order: start offset, duration and the idle gap before it (us)
    print('timeline (us): start  dur  gap  kernel')
    prev_end = t0
    for r in fr:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('  %8.1f %7.1f %6.1f  %s  grid=%s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'].split('(')[0][:70], r.get('Grid_Size', '')))
        prev_end = max(prev_end, e)
