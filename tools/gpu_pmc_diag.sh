#!/bin/bash
# usage (GPU box): tools/gpu_pmc_diag.sh <tag> [bench args...]
# Instruction-mix pass for the kernels BETWEEN the GEMMs: VALU / LDS / VMEM instruction counts, LDS issue stalls and the
# wave-time breakdown per kernel -> gpurun_out/<tag>_pmc_diag.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_diag_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes "$@" > /dev/null 2> $O/pmc_diag_$tag.err || { tail -5 $O/pmc_diag_$tag.err; exit 4; }
cd $R
python3 - $O/pmc_diag_$tag > $O/${tag}_pmc_diag.txt <<'PY'
import collections, csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0])))
a = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    a[r['Kernel_Name'].split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
out = []
for k, v in a.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    cyc = m['GRBM_GUI_ACTIVE'] / 8.0
    wc = m['SQ_WAVE_CYCLES'] * 4
    out.append((cyc * len(v['GRBM_GUI_ACTIVE']), k, len(v['GRBM_GUI_ACTIVE']), cyc, wc, m))
for tot, k, n, cyc, wc, m in sorted(out, reverse=True)[:28]:
    print('%-70s launches=%4d cycles/XCD=%9.0f (%.1f us at 2.4 GHz)' % (k, n, cyc, cyc / 2400.0))
    print('     waves/SIMD %.2f; wave time: parked %.1f %%, issue-stalled %.1f %% (LDS issue %.1f %%), issuing %.1f %%; per launch: VALU %.2f M, LDS %.2f M, VMEM-read %.2f M wave-instructions' % (
        wc / (cyc * 1024), 100 * m['SQ_WAIT_ANY'] * 4 / wc, 100 * m['SQ_WAIT_INST_ANY'] * 4 / wc, 100 * m.get('SQ_WAIT_INST_LDS', 0) * 4 / wc,
        100 * m['SQ_ACTIVE_INST_ANY'] * 4 / wc, m['SQ_INSTS_VALU'] / 1e6, m['SQ_INSTS_LDS'] / 1e6, m['SQ_INSTS_VMEM_RD'] / 1e6))
PY
rm -rf $O/pmc_diag_$tag
echo "diag done"
