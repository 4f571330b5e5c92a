#!/bin/bash
# On the GPU box: kernel timeline of the default bench loop (are there idle gaps between the kernels of a pass?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gt && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -- python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 5 "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for fn in glob.glob('/tmp/gt/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(fn)):
        n = r['Kernel_Name'].split('(')[0]
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n[:44]))
rows.sort()
# the timed loop: consecutive reduce<..,2>/stream<2>/drain triples
idx = [i for i, r in enumerate(rows) if 'error_reduce_kernel<true, 1, 2>' in r[2]]
lo, hi = idx[10], idx[40]
t0 = rows[lo][0]
gaps = []
for i in range(lo, hi):
    gaps.append((rows[i + 1][0] - rows[i][1]) / 1e3)
for s, e, n in rows[lo:lo + 12]:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  {n}")
print("mean gap between consecutive kernels (us):", round(sum(gaps) / len(gaps), 2), "max", round(max(gaps), 2), "n", len(gaps))
print("loop period (us):", round((rows[hi][0] - rows[lo][0]) / 1e3 / 30, 1))
PY
