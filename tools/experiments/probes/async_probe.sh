#!/bin/bash
# On the GPU box: default bench vs --async-drain, interleaved, plus a kernel trace of the async run (does the drain overlap?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
cd $R
for i in 1 2; do
  for f in "" "--async-drain"; do
    python bench.py --no-cpu-baseline --steps 60 $f 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${f:-sync}', round(d['ms_per_step'],4), round(d['kernels']['error_reduce_ms'],4), round(d['kernels']['poisson_call_ms'],4))"
  done
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/at && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/at -- python3 $R/bench.py --no-cpu-baseline --steps 12 --warmup 3 --async-drain > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for fn in glob.glob('/tmp/at/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(fn)):
        n = r['Kernel_Name'].split('(')[0]
        if any(x in n for x in ('error_reduce', 'poisson_stream', 'poisson_drain')):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n[:40], r.get('Queue_Id', r.get('Stream_Id', '?'))))
rows.sort()
t0 = rows[30][0]
for s, e, n, q in rows[30:48]:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  {n:42s} q{q}")
PY
