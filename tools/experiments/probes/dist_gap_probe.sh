#!/bin/bash
# On the GPU box: kernel timeline of the N>1 bench path forced on one rank over RCCL (env:// rendezvous set by hand so that
# rocprofv3 wraps python3 itself): how much of a step is idle GPU time, i.e. host-side issue overhead of the pipeline?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29593
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/dg && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/dg -- python3 $R/bench.py --gpus 1 --force-dist --steps 40 --warmup 5 "$@" > /tmp/dg.log 2>&1
tail -1 /tmp/dg.log | cut -c1-160
python3 - <<'PY'
import csv, glob
rows = []
for fn in glob.glob('/tmp/dg/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60]))
rows.sort()
idx = [i for i, r in enumerate(rows) if 'error_reduce_kernel' in r[2]]
lo, hi = idx[12], idx[36]
t0 = rows[lo][0]
busy = 0
cur_end = rows[lo][0]
for s, e, n in rows[lo:hi]:
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
span = rows[hi][0] - rows[lo][0]
print("steps 24: period %.1f us, GPU busy %.1f %%" % (span / 24e3, 100 * busy / span))
for s, e, n in rows[lo:lo + 16]:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  {n}")
PY
