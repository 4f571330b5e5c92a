#!/bin/bash
# On the GPU box: VALU instruction count + duration of the 16-byte error_reduce for the shipped lib and every _variants/*.so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for f in shipped $R/_variants/*.so; do
  if [ "$f" = shipped ]; then unset AMPLISOLVE_HIP_LIB; else export AMPLISOLVE_HIP_LIB=$f; fi
  rm -rf /tmp/pv && timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pv -- python3 $R/tools/experiments/probes/u16_valu.py > /dev/null 2>&1
  python3 - "$f" <<'PY'
import csv, glob, sys, collections
c = collections.defaultdict(list)
for fn in glob.glob('/tmp/pv/*/*counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        if 'error_reduce' in r['Kernel_Name']: c[r['Counter_Name']].append(float(r['Counter_Value']))
dur = []
for fn in glob.glob('/tmp/pv/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(fn)):
        if 'error_reduce' in r['Kernel_Name']: dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(sys.argv[1].split('/')[-1], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in c.items()}, 'dur_us(min/avg)', round(min(dur), 1), round(sum(dur) / len(dur), 1))
PY
done
