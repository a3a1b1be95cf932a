#!/bin/bash
# Developer recipe (GPU box): per-kernel time split of one LightGlue batch (32 pairs, N = 2048).
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/plg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/plg -o lg -- python $GRAFT_REPO_ROOT/tools/${PROF_SCRIPT:-gpu_time_lg.py} > /tmp/plg.log 2>&1
tail -1 /tmp/plg.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/plg/lg_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(r['Name'][:50].ljust(50), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:9.1f} us", f"{100*float(r['TotalDurationNs'])/tot:6.2f}%")
PY
