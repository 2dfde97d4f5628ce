#!/bin/bash
# per-kernel durations of any python script of this repository:  tools/kernel_trace.sh tools/general_time.py [filter]   (run on the GPU box)
REPO=${GRAFT_REPO_ROOT:-$PWD}
S=$1; F=${2:-.}
D=$(mktemp -d /tmp/ktrace.XXXXXX)
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $D -o g -- python $REPO/$S > $D/stdout.txt 2>/dev/null)
python - <<PY
import csv, glob, re
fs = glob.glob("$D/**/*kernel_stats.csv", recursive=True)
if not fs:
    print("no kernel_stats.csv"); raise SystemExit(1)
print(f"{'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}  kernel")
for r in csv.DictReader(open(fs[0])):
    if re.search(r"$F", r["Name"]):
        print(f"{r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.1f} {float(r['MinNs']) / 1e3:10.1f} {float(r['MaxNs']) / 1e3:10.1f}  {r['Name'][:140]}")
PY
rm -rf $D
