#!/bin/bash
# rocprofv3 kernel trace of tools/prof_target.py; usage: tools/prof_run.sh <tag> <prof_target args...>; result: gpurun_out/r3/prof_<tag>.csv
tag=$1; shift
out=gpurun_out/r3/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 tools/prof_target.py "$@" > $out/run.log 2>&1
tail -1 $out/run.log
f=$(find $out -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Name']
    if 'bild' in name or 'rocclr' in name:
        print(f"  {int(r['Calls']):5d} calls  avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}   {name[:110]}")
PY
cp "$f" gpurun_out/r3/prof_$tag.csv
