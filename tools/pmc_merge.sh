#!/bin/bash
# tools/pmc_merge.sh <tag> <workload-key> -- copy one workload's summaries from gpurun_out/<tag>/ into profiles/ and merge
# its measured traffic / VALU occupancy into profiles/pmc_traffic.json (units per dispatch come from the bench line).
set -e
TAG=$1; KEY=$2
ROOT=$(cd $(dirname $0)/.. && pwd); cd $ROOT
S=gpurun_out/$TAG
for f in kernel_stats pmc_fetch pmc_write pmc_inst; do cp $S/${KEY}_$f.txt profiles/${TAG}_${KEY}_$f.txt; done
cp $S/${KEY}_bench.json profiles/${TAG}_bench_${KEY}.json
python - $S/${KEY}_kernel_stats.txt $S/${KEY}_pmc_fetch.txt > /tmp/units_$KEY.json <<'PY'
import json, re, sys
# units per dispatch: PMC averages are per dispatch, so any consistent per-dispatch unit count works; take it from the
# bench line's profiler record (units / launches) of the same workload
import glob, os
key = os.path.basename(sys.argv[1]).replace("_kernel_stats.txt", "")
line = json.load(open(os.path.join(os.path.dirname(sys.argv[1]), key + "_bench.json")))
json.dump(line["kernel_units_per_launch"], sys.stdout)
PY
python tools/pmc_traffic.py $KEY $S/${KEY}_pmc_fetch.txt $S/${KEY}_pmc_write.txt /tmp/units_$KEY.json $S/${KEY}_pmc_inst.txt --source-hash $(cat $S/${KEY}_source_hash.txt) --into profiles/pmc_traffic.json
echo merged $KEY
