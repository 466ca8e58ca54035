#!/bin/bash
# rocprofv3 kernel-trace stats of one training configuration (eager launches): [TAG=name] tools/profile_train.sh <model> [extra bench_train args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
m=${1:-vgg}; shift
out=gpurun_out/prof_train_${TAG:-$m}
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o train -- python3 tools/bench_train.py --model $m --iters 2 "$@" > $out/log.txt 2>&1 < /dev/null
tail -2 $out/log.txt | cut -c1-400
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
aten=[r for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"]]
print("total kernel ms %.1f ; at::native+rocclr launches %d of %d, %.2f %% of GPU time"%(tot/1e6, sum(int(r["Calls"]) for r in aten), sum(int(r["Calls"]) for r in rows), 100*sum(float(r["TotalDurationNs"]) for r in aten)/tot))
for r in rows[:28]:
    print("%6d %9.2f ms %5.1f%% avg %8.1f us  %s"%(int(r["Calls"]), float(r["TotalDurationNs"])/1e6, float(r["Percentage"]), float(r["AverageNs"])/1e3, r["Name"][:110]))
PY
find $out -name "*kernel_trace.csv" -delete
