#!/bin/bash
# LDS bank-conflict counters per kernel (one PMC pass per family, no trace domains mixed in)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_lds
rm -rf $out; mkdir -p $out
for m in vgg dcgan; do
  timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/$m -o pmc -- python3 bench.py --model $m --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-families --no-train-leg --no-f32mfma-leg --no-make-gifs-leg --no-extra-legs --no-check --sustained-s 0 > $out/$m.log 2>&1 < /dev/null
  echo "$m rc=$?"
done
python3 tools/pmc_summary.py $out "*" > $out/summary.json 2> $out/summary.err
rm -f $out/*/*/*kernel_trace.csv $out/*/*kernel_trace.csv
ls -R $out | head -30
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_lds/summary.json"))
for k,v in d.items():
    if "SQ_LDS_IDX_ACTIVE" in v and v["SQ_LDS_IDX_ACTIVE"]["avg"]>0:
        a=v["SQ_LDS_IDX_ACTIVE"]["avg"]; c=v["SQ_LDS_BANK_CONFLICT"]["avg"]
        print("%-28s n=%4d idx_active %.3e conflict %.3e (%.1f%%) mfma_busy %.3e gui %.3e"%(k,v["SQ_LDS_IDX_ACTIVE"]["dispatches"],a,c,100*c/a,v.get("SQ_VALU_MFMA_BUSY_CYCLES",{}).get("avg",0),v.get("GRBM_GUI_ACTIVE",{}).get("avg",0)))
PY
