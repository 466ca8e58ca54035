#!/bin/bash
# rocprofv3 PMC passes (counters only, no trace domains besides kernel-trace) over the implicit-GEMM kernels of
# tools/pmc_igemm_workload.py: wave-cycle breakdown, LDS activity / conflicts, matrix-pipe busy cycles.  Run through gpurun from
# the repo root; prints one line per (pass, kernel) with the per-launch counter sums.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_igemm
rm -rf $out; mkdir -p $out
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/p$i -o pmc -- python3 tools/pmc_igemm_workload.py > $out/p$i.log 2>&1 < /dev/null
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_igemm/p*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if 'conv_igemm2' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k in acc:
        print(f.split('/')[-2], k, {c: round(v / 5) for c, v in acc[k].items()})
PY
