#!/bin/bash
# SQ counters of one layer of scripts/bench_conv.py: scripts/pmc_probe.sh <tag> <ONLY filter> <WHICH> [B]
tag=$1; only=$2; which=$3; b=${4:-32}
out=gpurun_out/probe_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ONLY="$only" WHICH="$which" B=$b REPS=2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/a -- python3 scripts/bench_conv.py > $out/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $out/b -- python3 scripts/bench_conv.py > $out/b.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("a","b"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            if "conv" not in k: continue
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
        for k,v in agg.items():
            print(k, {c: round(x/max(n[(k,c)],1)) for c,x in v.items()})
PY
