#!/bin/bash
# the round's rocprofv3 evidence in one go (run on the GPU box from the repo root): scripts/profile_round.sh <tag>
# kernel-trace stats of the default fp32 command (overlapped + serial) and of the bf16 leg, then the FETCH_SIZE /
# WRITE_SIZE counter passes (separate runs, --pmc only) of both.  Outputs under gpurun_out/prof_<tag>/.
tag=${1:-x}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="--steps 3 --warmup 1 --no-cpu-baseline --no-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/f32 -- python3 bench.py $A > $out/f32.log 2>&1
DT_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/f32_serial -- python3 bench.py $A > $out/f32_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bf16 -- python3 bench.py --precision bf16 --batch 64 $A > $out/bf16.log 2>&1
P="--steps 1 --warmup 1 --no-cpu-baseline --no-legs"
DT_OVERLAP_WGRAD=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmcf_f32 -- python3 bench.py $P > $out/pmcf_f32.log 2>&1
DT_OVERLAP_WGRAD=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmcw_f32 -- python3 bench.py $P > $out/pmcw_f32.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmcf_bf16 -- python3 bench.py --precision bf16 --batch 64 --graph off $P > $out/pmcf_bf16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmcw_bf16 -- python3 bench.py --precision bf16 --batch 64 --graph off $P > $out/pmcw_bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/infer -- python3 bench.py --mode infer --size 256 --batch 64 --steps 10 --warmup 2 > $out/infer.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmcf_infer -- python3 bench.py --mode infer --size 256 --batch 64 --steps 2 --warmup 1 > $out/pmcf_infer.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmcw_infer -- python3 bench.py --mode infer --size 256 --batch 64 --steps 2 --warmup 1 > $out/pmcw_infer.log 2>&1
# counter calibration on known-byte kernels (scripts/ubench/pmc_calib.hip, built in-tree before the call)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/calf -- scripts/ubench/pmc_calib > $out/calf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/calw -- scripts/ubench/pmc_calib > $out/calw.log 2>&1
# keep what is needed, drop the per-dispatch traces (the merge back is capped at 64 MiB)
find $out -name "*kernel_trace.csv" -path "*pmc*" -delete
find $out -name "*agent_info.csv" -delete
ls -la $out $out/*/* | head -60
