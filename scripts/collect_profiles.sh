#!/bin/bash
# copy the summaries of one scripts/profile_round.sh run into profiles/ (tracked): scripts/collect_profiles.sh <tag> <round> [suffix]
# kernel-stats CSVs get the suffix (e.g. _v2), the large per-dispatch counter CSVs overwrite the round's previous ones
tag=$1; rnd=$2; sfx=$3
src=gpurun_out/prof_$tag
cp $src/f32/*/*kernel_stats.csv profiles/${rnd}_bench_b32_kernel_stats$sfx.csv
cp $src/f32_serial/*/*kernel_stats.csv profiles/${rnd}_bench_b32_serial_kernel_stats$sfx.csv
cp $src/bf16/*/*kernel_stats.csv profiles/${rnd}_bench_bf16_b64_kernel_stats$sfx.csv
cp $src/infer/*/*kernel_stats.csv profiles/${rnd}_infer_fp32_b64_kernel_stats$sfx.csv
cp $src/pmcf_f32/*/*counter_collection.csv profiles/${rnd}_pmc_fetch_size.csv
cp $src/pmcw_f32/*/*counter_collection.csv profiles/${rnd}_pmc_write_size.csv
cp $src/pmcf_bf16/*/*counter_collection.csv profiles/${rnd}_pmc_fetch_size_bf16.csv
cp $src/pmcw_bf16/*/*counter_collection.csv profiles/${rnd}_pmc_write_size_bf16.csv
cp $src/calf/*/*counter_collection.csv profiles/${rnd}_pmc_calib_fetch.csv
cp $src/calw/*/*counter_collection.csv profiles/${rnd}_pmc_calib_write.csv
if [ -d $src/pmcf_infer ]; then
  cp $src/pmcf_infer/*/*counter_collection.csv profiles/${rnd}_pmc_fetch_size_infer.csv
  cp $src/pmcw_infer/*/*counter_collection.csv profiles/${rnd}_pmc_write_size_infer.csv
fi
python scripts/pmc_calib_report.py profiles/${rnd}_pmc_calib_fetch.csv profiles/${rnd}_pmc_calib_write.csv
python scripts/make_traffic_json.py profiles/${rnd}_pmc_fetch_size.csv profiles/${rnd}_pmc_write_size.csv \
    profiles/${rnd}_pmc_fetch_size_bf16.csv profiles/${rnd}_pmc_write_size_bf16.csv | tail -n 40
if [ -f profiles/${rnd}_pmc_fetch_size_infer.csv ]; then
  python scripts/make_traffic_json.py --out traffic_infer.json profiles/${rnd}_pmc_fetch_size_infer.csv profiles/${rnd}_pmc_write_size_infer.csv | tail -n 12
fi
