#!/bin/bash
# Kernel-trace statistics and HBM traffic counters for bench.py's headline workload, per the MI355X guide: the PMC passes are
# separate runs with --pmc only (never combined with a trace domain), one counter per pass; plus the known-bytes gather probe
# that calibrates FETCH_SIZE for the bucket-accumulation access pattern.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh TAG [bench.py args...]
set -e -o pipefail
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-secondary --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $B --steps 3 --warmup 1 "$@" > $OUT/bench_stats.json 2> $OUT/stats.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $B --steps 1 --warmup 1 "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $B --steps 1 --warmup 1 "$@" > $OUT/bench_write.json 2> $OUT/write.err
echo "write pass done"
python3 $R/tools/gather_probe.py > $OUT/gather_probe_plain.json 2> $OUT/probe.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/probe_fetch -- python3 $R/tools/gather_probe.py > $OUT/gather_probe_under_pmc.json 2>> $OUT/probe.err
echo "probe passes done"
python3 $R/tools/pmc_summary.py $OUT/pmc.json FETCH_SIZE=$OUT/fetch WRITE_SIZE=$OUT/write
python3 $R/tools/pmc_summary.py $OUT/pmc_probe.json FETCH_SIZE=$OUT/probe_fetch
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# keep the merge-back small: drop the raw traces
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/probe_fetch
ls -la $OUT
