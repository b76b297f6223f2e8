#!/bin/bash
# Kernel-trace statistics and HBM traffic counters for bench.py's headline workload, per the MI355X guide: the PMC passes are
# separate runs with --pmc only (never combined with a trace domain), one counter per pass; plus the known-bytes gather probe
# that calibrates FETCH_SIZE for the bucket-accumulation access pattern.
# Two kernel-trace passes: the multi-MSM pipeline at ONE internal stream (TKMK_MSM_STREAMS=1: every kernel alone on the device —
# the durations bench.py's roofline quotes from its serialised profiling pass) and at the default three (the timed region's form,
# --no-serial-pass so that the statistics hold only those launches).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh TAG [bench.py args...]
set -e -o pipefail
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-secondary --no-cpu-baseline"
export TKMK_MSM_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -- python3 $R/bench.py $B --steps 3 --warmup 1 "$@" > $OUT/bench_stats_1stream.json 2> $OUT/stats1.err
unset TKMK_MSM_STREAMS
echo "stats pass (1 stream) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3 -- python3 $R/bench.py $B --no-serial-pass --steps 3 --warmup 1 "$@" > $OUT/bench_stats_3stream.json 2> $OUT/stats3.err
echo "stats pass (3 streams) done"
python3 $R/bench.py $B --steps 5 --warmup 1 "$@" > $OUT/bench_plain.json 2> $OUT/plain.err
echo "plain run done"
export TKMK_MSM_STREAMS=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $B --steps 1 --warmup 1 "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $B --steps 1 --warmup 1 "$@" > $OUT/bench_write.json 2> $OUT/write.err
echo "write pass done"
unset TKMK_MSM_STREAMS
python3 $R/tools/gather_probe.py > $OUT/gather_probe_plain.json 2> $OUT/probe.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/probe_fetch -- python3 $R/tools/gather_probe.py > $OUT/gather_probe_under_pmc.json 2>> $OUT/probe.err
echo "probe passes done"
python3 $R/tools/pmc_summary.py $OUT/pmc.json FETCH_SIZE=$OUT/fetch WRITE_SIZE=$OUT/write
python3 $R/tools/pmc_summary.py $OUT/pmc_probe.json FETCH_SIZE=$OUT/probe_fetch
cp $(find $OUT/stats1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_1stream.csv
cp $(find $OUT/stats3 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_3stream.csv
# keep the merge-back small: drop the raw traces
rm -rf $OUT/stats1 $OUT/stats3 $OUT/fetch $OUT/write $OUT/probe_fetch
ls -la $OUT
