#!/bin/bash
# Run ON THE GPU BOX (via gpurun):  bash tools/collect_profiles.sh rNN
# Produces gpurun_out/prof_<tag>/ : kernel-trace stats of the default bench command, and two separate PMC
# passes (FETCH_SIZE, WRITE_SIZE -- they cannot share a pass on gfx950, and gpurun forbids --pmc together
# with the trace domains).  tools/summarize_profiles.py then turns them into profiles/<tag>_*.
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_kt.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --frames 1 --no-cpu-baseline > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --frames 1 --no-cpu-baseline > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --frames 1 --no-cpu-baseline > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*.csv" | head -20
