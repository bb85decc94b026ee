#!/bin/bash
# Run ON THE GPU BOX (via gpurun):  bash tools/collect_profiles.sh rNN
# Produces gpurun_out/prof_<tag>/ : kernel-trace stats of the default bench command, and separate PMC passes
# (FETCH_SIZE, WRITE_SIZE -- they cannot share a pass on gfx950, and gpurun forbids --pmc together with the trace
# domains) for the strict path and for the fused non-parity tier.  tools/summarize_profiles.py then turns them into
# profiles/<tag>_*.   The program after "--" is python3 itself (no env/bash hop: the profiler initialises the GPU).
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
ONE="--steps 1 --warmup 0 --frames 1 --no-cpu-baseline --no-extras"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
SQ2="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
# --no-extras: every launch in this trace is the headline frame size, so the per-kernel average is comparable with bench.py's avg_launch_ms
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_kt.json 2> $OUT/kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_f16 -- python3 bench.py --steps 5 --warmup 2 --tier fast_f16 > $OUT/bench_kt_f16.json 2> $OUT/kt_f16.err
for T in strict fast_f16; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$T -- python3 bench.py $ONE --tier $T > $OUT/bench_pmc_fetch_$T.json 2> $OUT/pmc_fetch_$T.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$T -- python3 bench.py $ONE --tier $T > $OUT/bench_pmc_write_$T.json 2> $OUT/pmc_write_$T.err
  rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_sq_$T -- python3 bench.py $ONE --tier $T > $OUT/bench_pmc_sq_$T.json 2> $OUT/pmc_sq_$T.err
  rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/pmc_sq2_$T -- python3 bench.py $ONE --tier $T > $OUT/bench_pmc_sq2_$T.json 2> $OUT/pmc_sq2_$T.err
done
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*.csv" | head -40
