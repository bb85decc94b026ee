#!/bin/bash
# Run ON THE GPU BOX (via gpurun):  bash tools/profile_process.sh <tag>
# Kernel trace + FETCH_SIZE / WRITE_SIZE passes (separate runs: gpurun forbids --pmc with trace domains, and the two
# counters cannot share a pass on gfx950) of ProcessSRCNN on a 4K RGB image -> gpurun_out/prof_<tag>/process_*.
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="tools/process_probe.py --reps 4 ${PROBE_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/process_kt -- python3 $ARGS > $OUT/process_kt.json 2> $OUT/process_kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/process_fetch -- python3 $ARGS > $OUT/process_fetch.json 2> $OUT/process_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/process_write -- python3 $ARGS > $OUT/process_write.json 2> $OUT/process_write.err
python3 $ARGS --reps 8 > $OUT/process_wall.json 2> $OUT/process_wall.err
find $OUT -name "*process*" -name "*.csv" | head
