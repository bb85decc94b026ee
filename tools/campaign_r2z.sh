#!/bin/bash
# round-2 final campaign: full GPU suite, profiles (kernel trace + PMC, both tiers), bench line, microbenchmarks, side workloads
O=gpurun_out/r2z; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/collect_profiles.sh r02 > $O/collect.log 2>&1
cp gpurun_out/prof_r02/bench.json $O/bench.json
timeout 300 ./tools/ubench/strict_tap > $O/strict_tap.txt 2>&1
timeout 300 ./tools/ubench/f16_coissue > $O/f16_coissue.txt 2>&1
python tools/fused_timeline.py > $O/fused_timeline.txt 2>&1
python tools/fast_error.py 2160 3840 > $O/fast_error_4k.txt 2>&1
python tools/fast_error.py >> $O/fast_error_4k.txt 2>&1
python tools/quick_bench.py > $O/quick_bench.txt 2>&1
for wl in tiled8k batch1080p frames-graph host-stream; do timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 > $O/side_$wl.json 2>> $O/side.err; done
timeout 300 python bench.py --tier fast_f16 --steps 10 --warmup 3 > $O/bench_f16.json 2>> $O/side.err
timeout 300 python bench.py --tier fast --steps 10 --warmup 3 > $O/bench_fast.json 2>> $O/side.err
ls $O
