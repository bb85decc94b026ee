#!/bin/bash
# round-2 campaign A: full GPU suite, bench line, band-ordered (Infinity-Cache) experiment, f16 co-issue ubench,
# 2 ranks on one GPU through RCCL (may be refused by RCCL: recorded either way)
O=gpurun_out/r2a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for mb in 64 128 256 512 1024; do
  SRCNN_MAX_WORKSPACE_MB=$mb timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 5 > $O/bench_ws${mb}.json 2>> $O/bench.err
done
timeout 300 ./tools/ubench/f16_coissue > $O/f16_coissue.txt 2>&1
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload tiled8k --tiled-size 1920x1080 --steps 2 --warmup 1 > $O/tiled_2ranks_1gpu.log 2>&1; echo "2rank rc=$?" >> $O/tiled_2ranks_1gpu.log
timeout 300 python bench.py --workload tiled8k --steps 3 --warmup 1 > $O/tiled8k_1rank.json 2>> $O/bench.err
ls -la $O
