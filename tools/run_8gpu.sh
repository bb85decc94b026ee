#!/bin/bash
# Multi-GPU runs of the two shardings on ONE node, one process per GPU:
#   frames  : resident 4K frames sharded across ranks, no data-path collective   (the headline line, weak scaling)
#   tiled8k : ONE 7680x4320 frame -> 15360x8640, one output band per rank + RCCL gatherv to rank 0, verified
#             against the whole-frame result once per run                        (strong scaling)
#   host-stream : BASELINE config #5 shape -- host-resident 4K frames through the two-slot hipGraph stream path on every
#             rank, barrier + max over ranks, summed MPix                        (weak scaling, PCIe inclusive)
# Usage: tools/run_8gpu.sh [--dry-run] [N ...]        default N = 1 2 4 8
#   --dry-run validates, on the CPU and without touching a device, the rank -> device mapping, per-rank frame
#   ownership (seed = 0x5C0DE000 + frame index), the band partition and the buffer sizes for every N.
# bench.py launches its own ranks (subprocess children created before anything touches the GPU: no exec from a HIP
# process); `python3 -m torch.distributed.run ... bench.py --gpus N` works as well.
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
DRY=0
if [ "${1:-}" = "--dry-run" ]; then DRY=1; shift; fi
NS=("$@"); [ ${#NS[@]} -eq 0 ] && NS=(1 2 4 8)
PORT=${MASTER_PORT:-29531}
for N in "${NS[@]}"; do
    if [ $DRY -eq 1 ]; then
        python3 - "$N" <<'PY'
import json, sys
sys.path.insert(0, ".")
from libsrcnn_amd import multigpu
n = int(sys.argv[1])
p = multigpu.plan(n)
print(json.dumps({"world": n,
                  "devices": [r["device"] for r in p["ranks"]],
                  "first_frame_seed": [r["frames"]["seeds"][0] for r in p["ranks"]],
                  "band_rows": [[r["band"]["row0"], r["band"]["rows"]] for r in p["ranks"]],
                  "band_MB": [round(r["band"]["band_bytes"] / 1e6, 1) for r in p["ranks"]],
                  "band_scratch_GB": [round(r["band"]["scratch_bytes"] / 1e9, 2) for r in p["ranks"]],
                  "frames_scratch_GB": round(p["ranks"][0]["frames"]["scratch_bytes"] / 1e9, 2),
                  "root_frame_MB": round(p["tiled_frame"]["root_bytes"] / 1e6, 1)}))
PY
        continue
    fi
    for WL in frames tiled8k host-stream; do
        echo "== N=$N workload=$WL" >&2
        # bench.py is its own launcher: without WORLD_SIZE in the environment it starts the N ranks itself (fresh children,
        # the parent never touches the GPU) and prints the one aggregated JSON line
        python3 bench.py --gpus "$N" --steps 5 --warmup 2 --workload $WL --no-extras --no-cpu-baseline
    done
    echo "== N=$N workload=host-stream, plain launches (no hipGraph replay)" >&2
    python3 bench.py --gpus "$N" --steps 5 --warmup 2 --workload host-stream --plain-launches --no-extras --no-cpu-baseline
done
if [ $DRY -eq 0 ]; then
    # the single-process model: ONE process drives every visible GPU (contexts + peer copies, no RCCL): ProcessSRCNN over
    # all devices, the node-tiled frame, the frame stream dealt over the contexts
    echo "== one process, all devices (tools/node_probe.py --devices all)" >&2
    python3 tools/node_probe.py --devices all
fi
