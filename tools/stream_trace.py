#!/usr/bin/env python3
"""The PCIe-inclusive frame stream (srcnn_y_upscale2x_f32_stream, 8 page-locked 4K frames, hipGraph per slot), three calls:
the command behind `rocprofv3 --kernel-trace --memory-copy-trace` timelines of the stream path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import libsrcnn_amd as S

S.init(0)
print(bench.pcie_inclusive(S, frames=8))
