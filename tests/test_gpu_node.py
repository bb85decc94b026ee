"""Node-level paths (one process, several contexts) on the one GPU the test box has: K VIRTUAL contexts that alias device 0.
Each test runs tests/node_worker.py in a process of its own (contexts are process-wide).  The bar is the same as
everywhere: the oracle's bytes / the whole-frame call's bits."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "node_worker.py")


def run(cmd, arg=0, env=None, timeout=900):
    e = dict(os.environ)
    e.pop("SRCNN_DEVICES", None)
    if env:
        e.update(env)
    r = subprocess.run([sys.executable, WORKER, cmd, str(arg)], env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_processsrcnn_over_4_virtual_contexts():
    """The reference's one ProcessSRCNN call uses its whole machine (src/libsrcnn.cpp:665,791-798,817-824); with 4 contexts
    ours deals the bands of a large image to all of them: byte-equal to the oracle, to the one-context result at 4K, and
    under 4 concurrent host threads."""
    r = run("process", 4)
    assert r["contexts"] == 4
    assert all(r["vs_oracle"]), r
    assert r["lanes_after_oracle_cases"] >= 4                    # every context took part
    assert not r["thread_errors"], r
    assert r["big_rc"] == 0 and r["big_equal"] and r["big_rows_vs_oracle"], r
    assert r["contexts_after_reinit"] == 1


@pytest.mark.parametrize("nctx", [3, 8])
def test_node_tiled_frame_equals_whole_frame(nctx):
    r = run("node_tiled", nctx)
    assert r["contexts"] == nctx
    assert all(c["equal"] for c in r["cases"]), r
    assert r["vs_oracle"]


def test_host_stream_dealt_over_contexts():
    r = run("stream", 2)
    assert r["contexts"] == 2 and all(r["equal"]) and r["single_frame"], r


@pytest.mark.parametrize("check", ["", "0"], ids=["default", "SRCNN_COMM_CHECK=0"])
def test_rccl_tiled_call_with_overlapped_sub_band_gather(check):
    """World = 1 on the REAL librccl (world > 1 runs on the stand-in: tests/test_gpu_comm_ranks.py): every sub-band count and
    ragged heights, the explicit-offset gather, the bounded waits (srcnn_comm_wait / srcnn_comm_set_timeout_ms); second run
    with the cross-rank table check (on by default since round 5) switched off."""
    r = run("comm_tiled", env={"SRCNN_COMM_CHECK": check})
    assert all(c["equal"] for c in r["cases"]), r
    assert r["gatherv_at"]
    assert r["comm_check_env"] == check
    assert r["timeout_prev"] == 60000 and r["timeout_prev2"] == 5000, r
    assert r["comm_wait"] == 0 and r["barrier_short_deadline"] == 0 and r["barrier_no_deadline"] == 0, r
    assert r["wait_without_comm"] == -204, r


def test_a_missed_comm_deadline_aborts_the_communicator_and_a_new_one_works():
    """no wait on a peer is unbounded.  The only way to miss a deadline on a one-GPU box is to make the
    stream slow: 0.5 s of kernels against a 40 ms deadline.  srcnn_comm_wait gives up at the deadline and returns SRCNN_E_COMM
    once ncclCommAbort has returned (on a real one-rank RCCL communicator; the abort itself waits for the device to drain OUR
    kernels here -- measured 0.56 s -- where a spinning send / recv kernel would be told to quit), later calls are refused at
    once, and destroy + init bring everything back -- including the tiled call, bit-equal to the whole-frame result."""
    r = run("comm_deadline")
    assert r["prev_timeout"] == 60000
    assert r["wait_rc"] == -204 and "deadline" in r["wait_error"], r
    assert 30 <= r["wait_ms"] < 5000, r
    assert r["barrier_after_rc"] == -204 and "aborted" in r["barrier_after_error"], r
    assert r["gather_after_rc"] == -204, r
    assert r["destroy_rc"] == 0 and r["barrier_new_comm_rc"] == 0 and r["wait_new_comm_rc"] == 0, r
    assert r["tiled_equal"], r


def test_more_callers_than_lanes_queue_up():
    r = run("queue", 2, env={"SRCNN_MAX_LANES": "1"})
    assert not r["errors"], r
    assert r["contexts"] == 2 and r["lanes"] == 2, r            # one lane per context, shared by the three callers


def test_env_srcnn_devices_self_init():
    r = run("env_devices", env={"SRCNN_DEVICES": "0,0,0"})
    assert r["contexts"] == 3 and r["equal"] and r["process_equal"] and r["lanes"] >= 3, r
    r = run("env_devices", env={"SRCNN_DEVICES": "all"})
    assert r["contexts"] == 1 and r["equal"] and r["process_equal"], r


@pytest.mark.parametrize("env", [{}, {"SRCNN_SHELL_UNFUSED": "1"},
                                 {"SRCNN_RESAMPLE_2PASS": "1", "SRCNN_SHELL_UNFUSED": "1"}, {"SRCNN_RS_TPB": "1"},
                                 {"SRCNN_MAX_WORKSPACE_MB": "48"}, {"SRCNN_RS_DMA": "0", "SRCNN_NUMA": "0"},
                                 # host-side switches (independent of one another, so they share runs)
                                 {"SRCNN_THP": "0", "SRCNN_PREFAULT_THREADS": "3", "SRCNN_SPIN_WAIT": "1", "SRCNN_DEVICE_WAIT_IN": "1"},
                                 {"SRCNN_PREFAULT": "0", "SRCNN_BANDS": "0.03,0.07,0.2,0.3,0.3,0.05,0.02", "SRCNN_CONV12_SPREAD": "0", "SRCNN_CONV3_OFF64": "1"},
                                 # the two resampler switches ALONE (they used to leave the fused shell selected, whose RGB
                                 # source only k_rs2d can read: every up-scaling call failed); the second run also deals the layer-1+2
                                 # tiles with the static stride instead of the tile queue
                                 {"SRCNN_RESAMPLE_2PASS": "1"}, {"SRCNN_CONV12_DMA": "0", "SRCNN_CONV12_QUEUE": "0", "SRCNN_CONV3_WDMA": "0"}],
                         ids=["default", "unfused-shell", "two-pass", "tpb1", "small-budget", "no-dma-resampler-no-numa",
                              "no-thp-3-prefaulters-runtime-waits-device-stage-in", "no-prefault-eight-bands-no-quarter-spread-conv3-64bit-offsets",
                              "two-pass-alone", "layer-kernel-fallbacks-static-stride"])
def test_processsrcnn_kernel_selections_all_bit_exact(env):
    """The fused colour shell / k_rs2d (default) and every fallback they replace produce the oracle's bytes; so does a
    workspace budget small enough to force many bands inside srcnn_process_u8."""
    r = run("process_paths", env=env)
    assert all(r["process"]), (env, r)
    assert all(r["y"]), (env, r)
