"""CPU: the build's staleness check is content-based, and bench.py never quotes an HBM-traffic counter that was measured
on a different text of the dominant kernel."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_stale_is_content_based(tmp_path, monkeypatch):
    from libsrcnn_amd import build as b
    assert os.path.exists(b.LIB) and os.path.exists(b.STAMP), "run __graft_entry__.build() first"
    assert not b.stale()
    # newer file times alone do not make the build stale ...
    os.utime(os.path.join(b.CSRC, "srcnn_capi.cpp"))
    assert not b.stale()
    # ... a different content does, even if the .so is newer than every source (the case of a prebuilt .so shipped with
    # sources that changed under it)
    stamp = open(b.STAMP).read()
    try:
        open(b.STAMP, "w").write("0" * 64 + "\n")
        os.utime(b.LIB)
        assert b.stale()
    finally:
        open(b.STAMP, "w").write(stamp)
    assert not b.stale()


def test_digest_covers_every_source_and_header():
    from libsrcnn_amd import build as b
    have = {os.path.basename(d) for d in b.DEPS}
    for f in os.listdir(b.CSRC):
        if f.endswith((".hip", ".cpp", ".h", ".hpp", ".inc")):
            assert f in have, "%s is compiled or included but not part of the staleness digest" % f
    assert set(b.SOURCES) <= have


def test_traffic_is_withheld_for_a_different_kernel_text(tmp_path, monkeypatch):
    import bench
    from libsrcnn_amd import build as b
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rec = {"hbm_bytes_per_launch": 123.0, "measured_at": "test", "kernel_source_sha256": b.kernel_source_sha("k_conv12_mfma")}
    (prof / "r03_pmc_conv12.json").write_text(json.dumps(rec))
    val, src, _whole = bench.traffic_record()
    assert val == 123.0 and "r03_pmc_conv12.json" in src
    rec["kernel_source_sha256"] = "f" * 64
    (prof / "r03_pmc_conv12.json").write_text(json.dumps(rec))
    val, src, _whole = bench.traffic_record()
    assert val is None and "WITHHELD" in src
    del rec["kernel_source_sha256"]
    (prof / "r03_pmc_conv12.json").write_text(json.dumps(rec))
    val, src, _whole = bench.traffic_record()
    assert val is None and "WITHHELD" in src


def test_kernel_source_sha_sees_the_kernel():
    from libsrcnn_amd import build as b
    a = b.kernel_source_sha("k_conv12_mfma")
    assert a and len(a) == 64
    assert b.kernel_source_sha("k_conv3") not in (None, a)
    assert b.kernel_source_sha("k_does_not_exist") is None
