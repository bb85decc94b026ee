#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed profiles/<tag>_* files and
profiles/pmc_conv12.json (read by bench.py for roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


try:
    import subprocess
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    head = "unknown"


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None      # the latest collection wins


def short(name):
    tier = ""
    if "<true" in name:
        tier = " [strict]"
    elif "<false" in name:
        tier = " [fast tier]"
    if "k_conv12_mfma<" in name:          # first template argument = RELAX mask: 0 strict, 3 fast tier, 1 / 2 single-layer relaxations
        r = name.split("k_conv12_mfma<")[1].split(",")[0].strip()
        return "k_conv12_mfma" + {"0": " [strict]", "3": " [fast tier]"}.get(r, " [relaxed %s]" % r)
    if "k_conv3<" in name:              # k_conv3<STRICT, OFF64, X64, SDMA>: the third argument marks the relaxed-x64 experiment
        targs = name.split("k_conv3<")[1].split(">")[0].replace(" ", "").split(",")
        if len(targs) >= 3 and targs[2] == "true":
            return "k_conv3 [relaxed x64]"
        return "k_conv3 [strict]"
    if "k_rs2d_dma" in name:
        return "k_rs2d_dma [plane -> plane, LDS-DMA]"
    if "k_rs2d<" in name:
        kind = name.split("k_rs2d<")[1].split(",")[0].strip()
        return {"0": "k_rs2d [plane -> plane]", "1": "k_rs2d [RGB -> upscaled Y]", "2": "k_rs2d [fused chroma resample + colour merge]"}.get(kind, "k_rs2d")
    for key in ("k_fused_f16", "k_conv12_mfma", "k_conv3_fast", "k_conv3", "k_resample_rows", "k_resample_cols", "k_rgb_split",
                "k_ycc_merge"):
        if key in name:
            return key + (" [fast tier]" if key in ("k_fused_f16", "k_conv3_fast") else tier)
    return name[:40]


lines = ["# rocprofv3 summary `%s` (MI355X, `python3 bench.py`, strict mode, 3840x2160 -> 7680x4320 frames)\n" % tag]
for sub, title in (("kt", "`bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras` (strict headline; every launch of a layer kernel is one band of a 3840x2160 -> 7680x4320 frame: one per frame under the default scratch cap)"), ("kt_f16", "`bench.py --tier fast_f16` (non-parity fused kernel)")):
    stats = one(sub + "/**/*_kernel_stats.csv")
    if not stats:
        continue
    name = tag + ("_kernel_stats.csv" if sub == "kt" else "_kernel_stats_f16.csv")
    shutil.copy(stats, os.path.join(dst, name))
    lines.append("\n## kernel-trace --stats, %s (`%s`)\n" % (title, name))
    lines.append("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|")
    for r in csv.DictReader(open(stats)):
        lines.append("| %s | %s | %.4f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6,
                                                       float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
pmc = {}
pmc_n = {}
for which in ("fetch", "write", "sq", "sq2"):
    for tier in ("strict", "fast_f16"):
        f = one("pmc_%s_%s/**/*_counter_collection.csv" % (which, tier))
        if not f:
            continue
        shutil.copy(f, os.path.join(dst, "%s_pmc_%s_%s.csv" % (tag, which, tier)))
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if tier == "fast_f16" and "fast tier" not in k:
                k += " (in the fast_f16 run: whole frame per launch)"       # never mixed with the strict run's banded launches
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            pmc.setdefault(k, {})[c] = sum(v) / len(v)
            pmc_n.setdefault(k, {})[c] = len(v)          # the PMC runs push ONE frame through: launches counted = launches per frame
if pmc:
    lines.append("\n## PMC, per LAUNCH, of one 3840x2160 -> 7680x4320 frame (separate passes; FETCH_SIZE/WRITE_SIZE are in KiB).  A launch of the "
                 "layer kernels is one band of the frame: launches per frame = " +
                 ", ".join("%s %d" % (k.split(" ")[0], n.get("FETCH_SIZE", 0)) for k, n in pmc_n.items() if ("strict" in k or "rs2d_dma" in k) and "fast_f16 run" not in k) + "\n")
    cols = sorted({c for v in pmc.values() for c in v})
    lines.append("| kernel | " + " | ".join(cols) + " |\n|---|" + "---|" * len(cols))
    for k, v in pmc.items():
        lines.append("| %s | " % k + " | ".join("%.6g" % v.get(c, float("nan")) for c in cols) + " |")
    k12 = pmc.get("k_conv12_mfma [strict]")
    if k12 and "FETCH_SIZE" in k12 and "WRITE_SIZE" in k12:
        n_out = 7680 * 4320
        lpf12 = max(1, pmc_n["k_conv12_mfma [strict]"].get("FETCH_SIZE", 1))          # launches (bands) per frame
        fetch, write = k12["FETCH_SIZE"] * 1024, k12["WRITE_SIZE"] * 1024              # per launch
        sys.path.insert(0, ROOT)
        from libsrcnn_amd import build as _b
        rec = {"kernel": "k_conv12_mfma", "tag": tag, "kernel_source_sha256": _b.kernel_source_sha("k_conv12_mfma"), "measured_at": "tools/collect_profiles.sh %s, code at commit %s" % (tag, head),
               "fetch_bytes": fetch, "write_bytes": write,
               "launches_per_frame": lpf12,
               "hbm_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": 132 * n_out / lpf12,
               "note": "FETCH_SIZE/WRITE_SIZE are KiB, separate passes.  The guide's x2 FETCH correction is for 16 B/lane streams; "
                       "our loads are 4 B/lane, calibrated on k_conv3 of the same run: it requests 5.65 GB (4.25 GB of unique "
                       "layer-2 planes x 1.33 halo) and FETCH_SIZE reads 4.73 GB -- a halved counter would imply 9.5 GB, more "
                       "than was requested -- so FETCH_SIZE is taken at face value.  WRITE_SIZE matches 128 B/px exactly."}
        # the other kernels of the strict path, same passes: what one 4K -> 8K frame moves through HBM in total
        # (per FRAME: per-launch average x launches per frame)
        path = {"k_conv12_mfma": {"fetch_bytes": fetch * lpf12, "write_bytes": write * lpf12, "launches_per_frame": lpf12,
                                  "kernel_source_sha256": rec["kernel_source_sha256"]}}
        for kname, key in (("k_conv3", "k_conv3 [strict]"), ("k_rs2d_dma", "k_rs2d_dma [plane -> plane, LDS-DMA]")):
            kk = pmc.get(key)
            if kk and "FETCH_SIZE" in kk and "WRITE_SIZE" in kk:
                lpf = max(1, pmc_n[key].get("FETCH_SIZE", 1))
                path[kname] = {"fetch_bytes": kk["FETCH_SIZE"] * 1024 * lpf, "write_bytes": kk["WRITE_SIZE"] * 1024 * lpf,
                               "launches_per_frame": lpf, "kernel_source_sha256": _b.kernel_source_sha(kname)}
        rec["path"] = path
        if len(path) == 3:
            total = sum(v["fetch_bytes"] + v["write_bytes"] for v in path.values())
            rec["whole_path"] = {"hbm_bytes_per_frame": total, "algorithmic_bytes_per_frame": 5 * n_out,
                                 "ratio": total / (5 * n_out),
                                 "note": "algorithmic = 1 B of low-res input + 4 B of output per output pixel (SURVEY 8d, upscale fused); the "
                                         "excess is the 32 layer-2 planes making one round trip through HBM between k_conv12_mfma and k_conv3"}
            lines.append("\nwhole strict path per frame: %.3f GB through HBM vs %.3f GB algorithmic (5 B/px) -> ratio %.1f"
                         % (total / 1e9, 5 * n_out / 1e9, total / (5 * n_out)))
        json.dump(rec, open(os.path.join(dst, tag + "_pmc_conv12.json"), "w"), indent=1)
        lines.append("\nconv12 HBM traffic per launch (%d per frame) = %.3f GB (fetch %.3f + write %.3f) vs algorithmic %.3f GB -> ratio %.3f"
                     % (lpf12, (fetch + write) / 1e9, fetch / 1e9, write / 1e9, 132 * n_out / lpf12 / 1e9, (fetch + write) * lpf12 / (132 * n_out)))
    kf = pmc.get("k_fused_f16 [fast tier]")
    if kf and "FETCH_SIZE" in kf and "WRITE_SIZE" in kf:
        n_out = 7680 * 4320
        lines.append("fused fp16 kernel HBM traffic per launch = fetch %.3f GB + write %.3f GB vs algorithmic %.3f GB (4 B in + 4 B out per pixel)"
                     % (kf["FETCH_SIZE"] * 1024 / 1e9, kf["WRITE_SIZE"] * 1024 / 1e9, 8 * n_out / 1e9))
b = one("bench.json")
if b and os.path.getsize(b):
    shutil.copy(b, os.path.join(dst, tag + "_bench.json"))
    lines.append("\n## bench line of the same build (`%s_bench.json`)\n\n```\n%s```" % (tag, open(b).read()))
open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

# ---- ProcessSRCNN (tools/profile_process.sh): the colour shell + the banded Y path as ONE call runs them ----
pstats = one("process_kt/**/*_kernel_stats.csv")
if pstats:
    shutil.copy(pstats, os.path.join(dst, tag + "_process_kernel_stats.csv"))
    ppmc = {}
    for which in ("fetch", "write"):
        f = one("process_%s/**/*_counter_collection.csv" % which)
        if f:
            shutil.copy(f, os.path.join(dst, "%s_process_pmc_%s.csv" % (tag, which)))
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                ppmc.setdefault(k, {})[which] = sum(v) / len(v)
    pl = ["# ProcessSRCNN(3840x2160 RGB, x2) on MI355X, `%s` (code at commit %s)\n" % (tag, head),
          "`rocprofv3 --kernel-trace --stats -- python3 tools/process_probe.py --reps 4`; FETCH_SIZE / WRITE_SIZE from separate "
          "`--pmc` passes, KiB per launch as rocprofv3 reports them (one launch = one band of the output).\n",
          "| kernel | launches | avg ms | total ms | % | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch |", "|---|---|---|---|---|---|---|"]
    calls = 4.0
    per_call = collections.OrderedDict()
    for r in csv.DictReader(open(pstats)):
        k = short(r["Name"])
        pm = ppmc.get(k, {})
        pl.append("| %s | %s | %.4f | %.2f | %s | %.0f | %.0f |" % (k, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6,
                                                              r["Percentage"], pm.get("fetch", float("nan")), pm.get("write", float("nan"))))
        per_call[k] = float(r["TotalDurationNs"]) / 1e6 / calls
    pl.append("\nDevice time per call: " + ", ".join("%s %.2f ms" % kv for kv in per_call.items()) + " = %.2f ms." % sum(per_call.values()))
    shell = sum(v for k, v in per_call.items() if "k_rs2d" in k or "split" in k or "merge" in k or "resample" in k)
    pl.append("Colour shell + resamplers (everything that is not a convolution layer): %.2f ms per call." % shell)
    wj = one("process_wall.json")
    if wj:
        pl.append("\nWall time of the call, un-profiled (`tools/process_probe.py --reps 8`): `%s`" % open(wj).read().strip())
    open(os.path.join(dst, tag + "_process_summary.md"), "w").write("\n".join(pl) + "\n")
    print("\n".join(pl))
