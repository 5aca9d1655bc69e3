"""Turns what profiles/collect_round.sh left under gpurun_out/ into the committed summaries:
    python profiles/summarise_round.py r03
-> profiles/<tag>_kernel_stats_<C>.csv   rocprofv3 --kernel-trace --stats, first rows
   profiles/<tag>_pmc_<C>.csv            mean PMC counters per launch: the render kernel and the
                                         classification kernel in front of it
   profiles/<tag>_bench.json             the bench line of the same build (+ _driver_cmd: --steps 20 --warmup 5)
   profiles/pmc_<C>.json                 what bench.py quotes as roofline.traffic / fp64_valu for each config,
                                         tied to the build by the hash of csrc/ and the kernel name"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench       # noqa: E402
import workloads   # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = os.path.join(ROOT, "gpurun_out")

for c in ("C2", "C3", "C5", "C4", "default"):
    # (an earlier collection's files may lie beside the last one's: the newest wins)
    for f in sorted(glob.glob(os.path.join(out, "%s_trace_%s" % (tag, c), "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime):
        rows = open(f).read().splitlines()[:6]
        name = "%s_kernel_stats_%s.csv" % (tag, c if c != "default" else "default_cmd")
        open(os.path.join(ROOT, "profiles", name), "w").write("\n".join(rows) + "\n")

for name in ("bench", "bench_driver_cmd"):
    b = os.path.join(out, "%s_%s.json" % (tag, name))
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, name)))

for c in ("C2", "C3", "C5", "C4"):
    f = os.path.join(out, "pmc_%s_%s" % (tag, c), "summary.csv")
    if not os.path.exists(f):
        continue
    shutil.copy(f, os.path.join(ROOT, "profiles", "%s_pmc_%s.csv" % (tag, c)))
    rows = list(csv.DictReader(open(f)))
    p = {r["counter"]: float(r["mean_per_launch"]) for r in rows if r["kernel"] == "render"}
    q = {r["counter"]: float(r["mean_per_launch"]) for r in rows if r["kernel"] == "classify"}
    if "WRITE_SIZE" not in p or "GRBM_GUI_ACTIVE" not in p:
        print(c, "incomplete counters:", sorted(p))
        continue
    line = json.load(open(os.path.join(out, "pmc_%s_%s" % (tag, c), "g1.json")))
    kernel = line["roofline"]["kernel"]
    cfg = workloads.CONFIGS[c]
    rendered = cfg["height"] // 32 * 32 * cfg["width"]
    w, f_ = p["WRITE_SIZE"] * 1024., p["FETCH_SIZE"] * 1024.
    wc, fc = q.get("WRITE_SIZE", 0.) * 1024., q.get("FETCH_SIZE", 0.) * 1024.
    gui = p["GRBM_GUI_ACTIVE"] / 8.                                   # summed over the 8 XCDs
    latest = {
        "config": c, "n_gpus": 1, "kernel": kernel, "csrc_sha16": bench.csrc_hash(),
        "source": "profiles/%s_pmc_%s.csv (rocprofv3 --pmc, WRITE_SIZE / FETCH_SIZE in passes of their own, mean per launch "
                  "of the default strict kernel; the counters are in KB)" % (tag, c),
        "write_bytes_per_launch": w, "fetch_bytes_per_launch_uncorrected": f_,
        "classification_launch": {"write_bytes": wc, "fetch_bytes_uncorrected": fc,
                                  "valu_wave_instructions": q.get("SQ_INSTS_VALU"), "waves": q.get("SQ_WAVES")} if q else None,
        "hbm_bytes_per_launch": w + f_ + wc + fc,
        "algorithmic_bytes_per_launch": {"f64_frame": rendered * 24, "u8_display_frame": rendered * 3},
        "note": "per frame: the render launch and the classification launch in front of it.  FETCH_SIZE is not doubled: the "
                "kernel's reads are scalar loads and scratch reloads, not the wide streaming reads the x2 gfx950 correction "
                "was calibrated on; WRITE_SIZE is exact for its 16-byte stores",
        "valu_busy_frac": p["SQ_ACTIVE_INST_VALU"] * 4. / (gui * 1024.),
        "valu_lanes_active_frac": p["SQ_THREAD_CYCLES_VALU"] / p["SQ_ACTIVE_INST_VALU"] / 64.,
        "valu_wave_instructions": p["SQ_INSTS_VALU"], "salu_wave_instructions": p["SQ_INSTS_SALU"],
        "wait_any_frac": p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"], "wait_inst_any_frac": p["SQ_WAIT_INST_ANY"] / p["SQ_WAVE_CYCLES"],
        "waves": p["SQ_WAVES"], "vmem_write_instructions": p.get("SQ_INSTS_VMEM_WR"),
        "valu_note": "busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); lanes = SQ_THREAD_CYCLES_VALU / "
                     "SQ_ACTIVE_INST_VALU / 64 (reads 0.90 also for a kernel in which every lane executes every "
                     "instruction: not a measure of idle ray slots, profiles/r02_handover.txt)",
    }
    json.dump(latest, open(os.path.join(ROOT, "profiles", "pmc_%s.json" % c), "w"), indent=1)
    print(c, "VALU/wave %.0f  SALU/wave %.0f  busy %.2f  wait_any %.2f  wait_inst %.2f  write MB %.1f (algorithmic %.1f)  fetch MB %.1f" % (
        p["SQ_INSTS_VALU"] / p["SQ_WAVES"], p["SQ_INSTS_SALU"] / p["SQ_WAVES"], latest["valu_busy_frac"], latest["wait_any_frac"],
        latest["wait_inst_any_frac"], (w + wc) / 1e6, rendered * 27 / 1e6, (f_ + fc) / 1e6))
