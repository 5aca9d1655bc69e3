"""Turns what profiles/collect_round.sh left under gpurun_out/ into the committed summaries:
    python profiles/summarise_round.py r02
-> profiles/<tag>_kernel_stats_<C>.csv   rocprofv3 --kernel-trace --stats, first rows
   profiles/<tag>_pmc_<C>.csv            mean PMC counters per launch of the render kernel
   profiles/<tag>_bench.json             the bench line of the same build
   profiles/pmc_latest.json              what bench.py quotes as roofline.traffic / fp64_valu for C2,
                                         tied to the build by the hash of csrc/ and the kernel name"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = os.path.join(ROOT, "gpurun_out")

for c in ("C2", "C3", "C5", "C4"):
    for f in glob.glob(os.path.join(out, "%s_trace_%s" % (tag, c), "**", "*kernel_stats.csv"), recursive=True):
        rows = open(f).read().splitlines()[:5]
        open(os.path.join(ROOT, "profiles", "%s_kernel_stats_%s.csv" % (tag, c)), "w").write("\n".join(rows) + "\n")

pmc = {}
for c in ("C2", "C3", "C5"):
    f = os.path.join(out, "pmc_%s_%s" % (tag, c), "summary.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(ROOT, "profiles", "%s_pmc_%s.csv" % (tag, c)))
        pmc[c] = {r["counter"]: float(r["mean_per_launch"]) for r in csv.DictReader(open(f))}

b = os.path.join(out, "%s_bench.json" % tag)
if os.path.exists(b):
    shutil.copy(b, os.path.join(ROOT, "profiles", "%s_bench.json" % tag))

if "C2" in pmc:
    import bench
    p = pmc["C2"]
    kernel = json.load(open(os.path.join(out, "pmc_%s_C2" % tag, "g1.json")))["roofline"]["kernel"]
    w, f = p["WRITE_SIZE"] * 1024., p["FETCH_SIZE"] * 1024.
    gui = p["GRBM_GUI_ACTIVE"] / 8.                                   # summed over the 8 XCDs
    latest = {
        "config": "C2", "n_gpus": 1, "kernel": kernel, "csrc_sha16": bench.csrc_hash(),
        "source": "profiles/%s_pmc_C2.csv (rocprofv3 --pmc, WRITE_SIZE / FETCH_SIZE in passes of their own, mean per launch "
                  "of the default strict kernel; the counters are in KB)" % tag,
        "write_bytes_per_launch": w, "fetch_bytes_per_launch_uncorrected": f, "hbm_bytes_per_launch": w + f,
        "algorithmic_bytes_per_launch": {"f64_frame": 1920 * 1056 * 24, "u8_display_frame": 1920 * 1056 * 3},
        "note": "FETCH_SIZE is not doubled: the kernel's reads are scalar loads and scratch reloads, not the wide "
                "streaming reads the x2 gfx950 correction was calibrated on; WRITE_SIZE is exact for its 16-byte stores",
        "valu_busy_frac": p["SQ_ACTIVE_INST_VALU"] * 4. / (gui * 1024.),
        "valu_lanes_active_frac": p["SQ_THREAD_CYCLES_VALU"] / p["SQ_ACTIVE_INST_VALU"] / 64.,
        "valu_wave_instructions": p["SQ_INSTS_VALU"], "salu_wave_instructions": p["SQ_INSTS_SALU"],
        "wait_any_frac": p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"], "wait_inst_any_frac": p["SQ_WAIT_INST_ANY"] / p["SQ_WAVE_CYCLES"],
        "waves": p["SQ_WAVES"],
        "valu_note": "busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); lanes = SQ_THREAD_CYCLES_VALU / "
                     "SQ_ACTIVE_INST_VALU / 64 (reads 0.90 also for a kernel in which every lane executes every "
                     "instruction: not a measure of idle ray slots, profiles/r02_handover.txt)",
    }
    json.dump(latest, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
    print(json.dumps(latest, indent=1))
for c, p in pmc.items():
    print(c, "VALU/wave %.0f  SALU/wave %.0f  wait_any %.2f  wait_inst %.2f  write MB %.1f" % (
        p["SQ_INSTS_VALU"] / p["SQ_WAVES"], p["SQ_INSTS_SALU"] / p["SQ_WAVES"], p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"],
        p["SQ_WAIT_INST_ANY"] / p["SQ_WAVE_CYCLES"], p["WRITE_SIZE"] / 1024.))
