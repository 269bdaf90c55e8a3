"""Copy the files of a tools/r4_final.sh run (gpurun_out/TAG) into profiles/ under their round-4 names and
derive the two summaries that are not plain copies (K1 kernel statistics, K1 traffic).
python tools/install_record.py TAG"""
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
dst = "profiles"
plain = {"bench.json": "r04_bench_line.json", "bench_kernel_stats.csv": "r04_bench_kernel_stats.csv",
         "classes58_kernel_stats.csv": "r04_kernel_classes_cfg2_g16_stats.csv",
         "bench_2ranks_one_gpu.json": "r04_bench_2ranks_one_gpu_rehearsal.json", "cfg3.json": "r04_bench_cfg3.json",
         "cfg3-cycle.json": "r04_bench_cfg3_cycle.json", "cfg4.json": "r04_bench_cfg4.json", "cfg5.json": "r04_bench_cfg5.json",
         "cfg4_dre.json": "r04_bench_cfg4_dre.json", "dre2_kernel_stats.csv": "r04_cfg4_dre_kernel_stats.csv",
         "bench_4ranks_one_gpu.json": "r04_bench_4ranks_one_gpu_rehearsal.json",
         "bench_5ranks_one_gpu.json": "r04_bench_5ranks_one_gpu_rehearsal.json"}
for a, b in plain.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))


def stats_block(csvf, logf, title, cmd):
    rows = list(csv.DictReader(open(csvf)))
    ev = [ln.strip() for ln in open(logf) if "us per launch" in ln]
    out = ["# %s: %s" % (title, cmd), "# HIP events in the same run: %s" % (ev[0] if ev else "n/a"),
           "%-100s %8s %14s %12s" % ("kernel", "calls", "avg_us", "percent")]
    for r in rows[:4]:
        out.append("%-100s %8s %14.2f %12s" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:6]))
    return "\n".join(out)


if not os.path.exists(os.path.join(src, "spmm58_kernel_stats.csv")):
    print("installed", tag, "(part b: plain copies only)")
    sys.exit(0)
txt = ["rocprofv3 --kernel-trace --stats, MI355X, round 4 (tools/r4_final.sh)",
       stats_block(os.path.join(src, "spmm58_kernel_stats.csv"), os.path.join(src, "spmm58.log"),
                   "cfg2 (n=29 930), 16 groups, m=16", "python tools/spmm_batch_pmc.py 58 16 200"), "",
       stats_block(os.path.join(src, "spmm236_kernel_stats.csv"), os.path.join(src, "spmm236.log"),
                   "cfg5 (n=499 850), 16 groups, m=16", "python tools/spmm_batch_pmc.py 236 16 50"), ""]
open(os.path.join(dst, "r04_spmm_kernel_stats.txt"), "w").write("\n".join(txt))


def mean_of(f):
    m = re.search(r"mean ([0-9.e+-]+) over", open(f).read())
    return float(m.group(1))


old = json.load(open(os.path.join(dst, "r03_spmm_traffic.json")))
old["_comment"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/r4_final.sh: python "
                   "tools/spmm_batch_pmc.py N 16), mean per dispatch of the batched K1 launch (16 groups, m = 16), MI355X round 4. "
                   "hbm_bytes_uncalibrated_2x_fetch = (2*FETCH_SIZE + WRITE_SIZE)*1024 is round 3's reading (the guide's factor 2 "
                   "for 16-B/lane streams); hbm_bytes is the calibrated reading of _note; hbm_bytes_lower uses 1x FETCH_SIZE. "
                   "Infinity-cache hits are counted, not excluded.")
old["_note"] = ("round 4: FETCH_SIZE / WRITE_SIZE re-collected (tools/r4_final.sh a) on the launch as the iteration issues it "
                "(FP32 Z_j in, FP32 w out).  FETCH_SIZE counts requests x 64 B (profiles/r04_pmc_calibration.txt): "
                "hbm_bytes = (FETCH - matrix/2) + matrix + WRITE, matrix = the value / index stream of the kernel")
for key, n in (("16x29930x16", "58"), ("16x499850x16", "236")):
    fk = mean_of(os.path.join(src, "spmm%s_FETCH_SIZE.txt" % n))
    wk = mean_of(os.path.join(src, "spmm%s_WRITE_SIZE.txt" % n))
    old[key].update(FETCH_SIZE_KB=fk, WRITE_SIZE_KB=wk, hbm_bytes=int((2 * fk + wk) * 1024),
                    hbm_bytes_lower=int((fk + wk) * 1024))
    # FP32 x gathers (64-B requests) are counted in full, the matrix stream (>= 128-B requests) at half
    # (profiles/r04_pmc_calibration.txt).  cfg5: multi-shift kernel, 18 B per non-zero once; cfg2 (second half of
    # round 4: the per-group kernel also reads the FP32 Z_j and writes the FP32 w): 10 B per non-zero and group
    matrix = 18.0 * 14369733 if n == "236" else 10.0 * 847488 * 16
    old[key].update(hbm_bytes_uncalibrated_2x_fetch=old[key]["hbm_bytes"],
                    hbm_bytes=int((fk * 1024 - matrix / 2) + matrix + wk * 1024))
    old[key]["kernel"] = ("ricadi::spmm_blocked_ms_kernel<false, float, true> (multi-shift: values of all groups from one read; 16-byte tile fill)"
                          if n == "236" else "ricadi::spmm_blocked_kernel<false, false, float> (one value array per group)") \
        + "; x gathered from the FP32-stored Z_j, w written as an FP32 panel" 
json.dump(old, open(os.path.join(dst, "r04_spmm_traffic.json"), "w"), indent=1)
# the bench line of the record was printed before these PMC passes were installed: its traffic fields are refreshed
# from them (same arithmetic as bench.py: traffic = hbm_bytes of the launch, ratio to the batched-form bytes)
line_f = os.path.join(dst, "r04_bench_line.json")
if os.path.exists(line_f):
    line = json.load(open(line_f))
    for obj, key in (("roofline", "16x29930x16"), ("roofline_cfg5", "16x499850x16")):
        if obj in line and key in old:
            line[obj]["traffic"] = old[key]["hbm_bytes"]
            line[obj]["traffic_source"] = "r04_spmm_traffic.json (PMC passes of this record, installed by tools/install_record.py)"
            line[obj]["traffic_over_batched_form"] = round(old[key]["hbm_bytes"] / line[obj]["algorithmic_bytes_batched_form"], 3)
            if "bytes_batched_form_as_stored" in line[obj]:
                line[obj]["traffic_over_batched_form_as_stored"] = round(
                    old[key]["hbm_bytes"] / line[obj]["bytes_batched_form_as_stored"], 3)
    json.dump(line, open(line_f, "w"))
print("installed", tag)
