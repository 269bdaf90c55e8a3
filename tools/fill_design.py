"""Fill the @@...@@ placeholders of DESIGN.md section 6 from profiles/r03_*.json (run after tools/install_record.py)."""
import json
import re
d = json.load(open("profiles/r03_bench_line.json"))
c3, c4, c5 = (json.load(open("profiles/r03_bench_%s.json" % w)) for w in ("cfg3", "cfg4", "cfg5"))
dre = json.load(open("profiles/r03_bench_cfg4_dre.json"))
r2 = json.load(open("profiles/r03_bench_2ranks_one_gpu_rehearsal.json"))
vals = {"VALUE": "%.1f" % d["value"], "MS": "%.1f" % d["ms_per_step"], "ITS": "%.1f" % d["config"]["gmres_iters_per_shift_solve"],
        "VPCIE": "%.1f" % d.get("value_pcie_inclusive", float("nan")), "MSPCIE": "%.1f" % d.get("ms_per_step_pcie_inclusive", float("nan")),
        "VFP64": "%.1f" % d["value_fp64_storage"], "VPY": "%.1f" % d["value_python_sweep_driver"],
        "CFG3": "%.1f" % c3["value"], "CFG4": "%.1f" % c4["value"], "CFG5": "%.2f" % c5["value"], "CFG5S": "%.1f" % (c5["ms_per_step"] / 1e3),
        "DRE_S": "%.0f" % (dre["ms_per_step"] / 1e3), "DRE_V": "%.1f" % dre["value"],
        "CPU": "%.1f" % d["cpu_baseline"]["value"], "CPU1": "%.1f" % d["cpu_baseline"]["single_core"]["value"], "R2": "%.0f" % r2["value"]}
for f in ("DESIGN.md", "README.md"):
    s = open(f).read()
    for k, v in vals.items():
        s = s.replace("@@%s@@" % k, v)
    left = re.findall(r"@@[A-Z0-9_]+@@", s)
    open(f, "w").write(s)
    print(f, "filled", len(vals), "left:", left)
