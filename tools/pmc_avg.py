"""Mean counter value per dispatch of one kernel from a rocprofv3 counter_collection.csv.
python tools/pmc_avg.py file.csv kernel_substring"""
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-28s mean %.4e over %d dispatches" % (k, sum(v) / len(v), len(v)))
