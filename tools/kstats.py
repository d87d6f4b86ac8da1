"""Top rows of a rocprofv3 kernel_stats.csv: name, calls, total ms, average us.   python tools/kstats.py file.csv [n]"""
import csv, sys
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= n: break
    print("%-52s %7s %10.1f ms %10.1f us" % (r["Name"].split("(")[0].replace("void ", "")[:52], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
