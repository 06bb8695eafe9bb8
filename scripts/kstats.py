"""Print the top rows of a rocprofv3 kernel_stats.csv found under a directory:  python scripts/kstats.py <dir> [n]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in list(csv.DictReader(open(f)))[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")
    print(f"{re.sub(r'[(].*', '', name)[:60]:60s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):5.1f}%")
