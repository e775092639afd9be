"""Print the rows of a rocprofv3 kernel_stats.csv whose kernel name contains a pattern: python tools/kstats.py <csv> <pattern>"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2].lower() in r["Name"].lower():
        print(f"   {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:9.1f} us  min {float(r['MinNs']) / 1e3:9.1f}")
