"""print a rocprofv3 kernel_stats csv (tools/quick_stats.sh) compactly"""
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0].replace('void ', '').replace('r3d::', '')
    print(f"{n:34s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
