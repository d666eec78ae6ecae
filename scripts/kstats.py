#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: calls, avg us, total us, short kernel name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "")[:60]
    print(f'{int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:10.1f} us  {float(r["TotalDurationNs"])/1e3:10.1f} us {100*float(r["TotalDurationNs"])/tot:5.1f}%  {name}')
