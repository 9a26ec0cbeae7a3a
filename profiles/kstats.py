"""print a rocprofv3 *_kernel_stats.csv compactly: python profiles/kstats.py <csv>"""
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"].replace("ycge::", "").replace("void ", "").split("(")[0]
    print(f"{n:40s} calls={row['Calls']:>4s} avg_us={float(row['AverageNs'])/1e3:10.1f} min={float(row['MinNs'])/1e3:10.1f} max={float(row['MaxNs'])/1e3:10.1f} {row['Percentage']:>6s}%")
