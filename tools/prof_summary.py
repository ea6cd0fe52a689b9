"""Summarise a rocprofv3 kernel_stats.csv: ms per step per kernel.  usage: prof_summary.py <csv> <steps_total>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
tot = sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:60]:
    name = r['Name'].split('(')[0][:95]
    print(f"{int(r['TotalDurationNs'])/n/1e6:8.3f} ms/step {int(r['Calls'])/n:6.1f} calls/step {float(r['Percentage']):6.2f}%  {name}")
print(f"total {tot/n/1e6:.3f} ms/step")
