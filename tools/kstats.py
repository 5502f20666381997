#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 kernel_stats.csv: tools/kstats.py <csv> [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print('%-66s calls/step %5.1f  ms/step %7.2f  avg %7.3f' % (r['Name'].split('(')[0][-66:], int(r['Calls']) / steps,
          float(r['TotalDurationNs']) / steps / 1e6, float(r['AverageNs']) / 1e6))
print('total ms/step %.2f' % (tot / steps / 1e6))
