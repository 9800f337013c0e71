"""Prints the top rows of the newest rocprofv3 kernel_stats.csv under a directory: python tools/kstats.py DIR [rows]"""
import csv, glob, os, sys
d = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 14
g = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)
f = max(g, key=os.path.getmtime)
tot = 0.0
for i, r in enumerate(csv.DictReader(open(f))):
    tot += float(r['TotalDurationNs'])
    if i < rows:
        print('%-78s calls=%6s avg=%10.1f us total=%9.2f ms %5s%%' % (r['Name'].split('(')[0].replace('void rlh::', '')[:78], r['Calls'],
              float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
print('all kernels: %.2f ms' % (tot / 1e6))
