"""Condenses a tools/profile_round.sh output directory into the text + JSON summaries that
are committed under profiles/ (per-kernel durations; HBM traffic per launch from the
FETCH_SIZE / WRITE_SIZE passes with the gfx950 correction: FETCH_SIZE counts 64 B per 128-B
request for wide coalesced reads, so read bytes = 2 * FETCH_SIZE KiB; WRITE_SIZE is exact)."""
import collections
import csv
import glob
import json
import os
import sys

O = sys.argv[1]
out = {}
import os.path
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)     # a reused tag keeps older runs around
f = [newest(O + '/stats/*/*kernel_stats.csv')]
print('== rocprofv3 --kernel-trace --stats (bench.py --steps 5 --warmup 2)')
for r in csv.DictReader(open(f[0])):
    name = r['Name'].split('(')[0].replace('void rlh::', '')
    print('%-58s calls=%4s avg=%10.1f us  total=%8.2f ms  %5s%%' % (
        name[:58], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
    out.setdefault('kernels', {})[name] = {'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3}
# per-dispatch durations of the two-operand Gram: those reading 2 blocks (grid identical; split by duration rank)
trace = newest(O + '/stats/*/*kernel_trace.csv')
durs = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    durs[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
acc = {}
for tag in ('fetch', 'write'):
    f = newest(O + '/%s/*/*counter_collection.csv' % tag)
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        a[r['Kernel_Name'].split('(')[0].replace('void rlh::', '')].append(float(r['Counter_Value']))
    acc[tag] = a
print('== HBM traffic per launch (GB): read = 2 * FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB')
for k in sorted(acc['fetch']):
    rd = [2 * v * 1024 / 1e9 for v in acc['fetch'][k]]
    wr = [v * 1024 / 1e9 for v in acc['write'].get(k, [0])]
    print('%-58s launches=%3d read avg=%7.3f max=%7.3f  write avg=%7.3f' % (
        k[:58], len(rd), sum(rd) / len(rd), max(rd), sum(wr) / len(wr)))
    out.setdefault('traffic_gb', {})[k] = {'read_avg': sum(rd) / len(rd), 'read_max': max(rd),
                                           'write_avg': sum(wr) / len(wr)}
g = [k for k in acc['fetch'] if k.startswith('gram_') and 'finalize' not in k]
if g:
    # the headline's two-operand X.dot(Y) launches: the Gram instance launched most often (the self-Gram and
    # the stacked multi-block Grams of the fused leg run under other template instances)
    gk = max(g, key=lambda k: len(acc['fetch'][k]))
    rd = sorted(2 * v * 1024 for v in acc['fetch'][gk])
    two = [v for v in rd if v > 0.75 * rd[-1]]
    wr = acc['write'][gk]
    out['gram_two_operand'] = {'kernel': gk, 'hbm_bytes_per_launch': int(sum(two) / len(two) + sum(wr) / len(wr) * 1024),
                               'launches': len(two)}
    d = sorted(durs.get('void rlh::' + gk, []))
    if d:
        twod = [v for v in d if v > 0.75 * d[-1]]
        out['gram_two_operand']['avg_us_rocprof'] = sum(twod) / len(twod)
        print('two-operand Gram launches (%s): avg %.1f us over %d dispatches (kernel trace)' % (gk, sum(twod) / len(twod), len(twod)))
json.dump(out, open(O + '/summary.json', 'w'), indent=1)
