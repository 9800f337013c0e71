"""Reads the per-unit stamps the persistent triangular solve writes with RLH_SPTRSV_TRACE=<file> (group 0 only,
100 MHz wall clock) and prints where a unit's life goes and how fast the front advances.
usage: RLH_SPTRSV_TRACE=/tmp/t.bin python tools/ilu_bench.py fe --m 16 ; python tools/trsv_trace.py /tmp/t.bin"""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
ok = t[:, 6] > 0
print('units %d, stamped %d' % (len(t), ok.sum()))
s = t[ok].astype(np.int64)
t0 = s[:, 0].min()
us = lambda x: x * 0.01
spans = (('setup (taken -> wait)', 0, 1), ('wait for the opening unit', 1, 4), ('gather (polls)', 4, 5), ('slice sum', 5, 8), ('store issued', 8, 9),
         ('barrier (other waves) + done', 10, 6), ('whole unit', 0, 6))
for nm, i, j in spans:
    col = s[:, j] - s[:, i]
    print('%-30s median %7.2f us  mean %7.2f  p90 %7.2f' % (nm, us(np.median(col)), us(col.mean()), us(np.percentile(col, 90))))
done = s[:, 6] - t0
print('first taken -> last done: %.1f us' % us(done.max()))
order = np.argsort(np.nonzero(ok)[0])
dd = np.diff(done)
print('done[u+1] - done[u]: median %.2f us, mean %.2f us' % (us(np.median(dd)), us(dd.mean())))
near = (t[ok][:, 7] >> np.uint64(32)).astype(np.int64)
near[near >= len(s)] = -1
xcc = ((t[ok][:, 7] >> np.uint64(24)) & np.uint64(0xff)).astype(int)
blk = (t[ok][:, 7] & np.uint64(0xffffff)).astype(int)
has = near >= 0
lat = s[has, 5] - s[near[has], 6]
print('gather end - done word of the last unit it needs: median %.2f us, p10 %.2f, p90 %.2f' % (us(np.median(lat)), us(np.percentile(lat, 10)), us(np.percentile(lat, 90))))
gap = np.arange(len(s))[has] - near[has]
print('units between a unit and the last unit it needs: median %d, mean %.1f' % (np.median(gap), gap.mean()))
print('XCDs seen in group 0:', np.unique(xcc), ' workgroups:', len(np.unique(blk)))
k = len(s) // 2
print('sample units around %d: (taken, wait end, gather end, done) relative us' % k)
for u in range(k, min(k + 12, len(s))):
    r = s[u]
    print('  unit %6d: ' % u + ' '.join('%9.2f' % us(r[i] - t0) for i in (0, 4, 5, 6)) + '  wg %d needs %d' % (blk[u], near[u]))
