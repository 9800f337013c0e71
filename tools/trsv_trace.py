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
names = ['setup (taken -> far wait)', 'far wait', 'far gather', 'to throttle', 'throttle wait', 'near gather (polls)', 'reduce + store + done']
d = np.stack([s[:, i + 1] - s[:, i] for i in range(6)] + [s[:, 6] - s[:, 0]], 1)
for i, nm in enumerate(names[:6] + ['whole unit']):
    col = d[:, i] if i < 6 else d[:, 6]
    print('%-28s median %7.2f us  mean %7.2f  p90 %7.2f' % (nm if i < 6 else 'whole unit', us(np.median(col)), us(col.mean()), us(np.percentile(col, 90))))
done = s[:, 6] - t0
print('first taken -> last done: %.1f us' % us(done.max()))
order = np.argsort(np.nonzero(ok)[0])
dd = np.diff(done)
print('done[u+1] - done[u]: median %.2f us, mean %.2f us' % (us(np.median(dd)), us(dd.mean())))
# lead: how long before it finishes is a unit taken / past its throttle
print('taken -> done lead: median %.1f us; throttle passed -> done: median %.2f us; near gather start -> done %.2f us'
      % (us(np.median(s[:, 6] - s[:, 0])), us(np.median(s[:, 6] - s[:, 4])), us(np.median(s[:, 6] - s[:, 4]))))
near = (t[ok][:, 7] >> np.uint64(32)).astype(np.int64)
near[near >= len(s)] = -1
xcc = ((t[ok][:, 7] >> np.uint64(24)) & np.uint64(0xff)).astype(int)
blk = (t[ok][:, 7] & np.uint64(0xffffff)).astype(int)
has = near >= 0
lat = s[has, 5] - s[near[has], 6]
print('near gather end - done word of the last unit it needs: median %.2f us, p10 %.2f, p90 %.2f' % (us(np.median(lat)), us(np.percentile(lat, 10)), us(np.percentile(lat, 90))))
gap = np.arange(len(s))[has] - near[has]
print('units between a unit and the last unit it needs: median %d, mean %.1f' % (np.median(gap), gap.mean()))
fin = s[:, 6] - s[:, 5]
print('near gather end (wave 0) -> done: median %.2f us' % us(np.median(fin)))
print('  of it: slice sum %.2f, right-hand side + store issued %.2f, LDS wait %.2f, barrier (the other waves) %.2f'
      % tuple(us(np.median(s[:, j] - s[:, i])) for i, j in ((5, 8), (8, 9), (9, 10), (10, 6))))
print('XCDs seen in group 0:', np.unique(xcc), ' workgroups:', len(np.unique(blk)))
k = len(s) // 2
print('sample units around %d: (taken, far-wait-end, far-end, thr-end, near-end, done) relative us' % k)
for u in range(k, min(k + 12, len(s))):
    r = s[u]
    print('  unit %6d: ' % u + ' '.join('%9.2f' % us(r[i] - t0) for i in (0, 2, 3, 4, 5, 6)) + '  wg %d needs %d' % (blk[u], near[u]))
