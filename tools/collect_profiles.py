"""Copies what tools/profile_r04.sh left under gpurun_out/<tag>/ into profiles/ (the committed evidence):
bench lines, kernel stats, counter summaries, the new-kernel timings; keeps hand-added sections of the old
<tag>_new_kernels.txt that the script does not regenerate.  usage: tools/collect_profiles.py r04 [forced_bench.json]"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
O = os.path.join(ROOT, 'gpurun_out', tag)
P = os.path.join(ROOT, 'profiles')
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
line = lambda f: open(f).read().strip().splitlines()[-1] + '\n'
open(os.path.join(P, tag + '_bench.json'), 'w').write(line(os.path.join(O, 'bench.json')))
forced = sys.argv[2] if len(sys.argv) > 2 else os.path.join(O, 'bench_forced.json')
open(os.path.join(P, tag + '_bench_forced_collectives.json'), 'w').write(line(forced))
shutil.copy(os.path.join(O, 'summary.txt'), os.path.join(P, tag + '_summary.txt'))
shutil.copy(os.path.join(O, 'summary.json'), os.path.join(P, tag + '_summary.json'))
shutil.copy(newest(O + '/stats/*/*kernel_stats.csv'), os.path.join(P, tag + '_kernel_stats.csv'))
summ = json.load(open(os.path.join(O, 'summary.json')))
# the two-operand Gram's traffic for bench.py's roofline.traffic_from_profile
g = summ['gram_two_operand']
d = json.load(open(os.path.join(P, 'gram_traffic.json')))
d['hbm_bytes_per_launch'] = int(g['hbm_bytes_per_launch'])
d['avg_us_rocprof_kernel_trace'] = g['avg_us_rocprof']
d['launches'] = g['launches']
json.dump(d, open(os.path.join(P, 'gram_traffic.json'), 'w'), indent=1)
log = open(os.path.join(ROOT, 'gpurun_out', 'profile_%s.log' % tag)).read()
i = log.index('== c5_solve_stats')
new = log[i:]
oldtxt = open(os.path.join(P, tag + '_new_kernels.txt')).read()
head = oldtxt[:oldtxt.index('== c5_solve_stats')]
keep = oldtxt[oldtxt.index('== fe_spmm_row_pairs'):] if '== fe_spmm_row_pairs' in oldtxt else ''
open(os.path.join(P, tag + '_new_kernels.txt'), 'w').write(head + new.rstrip('\n') + '\n\n' + keep)
print('written')
