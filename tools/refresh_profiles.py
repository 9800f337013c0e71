"""Copies the summaries of a tools/profile_r02.sh run (gpurun_out/<tag>/, gpurun_out/profile_<tag>.txt) into the
tracked profiles/ files: r02_summary.txt / .json, r02_bench.json, r02_kernel_stats.csv, gram_traffic.json, and the
rocprofv3 sections at the top of r02_new_kernels.txt / r02_spmm_wide.txt (the hand-written parts below them stay)."""
import glob, json, os, re, shutil, sys
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, 'gpurun_out', tag)
P = os.path.join(R, 'profiles')
shutil.copy(os.path.join(O, 'summary.txt'), os.path.join(P, 'r02_summary.txt'))
shutil.copy(os.path.join(O, 'summary.json'), os.path.join(P, 'r02_summary.json'))
shutil.copy(os.path.join(O, 'bench.json'), os.path.join(P, 'r02_bench.json'))
stats = max(glob.glob(O + '/stats/*/*kernel_stats.csv'), key=os.path.getmtime)
shutil.copy(stats, os.path.join(P, 'r02_kernel_stats.csv'))
summ = json.load(open(os.path.join(O, 'summary.json')))
bench = json.loads(open(os.path.join(O, 'bench.json')).read().strip().splitlines()[-1])
g = json.load(open(os.path.join(P, 'gram_traffic.json')))
name = max((k for k in summ['kernels'] if k.startswith('gram_') and 'finalize' not in k), key=lambda k: summ['kernels'][k]['calls'])
g['kernel'] = re.sub(r'^[^(]*', name, g['kernel'], count=1) if '(' in g['kernel'] else name
g['hbm_bytes_per_launch'] = bench['roofline']['traffic'] if bench['roofline'].get('traffic') else g['hbm_bytes_per_launch']
g['avg_us_rocprof_kernel_trace'] = summ['kernels'][name]['avg_us']
json.dump(g, open(os.path.join(P, 'gram_traffic.json'), 'w'), indent=1)
text = open(os.path.join(R, 'gpurun_out', 'profile_%s.txt' % tag)).read()
sec = {}
for m in re.finditer(r'^== (\w+)[^\n]*\n(?:  [^\n]*\n?)*', text, re.M):
    sec[m.group(1)] = m.group(0).rstrip('\n')


def replace_sections(path, names):
    s = open(path).read()
    for n in names:
        if n not in sec:
            print('missing section', n); continue
        s, k = re.subn(r'^== %s\b[^\n]*\n(?:  [^\n]*\n?)*' % n, lambda _: sec[n] + '\n', s, count=1, flags=re.M)
        if not k:
            print('section not found in', path, n)
    open(path, 'w').write(s)


replace_sections(os.path.join(P, 'r02_new_kernels.txt'), ['c5_stats', 'ilu_stats', 'ilu_fe_stats', 'pca_stats', 'c5_fetch', 'c5_mfma', 'pca_mfma'])
replace_sections(os.path.join(P, 'r02_spmm_wide.txt'), ['spmm_fe_stats', 'spmm_band_stats', 'spmm_fe_fetch', 'spmm_fe_write', 'spmm_band_fetch', 'spmm_band_write'])
print('kernel', name, g['avg_us_rocprof_kernel_trace'], 'bench avg_launch_ms', bench['roofline']['avg_launch_ms'])
