"""Prints VGPR/SGPR/LDS/occupancy per kernel: compiles each csrc/*.hip with
-Rpass-analysis=kernel-resource-usage (cross-compile, no GPU needed)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'raleigh_amd', 'csrc')

def main(files):
    tmp = tempfile.mkdtemp()
    for f in files:
        src = os.path.join(CSRC, f + '.hip')
        r = subprocess.run(['hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17',
                            '-I', os.path.join(ROOT, 'include'), '-I', CSRC, '-c', src, '-o',
                            os.path.join(tmp, f + '.o'), '-Rpass-analysis=kernel-resource-usage'],
                           capture_output=True, text=True)
        txt = r.stderr
        blocks = txt.split('Function Name: ')[1:]
        for b in blocks:
            name = b.split()[0]
            dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
            dn = dn.replace('rlh::', '')
            g = lambda k: re.search(k + r': (\d+)', b).group(1)
            print('%-7s %-95s v=%s a=%s s=%s scr=%s occ=%s lds=%s' % (
                f, dn[:95], g(' VGPRs'), g('AGPRs'), g('TotalSGPRs'), g(r'ScratchSize \[bytes/lane\]'),
                g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))

if __name__ == '__main__':
    main(sys.argv[1:] or ['gram', 'update', 'spmm', 'spmm_wide_s', 'spmm_wide_d', 'spmm_wide_c', 'spmm_wide_z', 'sptrsv', 'dense'])
