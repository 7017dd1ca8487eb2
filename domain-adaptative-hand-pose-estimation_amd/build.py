"""Build libmi355pose.so (HIP, gfx950 only) in-tree: python build.py [--force]

hipcc cross-compiles without a GPU.  Objects are cached under csrc/_build/ by source mtime.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libmi355pose.so')
SOURCES = ['api.hip', 'igemm.hip', 'igemm_fp8.hip', 'bn.hip', 'pool_layout.hip', 'pw21.hip', 'heatmap.hip', 'optim.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-ffp-contract=off']   # no implicit FMA contraction: keep fp32 parity with the ATen op order


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=True):
    bdir = os.path.join(CSRC, '_build')
    os.makedirs(bdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'igemm_common.h'), os.path.join(HERE, '..', 'include', 'mi355pose.h')]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(bdir, s.replace('.hip', '.o'))
        if force or _newer([src] + hdrs, obj):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-6000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(bdir, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _newer(objs, OUT):
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
