"""Build libmi355pose.so (HIP, gfx950 only) in-tree: python build.py [--force] [--variant NAME --flags "..."]

hipcc cross-compiles without a GPU.  Objects are cached under csrc/_build/ by source mtime.

After linking, the device code of the library is disassembled and checked (`isa_gate`): the build FAILS when it contains a
packed-fp32 arithmetic instruction (any VOP3P `v_pk_*_f32` opcode) that takes an operand half from the OTHER register of its
pair (an `op_sel:[...]` with a 1 in it, or an `op_sel_hi:[...]` with a 0 in it).  That instruction form lost its swapped operand in lanes
48-63 of the conv accumulate epilogue now and then on MI355X (round 2; established in round 3 by replacing only that
instruction in the kernel's assembly -- the src1-swapped form fails, the src0-swapped one did not in that kernel; the gate
refuses both: DESIGN.md section 7 "dropped addend", profiles/dropped_addend_repro.py).  The
library is compiled with -fno-slp-vectorize, so hipcc forms no packed fp32 arithmetic from scalar code at all; the gate
is what keeps it that way when flags or sources change.

`--variant NAME --flags "..."` builds an experiment copy (objects in csrc/_build_NAME/, library in scratch/ab/libNAME.so,
loaded with MI355_LIB=...): same sources, extra / replaced compiler flags, for same-box A/B runs.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libmi355pose.so')
SOURCES = ['api.hip', 'igemm.hip', 'pgemm.hip', 'igemm_fp8.hip', 'wgrad_fp8.hip', 'bn.hip', 'pool_layout.hip', 'pw21.hip', 'heatmap.hip', 'optim.hip']
HEADERS = ['common.h', 'igemm_common.h', 'fp8_common.h']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
LLVM_BIN = os.environ.get('MI355_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-ffp-contract=off',       # no implicit FMA contraction: keep fp32 parity with the ATen op order
         '-fno-slp-vectorize']      # no packed fp32 VALU formed from scalar code (see the module docstring / isa_gate)

_PK = re.compile(r'\bv_pk_[a-z0-9_]+_f32\b')             # every VOP3P fp32-pair arithmetic opcode (add, mul, fma and whatever else the ISA grows)
_OPSEL = re.compile(r'\bop_sel:\[([01,]+)\]')           # default [0,0(,0)]: low results from the low registers
_OPSEL_HI = re.compile(r'\bop_sel_hi:\[([01,]+)\]')     # default [1,1(,1)]: high results from the high registers


def _negated(flag):
    if flag.startswith('-fno-'):
        return '-f' + flag[5:]
    if flag.startswith('-f'):
        return '-fno-' + flag[2:]
    return None


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def device_disassembly(so_path):
    """Yield (kernel symbol, instruction text) for every instruction of every gfx950 code object bundled in the library."""
    objdump = os.path.join(LLVM_BIN, 'llvm-objdump')
    tmp = tempfile.mkdtemp(prefix='mi355_isa_')
    try:
        copy = os.path.join(tmp, 'lib.so')
        shutil.copy(so_path, copy)
        # `--offloading` writes one file per bundle entry next to its input: work on a copy in a scratch directory
        r = subprocess.run([objdump, '--offloading', copy], capture_output=True, text=True, cwd=tmp)
        if r.returncode != 0:
            raise RuntimeError('llvm-objdump --offloading failed:\n' + r.stderr[-2000:])
        cos = sorted(f for f in os.listdir(tmp) if 'amdgcn' in f and 'gfx950' in f)
        if not cos:
            raise RuntimeError('no gfx950 code object found in %s' % so_path)
        for co in cos:
            r = subprocess.run([objdump, '-d', '--mcpu=gfx950', os.path.join(tmp, co)], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('disassembly of %s failed:\n%s' % (co, r.stderr[-2000:]))
            sym = '?'
            for line in r.stdout.splitlines():
                m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
                if m:
                    sym = m.group(1)
                    continue
                yield sym, line.split('//')[0].strip()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def swaps_halves(ins):
    """True for a packed-fp32 arithmetic instruction that takes an operand half from the other register of its pair."""
    if not _PK.search(ins):
        return False
    m, h = _OPSEL.search(ins), _OPSEL_HI.search(ins)
    return bool((m and '1' in m.group(1)) or (h and '0' in h.group(1)))


def isa_gate(so_path):
    """Fail on packed-fp32 arithmetic whose low result reads the high register of an operand pair (module docstring).
    Returns the number of device instructions inspected."""
    n, hits = 0, []
    for sym, ins in device_disassembly(so_path):
        n += 1
        if swaps_halves(ins):
            hits.append('%s: %s' % (sym, ins))
    if n < 1000:
        raise RuntimeError('isa gate: only %d device instructions found in %s -- the disassembly step is broken' % (n, so_path))
    if hits:
        raise RuntimeError('isa gate: %d packed-fp32 instruction(s) with a register-swapping op_sel in %s (see build.py docstring); '
                           'restructure the source (mask / select on the packed words, or scalar adds) until they are gone:\n  %s'
                           % (len(hits), so_path, '\n  '.join(hits[:20])))
    return n


def build(force=False, verbose=True, extra_flags=(), variant=None, gate=True):
    bdir = os.path.join(CSRC, '_build' if not variant else '_build_' + variant)
    out = OUT
    if variant:
        out = os.path.join(HERE, '..', 'scratch', 'ab', 'lib%s.so' % variant)
        os.makedirs(os.path.dirname(out), exist_ok=True)
    os.makedirs(bdir, exist_ok=True)
    flags = [f for f in FLAGS if _negated(f) not in extra_flags] + list(extra_flags)      # an extra flag may switch a default off
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(HERE, '..', 'include', 'mi355pose.h'), os.path.abspath(__file__)]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(bdir, s.replace('.hip', '.o'))
        if force or _newer([src] + hdrs, obj):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + flags + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-6000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(bdir, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _newer(objs, out):
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
        if gate:
            try:
                isa_gate(out)
            except Exception:
                os.remove(out)          # a library that fails the gate must not be loadable
                raise
    return out


def build_host_sanitized(out_dir, driver_src, verbose=False):
    """CPU-box sanitizer job: the HOST pass of every .hip source compiled with AddressSanitizer +
    UndefinedBehaviorSanitizer (`-fno-gpu-sanitize`: the device code is built as usual, no GPU sanitizer is involved) and linked with
    `driver_src` into one executable, which is returned.  The driver exercises the host halves of the entry points --
    descriptor checks, tap / phase tables, split plans, grouped argument blocks, workspace sizing -- on a machine without a GPU:
    launches fail in the HIP runtime after the host code under test has run."""
    os.makedirs(out_dir, exist_ok=True)
    san = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer', '-g', '-O1']
    base = ['--offload-arch=gfx950', '-fno-gpu-sanitize', '-std=c++17', '-fPIC', '-Wno-unused-function', '-ffp-contract=off', '-fno-slp-vectorize'] + san

    def cc(src):
        obj = os.path.join(out_dir, os.path.basename(src).rsplit('.', 1)[0] + '.o')
        cmd = [HIPCC] + base + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('sanitizer build failed for %s:\n%s' % (src, r.stderr[-4000:]))
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, [os.path.join(CSRC, s) for s in SOURCES] + [driver_src]))
    exe = os.path.join(out_dir, 'host_sanitize_driver')
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-fno-gpu-sanitize'] + san + ['-o', exe] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('sanitizer link failed:\n' + r.stderr[-4000:])
    return exe


if __name__ == '__main__':
    a = sys.argv[1:]
    variant = a[a.index('--variant') + 1] if '--variant' in a else None
    xf = a[a.index('--flags') + 1].split() if '--flags' in a else []
    print(build(force='--force' in a, extra_flags=xf, variant=variant, gate='--no-gate' not in a))
