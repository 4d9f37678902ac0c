"""hipcc build recipe for csrc/ -> lib/libempanada_hip.so (gfx950 only).

In-tree, incremental (per-file objects under csrc/_obj, rebuilt when the source
or a header is newer).  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libempanada_hip.so')
ARCH = 'gfx950'

COMMON = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
          '-Wall', '-Wno-unused-function', '-Wno-unused-result']
# bit-exactness of the voting distance (DESIGN.md "distance arithmetic")
PER_FILE = {
    'postprocess.hip': ['-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt'],
}
SOURCES = ['abi.hip', 'conv_igemm.hip', 'layers.hip', 'pointrend.hip', 'postprocess.hip', 'pdl_net.hip', 'sparse.hip',
           'sepconv.hip', 'sepconv_precise.hip', 'conv_igemm256.hip', 'conv_igemm_s64.hip', 'conv3x3c64.hip', 'stem.hip', 'matcher.hip', 'ref32.hip', 'conv16x3.hip', 'conv_igemm_grouped.hip', 'conv16x3p.hip', 'sepconv_x3.hip']
# sources that #include another source: rebuilt when that one changes
INCLUDES = {'conv_igemm_grouped.hip': ['conv_igemm.hip']}


def _hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the HIP engine cannot be built (no CPU fallback exists)')


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(HERE, '..', 'include', 'empanada_hip.h'))
    return max(os.path.getmtime(h) for h in hs)


def build_all(verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr = _newest_header()
    objs, rebuilt = [], False
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        objs.append(o)
        dep = max([os.path.getmtime(s), hdr] + [os.path.getmtime(os.path.join(CSRC, d)) for d in INCLUDES.get(src, [])])
        if force or not os.path.exists(o) or os.path.getmtime(o) < dep:
            cmd = [hipcc] + COMMON + PER_FILE.get(src, []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f'hipcc failed on {src}')
        if verbose and out.strip():
            print(out.decode())
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build_all(verbose=True, force='--force' in sys.argv))
