"""
Build recipe of liblsqfitgp_hip.so: every HIP translation unit under
``lsqfitgp_amd/csrc`` is compiled for gfx950 with hipcc and linked into one
shared library that lives next to this file (in-tree, so that it travels to the
GPU box with the snapshot).  No PyTorch, no cmake: plain ``hipcc`` commands.
"""

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBNAME = 'liblsqfitgp_hip.so'
LIBPATH = os.path.join(HERE, LIBNAME)

# (source, extra flags)
UNITS = [
    ('api.hip', []),
    ('gemm.hip', []),
    # accumulators that alternate between MFMA and vector ops stay in VGPRs (no AGPR copies)
    ('potrf.hip', ['-mllvm', '-amdgpu-mfma-vgpr-form=1']),
    ('potf2m.hip', ['-mllvm', '-amdgpu-mfma-vgpr-form=1']),
    # the chain kernel (task graph of resident workers)
    ('chaint.hip', ['-mllvm', '-amdgpu-mfma-vgpr-form=1']),
    ('elementwise.hip', []),
    ('dist.hip', []),
    # the Gram kernels follow the reference's unfused square-then-add evaluation
    ('gram.hip', ['-ffp-contract=off']),
    ('gram_jet.hip', ['-ffp-contract=off']),
]

COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-Wno-pass-failed']


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: the HIP toolchain (ROCm) is required to build ' + LIBNAME)
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, variant=None, defines=()):
    """ compile and link; returns the library path.  ``variant`` / ``defines``: a second build of the same sources with other
    compile-time switches (objects under csrc/build_<variant>, library liblsqfitgp_hip_<variant>.so), for same-box A/B runs """
    hipcc = _hipcc()
    libpath = LIBPATH if variant is None else os.path.join(HERE, f'liblsqfitgp_hip_{variant}.so')
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith((".h", ".inc"))] + [os.path.join(HERE, "..", "include", "lsqfitgp_hip.h")]
    objdir = os.path.join(CSRC, 'build' if variant is None else f'build_{variant}')
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src, extra in UNITS:
        srcpath = os.path.join(CSRC, src)
        obj = os.path.join(objdir, os.path.splitext(src)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, [srcpath] + headers):
            cmd = [hipcc] + COMMON + extra + [f'-D{d}' for d in defines] + ['-c', srcpath, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, proc in procs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            raise RuntimeError('compilation failed: ' + ' '.join(cmd) + '\n' + out)
        if verbose and out.strip():
            print(out)
    if force or _stale(libpath, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', libpath] + objs + ['-ldl']
        if verbose:
            print(' '.join(cmd))
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError('link failed: ' + ' '.join(cmd) + '\n' + res.stdout)
    return libpath


if __name__ == '__main__':
    import sys
    # python -m lsqfitgp_amd._build [variant DEFINE[=VALUE] ...]
    if len(sys.argv) > 1:
        print(build_library(verbose=True, variant=sys.argv[1], defines=sys.argv[2:]))
    else:
        print(build_library(verbose=True))
