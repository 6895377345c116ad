"""In-tree build of libiiseg_hip.so (hipcc, gfx950 only).

`python -m iterative_inference_segm_amd.build` or `__graft_entry__.build()`.  hipcc
cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB = os.path.join(HERE, 'libiiseg_hip.so')
SOURCES = ['abi.hip', 'conv_igemm.hip', 'conv_taps.hip', 'conv_wino.hip', 'conv_wino_bf16.hip', 'conv_halo.hip', 'conv_small.hip', 'conv_halo_bf16.hip', 'conv_c8_bf16.hip', 'conv_c8_m16.hip', 'conv1x1_c8.hip', 'conv_f64.hip', 'conv_halo_f64.hip', 'conv_wino_f64.hip', 'pool_unpool.hip', 'deconv.hip', 'deconv_phase.hip', 'tail.hip', 'metrics.hip', 'bn.hip']
ARCH = 'gfx950'


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _digest(paths, extra=()):
    """Content key of a build product: sha256 over the bytes of its inputs (NOT their mtimes: a
    checkout or a copy to the GPU box does not preserve those), the flags and the compiler version."""
    h = hashlib.sha256()
    for e in extra:
        h.update(str(e).encode() + b'\0')
    for q in paths:
        with open(q, 'rb') as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


def _file_sha(path):
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        for blk in iter(lambda: f.read(1 << 20), b''):
            h.update(blk)
    return h.hexdigest()


def _stale(target, key):
    """True unless `target` exists, the input key stored next to it equals `key`, AND the stored hash
    of the product's own bytes matches the file (a truncated / corrupted object with an intact key
    file is rebuilt, not reused)."""
    try:
        with open(target + '.key') as f:
            words = f.read().split()
        return not os.path.exists(target) or len(words) < 2 or words[0] != key or words[1] != _file_sha(target)
    except OSError:
        return True


def _stamp(target, key):
    with open(target + '.key', 'w') as f:
        f.write(key + ' ' + _file_sha(target) + '\n')


def library_sha256():
    """sha256 of libiiseg_hip.so as it is on disk (printed by __graft_entry__.smoke: which binary was
    validated)."""
    return _file_sha(LIB)


_HIPCC_VERSION = {}


def _hipcc_version(hipcc):
    if hipcc not in _HIPCC_VERSION:
        r = subprocess.run([hipcc, '--version'], capture_output=True, text=True)
        _HIPCC_VERSION[hipcc] = r.stdout.strip()
    return _HIPCC_VERSION[hipcc]


# Per-source flags.  The bf16 C8 kernels (and the fp32 halo / Winograd kernels) are compiled with relaxed NaN handling: their ReLU / max-pool
# epilogues are v_max_f32 straight on MFMA results, and under strict NaN rules hipcc puts a quieting pass
# (v_max_f32 v, v, v) in front of every one of them and cannot fold the lane exchange of the pool into the
# max (conv1_1 of the DAE: 725 -> 589 vector instructions per wave).  Results for non-NaN data are the same
# bit for bit; a NaN in the activations gives an unspecified (never trapping) value instead of a quiet NaN.
# NOT for the float64 sources: conv_wino_f64.hip uses NaN as its never-equal marker.
EXTRA_FLAGS = {
    'conv_c8_bf16.hip': ['-fno-honor-nans'],
    'conv_c8_m16.hip': ['-fno-honor-nans'],
    'conv1x1_c8.hip': ['-fno-honor-nans'],
    'conv_halo.hip': ['-fno-honor-nans'],
    'conv_wino.hip': ['-fno-honor-nans'],
    'conv_halo_bf16.hip': ['-fno-honor-nans'],
    'conv_wino_bf16.hip': ['-fno-honor-nans'],
}

LAST = {}      # what the latest build() did: {'compiled': [...], 'reused': [...], 'linked': bool}


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libiiseg_hip.so.  Returns the path.  An object
    is reused (unless `force`) only when the content key stored next to it -- source + headers +
    flags + hipcc version -- matches; the library only when its key over the objects' keys does.
    `LAST` says what happened."""
    hipcc = _hipcc()
    LAST.clear()
    LAST.update(compiled=[], reused=[], linked=False)
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'conv_common.h'), os.path.join(CSRC, 'c8_common.h'), os.path.join(CSRC, 'conv_f64_common.h'), os.path.join(CSRC, 'column_io.h'), os.path.join(CSRC, 'tail_math.h'),
               os.path.join(INCLUDE, 'iiseg.h')]
    flags = ['-O3', '--offload-arch=' + ARCH, '-fPIC', '-std=c++17', '-I' + INCLUDE, '-I' + CSRC,
             '-Wall', '-Wno-unused-function']

    version = _hipcc_version(hipcc)
    keys = {}
    root = os.path.dirname(HERE)
    key_flags = [f.replace(root, '.') for f in flags]

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        # (the key names the include directories relative to the tree: the same tree under another
        # root -- the GPU box's scratch copy -- reuses its objects)
        extra = EXTRA_FLAGS.get(src, [])
        key = keys[src] = _digest([s] + headers, extra=key_flags + extra + [version])
        if force or _stale(o, key):
            cmd = [hipcc] + flags + extra + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr))
            _stamp(o, key)
            LAST['compiled'].append(src)
        else:
            LAST['reused'].append(src)
        return o

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    lib_key = hashlib.sha256('\n'.join(keys[s] for s in SOURCES).encode()).hexdigest()
    if force or LAST['compiled'] or _stale(LIB, lib_key):
        cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr)
        _stamp(LIB, lib_key)
        LAST['linked'] = True
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
