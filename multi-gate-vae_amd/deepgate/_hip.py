"""ctypes binding of libmgvae_hip.so (the C ABI declared in include/mgvae_hip.h).

There is deliberately NO fallback: if the library is missing or a launcher reports an error the
caller gets an exception.  Argument types are read from the header itself, so the Python side
cannot drift from the declared ABI.
"""
import ctypes  # noqa: F401 (re-exported for callers that build host arrays)
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(_HERE)                      # multi-gate-vae_amd/
_REPO_ROOT = os.path.dirname(_PKG_ROOT)
LIB_PATH = os.environ.get('MGV_LIB', os.path.join(_PKG_ROOT, 'csrc', 'libmgvae_hip.so'))   # MGV_LIB: A/B-test builds
HEADER_PATH = os.path.join(_REPO_ROOT, 'include', 'mgvae_hip.h')

_CTYPE = (('int64_t', ctypes.c_int64), ('uint64_t', ctypes.c_uint64), ('int32_t', ctypes.c_int32),
          ('float', ctypes.c_float), ('double', ctypes.c_double), ('int', ctypes.c_int))


class HipLibraryError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """{function name: [ctypes argument types]} for every `int mgv_*(...)` declaration."""
    with open(path) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'//[^\n]*', '', text)
    out = {}
    for m in re.finditer(r'\bint\s+(mgv_\w+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        types = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    types.append(ctypes.c_void_p)
                    continue
                for key, ct in _CTYPE:
                    if re.search(r'\b%s\b' % key, a):
                        types.append(ct)
                        break
                else:
                    raise HipLibraryError('cannot map C argument %r of %s' % (a, name))
        out[name] = types
    return out


_lib = None
_sigs = None


def load():
    """Load the shared library once; raise if it has not been built (python __graft_entry__.py)."""
    global _lib, _sigs
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError('%s not found: build it with `make -C %s` (or __graft_entry__.build()); '
                              'there is no CPU fallback for the DG_AE hot path' % (LIB_PATH, os.path.dirname(LIB_PATH)))
    lib = ctypes.CDLL(LIB_PATH)
    for marker in ('mgv_diag_ablation_build', 'mgv_diag_ablation_build_fwd'):
        # timing-ablation builds of the struct-stage kernels (-DMGV_ABL / -DMGV_ABLF, tools/run_abl*.sh) give wrong results by design
        if hasattr(lib, marker) and os.environ.get('MGV_ALLOW_ABLATION', '0') != '1':
            raise HipLibraryError('%s is a timing-ablation build (%s): refused outside tools/ (set MGV_ALLOW_ABLATION=1 for a timing run)'
                                  % (LIB_PATH, marker))
    sigs = parse_header()
    for name, types in sigs.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipLibraryError('%s is declared in %s but not exported by %s' % (name, HEADER_PATH, LIB_PATH))
        fn.argtypes = types
        fn.restype = ctypes.c_int
    _lib, _sigs = lib, sigs
    return lib


class _TensorPtr(ctypes.c_void_p):
    """A device pointer that keeps its tensor alive for as long as the pointer object lives — i.e. through the launcher call it is an
    argument of.  `ptr(x.to(torch.int32))` would otherwise free the temporary the moment ptr() returns, and a second temporary in the
    same argument list could be handed the same memory before the launch (it happened: the colour check of round 4)."""
    _keep = None


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    p = _TensorPtr(t.data_ptr())
    p._keep = t
    return p


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """The current torch stream of the current device as a hipStream_t (raw handle: `torch.cuda.current_stream()` builds a Python
    object and resolves the device index through three layers, ~9 us per launch)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_ERR = {-1: 'MGV_EINVAL (bad argument)', -2: 'MGV_EUNSUPPORTED (unsupported size)'}


_profile = None      # {launcher name: [(start event, end event), ...]} while profiling is on


def profile(enable):
    """Bracket every launcher call with a pair of HIP events on the stream it is enqueued on
    (bench.py reads per-launcher device time from them).  Returns the previous table."""
    global _profile
    old = _profile
    _profile = {} if enable else None
    return old


def profile_summary(table):
    """{name: (calls, total ms)}; synchronises."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(s.elapsed_time(e) for s, e in v)) for k, v in (table or {}).items()}


def profile_times(table, name):
    """Per-call device times (ms) of one launcher; synchronises."""
    torch.cuda.synchronize()
    return [s.elapsed_time(e) for s, e in (table or {}).get(name, [])]


def call(name, *args):
    """Call launcher `name`; the current torch stream is appended as the last argument."""
    lib = load()
    if _profile is not None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(lib, name)(*args, stream())
        e.record()
        _profile.setdefault(name, []).append((s, e))
    else:
        rc = getattr(lib, name)(*args, stream())
    if rc != 0:
        raise HipLibraryError('%s failed: %s' % (name, _ERR.get(rc, 'hipError_t %d' % rc)))


def call_value(name, *args):
    """Call a query entry point (no stream argument, the return value is the answer, not a status)."""
    return int(getattr(load(), name)(*args))


def check(t, dtype=torch.float32, name='tensor'):
    """Launchers take raw pointers: insist on what they assume."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError('%s must live on the GPU (got %s); the hot path has no CPU implementation' % (name, t.device))
    if t.dtype != dtype:
        raise HipLibraryError('%s must be %s (got %s)' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise HipLibraryError('%s must be contiguous' % name)
    return t
