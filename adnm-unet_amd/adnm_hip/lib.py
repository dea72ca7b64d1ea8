"""ctypes binding of libadnm_hip.so.  The prototypes are parsed from include/adnm_hip.h, so the
header is the single source of truth for the ABI.  There is NO fallback: if the library is
missing or a symbol is absent, importing a kernel raises."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libadnm_hip.so")
HEADER = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "adnm_hip.h")

F32, BF16 = 0, 1
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2

_CT = {
    "const void*": ctypes.c_void_p, "void*": ctypes.c_void_p, "const float*": ctypes.c_void_p, "float*": ctypes.c_void_p,
    "int64_t": ctypes.c_int64, "int": ctypes.c_int, "float": ctypes.c_float, "adnm_stream_t": ctypes.c_void_p,
    "const char*": ctypes.c_char_p, "char*": ctypes.c_char_p, "void": None,
    "float* const*": ctypes.c_void_p, "const float* const*": ctypes.c_void_p, "const int64_t*": ctypes.c_void_p, "const int*": ctypes.c_void_p,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|void\*|int64_t|int|void)\s+(adnm_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                t = a.rsplit(" ", 1)[0] if not a.endswith("*") else a
                t = t.replace(" *", "*")
                if t not in _CT:
                    raise ValueError(f"{name}: unknown C type '{t}' in '{a}'")
                types.append(t)
        protos[name] = (ret, types)
    return protos


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python adnm-unet_amd/build.py` (or __graft_entry__.build()). "
            "There is no PyTorch/CPU fallback for the HIP kernels.")
    # torch ships its own libamdhip64; import it FIRST so libadnm_hip.so binds to the same HIP runtime (and
    # therefore the same device context and streams) instead of pulling a second copy from /opt/rocm/lib.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, types) in parse_header().items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = _CT[ret]
        fn.argtypes = [_CT[t] for t in types]
    _lib = lib
    return lib


def last_error():
    return load().adnm_last_error().decode()


# bench.py sets this to a dict {entry_point: [(start_event, end_event), ...]} to time entry points with
# HIP events on the stream they are launched on (torch's current stream); None = no instrumentation.
EVENTS = None


def call(name, *args):
    """Invoke an int-returning entry point; raises RuntimeError (as the reference's torch ops
    do on a shape mismatch) when the library rejects the call."""
    ev = EVENTS
    if ev is not None and name in ev:
        import torch
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = getattr(load(), name)(*args)
        b.record()
        ev[name].append((a, b))
    else:
        rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {last_error()}")


def ptr_table(tensors):
    """Host array of device pointers (None -> NULL) for the `float* const*` parameters."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def i64_table(values):
    """Host array of int64 for the `const int64_t*` parameters."""
    return (ctypes.c_int64 * len(values))(*[int(v) for v in values])


def query(name, *args):
    return getattr(load(), name)(*args)
