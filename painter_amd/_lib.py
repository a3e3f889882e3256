"""ctypes binding of libpainter_hip.so.  Prototypes are parsed from include/painter_hip.h so the Python
side cannot drift from the C ABI.  There is NO fallback: if the library is missing the import of any
compute entry point raises (the product path never routes through PyTorch ops or the oracle)."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "painter_hip.h")
LIB_PATH = os.environ.get("PAINTER_AMD_LIB") or os.path.join(HERE, "lib", "libpainter_hip.so")      # PAINTER_AMD_LIB: A/B builds (diagnostics)

PA_F32, PA_BF16 = 0, 1
EPI_BIAS, EPI_BIAS_F32, EPI_BIAS_GELU, EPI_BIAS_RESID = 0, 1, 2, 3

_CT = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "hipStream_t": ctypes.c_void_p}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int64_t|int)\s+(pa_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    t = a.replace("const", " ").split()[0]
                    argtypes.append(_CT[t])
        protos[name] = (_CT[ret], argtypes)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "painter_amd: %s is missing -- run `python -m painter_amd.build` (or __graft_entry__.build()). "
                    "There is no CPU / PyTorch fallback for the hot path." % LIB_PATH)
            # torch bundles its own libamdhip64: it has to be in the process before ours resolves the HIP runtime, otherwise
            # two runtimes end up loaded and launches on torch's streams fail with hipErrorNoDevice.
            import torch  # noqa: F401
            dll = ctypes.CDLL(LIB_PATH)
            for name, (ret, args) in self._protos.items():
                fn = getattr(dll, name)          # AttributeError here = header/library drift: fail loudly
                fn.restype = ret
                fn.argtypes = args
            want = int(re.search(r"#define\s+PA_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
            got = dll.pa_abi_version()
            if got != want:          # e.g. a stale A/B build behind PAINTER_AMD_LIB: its argument lists would be shifted
                raise RuntimeError("painter_amd: %s has C ABI version %d, include/painter_hip.h declares %d -- rebuild it "
                                   "(python -m painter_amd.build --force)" % (LIB_PATH, got, want))
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.load(), name)


lib = _Lib()


def check(err, what=""):
    if err != 0:
        raise RuntimeError("painter_hip: %s failed with hipError %d" % (what, err))
