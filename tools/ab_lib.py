"""tools only: SFRON_LIB_NAME=libsfron_<variant>.so points the ctypes binding at another build of the library that sits beside
libsfron.so (tools/build_variant.sh) BEFORE the first lib() call.  The product package reads no environment variable."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def select():
    import sfron  # noqa: F401
    from sfron import _lib
    name = os.environ.get("SFRON_LIB_NAME")
    if name:
        path = os.path.join(os.path.dirname(_lib.LIB_PATH), name)
        if not os.path.exists(path):
            raise SystemExit(f"SFRON_LIB_NAME: {path} does not exist")
        _lib.LIB_PATH = path
        print(f"[ab] library: {name}", file=sys.stderr, flush=True)
    return _lib
