import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# A-B builds of the library (tools/build_variant.sh -> libsfron_<name>.so beside libsfron.so) can be put under the SAME parity tests:
#   SFRON_TEST_LIB=libsfron_<name>.so python -m pytest tests -m gpu ...
# Test infrastructure only (the package itself reads no environment variable); unset, every test loads the product library.
if os.environ.get("SFRON_TEST_LIB"):
    import sfron  # noqa: F401
    from sfron import _lib as _sfron_lib
    _p = os.path.join(os.path.dirname(_sfron_lib.LIB_PATH), os.environ["SFRON_TEST_LIB"])
    if not os.path.exists(_p):
        raise RuntimeError(f"SFRON_TEST_LIB: {_p} does not exist")
    _sfron_lib.LIB_PATH = _p
    print(f"[conftest] library under test: {_p}", file=sys.stderr)
