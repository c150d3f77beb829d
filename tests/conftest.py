import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _have_gpu():
    try:
        import ctypes
        lib = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return lib.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


HAVE_GPU = _have_gpu()


def _gpu_expected(config):
    """-m gpu (the driver's GPU tier) or RTPBR_EXPECT_GPU=1: a missing device is an ERROR, not a skip"""
    m = (config.getoption("-m") or "").replace(" ", "")
    return os.environ.get("RTPBR_EXPECT_GPU") == "1" or (m.startswith("gpu") and "notgpu" not in m)


_skipped_gpu = []


def pytest_collection_modifyitems(config, items):
    if HAVE_GPU:
        return
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if gpu_items and _gpu_expected(config):
        raise pytest.UsageError("GPU tests were requested (-m gpu / RTPBR_EXPECT_GPU=1) but no HIP device is visible: "
                                "the HIP-vs-oracle parity tests cannot run here")
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in gpu_items:
        it.add_marker(skip)
        _skipped_gpu.append(it.nodeid)


def pytest_terminal_summary(terminalreporter):
    if _skipped_gpu:
        terminalreporter.write_sep("=", "HIP path NOT exercised")
        terminalreporter.write_line(
            "%d gpu-marked tests (every HIP-vs-oracle parity test) were skipped: no HIP device here. "
            "They run with `pytest -m gpu` on an MI355X (gpurun); this run only checked the oracle, the host logic "
            "and the library's symbols." % len(_skipped_gpu))


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle_backend import oracle_api
    return oracle_api().lib
