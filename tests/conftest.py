import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_unavailable_reason():
    try:
        import torch
        if not torch.cuda.is_available():
            return "no GPU visible (torch.cuda.is_available() is False)"
        from gen6d_amd import lib
        lib.load()
    except Exception as e:          # missing / stale libgen6d_hip.so
        return f"HIP library not loadable: {e}"
    return None


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU (or without the built library) skips the gpu-marked tests instead of
    failing them; `-m gpu` on the GPU box runs them (there the library must load: a missing .so fails test_abi)."""
    if not any("gpu" in item.keywords for item in items):
        return
    reason = _gpu_unavailable_reason()
    if reason is None:
        return
    skip = pytest.mark.skip(reason=reason)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _apply_session_switches():
    """A/B variants of the product path for whole test runs (tests/test_networks_gpu.py::test_alternative_paths_keep_parity starts
    pytest with G6D_TEST_SWITCHES set): comma list of `knob:<name>=<value>` (library launch-policy knobs, include/gen6d_hip.h),
    `attr:<module>.<NAME>=<0|1>` (launch-structure attributes of the package) and `library_trunk` (tools/library_trunk.py: the MIOpen
    trunk).  The switches live HERE, in the test infrastructure: neither the package nor the library reads the environment."""
    spec = os.environ.get("G6D_TEST_SWITCHES", "")
    for item in filter(None, (t.strip() for t in spec.split(","))):
        if item == "library_trunk":
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import library_trunk
            library_trunk.install()
        elif item.startswith("knob:"):
            from gen6d_amd import lib
            name, val = item[5:].split("=")
            lib.set_knob(name, float(val))
        elif item.startswith("attr:"):
            import importlib
            path, val = item[5:].split("=")
            mod, attr = path.rsplit(".", 1)
            setattr(importlib.import_module(mod), attr, bool(int(val)))
        else:
            raise ValueError(f"G6D_TEST_SWITCHES: cannot parse {item!r}")


def pytest_sessionstart(session):
    """Achieved parity errors of the GPU tests are appended to gpurun_out/parity_r06.jsonl (tools/parity_table.py turns
    them into profiles/r05_parity.md)."""
    os.environ.setdefault("G6D_PARITY_LOG", os.path.join(ROOT, "gpurun_out", "parity_r06.jsonl"))
    _apply_session_switches()


@pytest.fixture(autouse=True)
def _knobs_back_to_defaults(request):
    """A GPU test that sets launch-policy knobs (directly through lib.set_knob, or by failing inside the `knob` fixture's scope) must
    not leave a non-default launch policy behind for the tests that follow in the same process."""
    yield
    if "gpu" in request.keywords:
        try:
            from gen6d_amd import lib
            lib.reset_knobs()
            _apply_session_switches()
        except Exception:
            pass


@pytest.fixture
def knob():
    """knob(name, value): set a launch-policy knob of the library for the duration of one test (product defaults are restored)."""
    from gen6d_amd import lib
    yield lib.set_knob
    lib.reset_knobs()
    _apply_session_switches()



@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load


def assert_pinned(g, inputs, weights, what=""):
    """The synthetic inputs / weights regenerated on THIS host are bit-identical to the ones the reference ran on when the
    fixture was made (make_golden*.py store their SHA-256): with that, `HIP vs golden` compares like with like and can be held
    to the same bar as `HIP vs oracle` (round 2's rotated copies went through an sgemm + grid_sample and
    differed between the build container and the GPU box)."""
    from gen6d_amd import synth
    got_i, got_w = synth.fingerprint(inputs), synth.fingerprint(weights)
    assert got_i == str(g["sha_inputs"]), f"{what}: synthetic inputs differ from the ones the golden fixture was generated from"
    assert got_w == str(g["sha_weights"]), f"{what}: synthetic weights differ from the ones the golden fixture was generated from"
