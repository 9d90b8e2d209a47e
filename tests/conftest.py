import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False,
                     help="also run the GPU tests marked `slow` (second variants of the two-process tests: ~2 min; the default -m gpu run must stay well "
                          "inside the driver's 20-minute limit)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    from rows import ALL_ROWS
    for r in ALL_ROWS:
        config.addinivalue_line("markers", f"row_{r}: covers SURVEY.md section-8 row {r} (tests/rows.py)")
    config.addinivalue_line("markers", "devlib: the test forces a code path through a dvlp_dev_* switch and therefore runs on libdemovlp_hip_dev.so (set automatically)")
    config.addinivalue_line("markers", "slow: long second variant of a two-process GPU test; skipped unless --runslow")


def _code_mentions_dev_switch(fn, depth=2, seen=None):
    """Does ``fn`` (or a module-level helper it calls, ``depth`` levels down) name a developer switch (`dvlp_dev_*`)?"""
    import inspect
    seen = set() if seen is None else seen
    fn = inspect.unwrap(fn)
    if id(fn) in seen or not hasattr(fn, "__code__"):
        return False
    seen.add(id(fn))
    try:
        if "dvlp_dev_" in inspect.getsource(fn) or "use_dev_library" in inspect.getsource(fn):
            return True
    except (OSError, TypeError):
        return False
    if depth > 0:
        g = getattr(fn, "__globals__", {})
        for name in fn.__code__.co_names:
            h = g.get(name)
            if inspect.isfunction(h) and h.__module__ == fn.__module__ and _code_mentions_dev_switch(h, depth - 1, seen):
                return True
    return False


def _item_uses_dev_switches(item):
    fns = [item.function] if hasattr(item, "function") else []
    fm = item.session._fixturemanager
    for name in getattr(item, "fixturenames", ()):
        if name == "_library_for_this_test":
            continue
        try:
            defs = fm.getfixturedefs(name, item)
        except TypeError:
            defs = fm.getfixturedefs(name, item.nodeid)
        fns += [d.func for d in defs or ()]
    return any(_code_mentions_dev_switch(f) for f in fns)


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    # tests that force a code path through a developer switch run on libdemovlp_hip_dev.so (the product library exports none);
    # everything else -- the parity tests proper -- stays on libdemovlp_hip.so
    from rows import rows_of
    for item in items:
        if _item_uses_dev_switches(item):
            item.add_marker(pytest.mark.devlib)
        for r in rows_of(getattr(item, "originalname", None) or item.name.split("[")[0]):
            item.add_marker(getattr(pytest.mark, "row_" + r))
    if config.getoption("--runslow"):
        return
    skip = pytest.mark.skip(reason="slow variant: run with --runslow")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _library_for_this_test(request):
    """Switch to the -DDVLP_DEV build for tests marked `devlib` (automatically: any test or fixture whose code names a `dvlp_dev_*`
    switch), and back to the product library afterwards.  Instantiated before every other fixture of the test."""
    if request.node.get_closest_marker("devlib") is None:
        yield
        return
    from demovlp_amd import _lib
    _lib.use_dev_library(True)
    try:
        yield
    finally:
        _lib.use_dev_library(False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
