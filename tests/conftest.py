import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def pytest_collection_modifyitems(config, items):
    # tests that wait for an oracle run of their own (oracle_pool.with_oracle) go last, the cheapest oracle first: everything
    # else runs while the oracles are at work, and nobody waits for the 7-minute one but the test that needs it
    import oracle_pool
    mine, rest = [], []
    for it in items:
        (mine if getattr(getattr(it, "function", None), "_oracle_case", None) is not None else rest).append(it)
    mine.sort(key=oracle_pool.cost_of)
    items[:] = rest + mine


def pytest_collection_finish(session):
    # the heavy parity tests' oracle runs set off together, each on a thread of its own (tests/oracle_pool.py)
    import oracle_pool
    oracle_pool.prefetch(session.items)


@pytest.fixture
def oracle_ref(request):
    """(case, ref) of a test decorated with oracle_pool.with_oracle: the problem its case function builds and the C oracle's
    state after every sweep of it."""
    import oracle_pool
    return oracle_pool.result_for(request.node)


def pytest_terminal_summary(terminalreporter):
    import oracle_pool
    if oracle_pool.TIMES:
        terminalreporter.write_line("oracle runs side by side (tests/oracle_pool.py), seconds of one host core each:")
        for sec, what in sorted(oracle_pool.TIMES, reverse=True):
            terminalreporter.write_line("  %7.1f s  %s" % (sec, what))
