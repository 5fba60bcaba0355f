"""The C oracle's runs of the heavy GPU parity tests, started together when the session starts (test infrastructure).

The oracle (oracle/gibbs_oracle.c: the reference's algorithm, one visit at a time) costs 0.4 ms per visit at C4's shape
and 3.6 ms at C5's on the GPU box's host -- a core's cache bandwidth, 6.5 MB of inverse covariances per visit -- and what a
chain's oracle does never depends on what the device did.  So a test that is decorated

    @with_oracle(case_fn)
    def test_x(N, D, ..., oracle_ref):
        case, ref = oracle_ref

names a function ``case_fn(**its parameters)`` that builds the problem from its seeds -- a dict with ``X, prior (m_0, k_0,
v_0, S_0), alpha, z0, K_max, cov_type, sweeps = [(u, order | None, power | None, n_visits | None), ...]`` -- and when the
collection is complete (tests/conftest.py: pytest_collection_finish) every selected test's case is built and run through
the oracle on a thread of its own (ctypes releases the GIL; the box has 256 cores, and 25 oracles on 25 cores finish when
the slowest of them does).  ``oracle_ref`` hands the test its case and the oracle's state after every sweep:
``ref[it] = {"z", "log_marg", "K", "counts"}``.  A test run on its own (-k ...) gets the same through the same path; with
ORACLE_POOL=0 every case is run when its test asks for it.
"""
import os
import threading

_lock = threading.Lock()
_futures = {}
TIMES = []          # (seconds, case) of the runs that have finished: reported at the end of the session


class _Job(object):
    """One case on a DAEMON thread of its own (a session that stops at its first failure must not wait for oracles it no
    longer wants: the process may leave while they are still inside the C call)."""

    def __init__(self, gate, case_fn, params):
        self.out, self.err = None, None
        self.done = threading.Event()
        self.t = threading.Thread(target=self._run, args=(gate, case_fn, params), daemon=True, name="oracle")
        self.t.start()

    def _run(self, gate, case_fn, params):
        import time
        with gate:
            t0 = time.time()
            try:
                self.out = run_case(case_fn, params)
                TIMES.append((time.time() - t0, "%s%s" % (case_fn.__name__, tuple(params.values()))))
            except BaseException as e:      # noqa: BLE001  (raised again in the test that asks for the result)
                self.err = e
            finally:
                self.done.set()

    def result(self):
        self.done.wait()
        if self.err is not None:
            raise self.err
        return self.out


def with_oracle(case_fn):
    def deco(test_fn):
        test_fn._oracle_case = case_fn
        return test_fn
    return deco


def run_case(case_fn, params):
    """Builds the case and runs every sweep of it through the C oracle."""
    from oracle import c_oracle
    case = case_fn(**params)
    m_0, k_0, v_0, S_0 = case["prior"]
    o = c_oracle.COracle(case["X"], m_0, k_0, v_0, S_0, case.get("alpha", 1.0), case["z0"], case["K_max"],
                         scipy_tables=case.get("scipy_tables", False), cov_type=case.get("cov_type", "full"))
    ref = []
    for sw in case["sweeps"]:
        u, order, power, n_visits = (tuple(sw) + (None,) * 4)[:4]
        o.sweep(u, order, power, n_visits)
        ref.append({"z": o.z, "log_marg": o.log_marg(), "K": o.K, "counts": o.counts})
    return case, ref


def _key(item):
    return item.nodeid


def _params(item):
    fn = item.function._oracle_case
    want = fn.__code__.co_varnames[:fn.__code__.co_argcount]
    have = item.callspec.params if hasattr(item, "callspec") else {}
    return {k: have[k] for k in want}


def cost_of(item):
    """What the item's case says its oracle costs (visits x K x D^2; 0 if it does not say)."""
    fn = item.function._oracle_case
    return getattr(fn, "cost", lambda **kw: 0)(**_params(item))


def prefetch(items):
    """Called once the selection is known: one thread per decorated test (at most half the cores, at most 48)."""
    if os.environ.get("ORACLE_POOL", "1") == "0":
        return
    todo = [it for it in items if getattr(getattr(it, "function", None), "_oracle_case", None) is not None]
    if not todo:
        return
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    workers = max(1, min(48, cores // 2, len(todo)))
    # the longest first (a case may say what it costs: cost = visits x K x D^2), so that with fewer workers than cases the
    # long ones are not what the session ends on
    todo.sort(key=lambda it: -cost_of(it))
    gate = threading.BoundedSemaphore(workers)
    with _lock:
        for it in todo:
            _futures[_key(it)] = _Job(gate, it.function._oracle_case, _params(it))


def result_for(item):
    with _lock:
        fut = _futures.pop(_key(item), None)
    if fut is not None:
        return fut.result()
    return run_case(item.function._oracle_case, _params(item))
