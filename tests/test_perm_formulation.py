"""
The parallel restatement of numpy's legacy shuffle that pybgmm_amd/csrc/kernels_perm.hip runs on the device
(``np.random.permutation(range(N))`` of pcrpmm.py:89), here in numpy on the CPU against ``np.random`` itself:

  * the DRAWS: which 32-bit words a step consumes (masked rejection) resolved 64 words at a time by the fixed-point
    iteration  acc <- [(w & mask) <= i - #accepted below]  with a cut behind the step that ends a mask's range;
  * the SWAPS: final[i] from pred / predV links (a stable sort of the steps by target) and pointer jumping;
  * the generator state left behind = the block of the stream the last consumed word lies in;
  * (round 5) the same links from BUCKETS of targets whose boundaries are the quantiles of the targets' law: one sort per
    bucket of ~1 536 pairs instead of one over all of them (what the generations in flight do on the device).
The GPU test (tests/test_gpu_parity.py::test_device_permutation_equals_numpy) checks the kernels against numpy directly.
"""
import numpy as np
import pytest


def draws_by_runs_of_64(words, n, i=None, p=0, J=None):
    """The steps i .. 1 (default: all of them, n - 1 .. 1) from words[p:]; returns (targets, index behind the last word used)."""
    J = np.zeros(n, dtype=np.int64) if J is None else J
    i = n - 1 if i is None else i
    while i >= 1:
        mask = (1 << int(i).bit_length()) - 1
        lowi = (mask >> 1) + 1
        a = (words[p:p + 64] & mask).astype(np.int64)
        acc = a <= i
        for _ in range(70):
            c = np.concatenate([[0], np.cumsum(acc)[:-1]])
            acc2 = a <= i - c
            if np.array_equal(acc2, acc):
                break
            acc = acc2
        c = np.concatenate([[0], np.cumsum(acc)[:-1]])
        s = i - c
        last = np.nonzero(acc & (s == lowi))[0]
        cut = last[0] + 1 if last.size else len(a)
        use = acc[:cut]
        J[s[:cut][use]] = a[:cut][use]
        i -= int(use.sum())
        p += cut
    return J, p


def permutation_from_targets(J, n):
    steps, keys = np.arange(1, n), J[1:]
    order = np.argsort(keys, kind="stable")
    ks, idx = keys[order], steps[order]
    m = n - 1
    pred = np.full(n, -1, dtype=np.int64)
    same_next = np.zeros(m, bool)
    same_next[:-1] = ks[1:] == ks[:-1]
    pred[idx[:-1][same_next[:-1]]] = idx[1:][same_next[:-1]]
    ptr = np.arange(n)
    start = np.ones(m, bool)
    start[1:] = ks[1:] != ks[:-1]
    q = np.nonzero(start)[0]
    v, first = ks[q], idx[q]
    second = np.where(same_next[q], idx[np.minimum(q + 1, m - 1)], -1)
    pv = np.where(first != v, first, second)
    ptr[v[pv >= 0]] = pv[pv >= 0]
    while True:
        nxt = ptr[ptr]
        if np.array_equal(nxt, ptr):
            break
        ptr = nxt
    final = np.empty(n, dtype=np.int64)
    final[0] = ptr[0]
    i, j = np.arange(1, n), J[1:]
    final[1:] = np.where(j == i, ptr[i], np.where(pred[i] >= 0, ptr[np.maximum(pred[i], 0)], j))
    return final


@pytest.mark.parametrize("n", [2, 3, 5, 63, 64, 65, 100, 1000, 4096, 4097, 65536, 65537, 100000])
@pytest.mark.parametrize("seed", [1, 7])
def test_parallel_restatement_of_the_legacy_shuffle(n, seed):
    rs = np.random.RandomState(seed)
    rs.random_sample(seed * 13)                                    # (a position inside a block)
    st = rs.get_state()
    ref = rs.permutation(n)
    after = rs.get_state()
    r2 = np.random.RandomState()
    r2.set_state(st)
    words = r2.randint(0, 2 ** 32, size=2 * n + 1248, dtype=np.uint32)   # (the stream's 32-bit outputs, one per call)
    J, used = draws_by_runs_of_64(words, n)
    assert np.array_equal(permutation_from_targets(J, n), ref)
    r3 = np.random.RandomState()
    r3.set_state(st)
    r3.randint(0, 2 ** 32, size=used, dtype=np.uint32)             # (numpy consumed exactly `used` words)
    assert np.array_equal(r3.get_state()[1], after[1]) and r3.get_state()[2] == after[2]
    assert used <= 2 * n + 64


# ---- round 5: the links by buckets of targets (kernels_perm.hip perm_bucket_bounds / perm_bucket_links_kernel) -----------
def bucket_bounds(n, per=1536):
    """Bucket b holds the targets [bnd[b], bnd[b + 1]); the boundaries are the quantiles of the law of the targets (step i hits
    v <= i with probability 1 / (i + 1), so v expects rho(v) = sum_{i >= max(v, 1)} 1 / (i + 1) hits): equal EXPECTED load."""
    NB = max(1, (n - 1 + per - 1) // per)
    inv = 1.0 / (np.arange(n, dtype=np.float64) + 1.0)
    suffix = np.cumsum(inv[::-1])[::-1]                      # suffix[i] = sum_{j >= i} 1 / (j + 1)
    rho = suffix.copy()
    rho[0] = suffix[1] if n > 1 else 0.0
    cum = np.concatenate([[0.0], np.cumsum(rho)])
    bnd = np.searchsorted(cum[1:], cum[-1] / NB * np.arange(1, NB), side="left")
    return np.concatenate([[0], bnd, [n]]).astype(np.int64)


def links_by_buckets(J, n, bnd):
    """pred / ptr of permutation_from_targets, bucket by bucket: the (target, step) pairs of a bucket sorted as ONE key
    (target << 32 | step -- steps are distinct, so no stability is asked of the sort), links from neighbours in that order."""
    pred = np.full(n, -1, dtype=np.int64)
    ptr = np.arange(n, dtype=np.int64)
    steps = np.arange(1, n, dtype=np.int64)
    which = np.searchsorted(bnd, J[1:], side="right") - 1
    loads = np.bincount(which, minlength=len(bnd) - 1)
    for b in np.unique(which):
        i = steps[which == b]
        key = np.sort((J[i].astype(np.int64) << 32) | i)
        v, st = key >> 32, key & 0xffffffff
        same_next = np.zeros(len(key), bool)
        same_next[:-1] = v[1:] == v[:-1]
        pred[st[:-1][same_next[:-1]]] = st[1:][same_next[:-1]]
        first = np.ones(len(key), bool)
        first[1:] = v[1:] != v[:-1]
        q = np.nonzero(first)[0]
        nxt = np.where(same_next[q], st[np.minimum(q + 1, len(key) - 1)], -1)
        pv = np.where(st[q] != v[q], st[q], nxt)
        ptr[v[q][pv >= 0]] = pv[pv >= 0]
    return pred, ptr, loads


@pytest.mark.parametrize("n", [4096, 4097, 65537, 300000])
def test_links_by_buckets_of_equal_expected_load(n):
    rs = np.random.RandomState(n % 97)
    st = rs.get_state()
    ref = rs.permutation(n)
    r2 = np.random.RandomState()
    r2.set_state(st)
    words = r2.randint(0, 2 ** 32, size=2 * n + 1248, dtype=np.uint32)
    J, _ = draws_by_runs_of_64(words, n)
    bnd = bucket_bounds(n)
    assert bnd[0] == 0 and bnd[-1] == n and np.all(np.diff(bnd) >= 0)
    pred, ptr, loads = links_by_buckets(J, n, bnd)
    # the buckets' loads: (n - 1) / NB on average, far from the 2 048 slots a bucket has on the device
    assert loads.sum() == n - 1 and loads.max() <= 2048 and loads.max() < 1.25 * (n - 1) / len(loads) + 64
    # the same permutation as the sort-based links give (and numpy)
    while True:
        nxt = ptr[ptr]
        if np.array_equal(nxt, ptr):
            break
        ptr = nxt
    final = np.empty(n, dtype=np.int64)
    final[0] = ptr[0]
    i, j = np.arange(1, n), J[1:]
    final[1:] = np.where(j == i, ptr[i], np.where(pred[i] >= 0, ptr[np.maximum(pred[i], 0)], j))
    assert np.array_equal(final, ref)


# ---- the rounds across segments (kernels_perm.hip perm_draw_kernel / perm_draw_chained_kernel) ---------------------------
def segment_run(words, seg0, seg_len, i0, low, J=None):
    """One segment (one wavefront on the device): the steps served by words[seg0 : seg0 + seg_len] when the first of them
    meets step i0; stops behind step `low`.  Returns (accepted, words used if step `low` was served here else -1)."""
    i, p, accepted = i0, 0, 0
    while i >= low and p < seg_len:
        mask = (1 << int(i).bit_length()) - 1
        lowi = (mask >> 1) + 1
        a = (words[seg0 + p:seg0 + min(p + 64, seg_len)] & mask).astype(np.int64)
        acc = a <= i
        for _ in range(70):
            c = np.concatenate([[0], np.cumsum(acc)[:-1]])
            acc2 = a <= i - c
            if np.array_equal(acc2, acc):
                break
            acc = acc2
        c = np.concatenate([[0], np.cumsum(acc)[:-1]])
        s = i - c
        last = np.nonzero(acc & (s == lowi))[0]
        cut = last[0] + 1 if last.size else len(a)
        use = acc[:cut]
        if J is not None:
            J[s[:cut][use]] = a[:cut][use]
        k = int(use.sum())
        i -= k
        accepted += k
        p += cut
    return accepted, (p if i < low else -1)


@pytest.mark.parametrize("n,seg,low", [(20000, 1024, 256), (65537, 1024, 256), (65537, 512, 4096)])
def test_rounds_over_segments_settle_on_the_sequential_draws(n, seg, low):
    """The stream cut into segments, every segment's start taken from the counts of the segments in front of it as the
    previous round left them (here: all at once, i.e. the device's rounds without their in-place reads), starting from a guess
    that is wrong everywhere: the round in which no count changes has every segment at its true start, and the targets written
    from there -- plus the serial tail below `low` -- are the sequential ones (draws_by_runs_of_64)."""
    rs = np.random.RandomState(5)
    words = rs.randint(0, 2 ** 32, size=2 * n + 1248, dtype=np.uint32)
    J_ref, used_ref = draws_by_runs_of_64(words, n)
    T = (len(words) + seg - 1) // seg
    lens = [min(seg, len(words) - t * seg) for t in range(T)]
    cnt = np.full(T, int(0.7 * seg), dtype=np.int64)             # (a guess: 70 % of the words accepted, everywhere)
    for rounds in range(1, 200):
        before = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        new = np.array([segment_run(words, t * seg, lens[t], max(n - 1 - int(before[t]), 0), low)[0] if n - 1 - before[t] >= low else 0
                        for t in range(T)], dtype=np.int64)
        if np.array_equal(new, cnt):
            break
        cnt = new
    assert rounds < 60, rounds
    J = np.zeros(n, dtype=np.int64)
    before = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    tail_from = None
    for t in range(T):
        i0 = n - 1 - int(before[t])
        if i0 >= low:
            acc, end = segment_run(words, t * seg, lens[t], i0, low, J)
            if end >= 0 and before[t] + acc == n - low:
                tail_from = t * seg + end
    assert tail_from is not None
    _, p = draws_by_runs_of_64(words, n, i=low - 1, p=tail_from, J=J)     # (the serial tail: one wavefront behind the rounds)
    assert np.array_equal(J[1:], J_ref[1:])
    assert p == used_ref
