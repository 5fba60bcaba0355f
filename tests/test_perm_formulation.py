"""
The parallel restatement of numpy's legacy shuffle that pybgmm_amd/csrc/kernels_perm.hip runs on the device
(``np.random.permutation(range(N))`` of pcrpmm.py:89), here in numpy on the CPU against ``np.random`` itself:

  * the DRAWS: which 32-bit words a step consumes (masked rejection) resolved 64 words at a time by the fixed-point
    iteration  acc <- [(w & mask) <= i - #accepted below]  with a cut behind the step that ends a mask's range;
  * the SWAPS: final[i] from pred / predV links (a stable sort of the steps by target) and pointer jumping;
  * the generator state left behind = the block of the stream the last consumed word lies in.
The GPU test (tests/test_gpu_parity.py::test_device_permutation_equals_numpy) checks the kernels against numpy directly.
"""
import numpy as np
import pytest


def draws_by_runs_of_64(words, n):
    J = np.zeros(n, dtype=np.int64)
    i, p = n - 1, 0
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
