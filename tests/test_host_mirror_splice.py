"""The in-place splice of the library's host mirrors in a structural hspf_graph_patch (holo_amd/csrc/spf_capi.hip: runs of
unchanged rows between two replaced ones move by the length changes in front of them — the runs that move towards the
front first, front to back, then those that move towards the back, back to front — then the new rows go into the gaps and
the row bounds are shifted), restated on numpy arrays and compared with a fresh splice (engine.splice_rows) on random
multi-row patches: the ORDER of the moves is what this pins (a run's new place must never reach into a run that has not
moved yet).  The C++ itself is compared with the device arrays on the GPU (tests/test_gpu_graph_build.py, host_row_ptr /
host_col)."""
import numpy as np
import pytest

from holo_amd import engine as E


def splice_in_place(row_ptr, col, vertices, new_rows):
    n = len(row_ptr) - 1
    m = len(vertices)
    e_old = int(row_ptr[n])
    sh = np.zeros(m + 1, np.int64)
    for j, v in enumerate(vertices):
        sh[j + 1] = sh[j] + len(new_rows[j]) - (int(row_ptr[v + 1]) - int(row_ptr[v]))
    e_new = e_old + int(sh[m])
    buf = np.full(max(e_old, e_new), 0xDEAD, np.int64)
    buf[:e_old] = col
    orp = row_ptr.astype(np.int64)
    run_begin = lambda i: 0 if i == 0 else int(orp[vertices[i - 1] + 1])     # noqa: E731
    run_end = lambda i: e_old if i == m else int(orp[vertices[i]])            # noqa: E731

    def move(i):
        a, b = run_begin(i), run_end(i)
        if b > a:
            buf[a + sh[i]:b + sh[i]] = buf[a:b].copy()                          # memmove
            # what the run left behind is dead: poison it, so that a later move that still needed it shows
            lo, hi = (b + sh[i], b) if sh[i] < 0 else (a, a + sh[i])
            buf[max(lo, a if sh[i] > 0 else b + sh[i]):hi] = 0xDEAD
    for i in range(m + 1):
        if sh[i] < 0:
            move(i)
    for i in range(m, -1, -1):
        if sh[i] > 0:
            move(i)
    for j, v in enumerate(vertices):
        a = int(orp[v]) + sh[j]
        buf[a:a + len(new_rows[j])] = new_rows[j]
    nrp = orp.copy()
    j = 0
    for v in range(vertices[0] + 1, n + 1):
        while j < m and vertices[j] < v:
            j += 1
        nrp[v] += sh[j]
    return nrp, buf[:e_new]


@pytest.mark.parametrize("seed", range(40))
def test_in_place_splice_equals_a_fresh_one(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 40))
    lens = rng.integers(0, 7, n)
    row_ptr = np.zeros(n + 1, np.uint32); row_ptr[1:] = np.cumsum(lens)
    col = rng.integers(0, n, int(row_ptr[-1])).astype(np.uint32)
    met = np.ones(len(col), np.uint32); vf = np.zeros(n, np.uint8)
    for _ in range(6):
        m = int(rng.integers(1, min(n, 8) + 1))
        vs = np.sort(rng.choice(n, size=m, replace=False))
        rows = [rng.integers(0, n, int(rng.integers(0, 12))).astype(np.uint32) for _ in range(m)]
        want_rp, want_col, _, _ = E.splice_rows(row_ptr, col, met, vf, vs, rows, [np.ones(len(r), np.uint32) for r in rows], vf[vs])
        got_rp, got_col = splice_in_place(row_ptr, col, vs.tolist(), rows)
        assert np.array_equal(got_rp, want_rp.astype(np.int64))
        assert np.array_equal(got_col, want_col.astype(np.int64))
        row_ptr, col = want_rp, want_col
        met = np.ones(len(col), np.uint32)
