"""ORACLE (test infrastructure only -- never imported by egtr_amd/): restatement of the assignment solver behind the
reference's Hungarian matcher.

The reference calls ``scipy.optimize.linear_sum_assignment`` (model/deformable_detr.py:2985-2992; scipy is unpinned in
its requirements.txt, 1.15.3 in this image).  scipy's solver lives in a third-party C++ file that is absent from
/root/reference (scipy/optimize/rectangular_lsap/rectangular_lsap.cpp), so its published algorithm is restated here:
the shortest-augmenting-path method of D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE
Trans. Aerospace and Electronic Systems 52(4), 2016, as scipy implements it --
  * a tall matrix (more rows than columns) is transposed first, and the result is reported sorted by row index;
  * float64 throughout; reduced cost  r = minVal + cost[i, j] - u[i] - v[j]  in exactly this operation order;
  * the list of remaining columns starts in REVERSE order (n-1 .. 0) and a chosen column is replaced by the last
    remaining one;
  * among columns of equal shortest-path cost the scan keeps the first one it meets unless a later one is unassigned
    (``spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1)``).
PINNED: tests/test_oracle_golden.py::test_lsa_restatement_equals_scipy compares this function with scipy itself, index
for index, on thousands of matrices including integer matrices full of ties, constant matrices, wide / tall / square /
empty shapes and the matcher's own cost matrices from the reference fixture.  The HIP kernel
(egtr_amd/csrc/matcher.hip) follows the same steps and is tested against scipy the same way on the GPU.
"""
import math

import numpy as np


def linear_sum_assignment(cost):
    """Returns (row_ind, col_ind) int64 arrays exactly like scipy.optimize.linear_sum_assignment(cost) (minimise)."""
    cost = np.asarray(cost, dtype=np.float64)
    if cost.ndim != 2:
        raise ValueError("expected a matrix (2-D array)")
    nr, nc = cost.shape
    if nr == 0 or nc == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    transpose = nc < nr
    if transpose:
        cost = np.ascontiguousarray(cost.T)
        nr, nc = nc, nr
    if np.isnan(cost).any() or np.isneginf(cost).any():
        raise ValueError("matrix contains invalid numeric entries")
    u = [0.0] * nr
    v = [0.0] * nc
    spc = [0.0] * nc
    path = [-1] * nc
    col4row = [-1] * nr
    row4col = [-1] * nc
    c = cost.tolist()
    for cur_row in range(nr):
        # ---- augmenting_path
        min_val = 0.0
        remaining = [nc - it - 1 for it in range(nc)]
        num_remaining = nc
        SR = [False] * nr
        SC = [False] * nc
        for j in range(nc):
            spc[j] = math.inf
        sink = -1
        i = cur_row
        while sink == -1:
            index = -1
            lowest = math.inf
            SR[i] = True
            ci, ui = c[i], u[i]
            for it in range(num_remaining):
                j = remaining[it]
                r = min_val + ci[j] - ui - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            min_val = lowest
            if min_val == math.inf:
                raise ValueError("cost matrix is infeasible")
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            num_remaining -= 1
            remaining[index] = remaining[num_remaining]
        # ---- dual update
        u[cur_row] += min_val
        for i2 in range(nr):
            if SR[i2] and i2 != cur_row:
                u[i2] += min_val - spc[col4row[i2]]
        for j2 in range(nc):
            if SC[j2]:
                v[j2] -= min_val - spc[j2]
        # ---- augment
        j = sink
        while True:
            i2 = path[j]
            row4col[j] = i2
            col4row[i2], j = j, col4row[i2]
            if i2 == cur_row:
                break
    if transpose:
        order = np.argsort(np.asarray(col4row), kind="stable")
        return np.asarray(col4row, dtype=np.int64)[order], order.astype(np.int64)
    return np.arange(nr, dtype=np.int64), np.asarray(col4row, dtype=np.int64)
