"""TEST INFRASTRUCTURE ONLY - never imported by the product package.

numpy restatements of the two third-party algorithms the reference's SORT path
calls but which are absent from /root/reference and from this image:

* filterpy.kalman.KalmanFilter  (environment.yml:35, unpinned; 1.4.5 was current)
  call sites: /root/reference/tracking/sort/sort.py:97 (ctor), :164 (update), :172 (predict)
* sklearn.utils.linear_assignment_.linear_assignment  (scikit-learn 0.22.2,
  environment.yml:16) call site: /root/reference/tracking/sort/sort.py:26,206

They are injected through ``sys.modules`` by ``oracle/gen_golden_sort.py`` so that the
reference's own control flow (sort.py / tracker_sort.py / utils.py) can run in this
container and produce the golden fixtures under tests/golden/.  The published
algorithms are restated from their documentation: the textbook Kalman filter in
Joseph form and the classic Munkres six-step procedure (row reduction, star zeros,
cover starred columns, prime uncovered zeros in row-major order, augmenting path,
min-uncovered adjustment).  PARITY NOTE: the tie-breaking of the Munkres restatement
is pinned only to that published description (no copy of sklearn 0.22.2 on disk);
optimal cost is checked against scipy.optimize.linear_sum_assignment in tests.
"""
import numpy as np


class KalmanFilter(object):
    """Linear Kalman filter with the attribute names sort.py touches (x, P, Q, F, H, R)."""

    def __init__(self, dim_x, dim_z, dim_u=0):
        self.dim_x = dim_x
        self.dim_z = dim_z
        self.x = np.zeros((dim_x, 1))
        self.P = np.eye(dim_x)
        self.Q = np.eye(dim_x)
        self.F = np.eye(dim_x)
        self.H = np.zeros((dim_z, dim_x))
        self.R = np.eye(dim_z)
        self._alpha_sq = 1.0
        self._I = np.eye(dim_x)
        self.inv = np.linalg.inv

    def predict(self):
        F = self.F
        self.x = np.dot(F, self.x)
        self.P = self._alpha_sq * np.dot(np.dot(F, self.P), F.T) + self.Q

    def update(self, z):
        z = np.asarray(z, dtype=float).reshape(self.dim_z, 1)
        H = self.H
        R = self.R
        y = z - np.dot(H, self.x)
        PHT = np.dot(self.P, H.T)
        S = np.dot(H, PHT) + R
        SI = self.inv(S)
        K = np.dot(PHT, SI)
        self.x = self.x + np.dot(K, y)
        I_KH = self._I - np.dot(K, H)
        self.P = np.dot(np.dot(I_KH, self.P), I_KH.T) + np.dot(np.dot(K, R), K.T)
        self.y, self.S, self.SI, self.K = y, S, SI, K


class _MunkresState(object):
    def __init__(self, cost):
        self.C = cost.copy()          # keeps the caller's dtype (float32 from sort.py:201)
        n, m = self.C.shape
        self.row_uncovered = np.ones(n, dtype=bool)
        self.col_uncovered = np.ones(m, dtype=bool)
        self.Z0_r = 0
        self.Z0_c = 0
        self.path = np.zeros((n + m, 2), dtype=int)
        self.marked = np.zeros((n, m), dtype=int)   # 1 = star, 2 = prime

    def clear_covers(self):
        self.row_uncovered[:] = True
        self.col_uncovered[:] = True


def _step1(s):
    s.C -= s.C.min(axis=1)[:, np.newaxis]
    for i, j in zip(*np.where(s.C == 0)):         # row-major order
        if s.col_uncovered[j] and s.row_uncovered[i]:
            s.marked[i, j] = 1
            s.col_uncovered[j] = False
            s.row_uncovered[i] = False
    s.clear_covers()
    return _step3


def _step3(s):
    starred = (s.marked == 1)
    s.col_uncovered[np.any(starred, axis=0)] = False
    if starred.sum() < s.C.shape[0]:
        return _step4
    return None


def _step4(s):
    Cz = (s.C == 0).astype(int)
    covered = Cz * s.row_uncovered[:, np.newaxis]
    covered *= s.col_uncovered.astype(int)
    n, m = s.C.shape
    while True:
        row, col = np.unravel_index(np.argmax(covered), (n, m))   # first uncovered zero, row-major
        if covered[row, col] == 0:
            return _step6
        s.marked[row, col] = 2
        star_col = np.argmax(s.marked[row] == 1)
        if s.marked[row, star_col] != 1:
            s.Z0_r, s.Z0_c = row, col
            return _step5
        col = star_col
        s.row_uncovered[row] = False
        s.col_uncovered[col] = True
        covered[:, col] = Cz[:, col] * s.row_uncovered.astype(int)
        covered[row] = 0


def _step5(s):
    count = 0
    path = s.path
    path[count, 0] = s.Z0_r
    path[count, 1] = s.Z0_c
    while True:
        row = np.argmax(s.marked[:, path[count, 1]] == 1)
        if s.marked[row, path[count, 1]] != 1:
            break
        count += 1
        path[count, 0] = row
        path[count, 1] = path[count - 1, 1]
        col = np.argmax(s.marked[path[count, 0]] == 2)
        if s.marked[row, col] != 2:
            col = -1
        count += 1
        path[count, 0] = path[count - 1, 0]
        path[count, 1] = col
    for i in range(count + 1):
        if s.marked[path[i, 0], path[i, 1]] == 1:
            s.marked[path[i, 0], path[i, 1]] = 0
        else:
            s.marked[path[i, 0], path[i, 1]] = 1
    s.clear_covers()
    s.marked[s.marked == 2] = 0
    return _step3


def _step6(s):
    if np.any(s.row_uncovered) and np.any(s.col_uncovered):
        minval = np.min(s.C[s.row_uncovered], axis=0)
        minval = np.min(minval[s.col_uncovered])
        s.C[np.logical_not(s.row_uncovered)] += minval
        s.C[:, s.col_uncovered] -= minval
    return _step4


def _hungarian(X):
    X = np.atleast_2d(X)
    transposed = X.shape[1] < X.shape[0]
    if transposed:
        X = X.T
    s = _MunkresState(X)
    step = None if 0 in X.shape else _step1
    while step is not None:
        step = step(s)
    results = np.array(np.where(s.marked == 1)).T
    if transposed:
        results = results[:, ::-1]
    return results


def linear_assignment(X):
    """Pairs (row, col) of a minimum-cost assignment, sorted by row (then col)."""
    indices = _hungarian(X).tolist()
    indices.sort()
    indices = np.array(indices, dtype=int)
    indices.shape = (-1, 2)
    return indices
