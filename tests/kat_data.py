"""Golden data transcribed from the reference's own test
/root/reference/tests/test_galerkin_coarse_matrix.cc (data only: matrices, index sets, POU)."""
import numpy as np
import scipy.sparse as sp

# :20-48  global 9x9 test matrix (MatrixMarket, 1-based in the source)
_TRIPLES = [(1, 1, 1), (1, 2, 18), (2, 1, 10), (2, 2, 2), (2, 3, 19), (3, 2, 11), (3, 3, 3), (3, 4, 20),
            (4, 3, 12), (4, 4, 4), (4, 5, 21), (5, 4, 13), (5, 5, 5), (5, 6, 22), (6, 5, 14), (6, 6, 6),
            (6, 7, 23), (7, 6, 15), (7, 7, 7), (7, 8, 24), (8, 7, 16), (8, 8, 8), (8, 9, 25), (9, 8, 17),
            (9, 9, 9)]
A_GLOBAL = np.zeros((9, 9))
for i, j, v in _TRIPLES:
    A_GLOBAL[i - 1, j - 1] = v

# :50-67  expected coarse matrix R A R^T
_C = [(1, 1, 29.52777777777778), (1, 2, 27.02777777777778), (1, 3, 7.277777777777778),
      (2, 1, 21.69444444444445), (2, 2, 28.11111111111111), (2, 3, 21.19444444444444), (2, 4, 8.166666666666666),
      (3, 1, 4.611111111111111), (3, 2, 18.52777777777778), (3, 3, 34.11111111111111), (3, 4, 36.91666666666666),
      (4, 2, 5.499999999999999), (4, 3, 31.58333333333333), (4, 4, 50.75)]
A0_EXPECTED = np.zeros((4, 4))
for i, j, v in _C:
    A0_EXPECTED[i - 1, j - 1] = v

# :104-151 additive 3x3 blocks (diag, then (0,1),(1,0),(1,2),(2,1))
_BLOCKS = {0: ((1, 2, 1.5), (18, 10, 19, 11)), 1: ((1.5, 4, 2.5), (20, 12, 21, 13)),
           2: ((2.5, 6, 3.5), (22, 14, 23, 15)), 3: ((3.5, 8, 9), (24, 16, 25, 17))}
# :153-181 parallel index sets: (global, owner?, public?) per local index
_INDEX = {0: [(0, True, False), (1, True, False), (2, True, True)],
          1: [(2, False, True), (3, True, False), (4, True, True)],
          2: [(4, False, True), (5, True, False), (6, True, True)],
          3: [(6, False, True), (7, True, False), (8, True, False)]}
# :219-247 hand-written partition of unity on the overlap-1 index sets
POU = {0: [1, 0.5, 0.5, 1. / 3], 1: [0.5, 1. / 3, 0.5, 0.5, 1. / 3],
       2: [0.5, 1. / 3, 0.5, 1. / 3, 0.5], 3: [0.5, 0.5, 1, 1. / 3]}


class ChainRank:
    def __init__(self, r):
        d, o = _BLOCKS[r]
        M = np.zeros((3, 3))
        M[0, 0], M[1, 1], M[2, 2] = d
        M[0, 1], M[1, 0], M[1, 2], M[2, 1] = o
        self.rank = r
        self.A = sp.csr_matrix(M)
        self.glob = np.array([g for g, _, _ in _INDEX[r]], dtype=np.int64)
        self.owner = np.array([o_ for _, o_, _ in _INDEX[r]], dtype=np.uint8)
        self.public = np.array([p for _, _, p in _INDEX[r]], dtype=np.uint8)


def chain():
    return [ChainRank(r) for r in range(4)]
