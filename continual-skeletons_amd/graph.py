"""Skeleton graph -> 3-partition adjacency ``A (3, V, V)`` float64 = [self, inward, outward].

Counterpart of the reference's ``datasets/graph.py:9-44`` with the bone tables of
``datasets/ntu_rgbd.py:3-35`` (NTU RGB+D, 25 joints) and ``datasets/kinetics.py:24-46``
(OpenPose, 18 joints).  Matrix convention: entry [j, i] is set for a bone i -> j and every column is
normalised by its in-degree, so ``x @ A[k]`` sums over A's ROW index (models/base.py:266).
"""
from typing import List, Sequence, Tuple

import numpy as np

# (child, parent) joint pairs, 1-based as in the NTU RGB+D documentation
_NTU_PAIRS = [
    (1, 2), (2, 21), (3, 21), (4, 3), (5, 21), (6, 5), (7, 6), (8, 7), (9, 21), (10, 9), (11, 10),
    (12, 11), (13, 1), (14, 13), (15, 14), (16, 15), (17, 1), (18, 17), (19, 18), (20, 19),
    (22, 23), (23, 8), (24, 25), (25, 12),
]
# (origin, neighbour) joint pairs, 0-based OpenPose-18 indexing
_KINETICS_PAIRS = [
    (4, 3), (3, 2), (7, 6), (6, 5), (13, 12), (12, 11), (10, 9), (9, 8), (11, 5), (8, 2), (5, 1),
    (2, 1), (0, 1), (15, 0), (14, 0), (17, 15), (16, 14),
]


def _incidence(links: Sequence[Tuple[int, int]], n: int) -> np.ndarray:
    m = np.zeros((n, n), dtype=np.float64)
    if links:
        src, dst = zip(*links)
        m[list(dst), list(src)] = 1.0
    return m


def _column_normalised(m: np.ndarray) -> np.ndarray:
    deg = m.sum(axis=0)
    inv = np.divide(1.0, deg, out=np.zeros_like(deg), where=deg > 0)
    return m * inv[None, :]


class Graph:
    """``Graph(inward, num_node).A`` -- same constructor and attributes as datasets/graph.py:35-44."""

    def __init__(self, inward: List[Tuple[int, int]], num_node: int):
        self.num_node = num_node
        self.self_link = [(i, i) for i in range(num_node)]
        self.inward = list(inward)
        self.outward = [(j, i) for (i, j) in self.inward]
        self.neighbor = self.inward + self.outward
        self.A = np.stack(
            (
                _incidence(self.self_link, num_node),
                _column_normalised(_incidence(self.inward, num_node)),
                _column_normalised(_incidence(self.outward, num_node)),
            )
        )


def ntu_graph() -> Graph:
    return Graph([(a - 1, b - 1) for a, b in _NTU_PAIRS], 25)


def kinetics_graph() -> Graph:
    return Graph(list(_KINETICS_PAIRS), 18)
