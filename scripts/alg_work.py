#!/usr/bin/env python3
"""Algorithmic work per MPC ADMM iteration / factorisation as a function of the horizon N:

    F_iter(N) = 4 nnzL(N) + 4 nnz(A) + 12 (n + m),   nnz(A) = 126 N - 18,  n + m = 68 N
    F_fac(N)  = sum over columns of (column count of L)^2

with L the Cholesky factor of the reduced KKT matrix P + sigma I + A' R A (24N x 24N) under the time-interleaved
ordering (X_1, f_0, X_2, f_1, ...), counted symbolically (structural non-zeros of the pattern, no numerical
cancellation).  SURVEY's probe gave nnzL(16) = 10 946 and F_fac(16) ~ 0.33 Mflop; this script reproduces those and
extends them to any N — bench.py uses the fitted linear forms printed at the end.
"""
import sys

import numpy as np


def pattern(N):
    n, m = 24 * N, 44 * N
    A = np.zeros((m, n), dtype=bool)
    for k in range(N):
        for i in range(12):
            A[12 * k + i, 12 * k + i] = True
        if k > 0:
            for i in range(12):
                A[12 * k + i, 12 * (k - 1) + i] = True
            for i in range(6):
                A[12 * k + i, 12 * (k - 1) + 6 + i] = True
        for f in range(4):
            for c in range(3):
                A[12 * k + 6 + c, 12 * (N + k) + 3 * f + c] = True       # linear rows: dt/m
                for r in range(3):
                    A[12 * k + 9 + r, 12 * (N + k) + 3 * f + c] = True   # angular rows: all 12 columns stored
                A[12 * N + 12 * k + 3 * f + c, 12 * (N + k) + 3 * f + c] = True
            r0, c0 = 24 * N + 20 * k + 5 * f, 12 * (N + k) + 3 * f
            for r, cols in enumerate(((0, 2), (0, 2), (1, 2), (1, 2), (2,))):
                for c in cols:
                    A[r0 + r, c0 + c] = True
    return A


def symbolic(N):
    A = pattern(N)
    n = 24 * N
    K = (A.T.astype(np.int32) @ A.astype(np.int32)) > 0
    K |= np.eye(n, dtype=bool)
    order = []
    for k in range(N):
        order += list(range(12 * k, 12 * k + 12)) + list(range(12 * (N + k), 12 * (N + k) + 12))
    K = K[np.ix_(order, order)]
    L = np.tril(K)
    for j in range(n):  # symbolic elimination
        rows = np.nonzero(L[j + 1:, j])[0] + j + 1
        if rows.size:
            L[np.ix_(rows, rows)] |= np.tril(np.ones((rows.size, rows.size), dtype=bool))
    cnt = L.sum(axis=0) - 1  # below-diagonal counts
    nnzL = int(cnt.sum())
    f_fac = float((cnt.astype(np.float64) ** 2).sum())
    return nnzL, f_fac, int(A.sum())


if __name__ == "__main__":
    Ns = [int(a) for a in sys.argv[1:]] or [8, 16, 24, 32]
    rows = []
    for N in Ns:
        nnzL, f_fac, nnzA = symbolic(N)
        f_iter = 4 * nnzL + 4 * nnzA + 12 * 68 * N
        rows.append((N, nnzL, nnzA, f_iter, f_fac))
        print("N=%d nnzL=%d nnzA=%d (126N-18=%d) F_iter=%.1f kflop F_fac=%.3f Mflop" % (N, nnzL, nnzA, 126 * N - 18,
                                                                                          f_iter / 1e3, f_fac / 1e6))
    if len(rows) >= 2:
        (n0, l0, _, _, f0), (n1, l1, _, _, f1) = rows[0], rows[-1]
        a, b = (l1 - l0) / (n1 - n0), l0 - n0 * (l1 - l0) / (n1 - n0)
        c, d = (f1 - f0) / (n1 - n0), f0 - n0 * (f1 - f0) / (n1 - n0)
        print("nnzL(N) = %.1f N %+.1f ;  F_fac(N) = %.1f N %+.1f" % (a, b, c, d))
