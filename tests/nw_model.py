"""Independent pure-Python model of the reference's Needleman-Wunsch variant
(SURVEY.md A.2), written traceback-FREE: (matches, length) are carried forward
along the chosen predecessor with two rolling rows.  TEST INFRASTRUCTURE: used to
cross-check the C oracle (which keeps the full matrices and walks the traceback,
like reference src/pairwiseSeqAlign.cpp:209-313) on small cases.
"""
NEG = -(2 ** 31) // 2
ORDER = "ARNDCQEGHILKMFPSTWYVBZX*"


def _w(x):  # wrap to int32
    return (x + 2 ** 31) % 2 ** 32 - 2 ** 31


def nw_identity(a, b, table, go=10, ge=4):
    """-> (matches, length, score).  table: 576 ints, row-major over ORDER."""
    m, n = len(a), len(b)
    ia = [ORDER.index(c) for c in a]
    ib = [ORDER.index(c) for c in b]
    # row 0
    M = [0] + [NEG] * n
    X = [NEG] * (n + 1)
    Y = [NEG] + [_w(-go - (j - 1) * ge) for j in range(1, n + 1)]
    P = [(0, j) for j in range(n + 1)]
    for i in range(1, m + 1):
        M2 = [NEG] + [0] * n
        X2 = [_w(-go - (i - 1) * ge)] + [0] * n
        Y2 = [NEG] + [0] * n
        P2 = [(0, i)] + [None] * n
        for j in range(1, n + 1):
            s = table[ia[i - 1] * 24 + ib[j - 1]]
            x = max(_w(M[j] - (go + ge)), _w(X[j] - ge))
            y = max(_w(M2[j - 1] - (go + ge)), _w(Y2[j - 1] - ge))
            d = max(_w(M[j - 1] + s), _w(X[j - 1] + s), _w(Y[j - 1] + s))
            X2[j], Y2[j] = x, y
            if d >= x and d >= y:
                M2[j] = d
                pm, pl = P[j - 1]
                P2[j] = (pm + (1 if a[i - 1] == b[j - 1] else 0), pl + 1)
            elif x >= y:
                M2[j] = x
                P2[j] = (P[j][0], P[j][1] + 1)
            else:
                M2[j] = y
                P2[j] = (P2[j - 1][0], P2[j - 1][1] + 1)
        M, X, Y, P = M2, X2, Y2, P2
    return P[n][0], P[n][1], M[n]
