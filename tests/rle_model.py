"""Closed form of rle::rle_one's block cuts (the form banzai_amd/csrc/rle1.hip implements),
restated with numpy so that it can be checked against the oracle's literal state machine on CPU."""
import numpy as np


def canon_len(L):
    return 5 * (L // 255) + (L % 255 if L % 255 < 4 else 5)


def cut_in_run(Lr, R):
    """Budget R < canon_len(Lr) inside a freshly chunked run of Lr bytes -> (full chunks, literals)."""
    nf = Lr // 255
    k = min(R // 5, nf)
    rho = R - 5 * k
    ell = 255 if k < nf else Lr - 255 * nf
    t = min(rho, 3) if ell >= 4 else rho
    return k, t


def split(data, M):
    """-> [(in_off, in_len, rle_len)]"""
    a = np.frombuffer(data, dtype=np.uint8)
    N = len(a)
    if N == 0:
        return []
    starts = np.flatnonzero(np.concatenate(([True], a[1:] != a[:-1])))
    RS = np.concatenate((starts, [N])).astype(np.int64)
    L = np.diff(RS)
    OL = 5 * (L // 255) + np.where(L % 255 < 4, L % 255, 5)
    PO = np.concatenate(([0], np.cumsum(OL)))
    J = len(L)
    blocks, s, j0 = [], 0, 0
    while s < N:
        Lr = int(RS[j0 + 1] - s)
        A = canon_len(Lr)
        if A > M:
            k, t = cut_in_run(Lr, M)
            blocks.append((s, 255 * k + t, 5 * k + t))
            s += 255 * k + t
            continue
        lim = M - A + PO[j0 + 1]
        x = min(int(np.searchsorted(PO, lim, side="right")) - 1, J)
        cum = int(A + PO[x] - PO[j0 + 1])
        if x == J:
            blocks.append((s, N - s, cum))
            s = N
        else:
            k, t = cut_in_run(int(RS[x + 1] - RS[x]), M - cum)
            blocks.append((s, int(RS[x]) - s + 255 * k + t, cum + 5 * k + t))
            s = int(RS[x]) + 255 * k + t
            j0 = x
    return blocks
