#!/usr/bin/env python3
"""L1 gains of the multi-level 1-D analysis (9/7 as dwt_kernels.hpp implements it, and 5/3) from the input to the
low / high outputs of level l, by applying the transform to the identity (whole-sample symmetric extension, N = 1024):
the largest magnitude a subband sample can take is  max|input| x G_x x G_y  (the 2-D transform is separable per level).
Prints the tables launch_plan.hpp's coef16_ok() embeds (rounded up)."""
import numpy as np

A1, A2, A3, A4 = -1.586134342059924, -0.052980118572961, 0.882911075530934, 0.443506852043971
N1, N2 = 1.230174104914001, 0.812893066


def refl(i, n):
    if i < 0:
        i = -i
    if i >= n:
        i = 2 * (n - 1) - i
    return i


def step97(x):
    n = x.shape[0]
    e, o = x[0::2].copy(), x[1::2].copy()
    h = n // 2
    en = lambda k: e[refl(2 * k, n) // 2] if refl(2 * k, n) % 2 == 0 else None
    # lifting on rows of the matrix (linear maps): d1 = o + A1 (e[k] + e[k+1]) with symmetric extension
    def E(k):   # even sample 2k
        return e[k] if 0 <= k < h else e[refl(2 * k, n) // 2]
    def O(arr, k):  # odd-type sample 2k+1 of arr
        i = refl(2 * k + 1, n)
        return arr[(i - 1) // 2]
    d1 = np.array([o[k] + A1 * (e[k] + E(k + 1)) for k in range(h)])
    s1 = np.array([e[k] + A2 * (O(d1, k - 1) + d1[k]) for k in range(h)])
    def S(arr, k):
        i = refl(2 * k, n)
        return arr[i // 2]
    d2 = np.array([d1[k] + A3 * (s1[k] + S(s1, k + 1)) for k in range(h)])
    s2 = np.array([s1[k] + A4 * (O(d2, k - 1) + d2[k]) for k in range(h)])
    return s2 * N2, d2 * N1


def step53(x):      # the linear part (the floors move a sample by less than 1 per lifting step)
    n = x.shape[0]
    e, o = x[0::2].copy(), x[1::2].copy()
    h = n // 2
    E = lambda k: e[k] if 0 <= k < h else e[refl(2 * k, n) // 2]
    d = np.array([o[k] - 0.5 * (e[k] + E(k + 1)) for k in range(h)])
    D = lambda k: d[(refl(2 * k + 1, n) - 1) // 2]
    s = np.array([e[k] + 0.25 * (D(k - 1) + d[k]) for k in range(h)])
    return s, d


for name, step in (("9/7", step97), ("5/3", step53)):
    N = 1024
    low = np.eye(N)
    gl, gh = [], []
    for l in range(8):
        low, high = step(low)
        gl.append(np.abs(low).sum(axis=1).max())
        gh.append(np.abs(high).sum(axis=1).max())
    print(name, "low ", ", ".join("%.4f" % v for v in gl))
    print(name, "high", ", ".join("%.4f" % v for v in gh))
