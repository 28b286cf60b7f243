"""K10: one-breakpoint RBF kernel change-point detection (oracle; test infrastructure).

Restates `ruptures.KernelCPD(kernel="rbf").fit(signal).predict(n_bkps=1)` as called
at reference `src/shoulder/humerus/surgical_neck.py:31-34` (ruptures 1.1.9, absent
from this image; parity UNPINNED).  Published algorithm: gamma = 1/median(pairwise
squared distances) (1.0 when the median is 0), K_ij = exp(-clip(gamma*d_ij^2, 1e-2,
1e2)), c(a,b) = sum_{i in [a,b)} K_ii - (1/(b-a)) sum_{i,j in [a,b)} K_ij,
t* = first argmin over t in [min_size, n-min_size] (min_size=2, jump=1) of
c(0,t)+c(t,n).  `clip=False` gives the un-clipped variant; tests assert both agree
on the fixtures.
"""
import numpy as np


def rbf_gram(signal, clip=True):
    s = np.asarray(signal, dtype=np.float64).reshape(len(signal), -1)
    d2 = ((s[:, None, :] - s[None, :, :]) ** 2).sum(axis=2)
    iu = np.triu_indices(len(s), k=1)
    med = np.median(d2[iu])
    gamma = 1.0 if med == 0 else 1.0 / med
    k = d2 * gamma
    if clip:
        k = np.clip(k, 1e-2, 1e2)
    return np.exp(-k), gamma


def kernel_cpd_one_bkp(signal, min_size=2, clip=True) -> int:
    K, _ = rbf_gram(signal, clip)
    n = len(K)
    diag = np.diag(K)

    def cost(a, b):
        return diag[a:b].sum() - K[a:b, a:b].sum() / (b - a)

    best_t, best = -1, np.inf
    for t in range(min_size, n - min_size + 1):
        c = cost(0, t) + cost(t, n)
        if c < best:
            best, best_t = c, t
    return best_t
