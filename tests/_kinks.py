"""Kink-resolved comparisons (test infrastructure).  The iwgan / wgan critic is piecewise linear (lrelu), the generator has
relu behind batch norm: a gradient evaluated in float32 and one evaluated in float64 at the SAME variables differ by O(1e-3)
when some pre-activation lies within float32 rounding of zero and the two evaluations take different sides of the kink
("Kinks").  `resolve` makes that explanation testable: it records every pre-activation the oracle sees
within `tol` of zero (oracle/gan_ref.py: KINK hook) and, when the plain comparison misses the bound, re-evaluates the oracle
with the derivative of one, two, ... of those entries flipped; the comparison passes only if SOME assignment of those
near-zero entries brings EVERY entry of every tensor within the bound."""
import itertools

import numpy as np

from oracle import gan_ref as G


class Kinks:
    def __init__(self, tol):
        self.tol, self.tower, self.near, self.flip = tol, 0, {}, frozenset()

    def mask(self, tag, layer, pre, default, lo):
        flat = pre.reshape(-1)
        idx = np.flatnonzero(np.abs(flat) < self.tol)
        if idx.size == 0:
            return default
        m = default
        for j in idx:
            key = (self.tower, tag, layer, int(j))
            self.near[key] = float(flat[j])
            if key in self.flip:
                if m is default:
                    m = default.copy()
                v = m.reshape(-1)                      # (a view: the copy is contiguous)
                v[j] = (1.0 + lo) - v[j]
        return m


def resolve(compute, worst, bound=1e-3, tol=1e-5, max_flips=2, max_cands=12):
    """compute() -> gradients (evaluated under G.KINK); worst(grads) -> largest relative deviation from the other side.
    Returns (grads, flipped keys or None when nothing resolves it, deviation, number of near-zero pre-activations)."""
    K = Kinks(tol)
    G.KINK = K
    try:
        ref = compute()
        w = worst(ref)
        if w < bound:
            return ref, (), w, len(K.near)
        cands = sorted(K.near, key=lambda k: abs(K.near[k]))[:max_cands]
        for r in range(1, max_flips + 1):
            for S in itertools.combinations(cands, r):
                K.flip = frozenset(S)
                g = compute()
                w2 = worst(g)
                if w2 < bound:
                    return g, tuple((k, K.near[k]) for k in S), w2, len(cands)
        return ref, None, w, len(cands)
    finally:
        G.KINK = None
