import numpy as np


def maxdiff(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def assert_close(name, got, want, tol):
    d = maxdiff(got, want)
    assert np.isfinite(np.asarray(got, np.float64)).all(), name + ': non-finite values'
    assert d <= tol, '%s: max |diff| = %.3e > %.1e' % (name, d, tol)
    return d


def t2n(t):
    return t.detach().cpu().numpy()
