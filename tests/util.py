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


def greedy_tokens_under_margin_rule(tokens, ref_dec, what='', margin_tol=1e-3):
    """free-running decoder tokens [T, n] against an oracle / reference decoder run on
    the same questions (`ref_dec` = dict with predicted_tokens [T, n], token_scores [T, n, V],
    token_validity [T, n, V]).  Per question the tokens must be identical up to the first step whose
    top-2 margin over the valid tokens is < margin_tol; a differing token anywhere else fails.
    Returns the indices of the questions whose layout differs (each at a proven near-tie)."""
    ref_tok = np.asarray(ref_dec['predicted_tokens'])
    sc = np.where(np.asarray(ref_dec['token_validity'], bool), np.asarray(ref_dec['token_scores'], np.float64),
                  -np.inf)
    top2 = np.sort(sc, axis=2)[:, :, -2:]
    margin = top2[:, :, 1] - top2[:, :, 0]
    T = tokens.shape[0]
    flipped = []
    for i in range(tokens.shape[1]):
        stop = (tokens[:, i] != ref_tok[:, i]) | (margin[:, i] < margin_tol)
        upto = int(np.argmax(stop)) if stop.any() else T
        assert np.array_equal(tokens[:upto, i], ref_tok[:upto, i]), what
        if (tokens[:, i] != ref_tok[:, i]).any():
            assert upto < T and margin[upto, i] < margin_tol, \
                '%s: token flip at a non-tie (question %d, step %d, top-2 margin %.3e)' % \
                (what, i, upto, margin[upto, i] if upto < T else float('nan'))
            flipped.append(i)
    return flipped
