"""Oracle: Viterbi smoothing (numpy restatement).  Test infrastructure only.

Follows /root/reference/inaSpeechSegmenter/pyannote_viterbi.py:118-224 on the
only path the segmenter exercises (no `consecutive`, no `constraint`, default
uniform `initial`), and viterbi_utils.py:29-49.
"""
import numpy as np


def pred2logemission(pred, eps=1e-10):
    # viterbi_utils.py:29-34
    pred = np.array(pred)
    ret = np.ones((len(pred), 2)) * eps
    ret[pred == 0, 0] = 1 - eps
    ret[pred == 1, 1] = 1 - eps
    return np.log(ret)


def log_trans_exp(exp, cost0=0, cost1=0):
    # viterbi_utils.py:36-42
    cost = -exp * np.log(10)
    ret = np.ones((2, 2)) * cost
    ret[0, 0] = cost0
    ret[1, 1] = cost1
    return ret


def diag_trans_exp(exp, dim):
    # viterbi_utils.py:44-49
    cost = -exp * np.log(10)
    ret = np.ones((dim, dim)) * cost
    for i in range(dim):
        ret[i, i] = 0
    return ret


def viterbi_decoding(emission, transition):
    """Most probable state path, float64 arithmetic, first-max tie-breaking.

    pyannote_viterbi.py: uniform initial :166-167; V[0] = E[0] + init :194;
    forward :202-214 (tmp[k,k'] = V[t-1,k] + T[k,k'], argmax over k);
    back-tracking :217-220; returns float64 ids (:110, via np.empty)."""
    emission = np.asarray(emission)
    T, K = emission.shape
    initial = np.log(np.ones((K,)) / K)
    V = np.empty((T, K))
    P = np.empty((T, K), dtype=int)
    V[0, :] = emission[0, :] + initial
    P[0, :] = np.arange(K)
    cols = np.arange(K)
    for t in range(1, T):
        tmp = (V[t - 1, :] + transition.T).T
        P[t, :] = np.argmax(tmp, axis=0)
        V[t, :] = emission[t, :] + tmp[P[t, :], cols]
    X = np.empty((T,), dtype=int)
    X[-1] = np.argmax(V[-1, :])
    for t in range(1, T):
        X[-(t + 1)] = P[-t, X[-t]]
    return X.astype(np.float64)
