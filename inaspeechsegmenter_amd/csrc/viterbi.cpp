// Host-side Viterbi smoothing (no GPU involved): compiled restatement of the reference's
// per-frame Python loop, pyannote_viterbi.py:118-224, on the unconstrained path used by
// segmenter.py:72-73 (energy detector, K=2) and :176 (CNN posteriors, K=2|3).
//
// Arithmetic contract kept from the reference:
//   * all scores are float64; float32 emissions (np.log(r) of float32 probabilities,
//     segmenter.py:176) are promoted value-by-value when added (:194, :214);
//   * initial = log(1/K) for every state (:166-167);
//   * tmp[k][k'] = V[t-1][k] + T[k][k'];  P[t][k'] = argmax_k tmp[k][k'] taking the FIRST
//     maximum (numpy argmax; a NaN counts as the maximum, as in numpy) (:207-211);
//   * V[t][k'] = E[t][k'] + tmp[P[t][k']][k'] (:214);
//   * back-tracking from argmax V[T-1] (:217-220).
#include "../../include/iss.h"
#include <cmath>
#include <cstdlib>
#include <vector>

namespace {

inline bool beats(double cand, double best) {      // numpy argmax ordering: NaN > everything, first wins
    if (std::isnan(best)) return false;
    return cand > best || std::isnan(cand);
}

// em(t, k): emission score of state k at step t, already promoted to float64
template <class EmFn>
int viterbi_core(EmFn em, int64_t T, int32_t K, const double* tr, int32_t* out) {
    if (!tr || !out || T < 0 || K < 1 || K > 16) return ISS_EINVAL;
    if (T == 0) return ISS_OK;
    std::vector<uint8_t> P((size_t)T * K);
    double va[16], vb[16];
    double* prev = va;
    double* cur = vb;
    const double init = std::log(1.0 / (double)K);
    for (int k = 0; k < K; ++k) { prev[k] = em(0, k) + init; P[k] = (uint8_t)k; }
    for (int64_t t = 1; t < T; ++t) {
        uint8_t* p = &P[(size_t)t * K];
        for (int j = 0; j < K; ++j) {
            int best = 0;
            double bv = prev[0] + tr[j];                 // tr[0*K + j]
            for (int k = 1; k < K; ++k) {
                double v = prev[k] + tr[k * K + j];
                if (beats(v, bv)) { bv = v; best = k; }
            }
            p[j] = (uint8_t)best;
            cur[j] = em(t, j) + bv;
        }
        double* tmp = prev; prev = cur; cur = tmp;
    }
    int best = 0;
    for (int k = 1; k < K; ++k) if (beats(prev[k], prev[best])) best = k;
    out[T - 1] = best;
    for (int64_t t = T - 1; t >= 1; --t) {
        best = P[(size_t)t * K + best];
        out[t - 1] = best;
    }
    return ISS_OK;
}

template <typename E>
int viterbi_impl(const E* em, int64_t T, int32_t K, const double* tr, int32_t* out) {
    if (!em) return ISS_EINVAL;
    return viterbi_core([em, K](int64_t t, int k) { return (double)em[t * K + k]; }, T, K, tr, out);
}

}  // namespace

extern "C" int iss_viterbi_f64(const double* em, int64_t T, int32_t K, const double* tr, int32_t* out) {
    return viterbi_impl<double>(em, T, K, tr, out);
}
extern "C" int iss_viterbi_f32(const float* em, int64_t T, int32_t K, const double* tr, int32_t* out) {
    return viterbi_impl<float>(em, T, K, tr, out);
}
// The energy detector in one call (segmenter.py:69-73 behind the threshold): raw activity `loge > threshold` compared in
// float64 like numpy compares a float32 array with a float64 scalar (a NaN threshold -- all-silent input -- is never
// exceeded), two-state emissions pred2logemission (viterbi_utils.py:29-34) whose two values the caller passes as numpy
// computed them (log(1e-10), log(1 - 1e-10)), smoothed by the same recursion as iss_viterbi_f64.  Saves the (T,2) float64
// emission array, its fancy-indexed fill and its np.log per file.
extern "C" int iss_energy_viterbi(const float* loge, int64_t T, double threshold, double log_eps, double log_1m_eps,
                                  const double* tr, int32_t* out) {
    if (!loge && T > 0) return ISS_EINVAL;
    return viterbi_core([=](int64_t t, int k) {
        const int active = (double)loge[t] > threshold ? 1 : 0;
        return k == active ? log_1m_eps : log_eps;
    }, T, 2, tr, out);
}
// Viterbi over nseg consecutive segments of one (sum(seg_len), K) float32 emission array, each smoothed on its own exactly
// like iss_viterbi_f32 (segmenter.py:168-178 runs the smoothing per `inlabel` segment): one call per network and device pass
// instead of one per segment.
extern "C" int iss_viterbi_segments_f32(const float* em, const int64_t* seg_len, int64_t nseg, int32_t K, const double* tr,
                                        int32_t* out) {
    if (nseg < 0 || (nseg > 0 && (!em || !seg_len))) return ISS_EINVAL;
    int64_t pos = 0;
    for (int64_t s = 0; s < nseg; ++s) {
        if (seg_len[s] < 0) return ISS_EINVAL;
        const int rc = viterbi_impl<float>(em + pos * K, seg_len[s], K, tr, out + pos);
        if (rc != ISS_OK) return rc;
        pos += seg_len[s];
    }
    return ISS_OK;
}
