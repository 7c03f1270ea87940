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

template <typename E>
int viterbi_impl(const E* em, int64_t T, int32_t K, const double* tr, int32_t* out) {
    if (!em || !tr || !out || T < 0 || K < 1 || K > 16) return ISS_EINVAL;
    if (T == 0) return ISS_OK;
    std::vector<uint8_t> P((size_t)T * K);
    double va[16], vb[16];
    double* prev = va;
    double* cur = vb;
    const double init = std::log(1.0 / (double)K);
    for (int k = 0; k < K; ++k) { prev[k] = (double)em[k] + init; P[k] = (uint8_t)k; }
    for (int64_t t = 1; t < T; ++t) {
        const E* e = em + t * K;
        uint8_t* p = &P[(size_t)t * K];
        for (int j = 0; j < K; ++j) {
            int best = 0;
            double bv = prev[0] + tr[j];                 // tr[0*K + j]
            for (int k = 1; k < K; ++k) {
                double v = prev[k] + tr[k * K + j];
                if (beats(v, bv)) { bv = v; best = k; }
            }
            p[j] = (uint8_t)best;
            cur[j] = (double)e[j] + bv;
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

}  // namespace

extern "C" int iss_viterbi_f64(const double* em, int64_t T, int32_t K, const double* tr, int32_t* out) {
    return viterbi_impl<double>(em, T, K, tr, out);
}
extern "C" int iss_viterbi_f32(const float* em, int64_t T, int32_t K, const double* tr, int32_t* out) {
    return viterbi_impl<float>(em, T, K, tr, out);
}
