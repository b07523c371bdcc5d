/* Oracle (C): the integer / histogram part of the calibration metrics.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain sequential restatement of
 *   common/evalutation/numpyfunctions.py:51-63  (_binary_calibration: digitize + three bincounts)
 *   common/evalutation/numpyfunctions.py:86-107 (uncertainty: 8 confusion x uncertain counts)
 * of the reference.  np.digitize(float32 p, float64 edges) - 1 is restated as the count of float32
 * thresholds t_k <= p, with t_k the smallest float32 whose float64 value is >= edge_k
 * (oracle/calib_oracle.py:float32_thresholds).  Sums of confidences are
 * accumulated in double in index order, which is exactly what np.bincount(weights=...) does, so the
 * result is bit-identical to the numpy path.  Built by oracle/Makefile into oracle/_build/.
 */
#include <stddef.h>
#include <stdint.h>

static inline int bin_of(float p, const float *thr, int n_bins)
{
    int b = 0;
    for (int k = 0; k < n_bins - 1; ++k) b += (p >= thr[k]);
    return b;
}

/* ids[i] = bin index of p[i] (no mask). */
void orc_bin_ids(const float *p, size_t n, const float *thr, int n_bins, uint8_t *ids)
{
    for (size_t i = 0; i < n; ++i) ids[i] = (uint8_t)bin_of(p[i], thr, n_bins);
}

/* Reliability histogram over the voxels with mask != 0 (mask may be NULL = all voxels). */
void orc_ece_hist(const float *p, const uint8_t *target, const uint8_t *mask, size_t n, const float *thr,
                  int n_bins, uint64_t *count, double *sum_conf, uint64_t *sum_pos)
{
    for (int b = 0; b < n_bins; ++b) { count[b] = 0; sum_conf[b] = 0.0; sum_pos[b] = 0; }
    for (size_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        int b = bin_of(p[i], thr, n_bins);
        count[b] += 1;
        sum_conf[b] += (double)p[i];
        sum_pos[b] += (target[i] != 0);
    }
}

/* out[t][0..7] = tp, tn, fp, fn, tpu, tnu, fpu, fnu for "uncertain" := unc > thr[t] (compare in double). */
void orc_unc_counts(const double *unc, const uint8_t *prediction, const uint8_t *target, const uint8_t *mask,
                    size_t n, const double *thr, int n_thr, uint64_t *out)
{
    for (int t = 0; t < n_thr * 8; ++t) out[t] = 0;
    for (size_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        int pr = prediction[i] != 0, tg = target[i] != 0;
        int cell = tg ? (pr ? 0 : 3) : (pr ? 2 : 1); /* tp=0 tn=1 fp=2 fn=3 */
        for (int t = 0; t < n_thr; ++t) {
            out[t * 8 + cell] += 1;
            out[t * 8 + 4 + cell] += (unc[i] > thr[t]);
        }
    }
}
