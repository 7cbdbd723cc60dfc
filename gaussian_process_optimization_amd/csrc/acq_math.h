// Acquisition formulas as device functions, shared by the batched kernels (grad.hip) and the fused one-row path (onerow.hip).
// Reference: GPyOpt/GPyOpt/util/general.py:113-129 (get_quantiles), acquisitions/EI.py:32-51, LCB.py:31-46, MPI.py:32-51,
// models/gpmodel.py:95-142 (clip at 1e-10, dsdx = dvdx / (2 sqrt(v))), acquisitions/LP.py:40-140.
#pragma once
#include "gphip_internal.h"
#include "../../include/gphip.h"

// Value f and the two partial derivatives of one acquisition at (mean, var) of the NORMALISED model:
//   d f / dx = c_m * (y_std * dmdx) + c_s * (ds_scale * dvdx)
// (un-normalisation gp.py:344-352; clip gpmodel.py:99; s floor general.py:121-124).
__device__ __forceinline__ void acq_terms(int type, double par, double fmin, double y_mean, double y_std, double mean,
                                          double var, double &f, double &c_m, double &c_s, double &ds_scale) {
    const double m = mean * y_std + y_mean;
    double v = var * (y_std * y_std);
    v = (v < 1e-10) ? 1e-10 : v;
    double s = sqrt(v);
    ds_scale = (y_std * y_std) / (2.0 * s);  // dsdx = dvdx / (2 sqrt(v)), gpmodel.py:140
    if (type == GP_ACQ_LCB) {
        f = -m + par * s;
        c_m = -1.0;
        c_s = par;
    } else {
        if (s < 1e-10) s = 1e-10;
        const double u = (fmin - m - par) / s;
        const double phi = exp(-0.5 * u * u) / 2.50662827463100050241576528481105;
        const double Phi = 0.5 * erfc(-u / 1.41421356237309504880168872420970);
        if (type == GP_ACQ_EI) {
            f = s * (u * Phi + phi);
            c_m = -Phi;
            c_s = phi;
        } else {
            f = Phi;
            c_m = -(phi / s);
            c_s = -(phi / s) * u;
        }
    }
}

// scipy.stats.norm.logcdf == cephes log_ndtr: log(ndtr(z)) for z > -20, the asymptotic series below it, -ndtr(-z) above 6.
__device__ __forceinline__ double gp_log_ndtr(double z) {
    if (z > 6.0) return -0.5 * erfc(z / 1.41421356237309504880168872420970);
    if (z > -20.0) return log(0.5 * erfc(-z / 1.41421356237309504880168872420970));
    const double log_lhs = -0.5 * z * z - log(-z) - 0.5 * log(2.0 * 3.14159265358979323846);
    double last_total = 0.0, right_hand_side = 1.0, numerator = 1.0, denom_factor = 1.0;
    const double denom_cons = 1.0 / (z * z);
    long sign = 1, i = 0;
    while (fabs(last_total - right_hand_side) > 2.220446049250313e-16) {
        i += 1;
        last_total = right_hand_side;
        sign = -sign;
        denom_factor *= denom_cons;
        numerator *= (double)(2 * i - 1);
        right_hand_side += (double)sign * numerator * denom_factor;
        if (i > 200) break;
    }
    return log_lhs + log(right_hand_side);
}

// AcquisitionLP._penalized_acquisition (LP.py:70-89) for one location: negacq = -acq(x) in, the penalised score out.
// transform: 0 = none (log(acq + 1e-50)), 1 = softplus (LP.py:77-83)
__device__ __forceinline__ double lp_value(double negacq, const double *x, int D, const double *Xb, int nb, const double *r0,
                                           const double *s0, int transform) {
    double f = -negacq;
    if (transform == 1)
        f = (f >= 40.0) ? log(f) : log(log1p(exp(f)));
    else
        f = log(f + 1e-50);
    f = -f;
    for (int k = 0; k < nb; ++k) {
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) {
            const double df = x[d] - Xb[k * D + d];
            d2 = fma(df, df, d2);
        }
        f -= gp_log_ndtr((sqrt(d2) - r0[k]) / s0[k]);
    }
    return f;
}

// Value and gradient of the penalised acquisition (LP.py:112-140): negacq = -acq(x) and dneg[D] = -d acq / dx in, overwritten
// by the penalised value's gradient; returns the penalised value.  The penaliser's gradient is the reference's: one scalar
// per (location, centre) summed over the batch and subtracted from every dimension (LP.py:91-103 has no direction factor).
__device__ __forceinline__ double lp_value_grad(double negacq, double *dneg, const double *x, int D, const double *Xb, int nb,
                                                const double *r0, const double *s0, int transform) {
    const double a = -negacq;
    double f, scale;
    if (transform == 1) {
        const double sp = log1p(exp(a));
        f = (a >= 40.0) ? log(a) : log(sp);
        scale = 1.0 / (sp * (1.0 + exp(-a)));
    } else {
        f = log(a + 1e-50);
        scale = 1.0 / a;
    }
    f = -f;
    double pen = 0.0;
    for (int k = 0; k < nb; ++k) {
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) {
            const double df = x[d] - Xb[k * D + d];
            d2 = fma(df, df, d2);
        }
        const double nm = sqrt(d2);
        const double z = (nm - r0[k]) / s0[k];
        f -= gp_log_ndtr(z);
        const double cdf = 0.5 * erfc(-z / 1.41421356237309504880168872420970);
        if (!(cdf < 1e-50)) pen += 1.0 / (s0[k] * 2.50662827463100050241576528481105 * cdf) * exp(-0.5 * z * z) / nm;
    }
    for (int d = 0; d < D; ++d) dneg[d] = scale * dneg[d] - pen;
    return f;
}
