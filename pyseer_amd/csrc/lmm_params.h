// lmm_params.h -- structures passed by value between the host side (api.hip) and the LMM kernel launchers (lmm_kernels.hip)
#pragma once

struct LmmLinOut {              // per-variant outputs of the linear-terms kernel, Vpad each (of the row AS STORED, i.e. complemented when flip[v])
    int *t11, *t01, *m;
    double *xky, *dg, *rss, *s1, *q1;
    const unsigned char *flip;  // 1 = the row was stored complemented (k_repack_bits), or null
};

struct LmmFinParams {
    int N, D, continuous;
    int n1, n0;                 // #(y == 1), #(y == 0)   (binary prefilter margins)
    double yc_sum, yc_sq;       // sum / sum of squares of the centred phenotype (Welch, group 0 by subtraction)
    double yKy, inv_scale;
    double pret, lrtt;
    double min_af, max_af; int af_on;
    double sumv;                // sum_i v_i (= 0 up to rounding when the intercept is in the covariate span): x.v of a complemented row
    // a-posteriori bound of the fixed-point contraction: |x^T (G - Gq/s) x| <= err_norm * m', m' = carriers of the row as stored
    double err_norm;            // spectral norm of the symmetrised quantisation error of the main-pass limbs, in units of G
    double tol;                 // relative bound on xKx above which a variant is re-contracted with the extra limbs (0 = never)
    double inv_scale_low;       // 1 / scale of the extra (low) limbs
};

struct LmmRefine {              // device-side work list of the variants whose bound exceeds tol
    int *list, *count;          // variant indices, number of entries
    unsigned long long *bound_max;   // max over the batch of the final relative bound (bit pattern of a non-negative double)
};
