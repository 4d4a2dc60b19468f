// lmm_params.h -- structures passed by value between the host side (api.hip) and the LMM kernel launchers (lmm_kernels.hip)
#pragma once

struct LmmLinOut {              // per-variant outputs of the linear-terms kernel, Vpad each
    int *t11, *t01, *m;
    double *xky, *dg, *rss, *s1, *q1;
};

struct LmmFinParams {
    int N, D, continuous;
    int n1, n0;                 // #(y == 1), #(y == 0)   (binary prefilter margins)
    double yc_sum, yc_sq;       // sum / sum of squares of the centred phenotype (Welch, group 0 by subtraction)
    double yKy, inv_scale;
    double pret, lrtt;
    double min_af, max_af; int af_on;
};
