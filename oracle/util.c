#include <stdlib.h>
#include <stdio.h>
/*
 * oracle/util.c -- CPU ORACLE (test infrastructure only): small helpers + the statrs 0.16.0 functions the
 * reference calls (Cargo.lock:1897-1899; the crate is not under /root/reference, formulas restated from
 * its published definitions and pinned by the reference's tests: src/util/stats.rs:46-71,
 * src/hla/caller.rs:1837-1845,1884-1898).
 */
#include "sp_oracle.h"
#include <math.h>
#include <string.h>

/* src/util/homopolymers.rs:18-23 : run-length collapse */
size_t osp_hpc(const uint8_t* seq, size_t n, uint8_t* out) {
    size_t o = 0;
    for (size_t i = 0; i < n; ++i) if (i == 0 || seq[i] != seq[i - 1]) out[o++] = seq[i];
    return o;
}

/* src/util/homopolymers.rs:25-42 : index of `position` in the collapsed string */
size_t osp_hpc_pos(const uint8_t* seq, size_t n, size_t position) {
    size_t total_length = 0, offset = 0, i = 0;
    while (i < n) {
        size_t l = 1;
        while (i + l < n && seq[i + l] == seq[i]) ++l;
        total_length += l;
        if (position < total_length) break;
        offset += 1;
        i += l;
    }
    return offset;
}

/* src/util/sequence.rs:9-23 */
int osp_revcomp(const char* in, size_t n, char* out) {
    for (size_t i = 0; i < n; ++i) {
        char c = in[n - 1 - i], r;
        switch (c) {
            case 'A': r = 'T'; break;
            case 'C': r = 'G'; break;
            case 'G': r = 'C'; break;
            case 'T': r = 'A'; break;
            case 'N': r = 'N'; break;
            default: return -1;
        }
        out[i] = r;
    }
    return 0;
}

/* statrs::function::factorial::ln_factorial : cached factorials up to 170!, ln_gamma(x+1) beyond */
double osp_ln_factorial(uint64_t n) {
    if (n <= 170) {
        double f = 1.0;
        for (uint64_t i = 2; i <= n; ++i) f *= (double)i;
        return log(f);
    }
    return lgamma((double)n + 1.0);
}

/* src/util/stats.rs:11-37 */
double osp_multinomial_ln_pmf(const double* probs, const uint64_t* obs, int n) {
    uint64_t total = 0;
    for (int i = 0; i < n; ++i) total += obs[i];
    double coeff = osp_ln_factorial(total);
    for (int i = 0; i < n; ++i) coeff -= osp_ln_factorial(obs[i]);
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc = acc + (double)obs[i] * log(probs[i]);
    return coeff + acc;
}

/* statrs Binomial::ln_pmf : ln C(n,x) + x ln p + (n-x) ln(1-p) */
double osp_binomial_ln_pmf(double p, uint64_t n, uint64_t x) {
    if (x > n) return -INFINITY;
    if (p == 0.0) return x == 0 ? 0.0 : -INFINITY;
    if (p == 1.0) return x == n ? 0.0 : -INFINITY;
    double ln_binom = osp_ln_factorial(n) - osp_ln_factorial(x) - osp_ln_factorial(n - x);
    return ln_binom + (double)x * log(p) + (double)(n - x) * log(1.0 - p);
}

/* statrs Binomial::cdf = I_{1-p}(n-x, x+1); evaluated here as the exact finite sum (same value to ~1e-14) */
double osp_binomial_cdf(double p, uint64_t n, uint64_t x) {
    if (x >= n) return 1.0;
    double acc = 0.0;
    for (uint64_t k = 0; k <= x; ++k) acc += exp(osp_binomial_ln_pmf(p, n, k));
    return acc > 1.0 ? 1.0 : acc;
}

/* statrs Normal::ln_pdf */
double osp_normal_ln_pdf(double mean, double sd, double x) {
    const double LN_SQRT_2PI = 0.91893853320467274178032973640561763986139747363778341281715;
    double d = (x - mean) / sd;
    return (-0.5 * d * d) - LN_SQRT_2PI - log(sd);
}


/* Diplotype::diplotype / pharmcat_diplotype (src/data_types/pgx_diplotype.rs:13-65) */
void osp_diplotype_string(const char* hap1, const char* hap2, int pharmcat, char* out, size_t cap) {
    const int b1 = pharmcat && strchr(hap1, '+') != NULL, b2 = pharmcat && strchr(hap2, '+') != NULL;
    snprintf(out, cap, "%s%s%s/%s%s%s", b1 ? "[" : "", hap1, b1 ? "]" : "", b2 ? "[" : "", hap2, b2 ? "]" : "");
}

/* InexactHaplotype::new + full_haplotype (src/data_types/pgx_diplotype.rs:138-196); states follow VariantAlleleRelationship's
 * declaration order (1 = Match, 2 = Unexpected, 3 = Missing, everything else prints '?') */
int osp_inexact_haplotype(const char* base, int n, const char* const* labels, const uint8_t* is_vi, const int32_t* states, char* out, size_t cap) {
    int core = 1, sub = 1, modified = 0;
    size_t len = 0;
    char* buf = (char*)malloc(cap + 2);
    len += (size_t)snprintf(buf + len, cap - len, "%s", base);
    for (int i = 0; i < n; ++i) {
        if (states[i] == 1) continue;
        sub = 0; if (is_vi[i]) core = 0;
        const char sign = states[i] == 2 ? '+' : (states[i] == 3 ? '-' : '?');
        if (len < cap) len += (size_t)snprintf(buf + len, cap - len, " %c%s", sign, labels[i]);
        modified = 1;
    }
    if (modified) snprintf(out, cap, "(%s)", buf); else snprintf(out, cap, "%s", buf);
    free(buf);
    return sub ? 3 : (core ? 2 : 1);
}
