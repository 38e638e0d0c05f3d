/*
 * vmaf_oracle.c -- CPU restatement of the VMAF feature arithmetic.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The hot path of the reference (yoseph007/PQA2) contains no arithmetic
 * of its own: app/vmaf_analyzer.py:406-419,446 spawns `ffmpeg -lavfi libvmaf=...` and
 * app/vmaf_analyzer.py:1027-1034,1057-1064 spawn FFmpeg's `psnr=` / `ssim=` filters.  The
 * arithmetic therefore lives in third-party code that is absent from /root/reference and not
 * version-pinned by it (hint only: "FFmpeg 7.1.1" at app/vmaf_analyzer.py:334, whose full
 * builds bundle libvmaf 3.0.0).  Neither ffmpeg nor libvmaf exists in the build container or
 * on the GPU box, and the reference ships no tests, sample media or golden scores.  This
 * file restates the PUBLISHED algorithms of
 *     libvmaf 3.0.0  src/feature/{vif.c,vif_tools.c,adm.c,adm_tools.c,motion.c,
 *                                  float_vif.c,float_adm.c,float_motion.c,picture_copy.c}
 *     FFmpeg 7.1     libavfilter/{vf_psnr.c,vf_ssim.c}
 * from their public definitions (SURVEY.md section 8(a) rows a4-VIF/ADM/motion, a5 and
 * Appendix C).  It is pinned only by (i) closed-form known answers, (ii) an independent
 * numpy/scipy float64 restatement (oracle/np_restatement.py) and (iii) the SVM anchor
 * (1,0,1,1,1,1) -> 97.428043 on the bundled model JSON.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product path (pqa2_amd/) never does.
 *
 * Build:  REAL=float  -> liboracle_f32.so  (libvmaf's own arithmetic type, scalar-C tap order,
 *                                            compiled with -ffp-contract=off)
 *         REAL=double -> liboracle_f64.so  (same algorithm in f64: the "truth" both the f32
 *                                            oracle and the HIP kernels are measured against)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * helpers
 * ---------------------------------------------------------------------------------------- */

/* libvmaf's mirror rule (vif_tools.c vif_filter1d_s, adm_tools.c dwt2_src_indices_filt_s,
 * convolution_internal.h convolution_edge_s): low edge reflects WITHOUT repeating the edge
 * sample (-i), high edge reflects WITH the edge sample repeated (2n - i - 1). */
static inline int mirror(int i, int n)
{
    if (i < 0) i = -i;
    else if (i >= n) i = 2 * n - i - 1;
    /* tiny planes (n < radius): fold again until inside; libvmaf never sees such sizes */
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - i - 1;
    }
    return i;
}

/* Gaussian taps: N taps, sigma = N/5, normalised in double then rounded to `real`.
 * Reproduces libvmaf's vif_filter1d_table (17/9/5/3) and motion FILTER_5_s to 1e-7. */
static void gaussian_taps(int n, real *f)
{
    double sigma = n / 5.0, sum = 0.0, t[32];
    for (int k = 0; k < n; ++k) {
        double d = k - n / 2;
        t[k] = exp(-0.5 * d * d / (sigma * sigma));
        sum += t[k];
    }
    for (int k = 0; k < n; ++k) f[k] = (real)(float)(t[k] / sum); /* table is stored as float */
}

ORC_EXPORT void orc_gaussian_taps(int n, double *out)
{
    real f[32];
    gaussian_taps(n, f);
    for (int k = 0; k < n; ++k) out[k] = (double)f[k];
}

/* picture_copy.c: luma -> float, 8-bit: v - 128 ; hbd: v / 2^(bpc-8) - 128 */
ORC_EXPORT void orc_picture_copy(const void *src, int stride_bytes, int bpc, int w, int h, real *dst)
{
    if (bpc <= 8) {
        const uint8_t *s = (const uint8_t *)src;
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j)
                dst[(size_t)i * w + j] = (real)s[(size_t)i * stride_bytes + j] + (real)-128.0;
    } else {
        const real scaler = (real)(1 << (bpc - 8));
        for (int i = 0; i < h; ++i) {
            const uint16_t *s = (const uint16_t *)((const uint8_t *)src + (size_t)i * stride_bytes);
            for (int j = 0; j < w; ++j)
                dst[(size_t)i * w + j] = (real)s[j] / scaler + (real)-128.0;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * VIF  (libvmaf vif.c compute_vif, vif_tools.c vif_filter1d_{s,sq_s,xy_s}, vif_dec2_s,
 *       vif_statistic_s)
 * ---------------------------------------------------------------------------------------- */

enum { VIF_PLAIN = 0, VIF_SQ = 1, VIF_XY = 2 };

/* separable filter, vertical pass first (per output row), then horizontal -- vif_filter1d_s.
 * mode SQ squares the sample in the vertical pass, XY multiplies the two sources there. */
/* integer_vif.c pads differently from vif_tools.c: reflect-101 on BOTH edges (pad_top_and_bottom,
 * PADDING_SQ_DATA: index n-1+i -> n-1-i).  border101 selects that rule so the float arithmetic can be
 * compared with the fixed-point extractor the default models name (see vmaf_int_oracle.c). */
static inline int mirror_vif(int i, int n, int border101)
{
    if (!border101) return mirror(i, n);
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

static void vif_filter1d(const real *f, int fw, const real *a, const real *b, int mode,
                         real *dst, int w, int h, int border101)
{
    real *tmp = (real *)malloc(sizeof(real) * (size_t)w);
    for (int i = 0; i < h; ++i) {
        for (int j = 0; j < w; ++j) {
            real accum = 0;
            for (int fi = 0; fi < fw; ++fi) {
                int ii = mirror_vif(i - fw / 2 + fi, h, border101);
                real v = a[(size_t)ii * w + j];
                real img = mode == VIF_PLAIN ? v : (mode == VIF_SQ ? v * v : v * b[(size_t)ii * w + j]);
                accum += f[fi] * img;
            }
            tmp[j] = accum;
        }
        for (int j = 0; j < w; ++j) {
            real accum = 0;
            for (int fj = 0; fj < fw; ++fj) {
                int jj = mirror_vif(j - fw / 2 + fj, w, border101);
                accum += f[fj] * tmp[jj];
            }
            dst[(size_t)i * w + j] = accum;
        }
    }
    free(tmp);
}

static void vif_dec2(const real *src, real *dst, int src_w, int src_h)
{
    int dw = src_w / 2, dh = src_h / 2;
    for (int i = 0; i < dh; ++i)
        for (int j = 0; j < dw; ++j)
            dst[(size_t)i * dw + j] = src[(size_t)(2 * i) * src_w + 2 * j];
}

static real real_log2(real x) { return sizeof(real) == 4 ? (real)log2f((float)x) : (real)log2((double)x); }

static void vif_statistic(const real *mu1, const real *mu2, const real *xx, const real *yy,
                          const real *xy, int w, int h, double gain_limit, double *num, double *den)
{
    const real sigma_nsq = 2;
    const real sigma_max_inv = (real)(4.0 / (255.0 * 255.0));
    const real eps = (real)1.0e-10;
    const real gl = (real)gain_limit;
    real accum_num = 0, accum_den = 0;
    for (int i = 0; i < h; ++i) {
        real inner_num = 0, inner_den = 0;
        for (int j = 0; j < w; ++j) {
            size_t k = (size_t)i * w + j;
            real m1 = mu1[k], m2 = mu2[k];
            real sigma1_sq = xx[k] - m1 * m1;
            real sigma2_sq = yy[k] - m2 * m2;
            real sigma12 = xy[k] - m1 * m2;
            sigma1_sq = sigma1_sq < 0 ? 0 : sigma1_sq;
            sigma2_sq = sigma2_sq < 0 ? 0 : sigma2_sq;
            real g = sigma12 / (sigma1_sq + eps);
            real sv_sq = sigma2_sq - g * sigma12;
            if (sigma1_sq < eps) { g = 0; sv_sq = sigma2_sq; sigma1_sq = 0; }
            if (sigma2_sq < eps) { g = 0; sv_sq = 0; }
            if (g < 0) { sv_sq = sigma2_sq; g = 0; }
            sv_sq = sv_sq < eps ? eps : sv_sq;
            g = g < gl ? g : gl;
            real num_val = real_log2((real)1 + (g * g * sigma1_sq) / (sv_sq + sigma_nsq));
            real den_val = real_log2((real)1 + sigma1_sq / sigma_nsq);
            if (sigma12 < 0) num_val = 0;
            if (sigma1_sq < sigma_nsq) { num_val = (real)1 - sigma2_sq * sigma_max_inv; den_val = 1; }
            inner_num += num_val;
            inner_den += den_val;
        }
        accum_num += inner_num;
        accum_den += inner_den;
    }
    *num = (double)accum_num;
    *den = (double)accum_den;
}

/* out[0..3] = num per scale, out[4..7] = den per scale.  ref/dis are picture_copy'd planes. */
ORC_EXPORT int orc_vif2(const real *ref, const real *dis, int w, int h, double vif_enhn_gain_limit,
                        int border101, double *out)
{
    size_t n = (size_t)w * h;
    real *buf = (real *)malloc(sizeof(real) * n * 9);
    if (!buf) return -1;
    real *ref_s = buf, *dis_s = buf + n, *mu1 = buf + 2 * n, *mu2 = buf + 3 * n, *xx = buf + 4 * n,
         *yy = buf + 5 * n, *xy = buf + 6 * n, *cur_ref = buf + 7 * n, *cur_dis = buf + 8 * n;
    memcpy(cur_ref, ref, sizeof(real) * n);
    memcpy(cur_dis, dis, sizeof(real) * n);
    static const int fwidth[4] = { 17, 9, 5, 3 };
    for (int scale = 0; scale < 4; ++scale) {
        real f[32];
        gaussian_taps(fwidth[scale], f);
        int fw = fwidth[scale];
        if (scale > 0) {
            /* filter the previous scale with THIS scale's kernel, keep even samples */
            vif_filter1d(f, fw, cur_ref, NULL, VIF_PLAIN, mu1, w, h, border101);
            vif_filter1d(f, fw, cur_dis, NULL, VIF_PLAIN, mu2, w, h, border101);
            vif_dec2(mu1, ref_s, w, h);
            vif_dec2(mu2, dis_s, w, h);
            w /= 2; h /= 2;
            memcpy(cur_ref, ref_s, sizeof(real) * (size_t)w * h);
            memcpy(cur_dis, dis_s, sizeof(real) * (size_t)w * h);
        }
        vif_filter1d(f, fw, cur_ref, NULL, VIF_PLAIN, mu1, w, h, border101);
        vif_filter1d(f, fw, cur_dis, NULL, VIF_PLAIN, mu2, w, h, border101);
        vif_filter1d(f, fw, cur_ref, NULL, VIF_SQ, xx, w, h, border101);
        vif_filter1d(f, fw, cur_dis, NULL, VIF_SQ, yy, w, h, border101);
        vif_filter1d(f, fw, cur_ref, cur_dis, VIF_XY, xy, w, h, border101);
        vif_statistic(mu1, mu2, xx, yy, xy, w, h, vif_enhn_gain_limit, &out[scale], &out[4 + scale]);
    }
    free(buf);
    return 0;
}

ORC_EXPORT int orc_vif(const real *ref, const real *dis, int w, int h, double vif_enhn_gain_limit,
                       double *out)
{
    return orc_vif2(ref, dis, w, h, vif_enhn_gain_limit, 0, out);
}

/* ------------------------------------------------------------------------------------------
 * ADM  (libvmaf adm.c compute_adm, adm_tools.c adm_dwt2_s, adm_decouple_s, adm_csf_s,
 *       adm_csf_den_scale_s, adm_cm_s, adm_tools.h dwt_quant_step)
 * ---------------------------------------------------------------------------------------- */

static const double DB2_LO[4] = { 0.482962913144690, 0.836516303737469, 0.224143868041857, -0.129409522550921 };
static const double DB2_HI[4] = { -0.129409522550921, -0.224143868041857, 0.836516303737469, -0.482962913144690 };

/* Watson DWT 7/9 noise-floor model, Y channel */
static const double WATSON_A = 0.495, WATSON_K = 0.466, WATSON_F0 = 0.401;
static const double WATSON_G[4] = { 1.501, 1.0, 0.534, 1.0 };
static const double BASIS_AMP[6][4] = {
    { 0.62171, 0.67234, 0.72709, 0.67234 },     { 0.34537, 0.41317, 0.49428, 0.41317 },
    { 0.18004, 0.22727, 0.28688, 0.22727 },     { 0.091401, 0.11792, 0.15214, 0.11792 },
    { 0.045943, 0.059758, 0.077727, 0.059758 }, { 0.023013, 0.030018, 0.039156, 0.030018 },
};

static real dwt_quant_step(int lambda, int theta, double view_dist, int display_h)
{
    real r = (real)(view_dist * display_h * M_PI / 180.0);
    real temp = (real)log10(pow(2.0, lambda + 1) * WATSON_F0 * WATSON_G[theta] / (double)r);
    real Q = (real)(2.0 * WATSON_A * pow(10.0, WATSON_K * (double)temp * (double)temp) / BASIS_AMP[lambda][theta]);
    return Q;
}

ORC_EXPORT void orc_adm_rfactors(double *out8) /* [scale][0]=h/v, [scale][1]=d */
{
    for (int s = 0; s < 4; ++s) {
        out8[2 * s] = (double)((real)1 / dwt_quant_step(s, 1, 3.0, 1080));
        out8[2 * s + 1] = (double)((real)1 / dwt_quant_step(s, 2, 3.0, 1080));
    }
}

typedef struct { real *a, *h, *v, *d; } band_t;

static void adm_dwt2(const real *src, band_t *dst, int w, int h)
{
    int ow = (w + 1) / 2, oh = (h + 1) / 2;
    real lo[4], hi[4];
    for (int k = 0; k < 4; ++k) { lo[k] = (real)(float)DB2_LO[k]; hi[k] = (real)(float)DB2_HI[k]; }
    real *tmplo = (real *)malloc(sizeof(real) * (size_t)w);
    real *tmphi = (real *)malloc(sizeof(real) * (size_t)w);
    for (int i = 0; i < oh; ++i) {
        int i0 = mirror(2 * i - 1, h), i1 = mirror(2 * i, h), i2 = mirror(2 * i + 1, h), i3 = mirror(2 * i + 2, h);
        for (int j = 0; j < w; ++j) {
            real s0 = src[(size_t)i0 * w + j], s1 = src[(size_t)i1 * w + j];
            real s2 = src[(size_t)i2 * w + j], s3 = src[(size_t)i3 * w + j];
            real acc = 0;
            acc += lo[0] * s0; acc += lo[1] * s1; acc += lo[2] * s2; acc += lo[3] * s3;
            tmplo[j] = acc;
            acc = 0;
            acc += hi[0] * s0; acc += hi[1] * s1; acc += hi[2] * s2; acc += hi[3] * s3;
            tmphi[j] = acc;
        }
        for (int j = 0; j < ow; ++j) {
            int j0 = mirror(2 * j - 1, w), j1 = mirror(2 * j, w), j2 = mirror(2 * j + 1, w), j3 = mirror(2 * j + 2, w);
            real s0 = tmplo[j0], s1 = tmplo[j1], s2 = tmplo[j2], s3 = tmplo[j3];
            real acc = 0;
            acc += lo[0] * s0; acc += lo[1] * s1; acc += lo[2] * s2; acc += lo[3] * s3;
            dst->a[(size_t)i * ow + j] = acc;
            acc = 0;
            acc += hi[0] * s0; acc += hi[1] * s1; acc += hi[2] * s2; acc += hi[3] * s3;
            dst->v[(size_t)i * ow + j] = acc;
            s0 = tmphi[j0]; s1 = tmphi[j1]; s2 = tmphi[j2]; s3 = tmphi[j3];
            acc = 0;
            acc += lo[0] * s0; acc += lo[1] * s1; acc += lo[2] * s2; acc += lo[3] * s3;
            dst->h[(size_t)i * ow + j] = acc;
            acc = 0;
            acc += hi[0] * s0; acc += hi[1] * s1; acc += hi[2] * s2; acc += hi[3] * s3;
            dst->d[(size_t)i * ow + j] = acc;
        }
    }
    free(tmplo);
    free(tmphi);
}

static inline real clamp01(real k) { return k < 0 ? 0 : (k > 1 ? 1 : k); }

/* adm_decouple_s over the whole plane (libvmaf restricts it to the cropped window + 1 tap;
 * values outside that window are never read, so computing everywhere is equivalent). */
static void adm_decouple(const band_t *ref, const band_t *dis, band_t *r, band_t *a, int w, int h,
                         double gain_limit)
{
    const real cos_1deg_sq = (real)(cos(1.0 * M_PI / 180.0) * cos(1.0 * M_PI / 180.0));
    const real eps = (real)1e-30;
    const real gl = (real)gain_limit;
    for (size_t k = 0; k < (size_t)w * h; ++k) {
        real oh = ref->h[k], ov = ref->v[k], od = ref->d[k];
        real th = dis->h[k], tv = dis->v[k], td = dis->d[k];
        real kh = clamp01(th / (oh + eps)), kv = clamp01(tv / (ov + eps)), kd = clamp01(td / (od + eps));
        real tmph = kh * oh, tmpv = kv * ov, tmpd = kd * od;
        real ot_dp = oh * th + ov * tv;
        real o_mag_sq = oh * oh + ov * ov;
        real t_mag_sq = th * th + tv * tv;
        int angle_flag = (ot_dp >= 0) && (ot_dp * ot_dp >= cos_1deg_sq * o_mag_sq * t_mag_sq);
        if (angle_flag) {
            if (tmph > 0) { real x = tmph * gl; tmph = x < th ? x : th; }
            else if (tmph < 0) { real x = tmph * gl; tmph = x > th ? x : th; }
            if (tmpv > 0) { real x = tmpv * gl; tmpv = x < tv ? x : tv; }
            else if (tmpv < 0) { real x = tmpv * gl; tmpv = x > tv ? x : tv; }
            if (tmpd > 0) { real x = tmpd * gl; tmpd = x < td ? x : td; }
            else if (tmpd < 0) { real x = tmpd * gl; tmpd = x > td ? x : td; }
        }
        r->h[k] = tmph; r->v[k] = tmpv; r->d[k] = tmpd;
        a->h[k] = th - tmph; a->v[k] = tv - tmpv; a->d[k] = td - tmpd;
    }
}

static real real_cbrt_pow(real x) { return sizeof(real) == 4 ? (real)powf((float)x, 1.0f / 3.0f) : (real)pow((double)x, 1.0 / 3.0); }
static real real_abs(real x) { return x < 0 ? -x : x; }

static void adm_window(int w, int h, double border_factor, int *left, int *top, int *right, int *bottom)
{
    *left = (int)(w * border_factor - 0.5);
    *top = (int)(h * border_factor - 0.5);
    *right = w - *left;
    *bottom = h - *top;
}

static real adm_csf_den_scale(const band_t *src, int scale, int w, int h, double border_factor)
{
    real rf[3];
    rf[0] = rf[1] = (real)1 / dwt_quant_step(scale, 1, 3.0, 1080);
    rf[2] = (real)1 / dwt_quant_step(scale, 2, 3.0, 1080);
    const real *bands[3] = { src->h, src->v, src->d };
    int left, top, right, bottom;
    adm_window(w, h, border_factor, &left, &top, &right, &bottom);
    real total = 0;
    for (int t = 0; t < 3; ++t) {
        real accum = 0;
        for (int i = top; i < bottom; ++i) {
            real inner = 0;
            for (int j = left; j < right; ++j) {
                real val = real_abs(bands[t][(size_t)i * w + j]) * rf[t];
                inner += val * val * val;
            }
            accum += inner;
        }
        total += real_cbrt_pow(accum) + real_cbrt_pow((real)((bottom - top) * (right - left)) / (real)32);
    }
    return total;
}

/* adm_csf_s on the additive image + adm_cm_s on the restored image */
static real adm_cm(const band_t *r, const band_t *a, int scale, int w, int h, double border_factor)
{
    real rf[3];
    rf[0] = rf[1] = (real)1 / dwt_quant_step(scale, 1, 3.0, 1080);
    rf[2] = (real)1 / dwt_quant_step(scale, 2, 3.0, 1080);
    const real *rb[3] = { r->h, r->v, r->d };
    const real *ab[3] = { a->h, a->v, a->d };
    size_t n = (size_t)w * h;
    real *flt = (real *)malloc(sizeof(real) * n * 3); /* csf_f = |csf_a| / 30 */
    for (int t = 0; t < 3; ++t)
        for (size_t k = 0; k < n; ++k) {
            real dst_val = rf[t] * ab[t][k];
            flt[t * n + k] = (real)(1.0f / 30.0f) * real_abs(dst_val);
        }
    int left, top, right, bottom;
    adm_window(w, h, border_factor, &left, &top, &right, &bottom);
    real accum[3] = { 0, 0, 0 };
    for (int i = top; i < bottom; ++i) {
        real inner[3] = { 0, 0, 0 };
        for (int j = left; j < right; ++j) {
            real thr = 0;
            for (int t = 0; t < 3; ++t) {
                real sum1 = 0;
                for (int fi = -1; fi <= 1; ++fi) {
                    int ii = mirror(i + fi, h);
                    for (int fj = -1; fj <= 1; ++fj) {
                        int jj = mirror(j + fj, w);
                        sum1 += flt[t * n + (size_t)ii * w + jj];
                    }
                }
                sum1 += flt[t * n + (size_t)i * w + j]; /* centre weight 1/15 = 2/30 */
                thr += sum1;
            }
            for (int t = 0; t < 3; ++t) {
                real x = real_abs(rb[t][(size_t)i * w + j] * rf[t]) - thr;
                x = x < 0 ? 0 : x;
                inner[t] += x * x * x;
            }
        }
        for (int t = 0; t < 3; ++t) accum[t] += inner[t];
    }
    free(flt);
    real total = 0;
    for (int t = 0; t < 3; ++t)
        total += real_cbrt_pow(accum[t]) + real_cbrt_pow((real)((bottom - top) * (right - left)) / (real)32);
    return total;
}

/* out[0..3] = num per scale, out[4..7] = den per scale */
ORC_EXPORT int orc_adm(const real *ref, const real *dis, int w, int h, double adm_enhn_gain_limit,
                       double *out)
{
    const double border_factor = 0.1;
    size_t n0 = (size_t)((w + 1) / 2) * ((h + 1) / 2);
    real *buf = (real *)malloc(sizeof(real) * n0 * 16);
    real *cur_ref = (real *)malloc(sizeof(real) * (size_t)w * h);
    real *cur_dis = (real *)malloc(sizeof(real) * (size_t)w * h);
    if (!buf || !cur_ref || !cur_dis) return -1;
    memcpy(cur_ref, ref, sizeof(real) * (size_t)w * h);
    memcpy(cur_dis, dis, sizeof(real) * (size_t)w * h);
    band_t rd = { buf, buf + n0, buf + 2 * n0, buf + 3 * n0 };
    band_t dd = { buf + 4 * n0, buf + 5 * n0, buf + 6 * n0, buf + 7 * n0 };
    band_t dr = { buf + 8 * n0, buf + 9 * n0, buf + 10 * n0, buf + 11 * n0 };
    band_t da = { buf + 12 * n0, buf + 13 * n0, buf + 14 * n0, buf + 15 * n0 };
    for (int scale = 0; scale < 4; ++scale) {
        adm_dwt2(cur_ref, &rd, w, h);
        adm_dwt2(cur_dis, &dd, w, h);
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        adm_decouple(&rd, &dd, &dr, &da, w, h, adm_enhn_gain_limit);
        real den_scale = adm_csf_den_scale(&rd, scale, w, h, border_factor);
        real num_scale = adm_cm(&dr, &da, scale, w, h, border_factor);
        out[scale] = (double)num_scale;
        out[4 + scale] = (double)den_scale;
        memcpy(cur_ref, rd.a, sizeof(real) * (size_t)w * h);
        memcpy(cur_dis, dd.a, sizeof(real) * (size_t)w * h);
    }
    free(buf); free(cur_ref); free(cur_dis);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * motion  (libvmaf float_motion.c extract, motion.c compute_motion / vmaf_image_sad_c,
 *          convolution.c convolution_f32_c_s with FILTER_5_s)
 * ---------------------------------------------------------------------------------------- */

ORC_EXPORT void orc_motion_blur(const real *ref, int w, int h, real *blur)
{
    real f[8];
    gaussian_taps(5, f);
    /* convolution_f32_c_s: vertical pass over the whole plane into tmp, then horizontal */
    real *tmp = (real *)malloc(sizeof(real) * (size_t)w * h);
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            real accum = 0;
            for (int k = 0; k < 5; ++k) accum += f[k] * ref[(size_t)mirror(i - 2 + k, h) * w + j];
            tmp[(size_t)i * w + j] = accum;
        }
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            real accum = 0;
            for (int k = 0; k < 5; ++k) accum += f[k] * tmp[(size_t)i * w + mirror(j - 2 + k, w)];
            blur[(size_t)i * w + j] = accum;
        }
    free(tmp);
}

ORC_EXPORT double orc_motion_sad(const real *blur_a, const real *blur_b, int w, int h)
{
    real accum = 0;
    for (int i = 0; i < h; ++i) {
        real line = 0;
        for (int j = 0; j < w; ++j) line += real_abs(blur_a[(size_t)i * w + j] - blur_b[(size_t)i * w + j]);
        accum += line;
    }
    return (double)(real)(accum / (real)((double)w * h));
}

/* ------------------------------------------------------------------------------------------
 * FFmpeg psnr filter (vf_psnr.c): exact integer SSE per plane
 * ---------------------------------------------------------------------------------------- */
ORC_EXPORT uint64_t orc_sse_plane(const void *a, int a_stride, const void *b, int b_stride, int bpc,
                                  int w, int h)
{
    uint64_t sse = 0;
    for (int i = 0; i < h; ++i) {
        if (bpc <= 8) {
            const uint8_t *pa = (const uint8_t *)a + (size_t)i * a_stride;
            const uint8_t *pb = (const uint8_t *)b + (size_t)i * b_stride;
            for (int j = 0; j < w; ++j) { int d = (int)pa[j] - (int)pb[j]; sse += (uint64_t)(d * d); }
        } else {
            const uint16_t *pa = (const uint16_t *)((const uint8_t *)a + (size_t)i * a_stride);
            const uint16_t *pb = (const uint16_t *)((const uint8_t *)b + (size_t)i * b_stride);
            for (int j = 0; j < w; ++j) { int64_t d = (int64_t)pa[j] - (int64_t)pb[j]; sse += (uint64_t)(d * d); }
        }
    }
    return sse;
}

/* ------------------------------------------------------------------------------------------
 * FFmpeg ssim filter (vf_ssim.c ssim_4x4xn_{8,16}bit, ssim_end1/ssim_end1x, ssim_plane):
 * 4x4 block sums, 8x8 windows on a 4-px grid, integer sums -> float ratio per window.
 * `a` is the main (distorted) input, `b` the reference, as in the filter.
 * ---------------------------------------------------------------------------------------- */
static float ssim_end1x(int64_t s1, int64_t s2, int64_t ss, int64_t s12, int max)
{
    int64_t ssim_c1 = (int64_t)(.01 * .01 * max * max * 64 + .5);
    int64_t ssim_c2 = (int64_t)(.03 * .03 * max * max * 64 * 63 + .5);
    int64_t vars = ss * 64 - s1 * s1 - s2 * s2;
    int64_t covar = s12 * 64 - s1 * s2;
    return (float)(2 * s1 * s2 + ssim_c1) * (float)(2 * covar + ssim_c2) /
           ((float)(s1 * s1 + s2 * s2 + ssim_c1) * (float)(vars + ssim_c2));
}

ORC_EXPORT double orc_ssim_plane(const void *a, int a_stride, const void *b, int b_stride, int bpc,
                                 int w, int h)
{
    int bw = w >> 2, bh = h >> 2;
    int max = (1 << bpc) - 1;
    if (bw < 2 || bh < 2) return 0.0;
    int64_t *sums = (int64_t *)malloc(sizeof(int64_t) * 4 * (size_t)bw * bh);
    for (int by = 0; by < bh; ++by)
        for (int bx = 0; bx < bw; ++bx) {
            int64_t s1 = 0, s2 = 0, ss = 0, s12 = 0;
            for (int y = 0; y < 4; ++y)
                for (int x = 0; x < 4; ++x) {
                    int64_t pa, pb;
                    if (bpc <= 8) {
                        pa = ((const uint8_t *)a)[(size_t)(4 * by + y) * a_stride + 4 * bx + x];
                        pb = ((const uint8_t *)b)[(size_t)(4 * by + y) * b_stride + 4 * bx + x];
                    } else {
                        pa = ((const uint16_t *)((const uint8_t *)a + (size_t)(4 * by + y) * a_stride))[4 * bx + x];
                        pb = ((const uint16_t *)((const uint8_t *)b + (size_t)(4 * by + y) * b_stride))[4 * bx + x];
                    }
                    s1 += pa; s2 += pb; ss += pa * pa; ss += pb * pb; s12 += pa * pb;
                }
            int64_t *s = sums + 4 * ((size_t)by * bw + bx);
            s[0] = s1; s[1] = s2; s[2] = ss; s[3] = s12;
        }
    double ssim = 0.0;
    for (int y = 1; y < bh; ++y) {
        double line = 0.0; /* ssim_endn_*: double accumulation of float window scores */
        for (int x = 0; x < bw - 1; ++x) {
            const int64_t *p00 = sums + 4 * ((size_t)(y - 1) * bw + x), *p01 = p00 + 4;
            const int64_t *p10 = sums + 4 * ((size_t)y * bw + x), *p11 = p10 + 4;
            line += ssim_end1x(p00[0] + p01[0] + p10[0] + p11[0], p00[1] + p01[1] + p10[1] + p11[1],
                               p00[2] + p01[2] + p10[2] + p11[2], p00[3] + p01[3] + p10[3] + p11[3], max);
        }
        ssim += line;
    }
    free(sums);
    return ssim / ((double)(bh - 1) * (bw - 1));
}

/* ------------------------------------------------------------------------------------------
 * one call per frame pair: everything the hot path produces for frame i
 *   feat[0..3]  vif num   feat[4..7]  vif den   feat[8..11] adm num   feat[12..15] adm den
 *   feat[16]    motion (SAD mean vs prev_blur, 0 when prev_blur == NULL)
 * blur_out (w*h reals) receives this frame's blurred reference for the next call.
 * ---------------------------------------------------------------------------------------- */
ORC_EXPORT int orc_frame_features2(const void *ref_luma, const void *dis_luma, int stride_bytes, int bpc,
                                   int w, int h, double vif_gain_limit, double adm_gain_limit,
                                   const real *prev_blur, real *blur_out, double *feat, int vif_border101)
{
    size_t n = (size_t)w * h;
    real *ref = (real *)malloc(sizeof(real) * n);
    real *dis = (real *)malloc(sizeof(real) * n);
    if (!ref || !dis) return -1;
    orc_picture_copy(ref_luma, stride_bytes, bpc, w, h, ref);
    orc_picture_copy(dis_luma, stride_bytes, bpc, w, h, dis);
    int rc = orc_vif2(ref, dis, w, h, vif_gain_limit, vif_border101, feat);
    if (!rc) rc = orc_adm(ref, dis, w, h, adm_gain_limit, feat + 8);
    orc_motion_blur(ref, w, h, blur_out);
    feat[16] = prev_blur ? orc_motion_sad(prev_blur, blur_out, w, h) : 0.0;
    free(ref); free(dis);
    return rc;
}

ORC_EXPORT int orc_frame_features(const void *ref_luma, const void *dis_luma, int stride_bytes, int bpc,
                                  int w, int h, double vif_gain_limit, double adm_gain_limit,
                                  const real *prev_blur, real *blur_out, double *feat)
{
    return orc_frame_features2(ref_luma, dis_luma, stride_bytes, bpc, w, h, vif_gain_limit, adm_gain_limit,
                               prev_blur, blur_out, feat, 0);
}

ORC_EXPORT int orc_real_size(void) { return (int)sizeof(real); }
