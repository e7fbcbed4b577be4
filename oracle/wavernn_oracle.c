/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, IEEE fp32) of the reference WaveRNN mel->wav hot
 * path, used ONLY as the parity checker by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  The shipped path
 * (tacotronv2_wavernn_chinese_amd/csrc/) never links, loads or calls this
 * file.
 *
 * PARITY PIN: the reference ships no tests / golden vectors, so
 * this restatement is pinned against outputs of the reference itself:
 * oracle/make_golden.py runs the unmodified reference generate() in the build
 * container and freezes tests/golden/ (npz files); tests/test_oracle_golden.py checks
 * this file against those fixtures (labels bit-identical, logits to 2e-5).
 *
 * Each function cites the reference lines it follows; paths are relative to
 * /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define WO_MODE_RAW 0
#define WO_MODE_MOL 1

/* noise modes for the RAW sampler */
#define WO_NOISE_EXPO 0    /* injected Exp(1) draws, reference-exact semantics */
#define WO_NOISE_ARGMAX 2  /* q == 1: greedy */

typedef struct {
    int rnn_dims;      /* 512 */
    int fc_dims;       /* 512 */
    int feat_dims;     /* 80  */
    int aux_dims;      /* res_out_dims / 4 = 32 */
    int compute_dims;  /* 128 */
    int res_out_dims;  /* 128 */
    int res_blocks;    /* 10  */
    int pad;           /* 2   */
    int n_up;          /* 3   */
    int up[4];         /* 5,5,11 */
    int n_classes;     /* 1024 RAW / 30 MOL */
    int mode;          /* WO_MODE_* */
} wo_dims;

/* Weight pointers, all row-major exactly as in the reference state_dict
 * (wavernn/models/fatchord_version.py:93-129). */
typedef struct {
    const float *conv_in_w;                       /* (C, F, 2*pad+1) */
    const float *bn0_w, *bn0_b, *bn0_m, *bn0_v;   /* (C) */
    const float *res_conv1_w;                     /* (nblk, C, C) */
    const float *res_conv2_w;                     /* (nblk, C, C) */
    const float *res_bn1;                         /* (nblk, 4, C): w,b,mean,var */
    const float *res_bn2;                         /* (nblk, 4, C) */
    const float *conv_out_w, *conv_out_b;         /* (R, C), (R) */
    const float *up_w[4];                         /* (2*s+1) each */
    const float *I_w, *I_b;                       /* (H, 1+F+A), (H) */
    const float *r1_wih, *r1_whh, *r1_bih, *r1_bhh; /* (3H,H) (3H,H) (3H) (3H) */
    const float *r2_wih, *r2_whh, *r2_bih, *r2_bhh; /* (3H,H+A) (3H,H) ... */
    const float *fc1_w, *fc1_b;                   /* (FC, H+A) */
    const float *fc2_w, *fc2_b;                   /* (FC, FC+A) */
    const float *fc3_w, *fc3_b;                   /* (n_classes, FC) */
} wo_weights;

static const float BN_EPS = 1e-5f; /* torch.nn.BatchNorm1d default */

/* ---------------------------------------------------------------- prologue */

static void bn_eval(float *x, int C, int T, const float *w, const float *b,
                    const float *m, const float *v, int relu) {
    /* nn.BatchNorm1d in eval(): generate() calls self.eval() (:170) */
    for (int c = 0; c < C; ++c) {
        float inv = 1.0f / sqrtf(v[c] + BN_EPS);
        for (int t = 0; t < T; ++t) {
            float y = (x[(size_t)c * T + t] - m[c]) * inv * w[c] + b[c];
            if (relu && y < 0.0f) y = 0.0f;
            x[(size_t)c * T + t] = y;
        }
    }
}

static void conv1x1(const float *w, const float *x, float *y, int Co, int Ci, int T) {
    /* nn.Conv1d(kernel_size=1, bias=False): ResBlock :15-16 */
    for (int co = 0; co < Co; ++co)
        for (int t = 0; t < T; ++t) {
            float acc = 0.0f;
            for (int ci = 0; ci < Ci; ++ci) acc += w[(size_t)co * Ci + ci] * x[(size_t)ci * T + t];
            y[(size_t)co * T + t] = acc;
        }
}

/* pad_tensor(side='both') (:281-291) + MelResNet.forward (:42-48) +
 * ResBlock.forward (:21-28).
 * mels (B, F, T) -> aux (B, T, R)   [time-major like :89's transpose] */
int wo_resnet(const wo_dims *d, const wo_weights *w, const float *mels, int B, int T, float *aux) {
    const int F = d->feat_dims, C = d->compute_dims, R = d->res_out_dims, P = d->pad;
    const int K = 2 * P + 1, Tp = T + 2 * P;
    float *xp = (float *)calloc((size_t)F * Tp, sizeof(float));
    float *x = (float *)malloc((size_t)C * T * sizeof(float));
    float *y = (float *)malloc((size_t)C * T * sizeof(float));
    float *o = (float *)malloc((size_t)R * T * sizeof(float));
    float *z = (float *)malloc((size_t)C * T * sizeof(float));
    if (!xp || !x || !y || !o || !z) return -1;
    for (int b = 0; b < B; ++b) {
        memset(xp, 0, (size_t)F * Tp * sizeof(float));
        for (int f = 0; f < F; ++f)
            memcpy(xp + (size_t)f * Tp + P, mels + ((size_t)b * F + f) * T, (size_t)T * sizeof(float));
        /* conv_in: Conv1d(F -> C, k=2*pad+1, no padding, no bias) :34 */
        for (int c = 0; c < C; ++c)
            for (int t = 0; t < T; ++t) {
                float acc = 0.0f;
                for (int f = 0; f < F; ++f)
                    for (int k = 0; k < K; ++k)
                        acc += w->conv_in_w[((size_t)c * F + f) * K + k] * xp[(size_t)f * Tp + t + k];
                x[(size_t)c * T + t] = acc;
            }
        bn_eval(x, C, T, w->bn0_w, w->bn0_b, w->bn0_m, w->bn0_v, 1);
        for (int l = 0; l < d->res_blocks; ++l) {
            const float *b1 = w->res_bn1 + (size_t)l * 4 * C, *b2 = w->res_bn2 + (size_t)l * 4 * C;
            conv1x1(w->res_conv1_w + (size_t)l * C * C, x, y, C, C, T);
            bn_eval(y, C, T, b1, b1 + C, b1 + 2 * C, b1 + 3 * C, 1);
            conv1x1(w->res_conv2_w + (size_t)l * C * C, y, z, C, C, T);
            bn_eval(z, C, T, b2, b2 + C, b2 + 2 * C, b2 + 3 * C, 0);
            for (size_t i = 0; i < (size_t)C * T; ++i) x[i] = z[i] + x[i]; /* x + residual :28 */
        }
        /* conv_out: Conv1d(C -> R, k=1) with bias :40 */
        for (int r = 0; r < R; ++r)
            for (int t = 0; t < T; ++t) {
                float acc = 0.0f;
                for (int c = 0; c < C; ++c) acc += w->conv_out_w[(size_t)r * C + c] * x[(size_t)c * T + t];
                o[(size_t)r * T + t] = acc + w->conv_out_b[r];
            }
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < R; ++r) aux[((size_t)b * T + t) * R + r] = o[(size_t)r * T + t];
    }
    free(xp); free(x); free(y); free(o); free(z);
    return 0;
}

/* UpsampleNetwork.forward mel branch (:85-88): for each scale s:
 * Stretch2d(s,1) (:57-61, nearest repeat) then Conv2d(1,1,(1,2s+1),
 * padding=(0,s), bias=False) (:73-80); crop indent = pad*prod(scales) (:88);
 * transpose to time-major (:89).
 * mels (B, F, T) -> up (B, T*hop, F) */
int wo_upsample(const wo_dims *d, const wo_weights *w, const float *mels, int B, int T, float *up) {
    const int F = d->feat_dims, P = d->pad;
    int hop = 1;
    for (int i = 0; i < d->n_up; ++i) hop *= d->up[i];
    const int Tp = T + 2 * P;
    const size_t Lp = (size_t)Tp * hop, L = (size_t)T * hop;
    const size_t indent = (size_t)P * hop;
    float *a = (float *)malloc(Lp * sizeof(float));
    float *c = (float *)malloc(Lp * sizeof(float));
    if (!a || !c) return -1;
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f) {
            size_t n = (size_t)Tp;
            for (size_t i = 0; i < n; ++i) a[i] = 0.0f;
            memcpy(a + P, mels + ((size_t)b * F + f) * T, (size_t)T * sizeof(float));
            for (int li = 0; li < d->n_up; ++li) {
                const int s = d->up[li], K = 2 * s + 1;
                const size_t n2 = n * (size_t)s;
                /* stretch into c */
                for (size_t i = 0; i < n; ++i)
                    for (int r = 0; r < s; ++r) c[i * s + r] = a[i];
                /* cross-correlation with zero padding s, into a */
                for (size_t i = 0; i < n2; ++i) {
                    float acc = 0.0f;
                    for (int k = 0; k < K; ++k) {
                        long j = (long)i + k - s;
                        if (j >= 0 && (size_t)j < n2) acc += w->up_w[li][k] * c[j];
                    }
                    a[i] = acc;
                }
                n = n2;
            }
            for (size_t t = 0; t < L; ++t) up[((size_t)b * L + t) * F + f] = a[indent + t];
        }
    free(a); free(c);
    return 0;
}

/* resnet_stretch = Stretch2d(total_scale, 1) (:70,:84): aux (B,T,R) -> (B,T*hop,R) */
int wo_stretch_aux(const float *aux, int B, int T, int R, int hop, float *out) {
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < hop; ++r)
                memcpy(out + (((size_t)b * T + t) * hop + r) * R, aux + ((size_t)b * T + t) * R,
                       (size_t)R * sizeof(float));
    return 0;
}

/* fold_with_overlap (:293-340).  x (1, total_len, feat) -> folded
 * (num_folds, target+2*overlap, feat).  Returns num_folds; if `folded` is
 * NULL only the count is computed. */
int wo_fold(const float *x, long total_len, int feat, int target, int overlap, float *folded) {
    long num_folds = (total_len - overlap) / (target + overlap);
    long extended = num_folds * (overlap + target) + overlap;
    long remaining = total_len - extended;
    if (remaining != 0) num_folds += 1; /* zero-pad 'after' :327-330 */
    if (!folded) return (int)num_folds;
    const long flen = target + 2L * overlap;
    for (long i = 0; i < num_folds; ++i) {
        long start = i * (target + overlap);
        for (long t = 0; t < flen; ++t) {
            float *dst = folded + ((size_t)i * flen + t) * feat;
            long src = start + t;
            if (src < total_len) memcpy(dst, x + (size_t)src * feat, (size_t)feat * sizeof(float));
            else memset(dst, 0, (size_t)feat * sizeof(float));
        }
    }
    return (int)num_folds;
}

/* --------------------------------------------------------------- hot loop */

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* y[r] = b[r] + sum_k W[r][k] * x[k] */
static void matvec(const float *W, const float *b, const float *x, float *y, int rows, int cols) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float *wr = W + (size_t)r * cols;
        float acc = 0.0f;
#ifdef WO_FAST /* timing build only: vectorised reduction (reassociates the sum) */
#pragma omp simd reduction(+ : acc)
#endif
        for (int k = 0; k < cols; ++k) acc += wr[k] * x[k];
        y[r] = acc + b[r];
    }
}

/* nn.GRUCell (gate row order [r; z; n], get_gru_cell :273-279):
 *   r = s(Wir x + bir + Whr h + bhr); z = s(Wiz x + biz + Whz h + bhz)
 *   n = tanh(Win x + bin + r * (Whn h + bhn)); h' = (1-z)*n + z*h           */
static void gru_cell(const float *wih, const float *whh, const float *bih, const float *bhh,
                     const float *x, int in_dim, float *h, int H, float *gi, float *gh) {
    matvec(wih, bih, x, gi, 3 * H, in_dim);
    matvec(whh, bhh, h, gh, 3 * H, H);
    for (int j = 0; j < H; ++j) {
        float r = sigmoidf_(gi[j] + gh[j]);
        float z = sigmoidf_(gi[H + j] + gh[H + j]);
        float n = tanhf(gi[2 * H + j] + r * gh[2 * H + j]);
        h[j] = (1.0f - z) * n + z * h[j];
    }
}

/*
 * The per-sample loop of WaveRNN.generate (:194-241).
 *   cond_m (B, L, F), cond_a (B, L, R)  -- full-rate conditioning rows exactly
 *                                          as generate() indexes them (:203-206)
 *   noise_mode / noise1 / noise2:
 *     RAW  WO_NOISE_EXPO   noise1 = q (L, B, n_classes) Exp(1) draws: the n=1
 *                          path of torch.multinomial = argmax(p / q) (:231-235)
 *          WO_NOISE_ARGMAX q == 1
 *     MOL  noise1 = u_mix (L, B, 10), noise2 = u_log (L, B)
 *          (wavernn/utils/distribution.py:106,118)
 *   x_forced (L, B) or NULL: teacher forcing -- the value fed back as x_t is
 *          x_forced[t] instead of the sample just drawn.
 *   labels (L, B) int32  [RAW class index | MOL mixture index]
 *   samples (L, B) fp32  value appended to `output` (:227,:236)
 *   logits_out (L, B, n_classes) or NULL
 *   margin_out (L, B) or NULL: relative gap between the winning and runner-up
 *          race scores (how close the step was to a tie), runner_out (L,B)
 */
static int loop_row(const wo_dims *d, const wo_weights *w, const float *cond_m, const float *cond_a, int B, int b,
                    long L, int noise_mode, const float *noise1, const float *noise2, const float *x_forced,
                    int32_t *labels, float *samples, float *logits_out, float *margin_out,
                    int32_t *runner_out) {
    const int H = d->rnn_dims, FC = d->fc_dims, F = d->feat_dims, A = d->aux_dims;
    const int R = d->res_out_dims, NC = d->n_classes;
    const int in_I = 1 + F + A;
    /* state of row b only: the rows of a batch never interact (:194-196 zero state per row, every op is row-wise) */
    float *h1 = (float *)calloc((size_t)H, sizeof(float));   /* :194 */
    float *h2 = (float *)calloc((size_t)H, sizeof(float));   /* :195 */
    float xprev = 0.0f;                                       /* :196 */
    float *cat = (float *)malloc((size_t)((H > FC ? H : FC) + A + in_I) * sizeof(float));   /* holds [x | a] of every layer */
    float *x = (float *)malloc((size_t)H * sizeof(float));
    float *gi = (float *)malloc((size_t)3 * H * sizeof(float));
    float *gh = (float *)malloc((size_t)3 * H * sizeof(float));
    float *f1 = (float *)malloc((size_t)FC * sizeof(float));
    float *f2 = (float *)malloc((size_t)FC * sizeof(float));
    float *lg = (float *)malloc((size_t)NC * sizeof(float));
    float *p = (float *)malloc((size_t)NC * sizeof(float));
    if (!h1 || !h2 || !cat || !x || !gi || !gh || !f1 || !f2 || !lg || !p) return -1;

    for (long t = 0; t < L; ++t) {
        {
            const float *m_t = cond_m + ((size_t)b * L + t) * F;
            const float *a_t = cond_a + ((size_t)b * L + t) * R;
            float *hb1 = h1, *hb2 = h2;
            /* x = I(cat[x, m_t, a1_t]) :208-209 */
            cat[0] = xprev;
            memcpy(cat + 1, m_t, (size_t)F * sizeof(float));
            memcpy(cat + 1 + F, a_t, (size_t)A * sizeof(float));
            matvec(w->I_w, w->I_b, cat, x, H, in_I);
            /* h1 = rnn1(x, h1); x = x + h1 :210-212 */
            gru_cell(w->r1_wih, w->r1_whh, w->r1_bih, w->r1_bhh, x, H, hb1, H, gi, gh);
            for (int j = 0; j < H; ++j) x[j] = x[j] + hb1[j];
            /* h2 = rnn2(cat[x, a2_t], h2); x = x + h2 :213-216 */
            memcpy(cat, x, (size_t)H * sizeof(float));
            memcpy(cat + H, a_t + A, (size_t)A * sizeof(float));
            gru_cell(w->r2_wih, w->r2_whh, w->r2_bih, w->r2_bhh, cat, H + A, hb2, H, gi, gh);
            for (int j = 0; j < H; ++j) x[j] = x[j] + hb2[j];
            /* x = relu(fc1(cat[x, a3_t])) :217-218 */
            memcpy(cat, x, (size_t)H * sizeof(float));
            memcpy(cat + H, a_t + 2 * A, (size_t)A * sizeof(float));
            matvec(w->fc1_w, w->fc1_b, cat, f1, FC, H + A);
            for (int j = 0; j < FC; ++j) f1[j] = f1[j] > 0.0f ? f1[j] : 0.0f;
            /* x = relu(fc2(cat[x, a4_t])) :220-221 */
            memcpy(cat, f1, (size_t)FC * sizeof(float));
            memcpy(cat + FC, a_t + 3 * A, (size_t)A * sizeof(float));
            matvec(w->fc2_w, w->fc2_b, cat, f2, FC, FC + A);
            for (int j = 0; j < FC; ++j) f2[j] = f2[j] > 0.0f ? f2[j] : 0.0f;
            /* logits = fc3(x) :223 */
            matvec(w->fc3_w, w->fc3_b, f2, lg, NC, FC);
            if (logits_out) memcpy(logits_out + ((size_t)t * B + b) * NC, lg, (size_t)NC * sizeof(float));

            float sample;
            int32_t label, runner = -1;
            float margin = 0.0f;
            if (d->mode == WO_MODE_RAW) {
                /* posterior = softmax(logits) :232; Categorical(probs) renormalises
                 * probs / probs.sum(-1); sample() -> multinomial(p,1,True) ==
                 * argmax(p / q), q ~ Exp(1) :233-235 */
                float mx = lg[0];
                for (int k = 1; k < NC; ++k) mx = lg[k] > mx ? lg[k] : mx;
                float sum = 0.0f;
                for (int k = 0; k < NC; ++k) { p[k] = expf(lg[k] - mx); sum += p[k]; }
                float sum2 = 0.0f;
                for (int k = 0; k < NC; ++k) { p[k] = p[k] / sum; sum2 += p[k]; }
                const float *q = (noise_mode == WO_NOISE_EXPO) ? noise1 + ((size_t)t * B + b) * NC : NULL;
                float best = -1.0f, second = -1.0f;
                int bi = 0, si = -1;
                for (int k = 0; k < NC; ++k) {
                    float v = p[k] / sum2;
                    if (q) v = v / q[k];
                    if (v > best) { second = best; si = bi; best = v; bi = k; }
                    else if (v > second) { second = v; si = k; }
                }
                label = bi; runner = si;
                margin = best > 0.0f ? (best - second) / best : 0.0f;
                /* sample = 2 * k / (n_classes - 1.) - 1. :235 */
                sample = 2.0f * (float)bi / ((float)NC - 1.0f) - 1.0f;
            } else {
                /* sample_from_discretized_mix_logistic, distribution.py:87-123,
                 * called with y = logits (1, 30, B) :226 */
                const int nr = NC / 3;
                const float *u1 = noise1 + ((size_t)t * B + b) * nr;
                const float u2 = noise2[(size_t)t * B + b];
                float best = -INFINITY, second = -INFINITY;
                int bi = 0, si = -1;
                for (int k = 0; k < nr; ++k) {
                    float v = lg[k] - logf(-logf(u1[k]));  /* :107 */
                    if (v > best) { second = best; si = bi; best = v; bi = k; }
                    else if (v > second) { second = v; si = k; }
                }
                label = bi; runner = si;
                margin = best - second;
                float mean = lg[nr + bi];                 /* one-hot select :113 */
                float ls = lg[2 * nr + bi];               /* :114-115 */
                const float lsmin = (float)log(1e-14);
                if (ls < lsmin) ls = lsmin;
                float xs = mean + expf(ls) * (logf(u2) - logf(1.0f - u2));  /* :119 */
                if (xs < -1.0f) xs = -1.0f;               /* :121 */
                if (xs > 1.0f) xs = 1.0f;
                sample = xs;
            }
            labels[(size_t)t * B + b] = label;
            samples[(size_t)t * B + b] = sample;
            if (margin_out) margin_out[(size_t)t * B + b] = margin;
            if (runner_out) runner_out[(size_t)t * B + b] = runner;
            xprev = x_forced ? x_forced[(size_t)t * B + b] : sample;  /* :228,:237 */
        }
    }
    free(h1); free(h2); free(cat); free(x); free(gi); free(gh);
    free(f1); free(f2); free(lg); free(p);
    return 0;
}

/* All B rows.  The rows are independent, so the order in which (row, step) pairs are evaluated is free: with few rows each
 * row is walked with the matvecs split over the OpenMP team (as a B = 1 call always was); with >= 4 rows the ROWS are split
 * over the team and every row runs its matvecs on one thread (the nested `parallel for` inside matvec is then serial: nested
 * parallelism is off by default) -- no fork/join per layer, and each core streams the weights from its own cache.  The
 * arithmetic of a row is the same instruction sequence either way. */
int wo_loop(const wo_dims *d, const wo_weights *w, const float *cond_m, const float *cond_a, int B,
            long L, int noise_mode, const float *noise1, const float *noise2, const float *x_forced,
            int32_t *labels, float *samples, float *logits_out, float *margin_out,
            int32_t *runner_out) {
    int rc = 0;
#ifdef _OPENMP
    if (B >= 4 && omp_get_max_threads() > 1) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            const int r = loop_row(d, w, cond_m, cond_a, B, b, L, noise_mode, noise1, noise2, x_forced, labels, samples, logits_out,
                                   margin_out, runner_out);
            if (r) {
#pragma omp atomic write
                rc = r;
            }
        }
        return rc;
    }
#endif
    for (int b = 0; b < B && !rc; ++b)
        rc = loop_row(d, w, cond_m, cond_a, B, b, L, noise_mode, noise1, noise2, x_forced, labels, samples, logits_out, margin_out,
                      runner_out);
    return rc;
}

int wo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void wo_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * A12 (secondary): wavernn/models/deepmind_version.py -- unconditioned dual-softmax (coarse/fine)
 * WaveRNN, generate(seq_len) :75-165.  Weights row-major as in its state_dict:
 *   R (3H, H) no bias :16 | O1 (S,S)+b, O2 (Q,S)+b, O3 (S,S)+b, O4 (Q,S)+b :19-22 |
 *   I_coarse (3S, 2) :25, I_fine (3S, 3) :26 | bias_u, bias_r, bias_e (H) :29-31;  S = H/2.
 * noise: q (seq_len, 2, Q) Exp(1) draws -- [t][0] for the coarse Categorical.sample(), [t][1] for the fine one
 * (torch.multinomial n=1 == argmax(p/q)); NULL = greedy.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int hidden;  /* 896 */
    int quant;   /* 256 */
    const float *R, *O1w, *O1b, *O2w, *O2b, *O3w, *O3b, *O4w, *O4b, *Ic, *If, *bu, *br, *be;
} wo_dm_model;

static int dm_sample(const float *logits, int Q, const float *q, float *p, float *margin, int32_t *runner) {
    float mx = logits[0];
    for (int k = 1; k < Q; ++k) mx = logits[k] > mx ? logits[k] : mx;
    float sum = 0.0f;
    for (int k = 0; k < Q; ++k) { p[k] = expf(logits[k] - mx); sum += p[k]; }   /* F.softmax :129,:149 */
    float sum2 = 0.0f;
    for (int k = 0; k < Q; ++k) { p[k] = p[k] / sum; sum2 += p[k]; }            /* Categorical renormalises */
    float best = -1.0f, second = -1.0f;
    int bi = 0, si = -1;
    for (int k = 0; k < Q; ++k) {
        float v = p[k] / sum2;
        if (q) v = v / q[k];
        if (v > best) { second = best; si = bi; best = v; bi = k; }
        else if (v > second) { second = v; si = k; }
    }
    if (margin) *margin = best > 0.0f ? (best - second) / best : 0.0f;
    if (runner) *runner = si;
    return bi;
}

int wo_dm_generate(const wo_dm_model *m, long seq_len, const float *noise, int32_t *coarse, int32_t *fine,
                   float *margin_out /* (seq_len,2) or NULL */, int32_t *runner_out /* (seq_len,2) or NULL */) {
    const int H = m->hidden, S = H / 2, Q = m->quant;
    float *h = (float *)calloc((size_t)H, sizeof(float));          /* get_initial_hidden :168-170 */
    float *Rh = (float *)malloc((size_t)3 * H * sizeof(float));
    float *t1 = (float *)malloc((size_t)S * sizeof(float));
    float *lg = (float *)malloc((size_t)Q * sizeof(float));
    float *p = (float *)malloc((size_t)Q * sizeof(float));
    float *zero = (float *)calloc((size_t)3 * H, sizeof(float));
    if (!h || !Rh || !t1 || !lg || !p || !zero) return -1;
    int oc = 0, of = 0;                                             /* out_coarse = out_fine = 0 :90-91 */
    for (long t = 0; t < seq_len; ++t) {
        const float pc = (float)oc / 127.5f - 1.0f, pf = (float)of / 127.5f - 1.0f;   /* :106-107 */
        matvec(m->R, zero, h, Rh, 3 * H, H);                        /* R(hidden), split 6 ways :116-119 */
        /* coarse gates :111-125 ; R_hidden layout: [u_c | u_f | r_c | r_f | e_c | e_f] */
        for (int j = 0; j < S; ++j) {
            const float Iu = m->Ic[(size_t)j * 2] * pc + m->Ic[(size_t)j * 2 + 1] * pf;
            const float Ir = m->Ic[(size_t)(S + j) * 2] * pc + m->Ic[(size_t)(S + j) * 2 + 1] * pf;
            const float Ie = m->Ic[(size_t)(2 * S + j) * 2] * pc + m->Ic[(size_t)(2 * S + j) * 2 + 1] * pf;
            const float u = sigmoidf_(Rh[j] + Iu + m->bu[j]);
            const float r = sigmoidf_(Rh[H + j] + Ir + m->br[j]);
            const float e = tanhf(r * Rh[2 * H + j] + Ie + m->be[j]);
            h[j] = u * h[j] + (1.0f - u) * e;
        }
        matvec(m->O1w, m->O1b, h, t1, S, S);                        /* O2(relu(O1(hidden_coarse))) :128 */
        for (int j = 0; j < S; ++j) t1[j] = t1[j] > 0.0f ? t1[j] : 0.0f;
        matvec(m->O2w, m->O2b, t1, lg, Q, S);
        oc = dm_sample(lg, Q, noise ? noise + ((size_t)t * 2 + 0) * Q : NULL, p, margin_out ? margin_out + t * 2 : NULL,
                       runner_out ? runner_out + t * 2 : NULL);
        coarse[t] = oc;
        const float cp = (float)oc / 127.5f - 1.0f;                 /* :135 */
        for (int j = 0; j < S; ++j) {                               /* fine gates :136-145 */
            const float *wu = m->If + (size_t)j * 3, *wr = m->If + (size_t)(S + j) * 3, *we = m->If + (size_t)(2 * S + j) * 3;
            const float Iu = wu[0] * pc + wu[1] * pf + wu[2] * cp;
            const float Ir = wr[0] * pc + wr[1] * pf + wr[2] * cp;
            const float Ie = we[0] * pc + we[1] * pf + we[2] * cp;
            const float u = sigmoidf_(Rh[S + j] + Iu + m->bu[S + j]);
            const float r = sigmoidf_(Rh[H + S + j] + Ir + m->br[S + j]);
            const float e = tanhf(r * Rh[2 * H + S + j] + Ie + m->be[S + j]);
            h[S + j] = u * h[S + j] + (1.0f - u) * e;
        }
        matvec(m->O3w, m->O3b, h + S, t1, S, S);                    /* O4(relu(O3(hidden_fine))) :148 */
        for (int j = 0; j < S; ++j) t1[j] = t1[j] > 0.0f ? t1[j] : 0.0f;
        matvec(m->O4w, m->O4b, t1, lg, Q, S);
        of = dm_sample(lg, Q, noise ? noise + ((size_t)t * 2 + 1) * Q : NULL, p, margin_out ? margin_out + t * 2 + 1 : NULL,
                       runner_out ? runner_out + t * 2 + 1 : NULL);
        fine[t] = of;
    }
    free(h); free(Rh); free(t1); free(lg); free(p); free(zero);
    return 0;
}
