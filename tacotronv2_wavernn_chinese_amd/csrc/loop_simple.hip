// WRNN_KERNEL_SIMPLE: the per-sample loop of WaveRNN.generate
// (wavernn/models/fatchord_version.py:194-241), one persistent workgroup per
// loop row (utterance or fold).  This is the straightforward, reference-ordered
// implementation: every layer is evaluated as written (no algebraic hoisting),
// weights are streamed from L2/Infinity-Cache each step in [in][out] layout so
// a wavefront reads 256 contiguous bytes per k.  It is the correctness anchor
// and the any-shape kernel (rnn_dims <= 1024, every activation vector in LDS):
// AUTO selects it when the team kernels (loop_team2.hip, loop_batch.hip: built
// for the reference hparams) cannot run.
//
// Thread j (strided over the 512) owns output row j of every layer (rows j,
// j+H, j+2H of the 3H-row GRU matrices, i.e. gates r, z, n of hidden unit j --
// gate order [r; z; n], get_gru_cell :273-279), so the GRU update needs no
// cross-thread exchange; activations live in LDS.
#include "device_util.h"
#include "wrnn_internal.h"

#define SIMPLE_THREADS 512

namespace {

// LDS carve-up (floats) for arbitrary dims: sizes rounded up to 4
struct SimpleLay {
    int cat, aux, xin, h1, h2, xa, xb, f1, f2, logits, red, total;
    __host__ __device__ SimpleLay(const WrnnDims &d) {
        auto r4 = [](int n) { return (n + 3) & ~3; };
        int o = 0;
        cat = o; o += r4(1 + d.F + d.A);     // [x_{t-1} | m_t (F) | a1_t (A)]              :208
        aux = o; o += r4(d.R);               // a_t (R) for this step
        xin = o; o += r4(d.H);               // I(...)                                      :209
        h1 = o; o += r4(d.H);
        h2 = o; o += r4(d.H);
        xa = o; o += r4(d.H + d.A);          // [x + h1 | a2_t]                             :212-213
        xb = o; o += r4(d.H + d.A);          // [x + h2 | a3_t]                             :216-217
        f1 = o; o += r4(d.FC + d.A);         // [relu(fc1) | a4_t]                          :218,:220
        f2 = o; o += r4(d.FC);               // relu(fc2)                                   :221
        logits = o; o += r4(d.NC);           //                                             :223
        red = o; o += 32;                    // [8] values | [8] indices | xfeed
        total = o;
    }
};

// acc[g] = sum_k wt[k][j + g * stride] * xs[k]   (weights [in][out]: consecutive threads read consecutive addresses)
template <int G>
__device__ __forceinline__ void matvec_t(const float *__restrict__ wt, int nrows, int gstride, const float *xs, int K, int j,
                                         float (&acc)[G]) {
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.0f;
#pragma unroll 8
    for (int k = 0; k < K; ++k) {
        const float xv = xs[k];
        const float *row = wt + (size_t)k * nrows + j;
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = fmaf(row[g * gstride], xv, acc[g]);
    }
}

__global__ void __launch_bounds__(SIMPLE_THREADS) loop_simple_kernel(WrnnLoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_simple[];
    float *lds = (float *)smem_simple;
    const WrnnDims d = a.d;
    const SimpleLay ly(d);
    float *s_cat = lds + ly.cat, *s_aux = lds + ly.aux, *s_xin = lds + ly.xin, *s_h1 = lds + ly.h1, *s_h2 = lds + ly.h2;
    float *s_xa = lds + ly.xa, *s_xb = lds + ly.xb, *s_f1 = lds + ly.f1, *s_f2 = lds + ly.f2, *s_logits = lds + ly.logits;
    float *s_redv = lds + ly.red;
    int *s_redi = (int *)(lds + ly.red + 8);
    float *s_xfeed = lds + ly.red + 16;
    const int H = d.H, FC = d.FC, F = d.F, A = d.A, R = d.R, NC = d.NC, HOP = d.HOP, ND = d.ND, P = d.P;
    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const float *w = a.w;
    const WrnnRow rw = a.rows[row];
    const float *mel_b = a.mels + (size_t)rw.utt * F * a.mel_T;
    const float *aux_b = a.aux_frames + (size_t)rw.utt * a.T * R;
    const float *ktab = w + a.off.ktab;
    const int lane = tid & 63, wave = tid >> 6;

    // h1 = h2 = 0, x = x_init or 0   (:194-196)
    for (int j = tid; j < H; j += SIMPLE_THREADS) { s_h1[j] = 0.0f; s_h2[j] = 0.0f; }
    if (tid == 0) *s_xfeed = a.x_init ? a.x_init[row] : 0.0f;
    __syncthreads();

    for (int64_t t = 0; t < rw.steps; ++t) {   // the row's own length (ragged batch) or the call's
        // ---- conditioning row for this step: m_t, a_t  (:203-206) ----------
        const int64_t pos = rw.start + t;
        const bool live = pos < a.total_len;  // fold padding 'after' is zeros (:327-330)
        const int i = live ? (int)(pos / HOP) : 0;
        const int r = live ? (int)(pos - (int64_t)i * HOP) : 0;
        for (int j = tid; j < F + R; j += SIMPLE_THREADS) {
            if (j < F) {
                float acc = 0.0f;
                if (live) {
                    for (int k = 0; k < ND; ++k) {
                        const int fr = i + k - P + a.mel_off;
                        const float mv = (fr >= 0 && fr < a.mel_T) ? mel_b[(size_t)j * a.mel_T + fr] : 0.0f;
                        acc = fmaf(ktab[r * ND + k], mv, acc);
                    }
                }
                s_cat[1 + j] = acc;
            } else {
                const int c = j - F;
                s_aux[c] = live ? aux_b[(size_t)i * R + c] : 0.0f;
            }
        }
        if (tid == 0) s_cat[0] = *s_xfeed;
        __syncthreads();
        for (int j = tid; j < A; j += SIMPLE_THREADS) {
            s_cat[1 + F + j] = s_aux[j];            // a1_t
            s_xa[H + j] = s_aux[A + j];             // a2_t
            s_xb[H + j] = s_aux[2 * A + j];         // a3_t
            s_f1[FC + j] = s_aux[3 * A + j];        // a4_t
        }
        __syncthreads();

        // ---- x = I(cat[x, m_t, a1_t])  (:208-209) ---------------------------
        for (int j = tid; j < H; j += SIMPLE_THREADS) {
            float acc1[1];
            matvec_t<1>(w + a.off.I_t, H, 0, s_cat, 1 + F + A, j, acc1);
            s_xin[j] = acc1[0] + w[a.off.I_b + j];
        }
        __syncthreads();

        // ---- h1 = rnn1(x, h1); x = x + h1  (:210-212); thread j owns gate rows j, j+H, j+2H ([r; z; n], :273-279) ----
        float h1n[2], x2[2];   // up to 2 units per thread (H <= 1024)
        {
            int u = 0;
            for (int j = tid; j < H; j += SIMPLE_THREADS, ++u) {
                float gi[3], gh[3];
                matvec_t<3>(w + a.off.r1_wih_t, 3 * H, H, s_xin, H, j, gi);
                matvec_t<3>(w + a.off.r1_whh_t, 3 * H, H, s_h1, H, j, gh);
                const float *bi = w + a.off.r1_bih, *bh = w + a.off.r1_bhh;
                const float rg = 1.0f / (1.0f + expf(-((gi[0] + bi[j]) + (gh[0] + bh[j]))));
                const float zg = 1.0f / (1.0f + expf(-((gi[1] + bi[H + j]) + (gh[1] + bh[H + j]))));
                const float ng = tanhf((gi[2] + bi[2 * H + j]) + rg * (gh[2] + bh[2 * H + j]));
                h1n[u] = (1.0f - zg) * ng + zg * s_h1[j];
                x2[u] = s_xin[j] + h1n[u];
            }
        }
        __syncthreads();  // everyone finished reading h1
        {
            int u = 0;
            for (int j = tid; j < H; j += SIMPLE_THREADS, ++u) { s_h1[j] = h1n[u]; s_xa[j] = x2[u]; }
        }
        __syncthreads();

        // ---- h2 = rnn2(cat[x, a2_t], h2); x = x + h2  (:213-216) -----------
        float h2n[2], x3[2];
        {
            int u = 0;
            for (int j = tid; j < H; j += SIMPLE_THREADS, ++u) {
                float gi[3], gh[3];
                matvec_t<3>(w + a.off.r2_wih_t, 3 * H, H, s_xa, H + A, j, gi);
                matvec_t<3>(w + a.off.r2_whh_t, 3 * H, H, s_h2, H, j, gh);
                const float *bi = w + a.off.r2_bih, *bh = w + a.off.r2_bhh;
                const float rg = 1.0f / (1.0f + expf(-((gi[0] + bi[j]) + (gh[0] + bh[j]))));
                const float zg = 1.0f / (1.0f + expf(-((gi[1] + bi[H + j]) + (gh[1] + bh[H + j]))));
                const float ng = tanhf((gi[2] + bi[2 * H + j]) + rg * (gh[2] + bh[2 * H + j]));
                h2n[u] = (1.0f - zg) * ng + zg * s_h2[j];
                x3[u] = x2[u] + h2n[u];
            }
        }
        __syncthreads();
        {
            int u = 0;
            for (int j = tid; j < H; j += SIMPLE_THREADS, ++u) { s_h2[j] = h2n[u]; s_xb[j] = x3[u]; }
        }
        __syncthreads();

        // ---- x = relu(fc1(cat[x, a3_t]))  (:217-218) ------------------------
        for (int j = tid; j < FC; j += SIMPLE_THREADS) {
            float acc1[1];
            matvec_t<1>(w + a.off.fc1_t, FC, 0, s_xb, H + A, j, acc1);
            s_f1[j] = fmaxf(acc1[0] + w[a.off.fc1_b + j], 0.0f);
        }
        __syncthreads();
        // ---- x = relu(fc2(cat[x, a4_t]))  (:220-221) ------------------------
        for (int j = tid; j < FC; j += SIMPLE_THREADS) {
            float acc1[1];
            matvec_t<1>(w + a.off.fc2_t, FC, 0, s_f1, FC + A, j, acc1);
            s_f2[j] = fmaxf(acc1[0] + w[a.off.fc2_b + j], 0.0f);
        }
        __syncthreads();
        // ---- logits = fc3(x)  (:223) -----------------------------------------
        for (int c = tid; c < NC; c += SIMPLE_THREADS) {
            float acc = 0.0f;
            const float *wt = w + a.off.fc3_t + c;
#pragma unroll 8
            for (int k = 0; k < FC; ++k) acc = fmaf(wt[(size_t)k * NC], s_f2[k], acc);
            const float lg = acc + w[a.off.fc3_b + c];
            s_logits[c] = lg;
            if (a.logits_out) a.logits_out[((size_t)t * a.n_rows + row) * NC + c] = lg;
        }
        __syncthreads();

        // ---- sample  (:225-237) ----------------------------------------------
        if (d.mode == WRNN_MODE_RAW) {
            // Categorical(softmax(logits)).sample() == argmax_k p_k / q_k, q ~ Exp(1)
            // (torch.multinomial n=1 path); evaluated in the log domain:
            // argmax_k logit_k - log q_k  (same ordering, no softmax pass needed).
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int c = tid; c < NC; c += SIMPLE_THREADS) {
                float v = s_logits[c];
                if (a.noise_mode == WRNN_NOISE_INJECTED) {
                    v -= logf(a.noise1[((size_t)t * a.n_rows + row) * NC + c]);
                } else if (a.noise_mode == WRNN_NOISE_PHILOX) {
                    const float u = wrnn_uniform_raw(a.seed, (uint64_t)t, (uint32_t)row, (uint32_t)c);
                    v -= logf(-logf(u));
                }
                if (v > bv) { bv = v; bi = c; }
            }
            wave_argmax(bv, bi);
            if (lane == 0) { s_redv[wave] = bv; s_redi[wave] = bi; }
            __syncthreads();
            if (tid == 0) {
                float v = s_redv[0];
                int k = s_redi[0];
                for (int q = 1; q < SIMPLE_THREADS / 64; ++q)
                    if (s_redv[q] > v || (s_redv[q] == v && s_redi[q] < k)) { v = s_redv[q]; k = s_redi[q]; }
                // sample = 2 * k / (n_classes - 1.) - 1.   (:235)
                const float smp = 2.0f * (float)k / ((float)NC - 1.0f) - 1.0f;
                if (a.labels_out) a.labels_out[(size_t)row * a.steps + t] = k;
                a.samples_out[(size_t)row * a.steps + t] = smp;
                *s_xfeed = a.x_forced ? a.x_forced[(size_t)t * a.n_rows + row] : smp;
            }
        } else {
            // sample_from_discretized_mix_logistic, wavernn/utils/distribution.py:87-123
            const int nr = NC / 3;
            if (wave == 0) {
                float v = -INFINITY;
                int k = 0x7fffffff;
                if (lane < nr) {
                    float u1;
                    if (a.noise_mode == WRNN_NOISE_INJECTED) u1 = a.noise1[((size_t)t * a.n_rows + row) * nr + lane];
                    else u1 = 1e-5f + wrnn_uniform(a.seed, (uint64_t)t, (uint32_t)row, (uint32_t)lane) * (1.0f - 2e-5f);
                    v = s_logits[lane] - logf(-logf(u1));   // :107
                    k = lane;
                }
                wave_argmax(v, k);
                if (lane == 0) {
                    float u2;
                    if (a.noise_mode == WRNN_NOISE_INJECTED) u2 = a.noise2[(size_t)t * a.n_rows + row];
                    else u2 = 1e-5f + wrnn_uniform(a.seed, (uint64_t)t, (uint32_t)row, 10u) * (1.0f - 2e-5f);
                    const float mean = s_logits[nr + k];                       // :113
                    const float ls = fmaxf(s_logits[2 * nr + k], -32.23619130191664f);  // log(1e-14) :114-115
                    float xs = mean + expf(ls) * (logf(u2) - logf(1.0f - u2));  // :119
                    xs = fminf(fmaxf(xs, -1.0f), 1.0f);                         // :121
                    if (a.labels_out) a.labels_out[(size_t)row * a.steps + t] = k;
                    a.samples_out[(size_t)row * a.steps + t] = xs;
                    *s_xfeed = a.x_forced ? a.x_forced[(size_t)t * a.n_rows + row] : xs;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

// LDS bytes the kernel needs for these dims (wrnn_create refuses configurations that do not fit a CU's 160 KB)
size_t wrnn_simple_lds_bytes(const WrnnDims &d) { return (size_t)SimpleLay(d).total * sizeof(float); }

hipError_t wrnn_launch_loop_simple(const WrnnLoopArgs &a, hipStream_t s) {
    (void)hipGetLastError();  // the runtime is shared with PyTorch: drop any stale sticky error of this thread
    const size_t lds = wrnn_simple_lds_bytes(a.d);
    hipError_t e = hipFuncSetAttribute((const void *)loop_simple_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(loop_simple_kernel, dim3(a.n_rows), dim3(SIMPLE_THREADS), lds, s, a);
    return hipGetLastError();
}
