// WRNN_KERNEL_SIMPLE: the per-sample loop of WaveRNN.generate
// (wavernn/models/fatchord_version.py:194-241), one persistent workgroup per
// loop row (utterance or fold).  This is the straightforward, reference-ordered
// implementation: every layer is evaluated as written (no algebraic hoisting),
// weights are streamed from L2/Infinity-Cache each step in [in][out] layout so
// a wavefront reads 256 contiguous bytes per k.  It is the correctness anchor
// and the any-shape fallback; the low-latency path is loop_team.hip.
//
// Thread j of the 512 owns output row j of every layer (rows j, j+H, j+2H of
// the 3H-row GRU matrices, i.e. gates r, z, n of hidden unit j -- gate order
// [r; z; n], get_gru_cell :273-279), so the GRU update needs no cross-thread
// exchange; activations live in LDS.
#include "device_util.h"
#include "wrnn_internal.h"

#define SIMPLE_THREADS 512

namespace {

struct SimpleLds {
    float cat[128];        // [x_{t-1} | m_t (F) | a1_t (A)]              :208
    float aux[128];        // a_t (R) for this step
    float xin[512];        // I(...)                                      :209
    float h1[512];
    float h2[512];
    float xa[544];         // [x + h1 | a2_t]                             :212-213
    float xb[544];         // [x + h2 | a3_t]                             :216-217
    float f1[544];         // [relu(fc1) | a4_t]                          :218,:220
    float f2[512];         // relu(fc2)                                   :221
    float logits[1024];    //                                             :223
    float redv[8];
    int redi[8];
    float xfeed;
};

template <int G>
__device__ __forceinline__ void matvec_t(const float *__restrict__ wt, int nrows, const float *xs, int K, int j,
                                         float (&acc)[G]) {
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.0f;
#pragma unroll 8
    for (int k = 0; k < K; ++k) {
        const float xv = xs[k];
        const float *row = wt + (size_t)k * nrows + j;
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = fmaf(row[g * SIMPLE_THREADS], xv, acc[g]);
    }
}

__global__ void __launch_bounds__(SIMPLE_THREADS) loop_simple_kernel(WrnnLoopArgs a) {
    __shared__ SimpleLds s;
    const WrnnDims d = a.d;
    const int H = d.H, FC = d.FC, F = d.F, A = d.A, R = d.R, NC = d.NC, HOP = d.HOP, ND = d.ND, P = d.P;
    const int j = threadIdx.x;
    const int row = blockIdx.x;
    const float *w = a.w;
    const WrnnRow rw = a.rows[row];
    const float *mel_b = a.mels + (size_t)rw.utt * F * a.T;
    const float *aux_b = a.aux_frames + (size_t)rw.utt * a.T * R;
    const float *ktab = w + a.off.ktab;
    const int lane = j & 63, wave = j >> 6;

    // h1 = h2 = 0, x = 0   (:194-196)
    s.h1[j] = 0.0f;
    s.h2[j] = 0.0f;
    if (j == 0) s.xfeed = 0.0f;
    __syncthreads();

    for (int64_t t = 0; t < a.steps; ++t) {
        // ---- conditioning row for this step: m_t, a_t  (:203-206) ----------
        const int64_t pos = rw.start + t;
        const bool live = pos < a.total_len;  // fold padding 'after' is zeros (:327-330)
        const int i = live ? (int)(pos / HOP) : 0;
        const int r = live ? (int)(pos - (int64_t)i * HOP) : 0;
        if (j < F) {
            float acc = 0.0f;
            if (live) {
                for (int k = 0; k < ND; ++k) {
                    const int fr = i + k - P;
                    const float mv = (fr >= 0 && fr < a.T) ? mel_b[(size_t)j * a.T + fr] : 0.0f;
                    acc = fmaf(ktab[r * ND + k], mv, acc);
                }
            }
            s.cat[1 + j] = acc;
        } else if (j >= 128 && j < 128 + R) {
            const int c = j - 128;
            s.aux[c] = live ? aux_b[(size_t)i * R + c] : 0.0f;
        }
        if (j == 0) s.cat[0] = s.xfeed;
        __syncthreads();
        if (j < A) {
            s.cat[1 + F + j] = s.aux[j];            // a1_t
            s.xa[H + j] = s.aux[A + j];             // a2_t
            s.xb[H + j] = s.aux[2 * A + j];         // a3_t
            s.f1[FC + j] = s.aux[3 * A + j];        // a4_t
        }
        __syncthreads();

        // ---- x = I(cat[x, m_t, a1_t])  (:208-209) ---------------------------
        float acc1[1];
        matvec_t<1>(w + a.off.I_t, H, s.cat, 1 + F + A, j, acc1);
        const float xin = acc1[0] + w[a.off.I_b + j];
        s.xin[j] = xin;
        __syncthreads();

        // ---- h1 = rnn1(x, h1); x = x + h1  (:210-212) ----------------------
        float gi[3], gh[3];
        matvec_t<3>(w + a.off.r1_wih_t, 3 * H, s.xin, H, j, gi);
        matvec_t<3>(w + a.off.r1_whh_t, 3 * H, s.h1, H, j, gh);
        float h1n;
        {
            const float *bi = w + a.off.r1_bih, *bh = w + a.off.r1_bhh;
            const float rg = 1.0f / (1.0f + expf(-((gi[0] + bi[j]) + (gh[0] + bh[j]))));
            const float zg = 1.0f / (1.0f + expf(-((gi[1] + bi[H + j]) + (gh[1] + bh[H + j]))));
            const float ng = tanhf((gi[2] + bi[2 * H + j]) + rg * (gh[2] + bh[2 * H + j]));
            h1n = (1.0f - zg) * ng + zg * s.h1[j];
        }
        const float x2 = xin + h1n;
        __syncthreads();  // everyone finished reading h1
        s.h1[j] = h1n;
        s.xa[j] = x2;
        __syncthreads();

        // ---- h2 = rnn2(cat[x, a2_t], h2); x = x + h2  (:213-216) -----------
        matvec_t<3>(w + a.off.r2_wih_t, 3 * H, s.xa, H + A, j, gi);
        matvec_t<3>(w + a.off.r2_whh_t, 3 * H, s.h2, H, j, gh);
        float h2n;
        {
            const float *bi = w + a.off.r2_bih, *bh = w + a.off.r2_bhh;
            const float rg = 1.0f / (1.0f + expf(-((gi[0] + bi[j]) + (gh[0] + bh[j]))));
            const float zg = 1.0f / (1.0f + expf(-((gi[1] + bi[H + j]) + (gh[1] + bh[H + j]))));
            const float ng = tanhf((gi[2] + bi[2 * H + j]) + rg * (gh[2] + bh[2 * H + j]));
            h2n = (1.0f - zg) * ng + zg * s.h2[j];
        }
        const float x3 = x2 + h2n;
        __syncthreads();
        s.h2[j] = h2n;
        s.xb[j] = x3;
        __syncthreads();

        // ---- x = relu(fc1(cat[x, a3_t]))  (:217-218) ------------------------
        matvec_t<1>(w + a.off.fc1_t, FC, s.xb, H + A, j, acc1);
        s.f1[j] = fmaxf(acc1[0] + w[a.off.fc1_b + j], 0.0f);
        __syncthreads();
        // ---- x = relu(fc2(cat[x, a4_t]))  (:220-221) ------------------------
        matvec_t<1>(w + a.off.fc2_t, FC, s.f1, FC + A, j, acc1);
        s.f2[j] = fmaxf(acc1[0] + w[a.off.fc2_b + j], 0.0f);
        __syncthreads();
        // ---- logits = fc3(x)  (:223) -----------------------------------------
        for (int c = j; c < NC; c += SIMPLE_THREADS) {
            float acc = 0.0f;
            const float *wt = w + a.off.fc3_t + c;
#pragma unroll 8
            for (int k = 0; k < FC; ++k) acc = fmaf(wt[(size_t)k * NC], s.f2[k], acc);
            const float lg = acc + w[a.off.fc3_b + c];
            s.logits[c] = lg;
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
            for (int c = j; c < NC; c += SIMPLE_THREADS) {
                float v = s.logits[c];
                if (a.noise_mode == WRNN_NOISE_INJECTED) {
                    v -= logf(a.noise1[((size_t)t * a.n_rows + row) * NC + c]);
                } else if (a.noise_mode == WRNN_NOISE_PHILOX) {
                    const float u = wrnn_uniform_raw(a.seed, (uint64_t)t, (uint32_t)row, (uint32_t)c);
                    v -= logf(-logf(u));
                }
                if (v > bv) { bv = v; bi = c; }
            }
            wave_argmax(bv, bi);
            if (lane == 0) { s.redv[wave] = bv; s.redi[wave] = bi; }
            __syncthreads();
            if (j == 0) {
                float v = s.redv[0];
                int k = s.redi[0];
                for (int q = 1; q < SIMPLE_THREADS / 64; ++q)
                    if (s.redv[q] > v || (s.redv[q] == v && s.redi[q] < k)) { v = s.redv[q]; k = s.redi[q]; }
                // sample = 2 * k / (n_classes - 1.) - 1.   (:235)
                const float smp = 2.0f * (float)k / ((float)NC - 1.0f) - 1.0f;
                if (a.labels_out) a.labels_out[(size_t)row * a.steps + t] = k;
                a.samples_out[(size_t)row * a.steps + t] = smp;
                s.xfeed = a.x_forced ? a.x_forced[(size_t)t * a.n_rows + row] : smp;
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
                    v = s.logits[lane] - logf(-logf(u1));   // :107
                    k = lane;
                }
                wave_argmax(v, k);
                if (lane == 0) {
                    float u2;
                    if (a.noise_mode == WRNN_NOISE_INJECTED) u2 = a.noise2[(size_t)t * a.n_rows + row];
                    else u2 = 1e-5f + wrnn_uniform(a.seed, (uint64_t)t, (uint32_t)row, 10u) * (1.0f - 2e-5f);
                    const float mean = s.logits[nr + k];                       // :113
                    const float ls = fmaxf(s.logits[2 * nr + k], -32.23619130191664f);  // log(1e-14) :114-115
                    float xs = mean + expf(ls) * (logf(u2) - logf(1.0f - u2));  // :119
                    xs = fminf(fmaxf(xs, -1.0f), 1.0f);                         // :121
                    if (a.labels_out) a.labels_out[(size_t)row * a.steps + t] = k;
                    a.samples_out[(size_t)row * a.steps + t] = xs;
                    s.xfeed = a.x_forced ? a.x_forced[(size_t)t * a.n_rows + row] : xs;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t wrnn_launch_loop_simple(const WrnnLoopArgs &a, hipStream_t s) {
    (void)hipGetLastError();  // the runtime is shared with PyTorch: drop any stale sticky error of this thread
    hipLaunchKernelGGL(loop_simple_kernel, dim3(a.n_rows), dim3(SIMPLE_THREADS), 0, s, a);
    return hipGetLastError();
}
