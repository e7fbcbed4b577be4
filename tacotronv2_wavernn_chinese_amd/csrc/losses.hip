// The two losses the reference's training script applies to WaveRNN.forward's output (wavernn_train.py:82,112-121):
//   RAW: F.cross_entropy(y_hat.transpose(1, 2).unsqueeze(-1), y.unsqueeze(-1))  -- mean over B*L of logsumexp(row) - row[y]
//   MOL: discretized_mix_logistic_loss(y_hat, y)  (wavernn/utils/distribution.py:16-84, num_classes = 65536,
//        log_scale_min = log(1e-14), reduce = True)  -- mean over B*L of -logsumexp_k(log_prob_k + log_softmax(logit)_k)
// Forward values only (the path here is inference; the numbers are what a training log would print).
// Both are HBM-read-bound row reductions: one wave per row (RAW, 4 KB per row) / one thread per row (MOL, 120 B per row),
// per-block partial sums in double, summed in a fixed order by a second kernel (deterministic, no atomics).
#include "wrnn_internal.h"

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// grid ceil(rows / 4), block 256: wave w of block b reduces row 4 b + w
__global__ void __launch_bounds__(256) ce_rows_kernel(const float *__restrict__ logits, const int32_t *__restrict__ y, int NC,
                                                      long n_rows, double *__restrict__ partial, int *__restrict__ bad) {
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    double nll = 0.0;
    if (row < n_rows) {
        const float *p = logits + (size_t)row * NC;
        float m = -INFINITY;
        for (int c = lane; c < NC; c += 64) m = fmaxf(m, p[c]);
        m = wave_max_f(m);
        float s = 0.0f;
        for (int c = lane; c < NC; c += 64) s += expf(p[c] - m);
        s = wave_sum_f(s);
        const int t = y[row];
        if (t < 0 || t >= NC) { if (lane == 0) atomicExch(bad, 1); }
        else nll = (double)((m + logf(s)) - p[t]);    // -log_softmax(row)[y]
    }
    if (lane == 0) part[wave] = nll;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }   // F.softplus (beta 1, threshold 20)
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// grid ceil(rows / 256), block 256: one thread per (b, t) row of y_hat (rows, 3 * nr_mix)
__global__ void __launch_bounds__(256) mol_rows_kernel(const float *__restrict__ y_hat, const float *__restrict__ yv, int nr_mix,
                                                       long n_rows, float num_classes, float log_scale_min,
                                                       double *__restrict__ partial) {
    __shared__ double part[4];
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    double loss = 0.0;
    if (row < n_rows) {
        const float *p = y_hat + (size_t)row * 3 * nr_mix;
        const float y = yv[row];
        // log_softmax(logit_probs)                                   (:77)
        float lm = -INFINITY;
        for (int k = 0; k < nr_mix; ++k) lm = fmaxf(lm, p[k]);
        float ls = 0.0f;
        for (int k = 0; k < nr_mix; ++k) ls += expf(p[k] - lm);
        const float lse_logit = lm + logf(ls);
        const float half_bin = 1.0f / (num_classes - 1.0f);
        const float log_half = logf((num_classes - 1.0f) / 2.0f);
        float lp[16];
        float mx = -INFINITY;
        for (int k = 0; k < nr_mix; ++k) {
            const float mean = p[nr_mix + k];
            const float lsc = fmaxf(p[2 * nr_mix + k], log_scale_min);          // :31
            const float cy = y - mean;                                          // :36
            const float inv = expf(-lsc);                                       // :37
            const float plus_in = inv * (cy + half_bin), min_in = inv * (cy - half_bin);
            const float cdf_delta = sigmoid_ref(plus_in) - sigmoid_ref(min_in); // :39-54
            const float log_cdf_plus = plus_in - softplus_f(plus_in);           // :45
            const float log_one_minus_cdf_min = -softplus_f(min_in);            // :49
            const float mid_in = inv * cy;
            const float log_pdf_mid = mid_in - lsc - 2.0f * softplus_f(mid_in); // :57
            const float c2 = cdf_delta > 1e-5f ? 1.0f : 0.0f;                   // :68-72, evaluated as the same blend of both arms
            const float inner_inner = c2 * logf(fmaxf(cdf_delta, 1e-12f)) + (1.0f - c2) * (log_pdf_mid - log_half);
            const float c1 = y > 0.999f ? 1.0f : 0.0f;
            const float inner = c1 * log_one_minus_cdf_min + (1.0f - c1) * inner_inner;
            const float c0 = y < -0.999f ? 1.0f : 0.0f;
            const float v = (c0 * log_cdf_plus + (1.0f - c0) * inner) + (p[k] - lse_logit);   // :75-77
            lp[k] = v;
            mx = fmaxf(mx, v);
        }
        float s = 0.0f;
        for (int k = 0; k < nr_mix; ++k) s += expf(lp[k] - mx);                 // log_sum_exp :6-12
        loss = -(double)(mx + logf(s));
    }
    // block sum in a fixed order
    for (int off = 32; off >= 1; off >>= 1) loss += __shfl_xor(loss, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ void __launch_bounds__(256) mean_kernel(const double *__restrict__ partial, long n, double inv_count, const int *__restrict__ bad,
                                                   float *__restrict__ out) {
    __shared__ double sh[256];
    double acc = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) acc += partial[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = *bad ? __builtin_nanf("") : (float)(sh[0] * inv_count);   // a target outside [0, n_classes): NaN (torch raises)
}

}  // namespace

hipError_t wrnn_launch_loss(int mode, const float *y_hat, const void *y, int NC, long n_rows, double *partial, int *bad, float *out,
                            hipStream_t s) {
    (void)hipGetLastError();
    long nblk;
    if (mode == WRNN_MODE_RAW) {
        nblk = (n_rows + 3) / 4;
        hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)nblk), dim3(256), 0, s, y_hat, (const int32_t *)y, NC, n_rows, partial, bad);
    } else {
        nblk = (n_rows + 255) / 256;
        hipLaunchKernelGGL(mol_rows_kernel, dim3((unsigned)nblk), dim3(256), 0, s, y_hat, (const float *)y, NC / 3, n_rows, 65536.0f,
                           -32.23619130191664f, partial);
    }
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, partial, nblk, 1.0 / (double)n_rows, bad, out);
    return hipGetLastError();
}
