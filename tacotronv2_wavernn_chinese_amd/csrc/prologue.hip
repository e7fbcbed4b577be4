// Prologue kernels: MelResNet (aux features) and the (5,5,11) mel upsampler.
//
// Reference (paths relative to /root/reference/):
//   pad_tensor 'both'            wavernn/models/fatchord_version.py:281-291
//   MelResNet.forward            :42-48   ResBlock.forward :21-28
//   Stretch2d.forward            :57-61
//   UpsampleNetwork.forward      :82-89
//
// Design notes (gfx950):
//  * BatchNorm1d runs in eval mode inside generate() (:170), so it is folded
//    into the adjacent conv weights at load time (api.hip).
//  * MelResNet is ~0.4 MMAC per frame and runs once per utterance: one
//    workgroup handles FT frames so every weight read is reused FT times;
//    weights are stored [in][out] so a wavefront reads 256 contiguous bytes.
//  * The three "stretch by s, then (2s+1)-tap conv" stages are linear and, after
//    the `indent` crop (:88), shift-invariant (edge reach 341 samples < indent
//    550): upsampled[t = hop*i + r, c] = sum_d ktab[r][d] * melpad[i + d, c],
//    d < ND (= 5).  ktab is built on the host in fp64 from the learned taps.
//    The materialising kernel below is a pure coalesced HBM writer
//    (832 B/sample); the loop kernels never call it -- they evaluate the same
//    5-tap form on the fly.
#include "wrnn_internal.h"

#define RESNET_FT 8      // frames per workgroup

// grid (ceil(T/FT), B), block = max(C, R) rounded up to a wave (thread c = output channel c; 128 for the reference hparams)
__global__ void __launch_bounds__(1024)
resnet_kernel(const float *__restrict__ w, WrnnPacked off, WrnnDims d, const float *__restrict__ mels, int T, int mel_T,
              int mel_off, float *__restrict__ aux_frames) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int F = d.F, C = d.C, R = d.R, P = d.P, KS = d.KS;
    const int FT = RESNET_FT;
    float *xp = (float *)smem;             // [(FT + KS - 1)][F]  padded mel window, frame-major
    float *xa = xp + (FT + KS - 1) * F;    // [FT][C]
    float *xb = xa + FT * C;               // [FT][C]
    const int b = blockIdx.y, t0 = blockIdx.x * FT, c = threadIdx.x;

    // pad_tensor(..., side='both'): zeros outside [0, T)   (:281-291).  A caller that hands over mels already padded
    // by `pad` frames each side (WaveRNN.forward, :143) sets mel_T = T + 2 pad, mel_off = pad: every frame is read.
    for (int i = threadIdx.x; i < (FT + KS - 1) * F; i += blockDim.x) {
        const int fr = i / F, f = i - fr * F;
        const int t = t0 + fr - P + mel_off;
        xp[i] = (t >= 0 && t < mel_T) ? mels[((size_t)b * F + f) * mel_T + t] : 0.0f;
    }
    __syncthreads();

    float acc[RESNET_FT];
    // conv_in (k = 2*pad+1, valid, no bias) + folded BN + ReLU   (:34-36,:43-45)
    if (c < C) {
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f] = 0.0f;
        const float *wt = w + off.conv_in_t;
        for (int fi = 0; fi < F; ++fi)
            for (int k = 0; k < KS; ++k) {
                const float wv = wt[(size_t)(fi * KS + k) * C + c];
#pragma unroll
                for (int f = 0; f < FT; ++f) acc[f] = fmaf(wv, xp[(f + k) * F + fi], acc[f]);
            }
        const float bv = w[off.conv_in_b + c];
#pragma unroll
        for (int f = 0; f < FT; ++f) xa[f * C + c] = fmaxf(acc[f] + bv, 0.0f);
    }
    __syncthreads();
    // residual blocks: x + BN2(conv2(relu(BN1(conv1(x)))))   (:21-28)
    for (int l = 0; l < d.NBLK; ++l) {
        if (c < C) {
            const float *w1 = w + off.res_w1_t + (size_t)l * C * C;
#pragma unroll
            for (int f = 0; f < FT; ++f) acc[f] = 0.0f;
            for (int ci = 0; ci < C; ++ci) {
                const float wv = w1[(size_t)ci * C + c];
#pragma unroll
                for (int f = 0; f < FT; ++f) acc[f] = fmaf(wv, xa[f * C + ci], acc[f]);
            }
            const float bv = w[off.res_b1 + (size_t)l * C + c];
#pragma unroll
            for (int f = 0; f < FT; ++f) xb[f * C + c] = fmaxf(acc[f] + bv, 0.0f);
        }
        __syncthreads();
        float res[RESNET_FT];
        if (c < C) {
            const float *w2 = w + off.res_w2_t + (size_t)l * C * C;
#pragma unroll
            for (int f = 0; f < FT; ++f) acc[f] = 0.0f;
            for (int ci = 0; ci < C; ++ci) {
                const float wv = w2[(size_t)ci * C + c];
#pragma unroll
                for (int f = 0; f < FT; ++f) acc[f] = fmaf(wv, xb[f * C + ci], acc[f]);
            }
            const float bv = w[off.res_b2 + (size_t)l * C + c];
#pragma unroll
            for (int f = 0; f < FT; ++f) res[f] = (acc[f] + bv) + xa[f * C + c];
        }
        __syncthreads();
        if (c < C) {
#pragma unroll
            for (int f = 0; f < FT; ++f) xa[f * C + c] = res[f];
        }
        __syncthreads();
    }
    // conv_out (1x1, bias)   (:40,:47)
    if (c < R) {
        const float *wo = w + off.conv_out_t;
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f] = 0.0f;
        for (int ci = 0; ci < C; ++ci) {
            const float wv = wo[(size_t)ci * R + c];
#pragma unroll
            for (int f = 0; f < FT; ++f) acc[f] = fmaf(wv, xa[f * C + ci], acc[f]);
        }
        const float bv = w[off.conv_out_b + c];
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            const int t = t0 + f;
            if (t < T) aux_frames[((size_t)b * T + t) * R + c] = acc[f] + bv;
        }
    }
}

hipError_t wrnn_launch_resnet(const wrnn_handle *h, const float *mels, int B, int T, int mel_T, int mel_off, float *aux_frames,
                              hipStream_t s) {
    (void)hipGetLastError();  // the runtime is shared with PyTorch: drop any stale sticky error of this thread
    const WrnnDims &d = h->d;
    dim3 grid((T + RESNET_FT - 1) / RESNET_FT, B);
    size_t lds = ((size_t)(RESNET_FT + d.KS - 1) * d.F + 2 * (size_t)RESNET_FT * d.C) * sizeof(float);
    const int nthr = (((d.C > d.R ? d.C : d.R) + 63) / 64) * 64;
    // wrnn_create admits any dims whose window fits a CU's 160 KB: above the default 64 KB the launch needs the attribute
    hipError_t e = hipFuncSetAttribute((const void *)resnet_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(resnet_kernel, grid, dim3(nthr), lds, s, h->wdev, h->off, d, mels, T, mel_T, mel_off, aux_frames);
    return hipGetLastError();
}

// Materialised conditioning (parity tests of rows A2-A5 only).
//   up  (B, T*HOP, F):  up[b][t][c]  = sum_d ktab[t % HOP][d] * melpad[b][t / HOP + d][c]
//   aux (B, T*HOP, R):  aux[b][t][:] = aux_frames[b][t / HOP][:]     (Stretch2d(total_scale,1) :70,:84)
// grid (ceil(T*HOP / 64), B), block 256: 64 positions x (F + R = 208 channels) per block, channel fastest.
__global__ void __launch_bounds__(256)
materialize_kernel(const float *__restrict__ w, WrnnPacked off, WrnnDims d, const float *__restrict__ mels,
                   const float *__restrict__ aux_frames, int T, int mel_T, int mel_off, float *__restrict__ up,
                   float *__restrict__ aux_up) {
    const int F = d.F, R = d.R, HOP = d.HOP, ND = d.ND, P = d.P;
    const int b = blockIdx.y;
    const long L = (long)T * HOP;
    const long t0 = (long)blockIdx.x * 64;
    const float *ktab = w + off.ktab;
    const int per = F + R;
    for (int idx = threadIdx.x; idx < 64 * per; idx += blockDim.x) {
        const int dt = idx / per, c = idx - dt * per;
        const long t = t0 + dt;
        if (t >= L) break;
        const int i = (int)(t / HOP), r = (int)(t - (long)i * HOP);
        if (c < F) {
            if (!up) continue;
            float acc = 0.0f;
            for (int k = 0; k < ND; ++k) {
                const int fr = i + k - P + mel_off;  // melpad[i + k] = mel[i + k - pad], zero outside
                const float mv = (fr >= 0 && fr < mel_T) ? mels[((size_t)b * F + c) * mel_T + fr] : 0.0f;
                acc = fmaf(ktab[r * ND + k], mv, acc);
            }
            up[((size_t)b * L + t) * F + c] = acc;
        } else {
            if (!aux_up) continue;
            const int rc = c - F;
            aux_up[((size_t)b * L + t) * R + rc] = aux_frames[((size_t)b * T + i) * R + rc];
        }
    }
}

hipError_t wrnn_launch_materialize(const wrnn_handle *h, const float *mels, const float *aux_frames, int B,
                                   int T, int mel_T, int mel_off, float *up, float *aux_up, hipStream_t s) {
    (void)hipGetLastError();  // the runtime is shared with PyTorch: drop any stale sticky error of this thread
    const WrnnDims &d = h->d;
    const long L = (long)T * d.HOP;
    dim3 grid((unsigned)((L + 63) / 64), B);
    hipLaunchKernelGGL(materialize_kernel, grid, dim3(256), 0, s, h->wdev, h->off, d, mels, aux_frames, T, mel_T, mel_off, up,
                       aux_up);
    return hipGetLastError();
}

// Small per-frame linear maps that push the conditioning through the layers it feeds
// (team kernel only): out[b][f][n] = bias[n] + sum_k in(b,f,k) * Wt[k][n].
// grid (ceil(frames/8), ceil(N/128), B), block 128: 8 frames x 128 outputs per workgroup.
template <int MODE>
__global__ void __launch_bounds__(128)
frame_linear_kernel(const float *__restrict__ src, size_t src_bstride, int ld, int valid, const float *__restrict__ Wt,
                    int ldw, const float *__restrict__ bias, float *__restrict__ out, size_t out_bstride, int frames,
                    int K, int N, int T, int P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xin = (float *)smem;  // [8][K]
    const int f0 = blockIdx.x * 8, n = blockIdx.y * 128 + threadIdx.x, b = blockIdx.z;
    const float *sb = src + (size_t)b * src_bstride;
    for (int i = threadIdx.x; i < 8 * K; i += blockDim.x) {
        const int ff = i / K, k = i - ff * K, f = f0 + ff;
        float v = 0.0f;
        if (f < frames) {
            if (MODE == 0) { if (f < valid) v = sb[(size_t)f * ld + k]; }
            else { const int fr = f - P; if (fr >= 0 && fr < T) v = sb[(size_t)k * T + fr]; }
        }
        xin[i] = v;
    }
    __syncthreads();
    if (n >= N) return;
    float acc[8];
#pragma unroll
    for (int ff = 0; ff < 8; ++ff) acc[ff] = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float wv = Wt[(size_t)k * ldw + n];
#pragma unroll
        for (int ff = 0; ff < 8; ++ff) acc[ff] = fmaf(wv, xin[ff * K + k], acc[ff]);
    }
    const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int ff = 0; ff < 8; ++ff)
        if (f0 + ff < frames) out[(size_t)b * out_bstride + (size_t)(f0 + ff) * N + n] = acc[ff] + bv;
}

hipError_t wrnn_launch_frame_linear(int mode, const float *src, size_t src_bstride, int ld, int valid, const float *Wt,
                                    int ldw, const float *bias, float *out, size_t out_bstride, int frames, int K,
                                    int N, int B, int T, int P, hipStream_t s) {
    (void)hipGetLastError();  // the runtime is shared with PyTorch: drop any stale sticky error of this thread
    dim3 grid((frames + 7) / 8, (N + 127) / 128, B);
    const size_t lds = (size_t)8 * K * sizeof(float);
    hipError_t e = hipFuncSetAttribute(mode == 0 ? (const void *)frame_linear_kernel<0> : (const void *)frame_linear_kernel<1>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (mode == 0)
        hipLaunchKernelGGL(frame_linear_kernel<0>, grid, dim3(128), lds, s, src, src_bstride, ld, valid, Wt, ldw, bias, out,
                           out_bstride, frames, K, N, T, P);
    else
        hipLaunchKernelGGL(frame_linear_kernel<1>, grid, dim3(128), lds, s, src, src_bstride, ld, valid, Wt, ldw, bias, out,
                           out_bstride, frames, K, N, T, P);
    return hipGetLastError();
}

// Team kernel conditioning records: REC[b][fi][j][28], fi in [0, T] (fi == T: zero conditioning, used for
// fold padding), j = hidden unit.  One record holds everything phase A of unit j needs during frame fi:
//   [0] CA[fi][j]   [1..3] VA[fi][r,z,n][j]   [4..8] CM[fi+dd][j], dd<5   [9+3dd+g] VM[fi+dd][g][j]   [24..27] pad
// (28 floats = 112 B: a conflict-free ds_read_b128 stride).  The 5-frame windows overlap, i.e. the tables are
// stored 5x redundantly in HBM so that the per-frame LDS refill is one straight 56 KB copy.
__global__ void __launch_bounds__(256)
pack_records_kernel(const float *__restrict__ CM, const float *__restrict__ CA, const float *__restrict__ VM,
                    const float *__restrict__ VA, float *__restrict__ rec, int T, int P) {
    const int fi = blockIdx.x, b = blockIdx.y;
    const int TP = T + 2 * P, T1 = T + 1;
    const float *cm = CM + (size_t)b * TP * 512, *ca = CA + (size_t)b * T1 * 512;
    const float *vm = VM + (size_t)b * TP * 1536, *va = VA + (size_t)b * T1 * 1536;
    float *out = rec + ((size_t)b * T1 + fi) * 512 * 28;
    const bool live = fi < T;
    for (int i = threadIdx.x; i < 512 * 28; i += blockDim.x) {
        const int j = i / 28, f = i - j * 28;
        float v = 0.0f;
        if (f == 0) v = ca[(size_t)fi * 512 + j];
        else if (f < 4) v = va[(size_t)fi * 1536 + (f - 1) * 512 + j];
        else if (f < 9) { if (live) v = cm[(size_t)(fi + f - 4) * 512 + j]; }
        else if (f < 24) { const int dd = (f - 9) / 3, g = (f - 9) - 3 * dd; if (live) v = vm[(size_t)(fi + dd) * 1536 + g * 512 + j]; }
        out[i] = v;
    }
}

hipError_t wrnn_launch_pack_records(const float *CM, const float *CA, const float *VM, const float *VA, float *rec, int B,
                                    int T, int P, hipStream_t s) {
    (void)hipGetLastError();  // the runtime is shared with PyTorch: drop any stale sticky error of this thread
    hipLaunchKernelGGL(pack_records_kernel, dim3(T + 1, B), dim3(256), 0, s, CM, CA, VM, VA, rec, T, P);
    return hipGetLastError();
}

// Per-step phase-A conditioning stream for WRNN_KERNEL_TEAM2: cond[row][t][j] = {cI, v_r, v_z, v_n} of hidden unit j
// at loop step t, i.e. the conditioning of the I layer and of GRU1's input gates with the (5,5,11) upsampling taps
// already applied:  c[t] = rec.A + sum_d k[phase(t)][d] * rec.M[d]   (records: pack_records_kernel).
// 8 KB per step, written once as a coalesced HBM stream (0.9 GB for a 5 s clip, ~0.2 ms at HBM speed) and read once
// per step by the 32 workgroups of a team (1 HBM read + 31 L2 hits).  grid (ceil(steps/64), rows), block 512 (= unit).
__global__ void __launch_bounds__(512)
cond_stream_kernel(const float *__restrict__ rec, const float *__restrict__ ktab, const WrnnRow *__restrict__ rows,
                   float4 *__restrict__ cond, int T, int HOP, long total_len, long seg0, long seg_len) {
    const int j = threadIdx.x, row = blockIdx.y;
    const WrnnRow rw = rows[row];
    const float *recb = rec + (size_t)rw.utt * (T + 1) * 512 * 28;
    // steps seg0 .. seg0+seg_len-1 of every row -> cond[row][t - seg0]
    const long t0 = seg0 + (long)blockIdx.x * 64;
    const long t1 = t0 + 64 < seg0 + seg_len ? t0 + 64 : seg0 + seg_len;
    int cur = -1;
    float4 a0, a1, a2, a3, a4, a5;
    a0 = a1 = a2 = a3 = a4 = a5 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long t = t0; t < t1; ++t) {
        const long pos = rw.start + t;
        const bool live = pos < total_len;            // fold padding 'after' = zero rows (fatchord_version.py:327-330)
        const int fi = live ? (int)(pos / HOP) : T;   // T = the all-zero conditioning record
        const int ph = live ? (int)(pos - (long)fi * HOP) : 0;
        if (fi != cur) {
            const float4 *r = (const float4 *)(recb + ((size_t)fi * 512 + j) * 28);
            a0 = r[0]; a1 = r[1]; a2 = r[2]; a3 = r[3]; a4 = r[4]; a5 = r[5];
            cur = fi;
        }
        const float k0 = ktab[ph * 5 + 0], k1 = ktab[ph * 5 + 1], k2 = ktab[ph * 5 + 2], k3 = ktab[ph * 5 + 3], k4 = ktab[ph * 5 + 4];
        // record: {CA, VAr, VAz, VAn | CM0..3 | CM4, VM0r, VM0z, VM0n | VM1r, VM1z, VM1n, VM2r | VM2z, VM2n, VM3r, VM3z | VM3n, VM4r, VM4z, VM4n}
        float4 c;
        c.x = fmaf(k4, a2.x, fmaf(k3, a1.w, fmaf(k2, a1.z, fmaf(k1, a1.y, fmaf(k0, a1.x, a0.x)))));
        c.y = fmaf(k4, a5.y, fmaf(k3, a4.z, fmaf(k2, a3.w, fmaf(k1, a3.x, fmaf(k0, a2.y, a0.y)))));
        c.z = fmaf(k4, a5.z, fmaf(k3, a4.w, fmaf(k2, a4.x, fmaf(k1, a3.y, fmaf(k0, a2.z, a0.z)))));
        c.w = fmaf(k4, a5.w, fmaf(k3, a5.x, fmaf(k2, a4.y, fmaf(k1, a3.z, fmaf(k0, a2.w, a0.w)))));
        cond[((size_t)row * seg_len + (t - seg0)) * 512 + j] = c;
    }
}

hipError_t wrnn_launch_cond_stream(const float *rec, const float *ktab, const WrnnRow *rows, float *cond, int n_rows, int T,
                                   int HOP, long total_len, long seg0, long seg_len, hipStream_t s) {
    (void)hipGetLastError();  // the runtime is shared with PyTorch: drop any stale sticky error of this thread
    dim3 grid((unsigned)((seg_len + 63) / 64), n_rows);
    hipLaunchKernelGGL(cond_stream_kernel, grid, dim3(512), 0, s, rec, ktab, rows, (float4 *)cond, T, HOP, total_len, seg0, seg_len);
    return hipGetLastError();
}

// Batch kernel conditioning records: REC32[b][fi][j][32] = the 24 floats of pack_records_kernel (same slots) followed by
// the per-frame constants of the layers behind phase A for hidden unit j: [24..26] C2[fi][r,z,n][j] (W_ih2[:,H:].a2 + b_ih2),
// [27] C3[fi][j] (fc1), [28] C4[fi][j] (fc2), [29..31] pad.  One 128-byte record is everything a workgroup needs per
// (frame, unit): the batch kernel reads 16 units x R rows of them per step instead of an 8 KB/step conditioning stream.
__global__ void __launch_bounds__(256)
pack_records32_kernel(const float *__restrict__ CM, const float *__restrict__ CA, const float *__restrict__ VM,
                      const float *__restrict__ VA, const float *__restrict__ C2, const float *__restrict__ C3,
                      const float *__restrict__ C4, float *__restrict__ rec, int T, int P) {
    const int fi = blockIdx.x, b = blockIdx.y;
    const int TP = T + 2 * P, T1 = T + 1;
    const float *cm = CM + (size_t)b * TP * 512, *ca = CA + (size_t)b * T1 * 512;
    const float *vm = VM + (size_t)b * TP * 1536, *va = VA + (size_t)b * T1 * 1536;
    const float *c2 = C2 + (size_t)b * T1 * 1536, *c3 = C3 + (size_t)b * T1 * 512, *c4 = C4 + (size_t)b * T1 * 512;
    float *out = rec + ((size_t)b * T1 + fi) * 512 * 32;
    const bool live = fi < T;
    for (int i = threadIdx.x; i < 512 * 32; i += blockDim.x) {
        const int j = i >> 5, f = i & 31;
        float v = 0.0f;
        if (f == 0) v = ca[(size_t)fi * 512 + j];
        else if (f < 4) v = va[(size_t)fi * 1536 + (f - 1) * 512 + j];
        else if (f < 9) { if (live) v = cm[(size_t)(fi + f - 4) * 512 + j]; }
        else if (f < 24) { const int dd = (f - 9) / 3, g = (f - 9) - 3 * dd; if (live) v = vm[(size_t)(fi + dd) * 1536 + g * 512 + j]; }
        else if (f < 27) v = c2[(size_t)fi * 1536 + (f - 24) * 512 + j];
        else if (f == 27) v = c3[(size_t)fi * 512 + j];
        else if (f == 28) v = c4[(size_t)fi * 512 + j];
        out[i] = v;
    }
}

hipError_t wrnn_launch_pack_records32(const float *CM, const float *CA, const float *VM, const float *VA, const float *C2,
                                      const float *C3, const float *C4, float *rec, int B, int T, int P, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(pack_records32_kernel, dim3(T + 1, B), dim3(256), 0, s, CM, CA, VM, VA, C2, C3, C4, rec, T, P);
    return hipGetLastError();
}

// The loop's row table, built on the device (no host staging buffer, no synchronisation in wrnn_generate):
// unbatched: row r = utterance r from position 0; batched (fold_with_overlap :332-338): row r = utterance 0 from
// position r * (target + overlap).
__global__ void rows_kernel(WrnnRow *rows, int32_t *order, int32_t *sched, int n_rows, int n_teams, int batched, long stride, long steps,
                            const int32_t *frames, int T, int hop) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_slots = (n_rows + n_teams - 1) / n_teams * n_teams;
    if (r >= n_rows) {
        // empty slots of the last (partial) pass: slot p * n_teams + team with no row behind it
        if (r < n_slots) {
            const int pass = r / n_teams, pos = r - pass * n_teams;
            const int src = pass * n_teams + ((frames && (pass & 1)) ? n_teams - 1 - pos : pos);
            if (src >= n_rows) sched[r] = -1;
        }
        return;
    }
    auto len = [&](int i) -> long {
        if (!frames) return steps;
        int f = frames[i];
        f = f < 1 ? 1 : (f > T ? T : f);
        return (long)f * hop;
    };
    WrnnRow w;
    w.utt = batched ? 0 : r;
    const long mine = len(r);
    w.steps = (int32_t)mine;
    w.start = batched ? (int64_t)r * stride : 0;
    rows[r] = w;
    // schedule order: longest first, ties by index (a rank scan: n_rows is at most a few thousand utterances)
    int rank = r;
    if (frames) {
        rank = 0;
        for (int j = 0; j < n_rows; ++j) {
            const long lj = len(j);
            rank += (lj > mine || (lj == mine && j < r)) ? 1 : 0;
        }
    }
    order[rank] = r;
    // TEAM2: rank k goes to team k % n_teams on even passes, to the mirrored team on odd passes of a ragged batch
    const int pass = rank / n_teams, pos = rank - pass * n_teams;
    sched[pass * n_teams + ((frames && (pass & 1)) ? n_teams - 1 - pos : pos)] = r;
    // a thread past n_rows handles its own empty slot above; slots < n_rows that no rank maps to exist only in the last pass
    if (r < n_slots) {
        const int p2 = r / n_teams, q2 = r - p2 * n_teams;
        const int src = p2 * n_teams + ((frames && (p2 & 1)) ? n_teams - 1 - q2 : q2);
        if (src >= n_rows) sched[r] = -1;
    }
}

hipError_t wrnn_launch_rows(WrnnRow *rows, int32_t *order, int32_t *sched, int n_rows, int n_teams, int batched, long stride, long steps,
                            const int32_t *frames, int T, int hop, hipStream_t s) {
    (void)hipGetLastError();
    if (n_teams < 1) n_teams = 1;
    const int n_slots = (n_rows + n_teams - 1) / n_teams * n_teams;
    hipLaunchKernelGGL(rows_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, s, rows, order, sched, n_rows, n_teams, batched, stride, steps,
                       frames, T, hop);
    return hipGetLastError();
}
