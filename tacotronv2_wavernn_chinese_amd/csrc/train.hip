// wrnn_train_step: WaveRNN.forward (fatchord_version.py:131-167) + the training script's loss (wavernn_train.py:82,112-121,
// wavernn/utils/distribution.py:16-84) + the BACKWARD pass through the loop layers, i.e. what `loss.backward()` computes for
// I, rnn1, rnn2, fc1, fc2, fc3 and for the conditioning tensors (mels_up, aux) the upsample network produced.
//
// Training is teacher-forced: every input x_t is known up front, so the layers are NOT evaluated sample by sample like
// generate() -- everything except the two recurrences is one batched fp32 GEMM over all B*L (batch, step) pairs:
//   XI  = [x | mels | a1] . W_I^T + b_I              GI1 = XI . W_ih1^T + b_ih1
//   H1  = GRU1 recurrence over t (h_t = cell(GI1[t], h_{t-1}))            X2 = XI + H1
//   GI2 = [X2 | a2] . W_ih2^T + b_ih2                H2 = GRU2 recurrence  X3 = X2 + H2
//   F1  = relu([X3 | a3] . W_fc1^T + b1)   F2 = relu([F1 | a4] . W_fc2^T + b2)   Y = F2 . W_fc3^T + b3
// and the same in reverse (dW = dOut^T . In, dIn = dOut . W) with BPTT through the two recurrences.  The concatenations are
// never materialised: a layer over [x | a] is two GEMMs on column blocks of the weight.
//
// Kernels here:
//   sgemm_kernel<TA,TB>   C = (beta) C + A.B (+ bias) (relu): 128x128x16 block tile, 4 waves x (2x2) v_mfma_f32_32x32x2f32 tiles,
//                         operands staged through LDS k-major (conflict-free operand reads), register double-buffered global loads,
//                         optional split-K over blockIdx.z (weight gradients).  fp32 in, fp32 accumulate: the bound is the fp32
//                         matrix peak (157 TFLOP/s); measured 111 TFLOP/s (forward / input gradients), 65 (weight gradients).
//   the recurrences       train_team.hip: one persistent XCD-team kernel per recurrence and direction (rnn_dims 512).  Fallback here:
//   gru_fwd_step_kernel   one launch per time step: gh = h_{t-1} . W_hh^T for 4 hidden units x 32 batch rows per workgroup
//                         (weights + h_{t-1} through LDS, v_mfma_f32_16x16x4f32), gates, h_t; saves r, z, n, gh_n, h for the backward.
//   gru_bwd_step_kernel   one launch per time step, t = L-1 .. 0: carry = dGH_{t+1} . W_hh (+ dH_{t+1} z), gate derivatives of step t.
//                         The step launches are captured ONCE per (B, L) into hipGraphs and replayed.
//   ce_grad / mol_grad    d(mean loss) / d(fc3 outputs) (the MOL one in double); col_sum (bias gradients); small elementwise helpers.
// Entry points: wrnn_train_step (forward + the script's loss + backward in one call), wrnn_train_forward / wrnn_train_backward (the same
// split where autograd splits it), wrnn_sync_status.
#include <cmath>
#include <cstdarg>
#include <cstdio>

#include "wrnn_internal.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

#define TEAM_H 512   // rnn_dims the team recurrence kernels are built for
#define GT_M 128
#define GT_N 128
#define GT_K 16
#define GT_LD (GT_M + 4)
#ifndef TN_WG_TARGET
#define TN_WG_TARGET 768   // workgroups a split-K weight-gradient GEMM aims for (slices x tiles): 3 per CU = the kernel's occupancy
                           // (measured per iteration: 256 -> 26.5 ms, 512 -> 25.5, 768 -> 25.3, 1024 -> 26.4)
#endif

// A(m,k) = TA ? A[k*lda + m] : A[m*lda + k];  B(k,n) = TB ? B[n*ldb + k] : B[k*ldb + n]
template <bool TA, bool TB>
__global__ void __launch_bounds__(256) sgemm_kernel(const float *__restrict__ A, long lda, const float *__restrict__ B, long ldb,
                                                    float *__restrict__ C, long ldc, int M, int N, int K, int beta,
                                                    const float *__restrict__ bias, int relu, int kper, long cslice) {
    // split K (weight gradients: K = B*L, small output): blockIdx.z owns k in [z * kper, (z + 1) * kper) and writes its partial
    // product to C + z * cslice; kper == 0: the whole K range
    if (kper > 0) {
        const long k_lo = (long)blockIdx.z * kper;
        A += TA ? k_lo * lda : k_lo;
        B += TB ? k_lo : k_lo * ldb;
        C += (long)blockIdx.z * cslice;
        K = K - k_lo < kper ? (int)(K - k_lo) : kper;
        if (K < 0) K = 0;
    }
    __shared__ float As[GT_K][GT_LD];
    __shared__ float Bs[GT_K][GT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)blockIdx.y * GT_M, n0 = (long)blockIdx.x * GT_N;
    float ra[8], rb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (TA) {   // m contiguous
                const int m = tid & 127, k = (tid >> 7) + 2 * i;
                const long gm = m0 + m;
                const int gk = k0 + k;
                ra[i] = (gm < M && gk < K) ? A[(long)gk * lda + gm] : 0.0f;
            } else {    // k contiguous
                const int k = tid & 15, m = (tid >> 4) + 16 * i;
                const long gm = m0 + m;
                const int gk = k0 + k;
                ra[i] = (gm < M && gk < K) ? A[gm * lda + gk] : 0.0f;
            }
            if (TB) {   // k contiguous
                const int k = tid & 15, n = (tid >> 4) + 16 * i;
                const long gn = n0 + n;
                const int gk = k0 + k;
                rb[i] = (gn < N && gk < K) ? B[gn * ldb + gk] : 0.0f;
            } else {    // n contiguous
                const int n = tid & 127, k = (tid >> 7) + 2 * i;
                const long gn = n0 + n;
                const int gk = k0 + k;
                rb[i] = (gn < N && gk < K) ? B[(long)gk * ldb + gn] : 0.0f;
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (TA) As[(tid >> 7) + 2 * i][tid & 127] = ra[i];
            else As[tid & 15][(tid >> 4) + 16 * i] = ra[i];
            if (TB) Bs[tid & 15][(tid >> 4) + 16 * i] = rb[i];
            else Bs[(tid >> 7) + 2 * i][tid & 127] = rb[i];
        }
    };
    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += GT_K) {
        __syncthreads();
        stash();
        __syncthreads();
        if (k0 + GT_K < K) fetch(k0 + GT_K);
        const int r = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < GT_K; kk += 2) {
            const float a0 = As[kk + kh][wm * 64 + r], a1 = As[kk + kh][wm * 64 + 32 + r];
            const float b0 = Bs[kk + kh][wn * 64 + r], b1 = Bs[kk + kh][wn * 64 + 32 + r];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // D[i][j] of a 32x32 tile: j = lane % 32, i = 8 * (v / 4) + 4 * (lane / 32) + v % 4
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const long gn = n0 + wn * 64 + tj * 32 + (lane & 31);
            if (gn >= N) continue;
            const float bv = bias ? bias[gn] : 0.0f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long gm = m0 + wm * 64 + ti * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
                if (gm >= M) continue;
                float o = acc[ti][tj][v] + bv;
                if (beta) o += C[gm * ldc + gn];
                if (relu) o = fmaxf(o, 0.0f);
                C[gm * ldc + gn] = o;
            }
        }
}

hipError_t gemm(hipStream_t s, bool ta, bool tb, const float *A, long lda, const float *B, long ldb, float *C, long ldc, int M, int N, int K,
                int beta = 0, const float *bias = nullptr, int relu = 0, int slices = 1, int kper = 0, long cslice = 0) {
    if (M <= 0 || N <= 0) return hipSuccess;
    dim3 grid((N + GT_N - 1) / GT_N, (M + GT_M - 1) / GT_M, slices);
    if (ta && tb) hipLaunchKernelGGL((sgemm_kernel<true, true>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, beta, bias, relu, kper, cslice);
    else if (ta) hipLaunchKernelGGL((sgemm_kernel<true, false>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, beta, bias, relu, kper, cslice);
    else if (tb) hipLaunchKernelGGL((sgemm_kernel<false, true>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, beta, bias, relu, kper, cslice);
    else hipLaunchKernelGGL((sgemm_kernel<false, false>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, beta, bias, relu, kper, cslice);
    return hipGetLastError();
}

// ---- small elementwise / reduction kernels --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) add_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ o, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}
// g *= (act > 0)   (backward of relu, :218,:221)
__global__ void __launch_bounds__(256) relu_bwd_kernel(float *__restrict__ g, const float *__restrict__ act, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && !(act[i] > 0.0f)) g[i] = 0.0f;
}
// dst[r][c] = src[c][r]   (rows x cols of dst)
__global__ void __launch_bounds__(256) transpose_kernel(const float *__restrict__ src, float *__restrict__ dst, int rows, int cols) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (bx + tx < rows && by + i < cols) t[i][tx] = src[(size_t)(by + i) * rows + bx + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (bx + i < rows && by + tx < cols) dst[(size_t)(bx + i) * cols + by + tx] = t[tx][i];
}
// out[n] = sum_m A[m * lda + n] in two stages with a fixed summation order: CS_CHUNKS row chunks x 64-column blocks of partial sums
// (double), then one thread per column adds the chunks in order.
#define CS_CHUNKS 64
__global__ void __launch_bounds__(256) col_sum_part_kernel(const float *__restrict__ A, long lda, long M, int N, double *__restrict__ part) {
    __shared__ double sh[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const long per = (M + CS_CHUNKS - 1) / CS_CHUNKS, m_lo = (long)blockIdx.y * per, m_hi = m_lo + per < M ? m_lo + per : M;
    double acc = 0.0;
    if (n < N)
        for (long m = m_lo + rl; m < m_hi; m += 4) acc += (double)A[m * lda + n];
    sh[rl][c] = acc;
    __syncthreads();
    if (rl == 0 && n < N) part[(size_t)blockIdx.y * N + n] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
}
__global__ void __launch_bounds__(256) col_sum_final_kernel(const double *__restrict__ part, int N, float *__restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    double acc = 0.0;
    for (int c = 0; c < CS_CHUNKS; ++c) acc += part[(size_t)c * N + n];
    out[n] = (float)acc;
}
// split-K epilogue: C = (beta) C + sum over slices (fixed order) of P[s]
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float *__restrict__ P, int S, long MN, int N, float *__restrict__ C, long ldc,
                                                            int beta) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float acc = 0.0f;
    for (int sl = 0; sl < S; ++sl) acc += P[(size_t)sl * MN + i];
    const long m = i / N, n = i - m * N;
    C[m * ldc + n] = beta ? C[m * ldc + n] + acc : acc;
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- GRU recurrence, forward: one launch per time step ---------------------------------------------------------------------
// nn.GRU, one layer (gate order r, z, n; get_gru_cell :273-279):  gi = GI[b, t] (input part incl. b_ih, precomputed),
// gh = W_hh h_{t-1} + b_hh;  r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h = (1 - z) n + z h_{t-1}.
// grid (H / 4, ceil(B / 32)), block 256.  A workgroup owns 4 hidden units (12 gate rows) for 32 batch rows:
//   gh[32 x 12] = h_{t-1}[32 x H] . W^T[H x 12]   on v_mfma_f32_16x16x4f32 (two 16-row tiles x one 16-column tile, 4 columns idle),
// K split over the 4 waves (the first version evaluated it with v_fma from LDS operands: LDS-read bound, 11.4 us per step; the MFMA
// takes one b32 LDS read per operand and K step).  Operands staged through LDS with row pitch H + 4 floats (conflict-free b32
// reads for "lane = row"), partial tiles of the 4 waves summed in wave order through LDS, then 128 threads = (row, unit) do the gates.
#define GR_UW 4
#define GR_HLD(H) ((H) + 4)
typedef float f4v __attribute__((ext_vector_type(4)));
// XCD-aware workgroup -> tile map.  Workgroups are dealt to the 8 XCDs round-robin (workgroup i runs on XCD i % 8) and every
// XCD has its own L2.  A step kernel's workgroup writes 4 consecutive floats per output row: with tile = blockIdx the 8
// workgroups that share a 128-byte line sit on 8 different XCDs and every line is written back in 8 partial pieces at the end of
// the kernel; with this map the workgroups of one XCD own one contiguous range of hidden units, i.e. whole lines.
__device__ __forceinline__ int xcd_tile(int bid, int nblk) { return (nblk & 7) ? bid : (bid & 7) * (nblk >> 3) + (bid >> 3); }
// HT = rnn_dims at compile time (the reference's 512: every load loop unrolls, all loads of a phase are in flight at once;
// with run-time trip counts the staging loop issues one L2 round trip per iteration: 14 us per step); HT = 0: any H % 16 == 0
template <int HT>
__global__ void __launch_bounds__(256) gru_fwd_step_kernel(const float *__restrict__ GI, const float *__restrict__ Whh, const float *__restrict__ bhh,
                                                           float *__restrict__ Hs /* (B, L, H) h_t */, float *__restrict__ HP /* (B, L, H) h_{t-1} */,
                                                           float *__restrict__ Rs, float *__restrict__ Zs, float *__restrict__ Ns,
                                                           float *__restrict__ GHN, int B, long L, int Hrt, long t) {
    const int H = HT ? HT : Hrt;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hld = GR_HLD(H);
    float *hs = sm;                                    // [32][H + 4]   h_{t-1}
    float *ws = sm + 32 * hld;                         // [12][H + 4]   W_hh rows g * 4 + u
    float *ps = ws + 12 * hld;                         // [4 waves][32 rows][16]  partial gh tiles
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const int j0 = xcd_tile(blockIdx.x, gridDim.x) * GR_UW, b0 = blockIdx.y * 32;
    // the (row, unit) threads of the gate phase request their inputs first: that round trip (every kernel starts with a cold L2:
    // the previous step's results were written back at its end) then runs under the staging instead of behind the MFMAs
    const int gb = tid & 31, gu = (tid >> 5) & 3;
    const bool gate_thread = tid < 128 && b0 + gb < B;
    float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f, bh_r = 0.f, bh_z = 0.f, bh_n = 0.f;
    if (gate_thread) {
        const float *gi = GI + ((size_t)(b0 + gb) * L + t) * 3 * H + j0 + gu;
        gi_r = gi[0]; gi_z = gi[H]; gi_n = gi[2 * H];
        bh_r = bhh[j0 + gu]; bh_z = bhh[H + j0 + gu]; bh_n = bhh[2 * H + j0 + gu];
    }
    {
        // staging with compile-time trip counts (8 rows and 3 weight rows per wave; `for (b = wv; b < 32; b += 4)` is not unrolled --
        // wv is a run-time value -- and then every iteration waits for its own L2 round trip: 11 of them were the kernel's 9 us)
        const int h4 = H / 4;
        if (HT) {
            constexpr int C4 = HT ? HT / 256 : 1;      // float4 per lane and row
            float4 th[8][C4], tw[3][C4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int b = wv + 4 * i;
                const float4 *src = (const float4 *)(HP + ((size_t)(b0 + b) * L + t) * H);
                const bool ok = b0 + b < B;
#pragma unroll
                for (int c = 0; c < C4; ++c) th[i][c] = ok ? src[ln + 64 * c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int row = wv + 4 * i;               // row = g * 4 + u
                const float4 *src = (const float4 *)(Whh + ((size_t)(row >> 2) * H + j0 + (row & 3)) * H);
#pragma unroll
                for (int c = 0; c < C4; ++c) tw[i][c] = src[ln + 64 * c];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int c = 0; c < C4; ++c) ((float4 *)(hs + (wv + 4 * i) * hld))[ln + 64 * c] = th[i][c];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < C4; ++c) ((float4 *)(ws + (wv + 4 * i) * hld))[ln + 64 * c] = tw[i][c];
        } else {
            for (int b = wv; b < 32; b += 4) {
                const float4 *src = (const float4 *)(HP + ((size_t)(b0 + b) * L + t) * H);
                float4 *dst = (float4 *)(hs + b * hld);
                const bool ok = b0 + b < B;
                for (int c = ln; c < h4; c += 64) dst[c] = ok ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            for (int row = wv; row < 3 * GR_UW; row += 4) {
                const float4 *src = (const float4 *)(Whh + ((size_t)(row >> 2) * H + j0 + (row & 3)) * H);
                float4 *dst = (float4 *)(ws + row * hld);
                for (int c = ln; c < h4; c += 64) dst[c] = src[c];
            }
        }
    }
    __syncthreads();
    {
        // wave wv: k in [wv * H/4, (wv + 1) * H/4).  A lane (m = ln % 16, kk = ln / 16) = h[tile * 16 + m][k + kk];
        // B lane (kk, n = ln % 16) = W[n][k + kk] (n < 12, else 0);  D lane: rows 4 * (ln / 16) + i, column ln % 16
        const int m = ln & 15, kk = ln >> 4, kq = H / 4;
        const float *a0p = hs + m * hld + wv * kq + kk, *a1p = a0p + 16 * hld;
        const float *bp = ws + (m < 12 ? m : 0) * hld + wv * kq + kk;
        const bool bz = m >= 12;
        f4v d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int k = 0; k < kq; k += 4) {
            const float bv = bz ? 0.0f : bp[k];
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0p[k], bv, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1p[k], bv, d1, 0, 0, 0);
        }
        float *pw = ps + wv * 512;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pw[(4 * kk + i) * 16 + m] = d0[i];
            pw[(16 + 4 * kk + i) * 16 + m] = d1[i];
        }
    }
    __syncthreads();
    if (gate_thread) {
        const int b = gb, u = gu;
        float gh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int o = b * 16 + g * 4 + u;
            gh[g] = (ps[o] + ps[512 + o]) + (ps[1024 + o] + ps[1536 + o]);
        }
        const int j = j0 + u;
        const size_t row = (size_t)(b0 + b) * L + t;
        const float ghr = gh[0] + bh_r, ghz = gh[1] + bh_z, ghn = gh[2] + bh_n;
        const float r = sigm(gi_r + ghr), z = sigm(gi_z + ghz);
        const float n = tanhf(gi_n + r * ghn);
        const float hprev = hs[b * hld + j];
        const float h = (1.0f - z) * n + z * hprev;
        Hs[row * H + j] = h;
        if (t + 1 < L) HP[(row + 1) * H + j] = h;
        Rs[row * H + j] = r; Zs[row * H + j] = z; Ns[row * H + j] = n; GHN[row * H + j] = ghn;
    }
}

// ---- GRU recurrence, backward: one launch per time step, t = L-1 .. 0 ------------------------------------------------------
// dH_t = dHext[b, t] (from the layers above: x_out = x_in + h) + carry_t,   carry_t = dH_{t+1} z_{t+1} + dGH_{t+1} . W_hh
// dn = dH (1 - z), dz = dH (h_{t-1} - n), dpn = dn (1 - n^2), dr = dpn gh_n, dpr = dr r (1 - r), dpz = dz z (1 - z)
// dGI_t = [dpr, dpz, dpn]   dGH_t = [dpr, dpz, dpn r]   CD_t = dH z      (dGI / dGH feed the batched weight / input GEMMs)
// WhhT = W_hh transposed, [H][3H]: the workgroup's 4 units are 4 contiguous rows.  Same grid as the forward step;
// carry[32 x 4] = dGH_{t+1}[32 x 3H] . WhhT^T[3H x 4] on the same MFMA shape (12 of 16 columns idle: the chip has more CUs than
// this launch has workgroups, so the idle columns cost no time), dGH staged in chunks of GB_KC columns.
#define GB_KC 768
template <int HT>
__global__ void __launch_bounds__(256) gru_bwd_step_kernel(const float *__restrict__ dHext, const float *__restrict__ WhhT,
                                                           const float *__restrict__ HP, const float *__restrict__ Rs, const float *__restrict__ Zs,
                                                           const float *__restrict__ Ns, const float *__restrict__ GHN, float *__restrict__ dGI,
                                                           float *__restrict__ dGH, float *__restrict__ CD /* (B, H) */, int B, long L, int Hrt,
                                                           long t) {
    const int H = HT ? HT : Hrt;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int G = 3 * H, gld = GB_KC + 4, wld = G + 4;
    float *gs = sm;                                    // [32][GB_KC + 4] chunk of dGH_{t+1}
    float *ws = sm + 32 * gld;                         // [4 units][3H + 4]
    float *ps = ws + GR_UW * wld;                      // [4 waves][32 rows][4 units]
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const int j0 = xcd_tile(blockIdx.x, gridDim.x) * GR_UW, b0 = blockIdx.y * 32;
    const bool have = t + 1 < L;
    // inputs of the gate-derivative phase requested first (see gru_fwd_step_kernel)
    const int gb = tid & 31, gu = (tid >> 5) & 3;
    const bool gate_thread = tid < 128 && b0 + gb < B;
    float e_dh = 0.f, e_r = 0.f, e_z = 0.f, e_n = 0.f, e_ghn = 0.f, e_hp = 0.f, e_cd = 0.f;
    if (gate_thread) {
        const size_t o = ((size_t)(b0 + gb) * L + t) * H + j0 + gu;
        e_dh = dHext[o]; e_r = Rs[o]; e_z = Zs[o]; e_n = Ns[o]; e_ghn = GHN[o]; e_hp = HP[o];
        if (have) e_cd = CD[(size_t)(b0 + gb) * H + j0 + gu];
    }
    if (have) {
        {
            const float4 *src = (const float4 *)(WhhT + (size_t)(j0 + wv) * G);   // wave u stages its own unit's row
            float4 *dst = (float4 *)(ws + wv * wld);
            if (HT) {
                constexpr int W4 = HT ? 3 * HT / 256 : 1;
                float4 tw[W4];
#pragma unroll
                for (int c = 0; c < W4; ++c) tw[c] = src[ln + 64 * c];
#pragma unroll
                for (int c = 0; c < W4; ++c) dst[ln + 64 * c] = tw[c];
            } else {
                for (int c = ln; c < G / 4; c += 64) dst[c] = src[c];
            }
        }
        const int m = ln & 15, kk = ln >> 4;
        const bool bz = m >= GR_UW;
        f4v d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        auto mfma_chunk = [&](int c0, int cw) {
            const int kq = cw / 4;                     // this wave's share of the chunk (cw % 16 == 0)
            const float *a0p = gs + m * gld + wv * kq + kk, *a1p = a0p + 16 * gld;
            const float *bp = ws + (bz ? 0 : m) * wld + c0 + wv * kq + kk;
#pragma unroll 8
            for (int k = 0; k < kq; k += 4) {
                const float bv = bz ? 0.0f : bp[k];
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0p[k], bv, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1p[k], bv, d1, 0, 0, 0);
            }
        };
        if (HT) {
            // compile-time trip counts, and BOTH chunks requested before the first LDS store: one memory round trip for the whole
            // dGH_{t+1} block (a `for (bb = wv; ...)` loop is not unrolled -- wv is a run-time value -- and waits per iteration)
            constexpr int NCH = HT ? 3 * HT / GB_KC : 1, K4 = GB_KC / 256;
            static_assert(HT == 0 || (3 * HT) % GB_KC == 0, "3H must be a multiple of the chunk");
            float4 tg[NCH][8][K4];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int bb = wv + 4 * i;
                    const float4 *src = (const float4 *)(dGH + ((size_t)(b0 + bb) * L + t + 1) * G + ch * GB_KC);
                    const bool ok = b0 + bb < B;
#pragma unroll
                    for (int c = 0; c < K4; ++c) tg[ch][i][c] = ok ? src[ln + 64 * c] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int c = 0; c < K4; ++c) ((float4 *)(gs + (wv + 4 * i) * gld))[ln + 64 * c] = tg[ch][i][c];
                __syncthreads();
                mfma_chunk(ch * GB_KC, GB_KC);
            }
        } else {
            for (int c0 = 0; c0 < G; c0 += GB_KC) {
                const int cw = G - c0 < GB_KC ? G - c0 : GB_KC;
                __syncthreads();
                for (int bb = wv; bb < 32; bb += 4) {
                    const float4 *src = (const float4 *)(dGH + ((size_t)(b0 + bb) * L + t + 1) * G + c0);
                    float4 *dst = (float4 *)(gs + bb * gld);
                    const bool ok = b0 + bb < B;
                    for (int c = ln; c < cw / 4; c += 64) dst[c] = ok ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                __syncthreads();
                mfma_chunk(c0, cw);
            }
        }
        if (m < GR_UW) {
            float *pw = ps + wv * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pw[(4 * kk + i) * 4 + m] = d0[i];
                pw[(16 + 4 * kk + i) * 4 + m] = d1[i];
            }
        }
    }
    __syncthreads();
    if (gate_thread) {
        const int b = gb, u = gu;
        const int j = j0 + u;
        const size_t row = (size_t)(b0 + b) * L + t;
        float carry = 0.0f;
        if (have) {
            const int o = b * 4 + u;
            carry = ((ps[o] + ps[128 + o]) + (ps[256 + o] + ps[384 + o])) + e_cd;
        }
        const float dH = e_dh + carry;
        const float r = e_r, z = e_z, n = e_n, ghn = e_ghn, hprev = e_hp;
        const float dn = dH * (1.0f - z), dz = dH * (hprev - n);
        const float dpn = dn * (1.0f - n * n);
        const float dpr = (dpn * ghn) * r * (1.0f - r), dpz = dz * z * (1.0f - z);
        float *gi = dGI + row * G, *gh = dGH + row * G;
        gi[j] = dpr; gi[H + j] = dpz; gi[2 * H + j] = dpn;
        gh[j] = dpr; gh[H + j] = dpz; gh[2 * H + j] = dpn * r;
        CD[(size_t)(b0 + b) * H + j] = dH * z;
    }
}

// ---- loss gradients: d(mean loss) / d(fc3 output) --------------------------------------------------------------------------
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// F.cross_entropy (mean): dY = (softmax(row) - onehot(y)) / n_rows.  One wave per row.
__global__ void __launch_bounds__(256) ce_grad_kernel(const float *__restrict__ logits, const int32_t *__restrict__ y, int NC, long n_rows,
                                                      float inv_n, float *__restrict__ dY) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float *p = logits + (size_t)row * NC;
    float m = -INFINITY;
    for (int c = lane; c < NC; c += 64) m = fmaxf(m, p[c]);
    m = wmax(m);
    float s = 0.0f;
    for (int c = lane; c < NC; c += 64) s += expf(p[c] - m);
    s = wsum(s);
    const int tgt = y[row];
    float *d = dY + (size_t)row * NC;
    for (int c = lane; c < NC; c += 64) d[c] = (expf(p[c] - m) / s - (c == tgt ? 1.0f : 0.0f)) * inv_n;
}
__device__ __forceinline__ double softplus_d(double x) { return x > 30.0 ? x : log1p(exp(x)); }
__device__ __forceinline__ double sigm_d(double x) { return 1.0 / (1.0 + exp(-x)); }
// discretized_mix_logistic_loss (distribution.py:16-84; num_classes 65536, log_scale_min log(1e-14), reduce=True -> mean):
// loss_row = -logsumexp_k(D_k + log_softmax(logit)_k);  w = softmax_k of that sum;  d/dlogit_j = softmax(logit)_j - w_j;
// D_k is the arm the reference's masks select (its blends multiply the other arm by 0):
//   y < -0.999: log s(plus)            dD/dplus = s(-plus)
//   y >  0.999: -softplus(min)         dD/dmin  = -s(min)
//   cdf_delta > 1e-5: log(cdf_delta)   dD/dplus = s'(plus) / cdf_delta, dD/dmin = -s'(min) / cdf_delta
//   else: mid - ls - 2 softplus(mid) - log((nc-1)/2)   dD/dmid = 1 - 2 s(mid), dD/dls (direct) = -1
// with plus/min/mid = exp(-ls) (y - mean +- 1/(nc-1) | 0): d/dmean = -exp(-ls), d/dls = -(value); ls = max(raw, ls_min) passes
// the gradient to raw where raw >= ls_min (torch.clamp).  One thread per row, evaluated in DOUBLE: cdf_delta is the difference of two
// sigmoids 1/65535 apart and its reciprocal scales the gradient -- in fp32 (the reference's arithmetic) that difference carries ~1e-3
// relative noise, which then dominates every parameter gradient of a MOL model (torch float32 vs float64: 1.1e-4 of the largest entry on
// I.weight, 3e-2 on d_mels_up for some batches).  The arm selection uses the fp32-rounded cdf_delta the forward loss kernel sees, so the
// gradient belongs to the loss value that is reported; 30 doubles per row cost nothing next to the GEMMs.
__global__ void __launch_bounds__(256) mol_grad_kernel(const float *__restrict__ y_hat, const float *__restrict__ yv, int nr, long n_rows,
                                                       float num_classes, float ls_min, float inv_n, float *__restrict__ dY) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= n_rows) return;
    const float *p = y_hat + (size_t)row * 3 * nr;
    float *d = dY + (size_t)row * 3 * nr;
    const double y = (double)yv[row];
    double lm = -INFINITY;
    for (int k = 0; k < nr; ++k) lm = fmax(lm, (double)p[k]);
    double lsum = 0.0;
    for (int k = 0; k < nr; ++k) lsum += exp((double)p[k] - lm);
    const double lse_logit = lm + log(lsum);
    const double hb = 1.0 / ((double)num_classes - 1.0), log_half = log(((double)num_classes - 1.0) / 2.0);
    double lp[16], dmean[16], dls[16];
    double mx = -INFINITY;
    for (int k = 0; k < nr; ++k) {
        const double mean = (double)p[nr + k], raw = (double)p[2 * nr + k];
        const double ls = fmax(raw, (double)ls_min);
        const double cy = y - mean, inv = exp(-ls);
        const double plus = inv * (cy + hb), mn = inv * (cy - hb), mid = inv * cy;
        const double sp = sigm_d(plus), sn = sigm_d(mn);
        const double cdf_delta = sp - sn;
        // the arm the FORWARD (fp32, losses.hip / the reference) takes: its cdf_delta is the fp32 difference of fp32 sigmoids
        const float invf = expf(-fmaxf(p[2 * nr + k], ls_min)), cyf = yv[row] - p[nr + k];
        const float cdf_f = 1.0f / (1.0f + expf(-(invf * (cyf + (float)hb)))) - 1.0f / (1.0f + expf(-(invf * (cyf - (float)hb))));
        double D, dplus = 0.0, dmin = 0.0, dmid = 0.0, ddirect = 0.0;
        if (yv[row] < -0.999f) { D = plus - softplus_d(plus); dplus = 1.0 - sp; }
        else if (yv[row] > 0.999f) { D = -softplus_d(mn); dmin = -sn; }
        else if (cdf_f > 1e-5f) { const double cd = fmax(cdf_delta, 1e-12); D = log(cd); dplus = sp * (1.0 - sp) / cd; dmin = -sn * (1.0 - sn) / cd; }
        else { D = mid - ls - 2.0 * softplus_d(mid) - log_half; dmid = 1.0 - 2.0 * sigm_d(mid); ddirect = -1.0; }
        dmean[k] = -inv * (dplus + dmin + dmid);
        dls[k] = (raw >= (double)ls_min) ? (-(dplus * plus + dmin * mn + dmid * mid) + ddirect) : 0.0;
        lp[k] = D + ((double)p[k] - lse_logit);
        mx = fmax(mx, lp[k]);
    }
    double s = 0.0;
    for (int k = 0; k < nr; ++k) s += exp(lp[k] - mx);
    for (int k = 0; k < nr; ++k) {
        const double w = exp(lp[k] - mx) / s;
        d[k] = (float)((exp((double)p[k] - lse_logit) - w) * (double)inv_n);
        d[nr + k] = (float)(-w * dmean[k] * (double)inv_n);
        d[2 * nr + k] = (float)(-w * dls[k] * (double)inv_n);
    }
}

}  // namespace

// Workspace of one (B, L) problem and the captured step graphs (owned by the handle).
struct WrnnTrainState {
    float *ws = nullptr;
    size_t ws_floats = 0;
    int B = 0;
    long L = 0;
    hipGraphExec_t g_fwd[2] = {nullptr, nullptr}, g_bwd[2] = {nullptr, nullptr};   // [GRU1, GRU2]
    const float *w_hh[2] = {nullptr, nullptr}, *b_hh[2] = {nullptr, nullptr};       // weight pointers baked into the graphs
    hipStream_t cap = nullptr;
    // persistent team kernels of the recurrences (train_team.hip): mailbox, team-formation words, residency verdict
    unsigned long long *mail = nullptr;
    unsigned *ctl = nullptr;
    int team_checked = 0;      // 0 = not yet, 1 = resident, -1 = not
    bool fwd_valid = false;    // the workspace holds the activations of a forward pass for (B, L)
};

void wrnn_train_state_free(WrnnTrainState *st) {
    if (!st) return;
    for (int i = 0; i < 2; ++i) {
        if (st->g_fwd[i]) (void)hipGraphExecDestroy(st->g_fwd[i]);
        if (st->g_bwd[i]) (void)hipGraphExecDestroy(st->g_bwd[i]);
    }
    if (st->cap) (void)hipStreamDestroy(st->cap);
    if (st->mail) (void)hipFree(st->mail);
    if (st->ctl) (void)hipFree(st->ctl);
    if (st->ws) (void)hipFree(st->ws);
    delete st;
}

namespace {

int tfail(wrnn_handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    h->err = buf;
    return code;
}
#define T_TRY(expr)                                                                                       \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) return tfail(h, WRNN_ERR_HIP, "wrnn_train_step: %s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

struct Gru {   // one recurrence's buffers
    float *GI, *H, *HP, *R, *Z, *N, *GHN, *dGI, *dGH, *CD, *WhhT;
};

// the L step launches of one recurrence as a graph (captured once per (B, L, weight pointers)), replayed on the caller's stream
template <class F>
hipError_t run_steps(WrnnTrainState *st, hipGraphExec_t *slot, bool rebuild, hipStream_t s, F launch_all) {
    hipError_t e;
    if (rebuild && *slot) { (void)hipGraphExecDestroy(*slot); *slot = nullptr; }
    if (!*slot) {
        if (!st->cap && (e = hipStreamCreateWithFlags(&st->cap, hipStreamNonBlocking)) != hipSuccess) return e;
        hipGraph_t g = nullptr;
        if ((e = hipStreamBeginCapture(st->cap, hipStreamCaptureModeThreadLocal)) != hipSuccess) return e;
        launch_all(st->cap);
        const hipError_t le = hipGetLastError();   // a launch refused inside the capture (e.g. an over-limit LDS size) is reported as itself
        if ((e = hipStreamEndCapture(st->cap, &g)) != hipSuccess) return le != hipSuccess ? le : e;
        if (le != hipSuccess) { if (g) (void)hipGraphDestroy(g); return le; }
        e = hipGraphInstantiate(slot, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return e;
    }
    return hipGraphLaunch(*slot, s);
}

}  // namespace

// phase bit 0: the forward pass (keeps every activation in the workspace); bit 1: the backward pass, from dY_ext (the gradient of
// the caller's loss w.r.t. the fc3 outputs) or, when that is null, from the gradient of the training script's own loss on y_dev
static int train_impl(wrnn_handle *h, int phase, const wrnn_loop_params *w, const wrnn_loop_params *g, const float *x_dev, const float *mels_up_dev,
                      const float *aux_dev, const void *y_dev, const float *dY_ext, int32_t B, int64_t L, float *loss_out_dev, float *logits_out_dev,
                      float *d_mels_up_dev, float *d_aux_dev, void *stream) {
    if (!h) return WRNN_ERR_INVALID;
    const bool do_fwd = (phase & 1) != 0, do_bwd = (phase & 2) != 0;
    if (!w || !x_dev || !mels_up_dev || !aux_dev || B < 1 || L < 1) return tfail(h, WRNN_ERR_INVALID, "wrnn_train_step: bad arguments");
    if (do_bwd && !g) return tfail(h, WRNN_ERR_INVALID, "wrnn_train_backward: no gradient outputs");
    if (do_bwd && !dY_ext && (!y_dev || !loss_out_dev)) return tfail(h, WRNN_ERR_INVALID, "wrnn_train_step: gradients need targets and a loss output");
    const WrnnDims &d = h->d;
    const int H = d.H, FC = d.FC, F = d.F, A = d.A, R = d.R, NC = d.NC;
    if (H % 16 != 0 || FC < 1) return tfail(h, WRNN_ERR_INVALID, "wrnn_train_step: rnn_dims must be a multiple of 16");
    const size_t lds_f = (size_t)(44 * GR_HLD(H) + 4 * 512) * sizeof(float), lds_b = (size_t)(32 * (GB_KC + 4) + GR_UW * (3 * H + 4) + 4 * 128) * sizeof(float);
    if (lds_f > 160u * 1024u || lds_b > 160u * 1024u)
        return tfail(h, WRNN_ERR_INVALID, "wrnn_train_step: rnn_dims too large for the step kernels' LDS tiles");
    T_TRY(hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)B * L;
    const int G = 3 * H, IN_I = 1 + F + A;
    if (!h->train) h->train = new WrnnTrainState();
    WrnnTrainState *st = h->train;
    // ---- workspace carve-up (floats) ----
    const size_t nH = (size_t)M * H, nG = (size_t)M * G, nF = (size_t)M * FC, nY = (size_t)M * NC;
    size_t need = 0;
    auto take = [&](size_t n) { size_t at = need; need += (n + 63) & ~(size_t)63; return at; };
    const size_t oXI = take(nH), oX2 = take(nH), oX3 = take(nH), oF1 = take(nF), oF2 = take(nF), oY = take(nY);
    size_t oG[2][11];
    for (int i = 0; i < 2; ++i) {
        oG[i][0] = take(nG); oG[i][1] = take(nH); oG[i][2] = take(nH); oG[i][3] = take(nH); oG[i][4] = take(nH); oG[i][5] = take(nH);
        oG[i][6] = take(nH); oG[i][7] = take(nG); oG[i][8] = take(nG); oG[i][9] = take((size_t)B * H); oG[i][10] = take((size_t)H * G);
    }
    const size_t odY = take(nY), odF2 = take(nF), odF1 = take(nF), odX3 = take(nH), odX2 = take(nH), odXI = take(nH);
    const int maxN = G > NC ? G : NC;
    const size_t sk_floats = (size_t)16 * G * (H + A);                      // split-K partial products (<= 16 slices of the largest dW)
    const size_t oCS = take((size_t)2 * CS_CHUNKS * maxN), oSK = take(sk_floats);
    const size_t oIMG = take((size_t)786432);                                // weight image of the recurrence at hand (team kernels)
    const bool fresh = need > st->ws_floats;
    if (!do_fwd && (fresh || !st->fwd_valid || st->B != B || st->L != L))
        return tfail(h, WRNN_ERR_STATE, "wrnn_train_backward: no matching wrnn_train_forward before it (same handle, B, L)");
    if (fresh) {
        if (st->ws) (void)hipFree(st->ws);
        st->ws = nullptr; st->ws_floats = 0;
        T_TRY(hipMalloc(&st->ws, need * sizeof(float)));
        st->ws_floats = need;
    }
    float *ws = st->ws;
    float *XI = ws + oXI, *X2 = ws + oX2, *X3 = ws + oX3, *F1 = ws + oF1, *F2 = ws + oF2, *Y = logits_out_dev ? logits_out_dev : ws + oY;
    Gru gr[2];
    for (int i = 0; i < 2; ++i)
        gr[i] = Gru{ws + oG[i][0], ws + oG[i][1], ws + oG[i][2], ws + oG[i][3], ws + oG[i][4], ws + oG[i][5], ws + oG[i][6], ws + oG[i][7],
                    ws + oG[i][8], ws + oG[i][9], ws + oG[i][10]};
    double *cs_part = (double *)(ws + oCS);
    float *sk_part = ws + oSK;
    float *dY = ws + odY, *dF2 = ws + odF2, *dF1 = ws + odF1, *dX3 = ws + odX3, *dX2 = ws + odX2, *dXI = ws + odXI;
    const float *whh[2] = {w->rnn1_w_hh, w->rnn2_w_hh}, *bhh[2] = {w->rnn1_b_hh, w->rnn2_b_hh};
    // the step graphs bake buffer and weight addresses: rebuild when the problem or the parameter storage changed
    const bool rebuild = fresh || st->B != B || st->L != L || st->w_hh[0] != whh[0] || st->w_hh[1] != whh[1] || st->b_hh[0] != bhh[0] ||
                         st->b_hh[1] != bhh[1];
    st->B = B; st->L = L;
    for (int i = 0; i < 2; ++i) { st->w_hh[i] = whh[i]; st->b_hh[i] = bhh[i]; }
    if (rebuild) {
        // ALL four graphs go at once: a forward-only call (wrnn_train_forward, or phase 1 of the split pass) re-captures only the
        // forward graphs, and a backward graph kept from the previous (B, L, workspace, weights) would replay with those baked in
        // -- stale grid, stale (possibly freed) pointers -- as soon as wrnn_train_backward follows (round-3 advisor finding)
        for (int i = 0; i < 2; ++i) {
            if (st->g_fwd[i]) { (void)hipGraphExecDestroy(st->g_fwd[i]); st->g_fwd[i] = nullptr; }
            if (st->g_bwd[i]) { (void)hipGraphExecDestroy(st->g_bwd[i]); st->g_bwd[i] = nullptr; }
        }
    }
    (void)hipGetLastError();
    if (do_fwd) T_TRY(hipMemsetAsync(h->err_dev, 0, 64, s));   // device error word of the team kernels (wrnn_sync_status)
    auto *fwd_k = H == 512 ? gru_fwd_step_kernel<512> : gru_fwd_step_kernel<0>;
    auto *bwd_k = H == 512 ? gru_bwd_step_kernel<512> : gru_bwd_step_kernel<0>;
    const dim3 sgrid(H / GR_UW, (B + 31) / 32);
    // The recurrences as ONE persistent team kernel each (train_team.hip) where that is possible: rnn_dims 512, 32-CU teams, all four
    // instantiations resident; otherwise the per-step kernels below, replayed from hipGraphs.
    int rpb = (B + h->n_teams - 1) / (h->n_teams > 0 ? h->n_teams : 1);
    if (rpb > 8) rpb = 8;
    const int nq = rpb <= 4 ? 1 : 2;
    if (st->team_checked == 0) {
        st->team_checked = -1;
        if (H == TEAM_H && h->team_ok && h->n_teams >= 1) {
            bool ok = true;
            for (int q = 1; q <= 2 && ok; ++q)
                for (int bw = 0; bw < 2 && ok; ++bw) {
                    int blocks = 0;
                    ok = wrnn_gru_team_occupancy(q, bw != 0, &blocks) == hipSuccess && blocks >= 1;
                }
            (void)hipGetLastError();
            if (ok) {
                const size_t mg = (size_t)h->n_teams * wrnn_gru_team_mail_granules(2, true);
                ok = hipMalloc(&st->mail, mg * sizeof(unsigned long long)) == hipSuccess && hipMalloc(&st->ctl, 256) == hipSuccess;
            }
            if (ok) st->team_checked = 1;
        }
    }
    const bool use_team = st->team_checked == 1 && H == TEAM_H && !h->train_force_steps;
    if (!use_team) {   // the per-step kernels' LDS sizes: only where they run (the attribute is per device: set per call, a host-side table write)
        T_TRY(hipFuncSetAttribute((const void *)fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
        T_TRY(hipFuncSetAttribute((const void *)bwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    }
    float *img = ws + oIMG;
    auto recur_team = [&](int i, bool bwd, const float *dHext) -> hipError_t {
        const Gru &q = gr[i];
        hipError_t e = wrnn_gru_team_pack(whh[i], img, bwd, s);
        if (e != hipSuccess) return e;
        WrnnGruTeamArgs ta{};
        ta.img = img; ta.bhh = bhh[i]; ta.GI = q.GI; ta.Hs = q.H; ta.HP = q.HP; ta.Rs = q.R; ta.Zs = q.Z; ta.Ns = q.N; ta.GHN = q.GHN;
        ta.dHext = dHext; ta.dGI = q.dGI; ta.dGH = q.dGH; ta.B = B; ta.L = L; ta.n_teams = h->n_teams; ta.rpb = rpb;
        ta.mail = st->mail; ta.ctl = st->ctl; ta.err = h->err_dev;
        if ((e = wrnn_team_gate_enter(h->cfg.device, s)) != hipSuccess) return e;   // team kernels of a device run one after the other
        e = hipMemsetAsync(st->mail, 0, (size_t)h->n_teams * wrnn_gru_team_mail_granules(nq, bwd) * sizeof(unsigned long long), s);
        if (e == hipSuccess) e = hipMemsetAsync(st->ctl, 0, 256, s);
        if (e == hipSuccess) e = wrnn_gru_team_launch(ta, nq, bwd, s);
        const hipError_t ge = wrnn_team_gate_leave(h->cfg.device, s);
        return e != hipSuccess ? e : ge;
    };
    const float *a1 = aux_dev, *a2 = aux_dev + A, *a3 = aux_dev + 2 * A, *a4 = aux_dev + 3 * A;   // aux channel split (:198-199)

    // ================= forward (:146-167) =================
    if (do_fwd) {
    st->fwd_valid = false;
    // x = I(cat[x, mels, a1])
    T_TRY(gemm(s, false, true, x_dev, 1, w->I_w, IN_I, XI, H, M, H, 1, 0, w->I_b));
    T_TRY(gemm(s, false, true, mels_up_dev, F, w->I_w + 1, IN_I, XI, H, M, H, F, 1));
    T_TRY(gemm(s, false, true, a1, R, w->I_w + 1 + F, IN_I, XI, H, M, H, A, 1));
    auto recur_fwd = [&](int i) -> hipError_t {
        const Gru &q = gr[i];
        hipError_t e = hipMemsetAsync(q.HP, 0, nH * sizeof(float), s);   // h_{-1} = 0 (:141-142); rows t > 0 are overwritten
        if (e != hipSuccess) return e;
        if (use_team) return recur_team(i, false, nullptr);
        return run_steps(st, &st->g_fwd[i], rebuild, s, [&](hipStream_t cs) {
            for (long t = 0; t < L; ++t)
                hipLaunchKernelGGL(fwd_k, sgrid, dim3(256), lds_f, cs, q.GI, whh[i], bhh[i], q.H, q.HP, q.R, q.Z, q.N, q.GHN, B, L, H, t);
        });
    };
    // rnn1 (:152-153)
    T_TRY(gemm(s, false, true, XI, H, w->rnn1_w_ih, H, gr[0].GI, G, M, G, H, 0, w->rnn1_b_ih));
    T_TRY(recur_fwd(0));
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((nH + 255) / 256)), dim3(256), 0, s, XI, gr[0].H, X2, (long)nH);
    // rnn2 over cat[x, a2] (:155-158)
    T_TRY(gemm(s, false, true, X2, H, w->rnn2_w_ih, H + A, gr[1].GI, G, M, G, H, 0, w->rnn2_b_ih));
    T_TRY(gemm(s, false, true, a2, R, w->rnn2_w_ih + H, H + A, gr[1].GI, G, M, G, A, 1));
    T_TRY(recur_fwd(1));
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((nH + 255) / 256)), dim3(256), 0, s, X2, gr[1].H, X3, (long)nH);
    // fc1, fc2 (relu), fc3 (:160-166)
    T_TRY(gemm(s, false, true, X3, H, w->fc1_w, H + A, F1, FC, M, FC, H, 0, w->fc1_b));
    T_TRY(gemm(s, false, true, a3, R, w->fc1_w + H, H + A, F1, FC, M, FC, A, 1, nullptr, 1));
    T_TRY(gemm(s, false, true, F1, FC, w->fc2_w, FC + A, F2, FC, M, FC, FC, 0, w->fc2_b));
    T_TRY(gemm(s, false, true, a4, R, w->fc2_w + FC, FC + A, F2, FC, M, FC, A, 1, nullptr, 1));
    T_TRY(gemm(s, false, true, F2, FC, w->fc3_w, FC, Y, NC, M, NC, FC, 0, w->fc3_b));
    if (y_dev && loss_out_dev)
        if (int rc = wrnn_loss(h, Y, y_dev, M, loss_out_dev, stream)) return rc;
    st->fwd_valid = true;
    }
    if (!do_bwd) return WRNN_OK;

    // ================= backward =================
    if (dY_ext) {
        dY = const_cast<float *>(dY_ext);   // read only below
    } else {
        const float inv_n = 1.0f / (float)M;
        if (d.mode == WRNN_MODE_RAW)
            hipLaunchKernelGGL(ce_grad_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, Y, (const int32_t *)y_dev, NC, M, inv_n, dY);
        else
            hipLaunchKernelGGL(mol_grad_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, Y, (const float *)y_dev, NC / 3, M, 65536.0f,
                               -32.23619130191664f, inv_n, dY);
    }
    auto colsum = [&](const float *Am, long lda, int N, float *out) {
        hipLaunchKernelGGL(col_sum_part_kernel, dim3((N + 63) / 64, CS_CHUNKS), dim3(256), 0, s, Am, lda, M, N, cs_part);
        hipLaunchKernelGGL(col_sum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, s, cs_part, N, out);
    };
    // weight gradients dW = dOut^T . In contract over all B*L rows with a small output: split K over slices so that the launch fills
    // the chip, partial products summed in slice order (deterministic)
    auto gemm_tn = [&](const float *Am, long lda, const float *Bm, long ldb, float *Cm, long ldc, int Mo, int No) -> hipError_t {
        const int tiles = ((Mo + GT_M - 1) / GT_M) * ((No + GT_N - 1) / GT_N);
        // slices x tiles <= TN_WG_TARGET workgroups, all co-resident (with ceil() 48 tiles x 6 slices = 288 workgroups ran as one full pass
        // over 256 CUs + a 12 % second one)
        int S = tiles >= TN_WG_TARGET / 2 ? 1 : TN_WG_TARGET / tiles;
        if (S > 64) S = 64;
        while (S > 1 && (size_t)S * Mo * No > sk_floats) --S;
        if (S <= 1) return gemm(s, true, false, Am, lda, Bm, ldb, Cm, ldc, Mo, No, (int)M);
        const long per = ((M + S - 1) / S + GT_K - 1) / GT_K * GT_K;
        // one launch, blockIdx.z = slice (slices past the end of K write zeros)
        hipError_t e = gemm(s, true, false, Am, lda, Bm, ldb, sk_part, No, Mo, No, (int)M, 0, nullptr, 0, S, (int)per, (long)Mo * No);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long)Mo * No + 255) / 256)), dim3(256), 0, s, sk_part, S, (long)Mo * No, No, Cm, ldc, 0);
        return hipGetLastError();
    };
    const unsigned eb_f = (unsigned)((nF + 255) / 256);
    // fc3
    T_TRY(gemm_tn(dY, NC, F2, FC, g->fc3_w, FC, NC, FC));
    colsum(dY, NC, NC, g->fc3_b);
    T_TRY(gemm(s, false, false, dY, NC, w->fc3_w, FC, dF2, FC, M, FC, NC));
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(eb_f), dim3(256), 0, s, dF2, F2, (long)nF);
    // fc2 over cat[F1, a4]
    T_TRY(gemm_tn(dF2, FC, F1, FC, g->fc2_w, FC + A, FC, FC));
    T_TRY(gemm_tn(dF2, FC, a4, R, g->fc2_w + FC, FC + A, FC, A));
    colsum(dF2, FC, FC, g->fc2_b);
    T_TRY(gemm(s, false, false, dF2, FC, w->fc2_w, FC + A, dF1, FC, M, FC, FC));
    if (d_aux_dev) T_TRY(gemm(s, false, false, dF2, FC, w->fc2_w + FC, FC + A, d_aux_dev + 3 * A, R, M, A, FC));
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(eb_f), dim3(256), 0, s, dF1, F1, (long)nF);
    // fc1 over cat[X3, a3]
    T_TRY(gemm_tn(dF1, FC, X3, H, g->fc1_w, H + A, FC, H));
    T_TRY(gemm_tn(dF1, FC, a3, R, g->fc1_w + H, H + A, FC, A));
    colsum(dF1, FC, FC, g->fc1_b);
    T_TRY(gemm(s, false, false, dF1, FC, w->fc1_w, H + A, dX3, H, M, H, FC));
    if (d_aux_dev) T_TRY(gemm(s, false, false, dF1, FC, w->fc1_w + H, H + A, d_aux_dev + 2 * A, R, M, A, FC));
    auto recur_bwd = [&](int i, const float *dHext) -> hipError_t {
        const Gru &q = gr[i];
        if (use_team) return recur_team(i, true, dHext);
        hipLaunchKernelGGL(transpose_kernel, dim3((H + 31) / 32, (G + 31) / 32), dim3(256), 0, s, whh[i], q.WhhT, H, G);   // WhhT[j][k] = W_hh[k][j]
        return run_steps(st, &st->g_bwd[i], rebuild, s, [&](hipStream_t cs) {
            for (long t = L - 1; t >= 0; --t)
                hipLaunchKernelGGL(bwd_k, sgrid, dim3(256), lds_b, cs, dHext, q.WhhT, q.HP, q.R, q.Z, q.N, q.GHN, q.dGI, q.dGH, q.CD, B, L,
                                   H, t);
        });
    };
    // rnn2: h2 enters x3 = x2 + h2, so dH2(ext) = dX3
    T_TRY(recur_bwd(1, dX3));
    T_TRY(gemm_tn(gr[1].dGI, G, X2, H, g->rnn2_w_ih, H + A, G, H));
    T_TRY(gemm_tn(gr[1].dGI, G, a2, R, g->rnn2_w_ih + H, H + A, G, A));
    T_TRY(gemm_tn(gr[1].dGH, G, gr[1].HP, H, g->rnn2_w_hh, H, G, H));
    colsum(gr[1].dGI, G, G, g->rnn2_b_ih);
    colsum(gr[1].dGH, G, G, g->rnn2_b_hh);
    T_TRY(hipMemcpyAsync(dX2, dX3, nH * sizeof(float), hipMemcpyDeviceToDevice, s));        // residual x3 = x2 + h2
    T_TRY(gemm(s, false, false, gr[1].dGI, G, w->rnn2_w_ih, H + A, dX2, H, M, H, G, 1));
    if (d_aux_dev) T_TRY(gemm(s, false, false, gr[1].dGI, G, w->rnn2_w_ih + H, H + A, d_aux_dev + A, R, M, A, G));
    // rnn1
    T_TRY(recur_bwd(0, dX2));
    T_TRY(gemm_tn(gr[0].dGI, G, XI, H, g->rnn1_w_ih, H, G, H));
    T_TRY(gemm_tn(gr[0].dGH, G, gr[0].HP, H, g->rnn1_w_hh, H, G, H));
    colsum(gr[0].dGI, G, G, g->rnn1_b_ih);
    colsum(gr[0].dGH, G, G, g->rnn1_b_hh);
    T_TRY(hipMemcpyAsync(dXI, dX2, nH * sizeof(float), hipMemcpyDeviceToDevice, s));        // residual x2 = xI + h1
    T_TRY(gemm(s, false, false, gr[0].dGI, G, w->rnn1_w_ih, H, dXI, H, M, H, G, 1));
    // I over cat[x, mels, a1]
    T_TRY(gemm_tn(dXI, H, x_dev, 1, g->I_w, IN_I, H, 1));
    T_TRY(gemm_tn(dXI, H, mels_up_dev, F, g->I_w + 1, IN_I, H, F));
    T_TRY(gemm_tn(dXI, H, a1, R, g->I_w + 1 + F, IN_I, H, A));
    colsum(dXI, H, H, g->I_b);
    if (d_mels_up_dev) T_TRY(gemm(s, false, false, dXI, H, w->I_w + 1, IN_I, d_mels_up_dev, F, M, F, H));
    if (d_aux_dev) T_TRY(gemm(s, false, false, dXI, H, w->I_w + 1 + F, IN_I, d_aux_dev, R, M, A, H));
    T_TRY(hipGetLastError());
    return WRNN_OK;
}

extern "C" int wrnn_train_step(wrnn_handle *h, const wrnn_loop_params *w, const wrnn_loop_params *g, const float *x_dev, const float *mels_up_dev,
                               const float *aux_dev, const void *y_dev, int32_t B, int64_t L, float *loss_out_dev, float *logits_out_dev,
                               float *d_mels_up_dev, float *d_aux_dev, void *stream) {
    return train_impl(h, g ? 3 : 1, w, g, x_dev, mels_up_dev, aux_dev, y_dev, nullptr, B, L, loss_out_dev, logits_out_dev, d_mels_up_dev, d_aux_dev, stream);
}

extern "C" int wrnn_train_forward(wrnn_handle *h, const wrnn_loop_params *w, const float *x_dev, const float *mels_up_dev, const float *aux_dev,
                                  int32_t B, int64_t L, float *logits_out_dev, void *stream) {
    if (h && !logits_out_dev) return tfail(h, WRNN_ERR_INVALID, "wrnn_train_forward: logits_out_dev is the result");
    return train_impl(h, 1, w, nullptr, x_dev, mels_up_dev, aux_dev, nullptr, nullptr, B, L, nullptr, logits_out_dev, nullptr, nullptr, stream);
}

extern "C" int wrnn_train_backward(wrnn_handle *h, const wrnn_loop_params *w, const wrnn_loop_params *g, const float *d_logits_dev,
                                   const float *x_dev, const float *mels_up_dev, const float *aux_dev, int32_t B, int64_t L,
                                   float *d_mels_up_dev, float *d_aux_dev, void *stream) {
    if (h && !d_logits_dev) return tfail(h, WRNN_ERR_INVALID, "wrnn_train_backward: d_logits_dev missing");
    return train_impl(h, 2, w, g, x_dev, mels_up_dev, aux_dev, nullptr, d_logits_dev, B, L, nullptr, nullptr, d_mels_up_dev, d_aux_dev, stream);
}

// Waits for `stream` and reports the device error word of the team kernels launched on this handle since the last call
// (WRNN_ERR_BUSY / WRNN_ERR_TIMEOUT): what wrnn_last_timing does for wrnn_generate, for callers of wrnn_train_step.
extern "C" int wrnn_sync_status(wrnn_handle *h, void *stream) {
    if (!h) return WRNN_ERR_INVALID;
    T_TRY(hipSetDevice(h->cfg.device));
    T_TRY(hipStreamSynchronize((hipStream_t)stream));
    unsigned errw = 0;
    T_TRY(hipMemcpy(&errw, h->err_dev, sizeof(errw), hipMemcpyDeviceToHost));
    if (errw == WRNN_DEVERR_BUSY)
        return tfail(h, WRNN_ERR_BUSY, "a team kernel's workgroups did not all become resident: the GPU is shared with another kernel (retry)");
    if (errw) return tfail(h, WRNN_ERR_TIMEOUT, "device-side bounded spin gave up (code %u)", errw);
    return WRNN_OK;
}

// developer / test switch: != 0 makes wrnn_train_step use the per-step kernels (hipGraph replay) even where the team kernels can run
extern "C" int wrnn_train_force_step_kernels(wrnn_handle *h, int32_t on) {
    if (!h) return WRNN_ERR_INVALID;
    h->train_force_steps = on != 0;
    return WRNN_OK;
}
