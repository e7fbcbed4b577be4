// Developer probe for the batched loop kernel (loop_batch.hip): v_mfma_f32_4x4x1_16b_f32 on gfx950.
//  (1) operand / result lane layout: 16 blocks b = lane>>2; A lane (b, i = lane&3) = A_b[i]; B lane (b, j = lane&3) = B_b[j];
//      D vgpr r of lane (b, j) = A_b[r] * B_b[j] + C  -- checked against that formula on random data;
//  (2) exactness: a K-chain of MFMAs against an fmaf chain in the same order (bitwise);
//  (3) issue rate: cycles per MFMA for 1 / 2 waves per SIMD with NCH independent accumulator chains,
//      with the B operand held in registers and with the B operand read from LDS (ds_read_b128 per 4 MFMAs);
//  (4) the kp reduction used after the K loop: 2 x permlane32_swap + 1 x permlane16_swap + 2 row_ror DPP adds;
//  (5) co-issue: NV independent v_fma_f32 (and one DPP add) of the SAME wave behind every MFMA -- does VALU work (the folds of the
//      previous phase, gates, noise) hide under the matrix pipe, or does it add to the 8 cycles per MFMA?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__global__ void layout_kernel(const float *a, const float *b, const float *c, float *d, int K) {
    const int l = threadIdx.x;
    f4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = c[r * 64 + l];
    for (int k = 0; k < K; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k * 64 + l], b[k * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[r * 64 + l] = acc[r];
}

template <int NCH, bool LDSB>
__global__ void __launch_bounds__(512) rate_kernel(float *o, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float xs[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) xs[i] = 1.0f + 1e-3f * (float)(i & 255);
    f4 acc[NCH];
    float w[32];
    for (int i = 0; i < 32; ++i) w[i] = o[threadIdx.x + i];
    for (int c = 0; c < NCH; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int S = 0; S < 8; ++S) {
            f4 bq[2];
            if (LDSB) {
                bq[0] = ((const f4 *)xs)[S * 64 + lane];
                bq[1] = ((const f4 *)xs)[(8 + S) * 64 + lane];
            } else {
                bq[0] = (f4){w[0], w[1], w[2], w[3]};
                bq[1] = (f4){w[4], w[5], w[6], w[7]};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[(4 * S + e) & 31], bq[c & 1][e], acc[c], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    o[threadIdx.x] = s;
    if (lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// the kp reduction: input 4 VGPRs d[i] with lane (kp = lane>>2, j = lane&3); output: lane (rho = lane>>4, *, j) holds
// sum over kp of d[unit(rho)], unit(rho) = {0, 2, 1, 3}[rho]
__global__ void reduce_kernel(const float *in, float *out) {
    const int l = threadIdx.x;
    float d0 = in[l], d1 = in[64 + l], d2 = in[128 + l], d3 = in[192 + l];
    u2v p = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), __float_as_uint(d1), false, false);
    const float s01 = __uint_as_float(p.x) + __uint_as_float(p.y);
    u2v q = __builtin_amdgcn_permlane32_swap(__float_as_uint(d2), __float_as_uint(d3), false, false);
    const float s23 = __uint_as_float(q.x) + __uint_as_float(q.y);
    u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
    float t = __uint_as_float(r.x) + __uint_as_float(r.y);
    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xf, 0xf, true));   // row_ror:4
    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xf, 0xf, true));   // row_ror:8
    out[l] = t;
}


// (5) one wave per SIMD, 3 accumulator chains, B from registers; after every MFMA: NV v_fma_f32 on independent registers and
// (DPP) one row_ror DPP add; scheduling barriers pin the order
template <int NV, bool DPP>
__global__ void __launch_bounds__(256) mix_kernel(float *o, unsigned long long *cyc, int iters) {
    f4 acc[3];
    float w[32], x[8];
    for (int i = 0; i < 32; ++i) w[i] = o[threadIdx.x + i];
    for (int i = 0; i < 8; ++i) x[i] = o[threadIdx.x + 40 + i];
    float dp = o[threadIdx.x + 50];
    const float k1 = o[threadIdx.x + 51], k2 = o[threadIdx.x + 52];
    for (int c = 0; c < 3; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[s], w[(s + 1 + c) & 31], acc[c], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v]) : "v"(k1), "v"(k2));
                if (DPP) asm volatile("s_nop 0\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(dp));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sacc = dp;
    for (int c = 0; c < 3; ++c) sacc += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    for (int i = 0; i < 8; ++i) sacc += x[i];
    o[threadIdx.x] = sacc;
    if (lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int NV, bool DPP>
void mix(float *d, unsigned long long *c) {
    const int iters = 500;
    hipLaunchKernelGGL((mix_kernel<NV, DPP>), dim3(1), dim3(256), 0, 0, d, c, iters);
    unsigned long long h[8];
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("co-issue: MFMA + %d v_fma_f32%s behind each: %.2f cycles per MFMA (wave 0)\n", NV, DPP ? " + 1 DPP add" : "", (double)h[0] / (iters * 96.0));
}

template <int NCH, bool LDSB>
void rate(const char *name, float *d, unsigned long long *c) {
    const int iters = 500;
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL((rate_kernel<NCH, LDSB>), dim3(1), dim3(threads), 0, 0, d, c, iters);
        unsigned long long h[8];
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-34s %d waves/SIMD: %.2f cycles per MFMA per wave (wave 0), %.2f per SIMD\n", name, threads / 256,
               (double)h[0] / (iters * 32.0 * NCH), (double)h[0] / (iters * 32.0 * NCH) / (threads / 256));
    }
}

int main() {
    const int K = 64;
    float *ha = (float *)malloc(K * 64 * 4), *hb = (float *)malloc(K * 64 * 4), hc[256], hd[256];
    srand(7);
    for (int i = 0; i < K * 64; ++i) { ha[i] = (float)rand() / RAND_MAX - 0.5f; hb[i] = (float)rand() / RAND_MAX - 0.5f; }
    for (int i = 0; i < 256; ++i) hc[i] = (float)rand() / RAND_MAX;
    float *da, *db, *dc, *dd;
    hipMalloc(&da, K * 64 * 4); hipMalloc(&db, K * 64 * 4); hipMalloc(&dc, 1024); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, K * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, K * 64 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice);
    for (int kk : {1, K}) {
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dd, kk);
        hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        int bad = 0, inexact = 0;
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l) {
                const int b = l >> 2;
                float ref = hc[r * 64 + l];
                for (int k = 0; k < kk; ++k) ref = fmaf(ha[k * 64 + 4 * b + r], hb[k * 64 + l], ref);
                const float got = hd[r * 64 + l];
                if (fabsf(got - ref) > 1e-5f * (1.0f + fabsf(ref))) ++bad;
                if (got != ref) ++inexact;
            }
        printf("layout K=%d: %d wrong, %d not bit-equal to the fmaf chain (of 256)\n", kk, bad, inexact);
    }
    {   // reduction check
        float hin[256], hout[64];
        for (int i = 0; i < 256; ++i) hin[i] = (float)(rand() % 1000);
        hipMemcpy(dc, hin, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(64), 0, 0, dc, dd);
        hipMemcpy(hout, dd, 256, hipMemcpyDeviceToHost);
        const int umap[4] = {0, 2, 1, 3};
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const int rho = l >> 4, j = l & 3, u = umap[rho];
            float ref = 0;
            for (int kp = 0; kp < 16; ++kp) ref += hin[u * 64 + kp * 4 + j];
            if (ref != hout[l]) ++bad;
        }
        printf("kp reduction (2x swap32, swap16, row_ror 4/8; unit map {0,2,1,3}): %d wrong of 64\n", bad);
    }
    float *d; unsigned long long *c;
    hipMalloc(&d, 4096 * 4); hipMemset(d, 0, 4096 * 4); hipMalloc(&c, 64);
    rate<1, false>("4x4x1_16b, 1 chain, B in regs", d, c);
    rate<2, false>("4x4x1_16b, 2 chains, B in regs", d, c);
    rate<6, false>("4x4x1_16b, 6 chains, B in regs", d, c);
    rate<2, true>("4x4x1_16b, 2 chains, B from LDS", d, c);
    rate<6, true>("4x4x1_16b, 6 chains, B from LDS", d, c);
    mix<0, false>(d, c); mix<1, false>(d, c); mix<2, false>(d, c); mix<3, false>(d, c); mix<4, false>(d, c); mix<2, true>(d, c);
    return 0;
}
