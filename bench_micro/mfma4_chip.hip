// Developer probe: does a SECOND wave on a SIMD add v_mfma_f32_4x4x1_16b_f32 throughput?  Every CU runs one workgroup of 256 threads (one wave per SIMD) or 512 threads (two per SIMD) that
// issues nothing but MFMAs (6 independent accumulator chains, operands in registers); chip-wide TFLOP/s from HIP events, cycles per MFMA per wave from s_memtime, and the SIMD every wave of
// workgroup 0 ran on (HW_REG_HW_ID bits 5:4).  Written in round 5 to settle what bench_micro/mfma4_probe (round 2: "4.07 cycles per SIMD with two waves") means for the batch kernel's two roles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) rate(float *o, unsigned long long *cyc, unsigned *simd, int iters) {
    f4 acc[6];
    float w[32];
    for (int i = 0; i < 32; ++i) w[i] = o[(threadIdx.x + i) & 1023];
    for (int c = 0; c < 6; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s)
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[s], w[(s + c + 1) & 31], acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < 6; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) o[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
        cyc[threadIdx.x >> 6] = t1 - t0;
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        simd[threadIdx.x >> 6] = (id >> 4) & 3;
    }
}

int main() {
    float *d; unsigned long long *c; unsigned *sd;
    hipMalloc(&d, 4096); hipMemset(d, 0, 4096); hipMalloc(&c, 64); hipMalloc(&sd, 32);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(rate, dim3(cus), dim3(threads), 0, 0, d, c, sd, 100);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate, dim3(cus), dim3(threads), 0, 0, d, c, sd, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8]; unsigned hs[8];
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hs, sd, sizeof(hs), hipMemcpyDeviceToHost);
        const double mf = (double)cus * (threads / 64) * iters * 192.0;
        printf("%d waves per SIMD, %d CUs: %.1f TFLOP/s chip-wide (512 FLOP per MFMA), %.2f cycles per MFMA per wave (wave 0), clock %.2f GHz; SIMD of waves 0..%d:", threads / 256, cus,
               mf * 512.0 / (ms * 1e-3) / 1e12, (double)h[0] / (iters * 192.0), (double)h[0] / (ms * 1e-3) / 1e9, threads / 64 - 1);
        for (int i = 0; i < threads / 64; ++i) printf(" %u", hs[i]);
        printf("\n");
    }
    return 0;
}
