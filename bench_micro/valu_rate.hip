// Developer probe: VALU issue rate on one CU of gfx950 with 1 and 2 waves per SIMD, for the instruction kinds the
// loop kernel is made of.  Prints cycles per wave-instruction (as seen by one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(512) k(float *o, unsigned long long *cyc, int iters) {
    f2 a[8];
    float s[8];
    for (int i = 0; i < 8; ++i) { a[i].x = o[threadIdx.x + i]; a[i].y = o[threadIdx.x + 8 + i]; s[i] = a[i].x; }
    f2 m; m.x = 1.0001f; m.y = 0.9999f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) a[i] = __builtin_elementwise_fma(a[i], m, m);                  // v_pk_fma_f32, independent
                if (KIND == 1) s[i] = __builtin_fmaf(s[i], 1.0001f, 0.5f);                   // v_fma_f32
                if (KIND == 2) a[0] = __builtin_elementwise_fma(a[0], m, m);                  // v_pk_fma_f32, dependent chain
                if (KIND == 3) s[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s[i]), 0xB1, 0xf, 0xf, true));  // dpp add
                if (KIND == 4) { u2v r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s[i]), __float_as_uint(s[(i + 1) & 7]), false, false);
                                 s[i] = __uint_as_float(r2.x); s[(i + 1) & 7] = __uint_as_float(r2.y); }
                if (KIND == 5) s[i] = __expf(s[i]);                                           // v_exp_f32 (+mul)
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + s[i];
    o[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int KIND>
void run(const char *name, float *d, unsigned long long *c) {
    const int iters = 2000;
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, c, iters);
        unsigned long long h[8];
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-28s %d waves/SIMD: %.2f cycles per instruction (wave 0)\n", name, threads / 256, (double)h[0] / (iters * 32.0));
    }
}
int main() {
    float *d; unsigned long long *c;
    hipMalloc(&d, 4096 * 4); hipMemset(d, 0, 4096 * 4); hipMalloc(&c, 64);
    run<0>("v_pk_fma_f32 independent", d, c);
    run<1>("v_fma_f32 independent", d, c);
    run<2>("v_pk_fma_f32 dependent", d, c);
    run<3>("v_add_f32_dpp", d, c);
    run<4>("v_permlane32_swap", d, c);
    run<5>("v_exp_f32 (+v_mul)", d, c);
    return 0;
}
