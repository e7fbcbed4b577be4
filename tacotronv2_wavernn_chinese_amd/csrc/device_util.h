// Device helpers shared by the loop kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- Philox4x32-10 counter RNG (Salmon et al. 2011) -----------------------
// Production sampling noise: one stream per (seed, step, row, class/4); no
// state, so any workgroup can evaluate any draw.
struct Philox4 { uint32_t x, y, z, w; };

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return Philox4{c0, c1, c2, c3};
}

// uniform in (0,1): 23 random bits, never 0 or 1.  (m + 0.5) * 2^-23 with m < 2^23 is exact in fp32; with 24 bits the
// top value 2^24 - 0.5 rounds to 2^24, i.e. u == 1, -log u == 0 and a class of probability zero wins the race once per
// 2^24 draws -- found by the 110 275-step Philox parity test (8 such draws in 113 M).
__host__ __device__ inline float u01_from_bits(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// The uniform draw for (step t, row, element k): element k of the 1024 RAW
// classes, or k < 10 mixture uniforms / k == 10 logistic uniform for MOL.
__host__ __device__ inline float wrnn_uniform(uint64_t seed, uint64_t t, uint32_t row, uint32_t k) {
    const Philox4 p = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), row, k >> 2, (uint32_t)seed,
                                    (uint32_t)(seed >> 32));
    const uint32_t sel = k & 3u;
    const uint32_t bits = sel == 0 ? p.x : sel == 1 ? p.y : sel == 2 ? p.z : p.w;
    return u01_from_bits(bits);
}

// RAW sampling noise: the uniform behind q_k of (step t, row, class k).  One Philox block serves classes
// (2j, 2j+1) for steps (2s, 2s+1): block counter (t>>1, row, k>>1), element ((t&1)<<1)|(k&1) -- a kernel that
// owns a class pair evaluates the block every other step.
__host__ __device__ inline Philox4 wrnn_raw_block(uint64_t seed, uint64_t t, uint32_t row, uint32_t k) {
    const uint64_t th = t >> 1;
    return philox4x32_10((uint32_t)th, (uint32_t)(th >> 32), row, k >> 1, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__host__ __device__ inline float wrnn_uniform_raw(uint64_t seed, uint64_t t, uint32_t row, uint32_t k) {
    const Philox4 p = wrnn_raw_block(seed, t, row, k);
    const uint32_t sel = (uint32_t)((t & 1u) << 1) | (k & 1u);
    return u01_from_bits(sel == 0 ? p.x : sel == 1 ? p.y : sel == 2 ? p.z : p.w);
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
// tanh via exp: tanh(x) = 1 - 2/(exp(2x)+1); accurate to ~2 ulp in fp32 for the GRU range
__device__ __forceinline__ float tanh_f(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}

// wave64 argmax reduction (value, index); ties -> lowest index, like argmax in the reference path
__device__ __forceinline__ void wave_argmax(float &v, int &i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(i, off, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}
