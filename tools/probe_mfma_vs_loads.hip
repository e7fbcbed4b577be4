// Developer probe (GPU box), standalone:  hipcc --offload-arch=gfx950 -O3 -o probe tools/probe_mfma_vs_loads.hip && ./probe
//
// Question (profiles/r04_batch_cs_experiments.txt, "WHERE the x3 / f1 exchanges lose their ~900 cycles"): how long does a "look" -- 8 x 16-byte sc1 loads per lane
// of a 32 KB L2-resident block that every workgroup of the XCD reads -- take for a wave while the OTHER wave of its SIMD issues v_mfma_f32_4x4x1 back to back?
// One workgroup per CU, 8 waves: waves 0-3 look (one per SIMD), waves 4-7 keep the SIMD's matrix / vector pipe busy in one of these ways:
//   0 idle (s_sleep)   1 MFMA, accumulators in architectural VGPRs   2 MFMA, accumulators in AGPRs   3 v_fma_f32 chains (no MFMA)
//   6 the MFMA loop of an S wave of loop_batch_cs.hip (96 resident A registers, 6 accumulator chains in VGPRs, two 16-byte LDS reads per 24 MFMAs)
//   4 MFMA (VGPR form) with an s_nop 7 between the instructions (~half the issue density)      5 MFMA (AGPR form) + the B operand read from LDS every 4 MFMAs
// The looking waves check what they fetched in one of these ways (LCHK): 0 a few VALU behind one wait   1 C++ `ok = ok && tag == ...` per load (hipcc makes a
// branchy sequence of it here: 2 240 cycles even beside an idle wave -- not what the kernel's ISA looks like)   2 a v_min3 chain   3 no loads: 64 dependent v_add_u32
// Prints, per mode, the mean cycles of a look and the busy waves' instruction rate.  `./probe 256 all` prints the sibling-mode table of round 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LOOKS = 200;     // looks per loader wave
constexpr int NLOAD = 8;       // 16-byte loads per lane per look (R = 8)

template <int MODE, bool BIG = false, int LCHK = 0>
__global__ __launch_bounds__(512) void probe(unsigned *blk, unsigned *out_cyc, unsigned *out_busy, float seed, int sync, int pub) {
    __shared__ int done;   // loader waves that have finished
    __shared__ f4 ldsb[64 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (BIG) asm volatile("v_mov_b32 v255, 0" ::: "v255");   // the kernel allocates all 256 VGPRs: with two waves per SIMD the register file is full, as in loop_batch_cs.hip
    if (tid == 0) done = 0;
    if (blockIdx.x == 0 && lane == 0) {   // which SIMD each wave of a workgroup sits on (HW_ID bits 5:4)
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out_busy[gridDim.x * 8 + wave] = hw;
    }
    for (int i = tid; i < 64 * 8; i += 512) {   // operands with busy mantissas (seed = 1: constants)
        const float r0 = seed == 1.0f ? 1.0f : __uint_as_float(0x3f000000u | ((unsigned)(i * 2654435761u) >> 9));
        ldsb[i] = (f4){r0, r0 * 1.37f, r0 * 0.71f, r0 * 1.93f};
    }
    __syncthreads();
    if (wave < 4) {
        __builtin_amdgcn_s_setprio(3);   // as the C waves of loop_batch_cs.hip
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)blk, 0, 32768, 0x00020000);
        const unsigned voff = (unsigned)(wave * 64 + lane) * 16u;
        unsigned total = 0, sink = 0;
        for (int it = 0; it < LOOKS; ++it) {
            __builtin_amdgcn_s_sleep(20);
            if (sync) {   // every workgroup of the device starts its look in the same ~100 cycles (s_memrealtime: one 100 MHz clock for the whole device)
                const unsigned long long target = (__builtin_amdgcn_s_memrealtime() / 200ull + 1ull) * 200ull;
                while (__builtin_amdgcn_s_memrealtime() < target) __builtin_amdgcn_s_sleep(1);
            }
            if (pub && lane < 32) {   // as a publish: 128 scattered 8-byte granules per workgroup, then the look ~a store round trip later
                const unsigned idx = ((unsigned)(wave * 32 + lane) * 32u + (blockIdx.x & 31u)) * 8u;
                const unsigned long long v = ((unsigned long long)it << 32) | lane;
                asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(idx), "v"(v), "s"(blk) : "memory");
                }
            if (pub) __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned t0 = (unsigned)__builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned tag0 = 0x01010101u;
            u4v g[NLOAD];
#pragma unroll
            for (int m = 0; m < NLOAD; ++m) g[m] = LCHK == 3 ? (u4v){tag0, tag0, tag0, tag0} : __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)m * 4096u, 16);
            const unsigned tag = 0x01010101u;   // what hipMemset(blk, 1) left in every word
            if (LCHK == 0) {          // one wait, a few VALU
#pragma unroll
                for (int m = 0; m < NLOAD; ++m) sink += g[m].x ^ g[m].w;
                asm volatile("" :: "v"(sink));
            } else if (LCHK == 1) {   // the check of loop_batch_cs.hip: per load a wait, two compares, two scalar ANDs (~45 instructions)
                bool ok = true;
#pragma unroll
                for (int m = 0; m < NLOAD; ++m) ok = ok && g[m].y == tag && g[m].w == tag;
                if (!__all(ok)) sink += 1;
            } else if (LCHK == 2) {   // one wait, a v_min3 tree over the 16 tag words, one compare (~10 instructions)
                unsigned mn = 0xffffffffu;
#pragma unroll
                for (int m = 0; m < NLOAD; m += 2) {
                    const unsigned a = g[m].y < g[m].w ? g[m].y : g[m].w, b = g[m + 1].y < g[m + 1].w ? g[m + 1].y : g[m + 1].w;
                    const unsigned c = a < b ? a : b;
                    mn = mn < c ? mn : c;
                }
                if (!__all(mn == tag)) sink += 1;
            } else {                  // LCHK 3: no loads at all -- 64 independent VALU instructions: what does ONE instruction of this wave cost?
#pragma unroll
                for (int k = 0; k < 64; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(sink) : "v"(lane));
            }
            __builtin_amdgcn_sched_barrier(0);
            const unsigned t1 = (unsigned)__builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
            total += t1 - t0;
        }
        if (lane == 0) out_cyc[blockIdx.x * 4 + wave] = total / LOOKS + (sink == 0x12345u ? 1u : 0u);
        if (lane == 0) atomicAdd(&done, 1);
    } else {
        f4 acc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
        float a = seed + lane, b = seed * 0.5f;
        float wreg[96];   // mode 6: resident "weights", as the S waves of loop_batch_cs.hip
        if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 96; ++i) { wreg[i] = seed == 1.0f ? (float)(i + 1) + lane : __uint_as_float(0x3f000000u | ((unsigned)((i * 64 + lane) * 2246822519u) >> 9)); asm volatile("" : "+v"(wreg[i])); }
        }
        float fa[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) fa[i] = seed + i;
        unsigned iters = 0;
        const unsigned t0 = (unsigned)__builtin_readcyclecounter();
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
            if (MODE == 0) {
                __builtin_amdgcn_s_sleep(8);
            } else if (MODE == 1 || MODE == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                        if (MODE == 4) asm volatile("s_nop 7");
                    }
            } else if (MODE == 2 || MODE == 5) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (MODE == 5) { const f4 v = ldsb[r * 64 + lane]; b = v.x; }
#pragma unroll
                    for (int i = 0; i < 6; ++i) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                }
            } else if (MODE == 6) {
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {
                    const f4 b0 = ldsb[sl * 64 + lane], b1 = ldsb[((sl + 3) & 7) * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int gt = 0; gt < 3; ++gt) {
                            asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[2 * gt]) : "v"(wreg[(sl * 4 + e) * 3 + gt]), "v"(b0[e]));
                            asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[2 * gt + 1]) : "v"(wreg[(sl * 4 + e) * 3 + gt]), "v"(b1[e]));
                        }
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 12; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fa[i]) : "v"(a), "v"(b));
            }
            ++iters;
        }
        const unsigned t1 = (unsigned)__builtin_readcyclecounter();
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
        for (int i = 0; i < 12; ++i) s += fa[i];
        if (lane == 0) {
            out_busy[(blockIdx.x * 4 + (wave - 4)) * 2] = iters + (s == 1.2345f ? 1u : 0u);
            out_busy[(blockIdx.x * 4 + (wave - 4)) * 2 + 1] = t1 - t0;
        }
    }
}

template <int MODE, bool BIG = false, int LCHK = 0>
static void run(const char *what, int per_iter, unsigned *blk, unsigned *d_cyc, unsigned *d_busy, int wgs, int sync, int pub, float seed = 1.0f) {
    CHECK(hipMemset(d_cyc, 0, wgs * 4 * sizeof(unsigned)));
    CHECK(hipMemset(d_busy, 0, wgs * 8 * sizeof(unsigned)));
    CHECK(hipFuncSetAttribute((const void *)probe<MODE, BIG, LCHK>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));   // one workgroup per CU
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<MODE, BIG, LCHK>), dim3(wgs), dim3(512), 96 * 1024, 0, blk, d_cyc, d_busy, seed, sync, pub);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned> cyc(wgs * 4), busy(wgs * 8);
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(busy.data(), d_busy, busy.size() * 4, hipMemcpyDeviceToHost));
    double m = 0, mx = 0, rate = 0;
    for (unsigned c : cyc) { m += c; if (c > mx) mx = c; }
    for (int i = 0; i < wgs * 4; ++i) rate += busy[2 * i + 1] ? (double)busy[2 * i] * per_iter / busy[2 * i + 1] : 0.0;
    printf("check %d %s sync %d pub %d mode %d %-52s look: mean %7.0f max %7.0f cycles;  busy wave: %.3f instr/cycle\n", LCHK, BIG ? "256 VGPRs" : "few VGPRs", sync, pub, MODE, what, m / cyc.size(), mx, rate / (wgs * 4));
}

int main(int argc, char **argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    unsigned *blk, *d_cyc, *d_busy;
    CHECK(hipMalloc(&blk, 32768));
    CHECK(hipMemset(blk, 1, 32768));
    CHECK(hipMalloc(&d_cyc, wgs * 4 * sizeof(unsigned)));
    CHECK(hipMalloc(&d_busy, (wgs * 8 + 8) * sizeof(unsigned)));
    {
        CHECK(hipFuncSetAttribute((const void *)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(512), 96 * 1024, 0, blk, d_cyc, d_busy, 1.0f, 0, 0);
        CHECK(hipDeviceSynchronize());
        unsigned hw[8];
        CHECK(hipMemcpy(hw, d_busy + wgs * 8, sizeof(hw), hipMemcpyDeviceToHost));
        printf("workgroup 0: wave -> SIMD:");
        for (int w = 0; w < 8; ++w) printf(" %d->%u", w, (hw[w] >> 4) & 3u);
        printf("   (CU %u)\n", (hw[0] >> 8) & 15u);
    }
    printf("%d workgroups x 512 threads, %d looks of %d x 16 B per lane (32 KB per workgroup and look)\n", wgs, LOOKS, NLOAD);
    // argv[2] = "all": the round-4 table (sibling modes x register allocation); default: how the look's CHECK code fares under the neighbour's MFMAs
    if (argc > 2) {
        for (int sync = 0; sync < 2; ++sync) {
            run<0>("other wave idle", 0, blk, d_cyc, d_busy, wgs, sync, 0);
            run<1>("other wave: MFMA 4x4x1, accumulators in VGPRs", 24, blk, d_cyc, d_busy, wgs, sync, 0);
            run<2>("other wave: MFMA 4x4x1, accumulators in AGPRs", 24, blk, d_cyc, d_busy, wgs, sync, 0);
            run<3>("other wave: v_fma_f32 chains", 96, blk, d_cyc, d_busy, wgs, sync, 0);
            run<6>("other wave: as an S wave (96 A regs, B from LDS)", 192, blk, d_cyc, d_busy, wgs, sync, 0);
            run<6, true>("other wave: as an S wave (96 A regs, B from LDS)", 192, blk, d_cyc, d_busy, wgs, sync, 0);
            run<6, true>("other wave: as an S wave, operands with random mantissas", 192, blk, d_cyc, d_busy, wgs, sync, 0, 0.5f);
        }
        return 0;
    }
    run<0, true, 0>("idle", 0, blk, d_cyc, d_busy, wgs, 1, 0);
    run<1, true, 0>("MFMA", 24, blk, d_cyc, d_busy, wgs, 1, 0);
    run<6, true, 0>("S-wave loop", 192, blk, d_cyc, d_busy, wgs, 1, 0);
    run<0, true, 1>("idle", 0, blk, d_cyc, d_busy, wgs, 1, 0);
    run<1, true, 1>("MFMA", 24, blk, d_cyc, d_busy, wgs, 1, 0);
    run<6, true, 1>("S-wave loop", 192, blk, d_cyc, d_busy, wgs, 1, 0);
    run<0, true, 2>("idle", 0, blk, d_cyc, d_busy, wgs, 1, 0);
    run<1, true, 2>("MFMA", 24, blk, d_cyc, d_busy, wgs, 1, 0);
    run<6, true, 2>("S-wave loop", 192, blk, d_cyc, d_busy, wgs, 1, 0);
    run<0, true, 3>("idle", 0, blk, d_cyc, d_busy, wgs, 0, 0);
    run<1, true, 3>("MFMA", 24, blk, d_cyc, d_busy, wgs, 0, 0);
    run<3, true, 3>("v_fma", 96, blk, d_cyc, d_busy, wgs, 0, 0);
    run<6, true, 3>("S-wave loop", 192, blk, d_cyc, d_busy, wgs, 0, 0);
    return 0;
}
