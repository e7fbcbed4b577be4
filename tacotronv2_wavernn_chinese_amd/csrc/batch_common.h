// Building blocks of the batch kernel (loop_batch.hip): team / mailbox primitives, the
// v_mfma_f32_4x4x1 loops with hand-made software pipelining, the K-phase fold, the LDS carve-up.  See loop_batch.hip for the
// mapping these pieces implement.
#pragma once
#include "device_util.h"
#include "wrnn_internal.h"

#define TB_WGS 32
#define TB_THREADS 256
#define TB_SPIN_MAX 400000u

typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
// LDS pointers with an OPAQUE per-thread base: the base (one VGPR) goes through an empty asm so that the compiler cannot fold the
// buffer's constant LDS address into it -- every access is then "base + 16-bit immediate".  With foldable bases hipcc hoisted
// ~50 different precomputed LDS addresses out of the step loop and kept each in its own VGPR (the constants exceed the DS
// offset field once the buffer address is part of them): measured in the ISA, 50 distinct ds_read_b128 address registers.
typedef const f4 __attribute__((address_space(3))) *lds_cf4p;
typedef float f2v __attribute__((ext_vector_type(2)));
typedef f2v __attribute__((address_space(3))) *lds_f2p;
typedef const float __attribute__((address_space(3))) *lds_cfp;
__device__ __forceinline__ unsigned launder(unsigned v) { asm volatile("" : "+v"(v)); return v; }

namespace {

__device__ __forceinline__ unsigned xcc_idb() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ void st_granule(u64 *base, unsigned idx, unsigned tag, unsigned payload) {
    const u64 v = ((u64)tag << 32) | payload;
    const unsigned off = idx * 8u;
    asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}
// 16-byte sc1 load (L1 bypass) of two adjacent granules; compiler-tracked (s_waitcnt vmcnt inserted by hipcc).  The per-thread
// part of the address (tid * 16) is the VGPR offset, everything wave-uniform (region, parity, slice) goes into the SGPR offset:
// the instruction's immediate offset has 12 bits, so slice offsets folded into the VGPR cost one register per slice.
__device__ __forceinline__ u4v ld_pair(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16 /* sc1 */);
}
__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
// Weights parked in the accumulator half of the register file ("a" constraint = AGPR class for the value's whole life):
// hipcc allocates at most 256 architectural VGPRs per wave and uses AGPRs only as spill slots, so 320 weights + working set
// as plain floats overflow into scratch (129 registers, reloaded on the serial chain: measured 13 000 cycles in phase B).
// A value of AGPR class feeds the MFMA's A operand directly (`v_mfma_f32_4x4x1_16b_f32 a[0:3], a93, v9, a[0:3]`: the operand
// class of the builtin is "VGPR or AGPR").  Fetching it into a VGPR first (v_accvgpr_read) costs ~20 cycles per MFMA instead
// of 8: the read waits for the in-flight MFMAs that write accumulators (measured: 4 046 vs ~1 700 cycles for W_hh1 at R = 8).
__device__ __forceinline__ void apark(float &dst, float v) { asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(v)); }
__device__ __forceinline__ float aget(const float &a) { return a; }
template <bool AG>
__device__ __forceinline__ float wget(const float &w) { return AG ? aget(w) : w; }

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// fold the 16 K phases: in d[i] lane (kp, j) = partial of unit i, batch row j; out: lane (rho = lane>>4, *, j) = full sum
// of unit {0, 2, 1, 3}[rho] for batch row j, replicated over the 4 lanes-of-four of the row
__device__ __forceinline__ float fold_kp(f4 d) {
    const u2v p = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[0]), __float_as_uint(d[1]), false, false);
    const float s01 = __uint_as_float(p.x) + __uint_as_float(p.y);
    const u2v q = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[2]), __float_as_uint(d[3]), false, false);
    const float s23 = __uint_as_float(q.x) + __uint_as_float(q.y);
    const u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
    float t = __uint_as_float(r.x) + __uint_as_float(r.y);
    t += dppf<0x124>(t);   // row_ror:4
    t += dppf<0x128>(t);   // row_ror:8
    return t;
}
__device__ __forceinline__ float wave_max_b(float v) {   // max over 64 lanes, valid in lane 63 (see loop_team2.hip)
    asm volatile(
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// ---- LDS carve-up (floats) -----------------------------------------------------------------------------------------
template <int NQ>
struct Lay {
    static constexpr int R = 4 * NQ;
    static constexpr int VEC = R * 512;            // one activation vector for R rows, B-operand order [rq][S][kp][j][e]
    static constexpr int L_FC3 = 0;                // [4 waves][2 sets][8 S][64 lanes][4 e]: A operands of the fc3 slice
    static constexpr int L_WN = 16384;             // [4 waves][8 S][64 lanes][4 e]: A operands of gate n of W_hh2 (shadow path)
    static constexpr int L_CST = L_WN + 8192;      // [12][256]: per-thread constants (read once per step, not worth registers)
    static constexpr int L_P = L_CST + 12 * 256;   // x2, later fc2 outputs
    static constexpr int L_Q = L_P + VEC;          // x3
    static constexpr int L_H1 = L_Q + VEC;         // h1', later fc1 outputs        (h2' = x3 - x2 is formed on the fly)
    static constexpr int L_XN = L_H1 + VEC;        // [16] x_{t-1} of every batch row
    static constexpr int L_MISC = L_XN + 16;       // [16]
    static constexpr int L_PROF = L_MISC + 16;     // [4 waves][24] phase-cycle accumulators of the instrumented build (in LDS, not in
                                                   // registers: the kernel has none to spare)
    static constexpr int L_TOTAL = L_PROF + 96;
    static_assert(L_TOTAL * 4 <= 163840, "LDS budget");
    // mailbox regions per team (granules); every region is double-buffered by step parity
    static constexpr unsigned RG = (unsigned)VEC;
    static constexpr unsigned G_X2 = 0, G_H1 = 2 * RG, G_X3 = 4 * RG, G_F1 = 6 * RG, G_F2 = 8 * RG, G_PR = 10 * RG;
    static constexpr unsigned PRG = (unsigned)R * 128u;
    static constexpr unsigned MAIL = 10 * RG + 2 * PRG;
    static_assert(MAIL <= WRNN_BATCH_MAIL_GRANULES, "mailbox budget");
    static constexpr int NM = R;                   // 16-byte loads per thread per gathered vector (256 threads)
};
// slots of L_CST
constexpr int C_A0 = 0, C_A1 = 1, C_A2 = 2, C_A3 = 3, C_B30 = 4, C_B31 = 5, C_H1R = 6, C_H1Z = 7, C_H1N = 8, C_H2R = 9, C_H2Z = 10, C_H2N = 11;
constexpr int M_DEAD = 0, M_TEAM = 1, M_RANK = 2;

// All-gather of NV published vectors (R x 512 granules each, mailbox order [rq][wl][S][iu][j][e]; slice m = (rq, wl) holds
// what wave wl of EVERY workgroup published for row quad rq).  Every load of every vector is in flight at once -- one L2
// round trip when the producers are done, which they normally are: the shadow work of the window sits between the publish
// and this poll.  If a slice came back incomplete, only the last slice (published by the wave that is dispatched last) is
// polled until it is complete, then everything is fetched again: no per-slice state is kept (the per-slice retry masks
// of the first version cost ~16 SGPR pairs and pushed the 8-row kernel into scratch spills).
// voff = tid * 16; soff[v] = byte offset of vector v's region (wave-uniform).
// PRE: the caller already requested every slice once (gather_issue, in the middle of the shadow work of the window, when the
// producers are normally done): the first look then costs no round trip of its own.
template <int NM, int NV>
__device__ __forceinline__ void gather_issue(__amdgpu_buffer_rsrc_t rs, unsigned voff, const unsigned (&soff)[NV], u4v (&g)[NV][NM]) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int m = 0; m < NM; ++m) g[v][m] = ld_pair(rs, voff, soff[v] + m * 4096u);
    __builtin_amdgcn_sched_barrier(0);
}
// developer build (-DWRNN_COUNT_SLOW): how often an exchange's first look comes back stale (workgroup 0, wave 0; printed by the kernel)
#ifdef WRNN_COUNT_SLOW
__device__ unsigned wrnn_dbg_slow[64];
#define WRNN_SLOW_NOTE(code, slow) do { if (blockIdx.x == 0 && threadIdx.x == 0) { atomicAdd(&wrnn_dbg_slow[32 + ((code) & 31)], 1u); \
    if (slow) atomicAdd(&wrnn_dbg_slow[(code) & 31], 1u); } } while (0)
#else
#define WRNN_SLOW_NOTE(code, slow)
#endif
template <int NM, int NV, bool PRE = false>
__device__ __forceinline__ void gather_vecs(__amdgpu_buffer_rsrc_t rs, unsigned voff, const unsigned (&soff)[NV], unsigned tag, u4v (&g)[NV][NM],
                                            bool &dead, unsigned *err, unsigned code) {
    if (!PRE) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int m = 0; m < NM; ++m) g[v][m] = ld_pair(rs, voff, soff[v] + m * 4096u);
    }
    unsigned spins = 0;
#ifdef WRNN_COUNT_SLOW
    bool first_ = true;
#endif
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int m = 0; m < NM; ++m) ok = ok && g[v][m].y == tag && g[v][m].w == tag;
#ifdef WRNN_COUNT_SLOW
        if (first_) { WRNN_SLOW_NOTE(code, !__all(ok)); first_ = false; }
#endif
        if (__all(ok) || dead) break;
        // wait on the sentinel slice of the last vector, then look at everything again
        // (three sentinel loads in flight instead of one -- the arrival noticed a third of a round trip after it happened -- measured
        //  +0.2 % at R = 4, -0.7 % at R = 8, nothing in the training kernels: more polls are more L2 traffic; not kept)
        for (;;) {
            if (++spins > TB_SPIN_MAX) { dead = true; if ((threadIdx.x & 63) == 0) atomicExch(err, code); break; }
            __builtin_amdgcn_s_sleep(1);
            const u4v sv = ld_pair(rs, voff, soff[NV - 1] + (NM - 1) * 4096u);
            if (__all(sv.y == tag && sv.w == tag)) break;
        }
        if (dead) break;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int m = 0; m < NM; ++m) g[v][m] = ld_pair(rs, voff, soff[v] + m * 4096u);
    }
}

// The MFMA loops below are software pipelined by hand: the LDS operands of slab S + D are requested before the MFMAs of slab S
// are issued, and scheduling barriers on either side of the MFMA group keep hipcc from sinking the reads next to their first
// use (which it does otherwise: `ds_read x2, s_waitcnt, 24 MFMAs` per slab exposes one LDS round trip per slab, 8 per phase --
// measured 2 092 cycles for the 192 MFMAs of phase B against the 1 555 the MFMA pipe needs).
struct NoMid { __device__ __forceinline__ void operator()() const {} };

// acc[g][q] += W_g (32 slabs at w[g*32 ..]) . x for NG weight rows sharing the B operand; xv = LDS vector as f4 [rq][S][lane];
// mid() runs before slab MS (the early request of the next exchange's granules)
template <int NQ, int NG, bool AG, int D, class F, int MS = 4, bool EVERY = false>
__device__ __forceinline__ void mfma_gates(const float *w, lds_cf4p xv, f4 (&acc)[NG][NQ], F mid) {
    f4 ring[D][NQ];
#pragma unroll
    for (int dd = 0; dd < D; ++dd)
#pragma unroll
        for (int q = 0; q < NQ; ++q) ring[dd][q] = xv[(q * 8 + dd) * 64];
#pragma unroll
    for (int S = 0; S < 8; ++S) {
        if (EVERY || S == MS) mid();   // EVERY: a hook in front of every slab (loop_batch_cs.hip: the shadow wave yields to the critical one)
        f4 b[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) b[q] = ring[S % D][q];
        if (S + D < 8) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) ring[S % D][q] = xv[(q * 8 + S + D) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int gt = 0; gt < NG; ++gt) {
                const float wa = wget<AG>(w[gt * 32 + 4 * S + e]);
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[gt][q] = mfma4(wa, b[q][e], acc[gt][q]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// one weight row set (32 slabs at w[..]); the K sum is split over NP independent accumulator chains
template <int NQ, int NP, bool AG, int D>
__device__ __forceinline__ void mfma_single(const float *w, lds_cf4p xv, f4 (&sum)[NQ]) {
    f4 acc[NP][NQ];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[p][q] = (f4){0.f, 0.f, 0.f, 0.f};
    f4 ring[D][NQ];
#pragma unroll
    for (int dd = 0; dd < D; ++dd)
#pragma unroll
        for (int q = 0; q < NQ; ++q) ring[dd][q] = xv[(q * 8 + dd) * 64];
#pragma unroll
    for (int S = 0; S < 8; ++S) {
        f4 b[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) b[q] = ring[S % D][q];
        if (S + D < 8) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) ring[S % D][q] = xv[(q * 8 + S + D) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float wa = wget<AG>(w[4 * S + e]);
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[e % NP][q] = mfma4(wa, b[q][e], acc[e % NP][q]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        sum[q] = acc[0][q];
#pragma unroll
        for (int p = 1; p < NP; ++p) sum[q] += acc[p][q];
    }
}

}  // namespace

// phase-cycle instrumentation (WRNN_TEAM_PROF=1): the scheduling barriers keep the (side-effect free) MFMAs and VALU work of a
// phase on its own side of the time stamp
#define PB(i)                                                                  \
    do {                                                                       \
        if (PROF) {                                                            \
            __builtin_amdgcn_sched_barrier(0);                                 \
            const unsigned now_ = (unsigned)__builtin_readcyclecounter();      \
            __builtin_amdgcn_sched_barrier(0);                                 \
            if (lane == 0) prof_lds[wl * 24 + (i)] += now_ - prof_last;        \
            prof_last = now_;                                                  \
        }                                                                      \
    } while (0)
