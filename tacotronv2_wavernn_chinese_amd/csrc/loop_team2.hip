// WRNN_KERNEL_TEAM2: the per-sample loop (fatchord_version.py:194-241) with WAVE SPECIALISATION.
//
// Team structure and exchange protocol (one team = the 32 workgroups of one XCD,
// fp32 weights resident on chip, 8-byte {tag,value} granules through the XCD's L2, 4 exchanges on the critical
// path + 1 off it, 5 workgroup barriers per step).  What changes is who does what inside a workgroup.
//
// Measured on loop_team.hip (profiles/): with one wave per SIMD the kernel is bound by single-wave
// instruction issue (~1 instruction / 4 cycles): ~2100 instructions per step, of which only ~35 % belong to the
// serial chain x -> GRU1 -> GRU2 -> fc1 -> fc2 -> fc3 -> sample; the rest is work that never waits on x_t
// (W_hh1.h1, W_hh2.h2, conditioning and noise of the next step, the gh1 gather) plus 256 v_accvgpr_read for
// weights parked in AGPRs.  So: 8 waves per workgroup, two per SIMD, and the two waves of a SIMD get different jobs:
//   waves 0-3 "C" (critical): W_ih2 (96) + fc1 (32) + fc2 (32) weights in VGPRs, fc3 slice in LDS;
//                             phases B, C, D, E; each C wave reduces a quarter of the race winners;
//   waves 4-7 "S" (shadow):   W_hh1 (96) + W_hh2 gates r,z (64) in VGPRs, W_hh2 gate n (32) in LDS;
//                             window 2: gh1 = W_hh1.h1 (published), window 3: sampling noise of step t+1 (16 lanes of a
//                             quarter-wave evaluate 16 different Philox blocks -> one evaluation per lane per 32 steps),
//                             window 4: conditioning of step t+1 (LDS <- registers <- HBM stream, prefetched two steps
//                             ahead), window 5: gh1 gather, gh2 = W_hh2.h2, per-frame constants.
// All 512 threads share phase A (unit j = tid), the exchange polls (granule tid) and the final 4-way merge of the
// race.  No AGPR parking: 2 waves/SIMD x 256 registers hold 160 weights + the working set.  S hands its results to C
// through small LDS slots between the same 5 barriers.  The phase-A conditioning {cI, v_r, v_z, v_n}[512] of every step
// is precomputed by cond_stream_kernel (prologue.hip) into an 8 KB/step HBM stream instead of being rebuilt from
// per-frame records in LDS: 57 KB of LDS and ~70 instructions per step saved for 1.6 GB/s of HBM traffic.
//
// Thread map inside a role (wl = wave & 3, lane l: quarter r4 = l>>4, q = l&15): quarter-wave (wl, r4) owns hidden
// unit / fc row u = 16 g + 4 wl + r4 and columns 32q..32q+31 of every row it owns.
#include "device_util.h"
#include "wrnn_internal.h"

#define T2_WGS 32
#define T2_THREADS 512
#define T2_SPIN_MAX 300000u

typedef unsigned long long u64;

namespace {

__device__ __forceinline__ unsigned xcc_id2() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// 8-byte granule {tag (hi), payload (lo)}: plain store (-> the XCD's L2), sc1 load (bypasses L1).  Same-XCD only.
__device__ __forceinline__ void st_granule(u64 *base, unsigned idx, unsigned tag, unsigned payload) {
    const u64 v = ((u64)tag << 32) | payload;
    const unsigned off = idx * 8u;
    asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void peek_n(const u64 *base, unsigned idx, unsigned stride, u64 (&g)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] = __hip_atomic_load(base + idx + (unsigned)i * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int N, int SHIFT>
__device__ __forceinline__ bool tags_ok(const u64 (&g)[N], unsigned tag) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) ok = ok && ((unsigned)(g[i] >> SHIFT) == tag);
    return ok;
}
// wave-uniform completion loop (see loop_team.hip); bounded
template <int N, int SHIFT>
__device__ __forceinline__ void finish_n(const u64 *base, unsigned idx, unsigned stride, unsigned tag, u64 (&g)[N], bool &dead,
                                         unsigned *err, unsigned code) {
    unsigned spins = 0;
    while (!dead && !__all(tags_ok<N, SHIFT>(g, tag))) {
        if (++spins > T2_SPIN_MAX) { dead = true; if ((threadIdx.x & 63) == 0) atomicExch(err, code); break; }
        peek_n<N>(base, idx, stride, g);
    }
}

// One exchange, one granule per thread.  Shipped (round 4): ONE look right behind the publish, then the bounded re-read loop.
// Rounds 1-3 issued two staggered looks (A at once, B `T2_STAGGER` sleeps later) and meant to examine B only if A came back
// incomplete -- but written as `return all_ok(A) ? A : finish(B)` the compiler merged the two results with a v_cndmask behind
// `s_waitcnt vmcnt(0)`, so every exchange waited for look B as well; and because B could still be in flight on the way out, every
// later reuse of its registers (among them the first instruction of the step loop) got a conservative vmcnt(0).  Measured in one
// session (profiles/r04_team2_experiments.txt, ksamples/s at B = 1): round-3 code 300.5; two looks with B really examined late
// (T2_LATE_B 1: every path's result pinned to its own load, B retired behind the exchange's barrier) 296 - 297 with T2_STAGGER 3,
// 293 with 2, 303 with 5; one look (T2_LATE_B 2) 303; one look behind an s_sleep (T2_PRESLEEP 1 / 2 / 4) 304 / 303 / 295.  The
// staggered second look buys nothing: a look that arrives before the data costs one more round trip whichever way it is issued.
#ifndef T2_STAGGER
#define T2_STAGGER 3
#endif
#ifndef T2_DEFER_OUT
#define T2_DEFER_OUT 0   // developer knob, see the merge at the end of the step
#endif
#ifndef T2_PRESLEEP
#define T2_PRESLEEP 0   // developer knob: s_sleep units in front of the first look
#endif
#ifndef T2_LATE_B
#define T2_LATE_B 2   // 2 = one look + re-read loop (shipped); 1 = two staggered looks, B examined only when A is stale; 0 = the round-3 code
#endif
__device__ __forceinline__ unsigned take_granule(const u64 *base, unsigned idx, unsigned tag, bool &dead, unsigned *err, unsigned code, u64 (&gb)[1]) {
    u64 ga[1];
#if T2_PRESLEEP
    __builtin_amdgcn_s_sleep(T2_PRESLEEP);
#endif
    peek_n<1>(base, idx, 1, ga);
#if T2_LATE_B == 2   // one look, then the re-read loop
    finish_n<1, 32>(base, idx, 1, tag, ga, dead, err, code);
    return (unsigned)ga[0];
#else
    __builtin_amdgcn_s_sleep(T2_STAGGER);
    peek_n<1>(base, idx, 1, gb);
    if (__all(tags_ok<1, 32>(ga, tag))) {
        unsigned r = (unsigned)ga[0];
#if T2_LATE_B
        asm volatile("" : "+v"(r));
#endif
        return r;
    }
#if T2_LATE_B
    // `gb` keeps its single definition (the load above): re-reads go into their own registers -- a variable that is loaded on one
    // path and re-loaded on another becomes a phi, and the copy into the phi's register is a wait for the load on EVERY path
    if (__all(tags_ok<1, 32>(gb, tag))) { unsigned r = (unsigned)gb[0]; asm volatile("" : "+v"(r)); return r; }
    u64 gc[1];
    peek_n<1>(base, idx, 1, gc);
    finish_n<1, 32>(base, idx, 1, tag, gc, dead, err, code);
    unsigned r = (unsigned)gc[0];
    asm volatile("" : "+v"(r));   // every path hands over a value that is waited for on that path
    return r;
#else
    finish_n<1, 32>(base, idx, 1, tag, gb, dead, err, code);
    return (unsigned)gb[0];
#endif
#endif
}
// look B has returned by the time the gathered value is in LDS and the workgroup barrier is passed: waiting for it HERE costs nothing and
// leaves no load pending whose registers the compiler would have to guard with a vmcnt(0) somewhere on the serial chain
__device__ __forceinline__ void retire_look(u64 (&gb)[1]) {
#if T2_LATE_B
    asm volatile("" ::"v"(gb[0]));
#endif
}

template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum(float v) {   // sum over the 16 lanes of a DPP row, result in every lane
    v += dpp_get<0xB1>(v);
    v += dpp_get<0x4E>(v);
    v += dpp_get<0x141>(v);
    v += dpp_get<0x140>(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {   // max over 64 lanes, valid in lane 63
    // One v_max_f32_dpp per step (the compiler's fmaxf + update_dpp form costs 4 VALU per step: copy, DPP move, two
    // canonicalising maxes).  s_nop 1 = the two wait states a DPP read of a just-written VGPR needs.
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

// activation vectors in LDS: element j -> plane p=(j>>2)&7, slot q=j>>5 (8 conflict-free ds_read_b128 per lane).  A plane
// is 64 floats + 4 of padding: the ds_write_b32 of 32 consecutive elements then touches 32 distinct banks (with 64-float
// planes the 8 planes of a 32-lane group alias onto 4 banks: 8-way conflict, 27 % of the LDS-active cycles in round 1).
#define XB_PLANE 68
#define XB_VEC (8 * XB_PLANE)
__device__ __forceinline__ int perm(int j) { return ((j >> 2) & 7) * XB_PLANE + (j >> 5) * 4 + (j & 3); }

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// three rows (GRU gates) x the lane's 32-float chunk, weights w[0..95] in VGPRs; the chunk is consumed in two
// halves of 16 floats to keep the live register set small
__device__ __forceinline__ void dot32x3(const float *w, const float *vec, int q, float &o0, float &o1, float &o2) {
    const float4 *p = (const float4 *)vec + q;
    f2 a = mk2(0.f, 0.f), b = mk2(0.f, 0.f), c = mk2(0.f, 0.f);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = p[(4 * h + k) * (XB_PLANE / 4)];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int kk = 4 * h + k;
            const f2 xl = mk2(x[k].x, x[k].y), xh = mk2(x[k].z, x[k].w);
            a = pkfma(mk2(w[4 * kk + 0], w[4 * kk + 1]), xl, a);
            b = pkfma(mk2(w[32 + 4 * kk + 0], w[32 + 4 * kk + 1]), xl, b);
            c = pkfma(mk2(w[64 + 4 * kk + 0], w[64 + 4 * kk + 1]), xl, c);
            a = pkfma(mk2(w[4 * kk + 2], w[4 * kk + 3]), xh, a);
            b = pkfma(mk2(w[32 + 4 * kk + 2], w[32 + 4 * kk + 3]), xh, b);
            c = pkfma(mk2(w[64 + 4 * kk + 2], w[64 + 4 * kk + 3]), xh, c);
        }
    }
    o0 = a.x + a.y; o1 = b.x + b.y; o2 = c.x + c.y;
}
// W_hh2 on the S waves: gate rows r, z from VGPRs w[0..63], gate row n from this thread's LDS slots
__device__ __forceinline__ void dot32x3_mixed(const float *w, const float4 *wl_, const float *vec, int q, float &o0, float &o1, float &o2) {
    const float4 *p = (const float4 *)vec + q;
    f2 a = mk2(0.f, 0.f), b = mk2(0.f, 0.f), c = mk2(0.f, 0.f);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 x[4], wn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[k] = p[(4 * h + k) * (XB_PLANE / 4)]; wn[k] = wl_[(4 * h + k) * 256]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int kk = 4 * h + k;
            const f2 xl = mk2(x[k].x, x[k].y), xh = mk2(x[k].z, x[k].w);
            a = pkfma(mk2(w[4 * kk + 0], w[4 * kk + 1]), xl, a);
            b = pkfma(mk2(w[32 + 4 * kk + 0], w[32 + 4 * kk + 1]), xl, b);
            c = pkfma(mk2(wn[k].x, wn[k].y), xl, c);
            a = pkfma(mk2(w[4 * kk + 2], w[4 * kk + 3]), xh, a);
            b = pkfma(mk2(w[32 + 4 * kk + 2], w[32 + 4 * kk + 3]), xh, b);
            c = pkfma(mk2(wn[k].z, wn[k].w), xh, c);
        }
    }
    o0 = a.x + a.y; o1 = b.x + b.y; o2 = c.x + c.y;
}
__device__ __forceinline__ float dot32(const float *w, const float *vec, int q) {
    const float4 *p = (const float4 *)vec + q;
    f2 s0 = mk2(0.f, 0.f), s1 = mk2(0.f, 0.f);
    float4 x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = p[k * (XB_PLANE / 4)];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s0 = pkfma(mk2(w[4 * k + 0], w[4 * k + 1]), mk2(x[k].x, x[k].y), s0);
        s1 = pkfma(mk2(w[4 * k + 2], w[4 * k + 3]), mk2(x[k].z, x[k].w), s1);
    }
    return (s0.x + s0.y) + (s1.x + s1.y);
}
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// ---- LDS carve-up (floats); per-thread-base + constant families first (16-bit DS offsets) ---------------
constexpr int L_COND = 0;                      // [512][4]  {cI, v_r, v_z, v_n} of the coming step (HBM stream -> everyone)
constexpr int L_CSTA = L_COND + 512 * 4;       // [512][4]  {W_I[:,0], u_r, u_z, u_n}
constexpr int L_GH1 = L_CSTA + 512 * 4;        // [3][512]  W_hh1.h1 + b_hh1 gathered from the team
constexpr int L_CSTQ = L_GH1 + 1536;           // [16 quarters][16] b_hh1 rzn | b_hh2 rzn | b3 x2 | c2 rzn | c3 | c4
constexpr int L_HAND = L_CSTQ + 256;           // [16 quarters][8]  S -> C: gh2 rzn | . | noise[2 parities][2]
constexpr int L_MISC = L_HAND + 128;           // scratch words
constexpr int L_XB = L_MISC + 64;              // 6 activation vectors x XB_VEC (plane order, padded planes)
constexpr int XB_H1 = 0, XB_X2 = 1, XB_X3 = 2, XB_H2 = 3, XB_F1 = 4, XB_F2 = 5;
constexpr int L_SW = L_XB + 6 * XB_VEC;           // [8 planes][256 S-threads][4]: the n-gate row of W_hh2 (32 weights / S thread)
constexpr int L_FC3 = L_SW + 8 * 256 * 4;      // [4 C-waves][2 rows][8 planes][64 lanes][4]
constexpr int L_TOTAL = L_FC3 + 16384;
static_assert(L_TOTAL * 4 <= 163840, "LDS budget");
constexpr int M_XF = 9, M_DEAD = 10;  // misc slots: fed-back sample, bail-out flag

constexpr unsigned G_X3 = 0, G_F1 = 1024, G_F2 = 2048, G_PR = 3072, G_GH = 4096;  // mailbox regions (granules)

}  // namespace

#define P2(i)                                                               \
    do {                                                                    \
        if (PROF) {                                                         \
            const u64 now_ = __builtin_readcyclecounter();                  \
            prof_acc[i] += now_ - prof_last;                                \
            prof_last = now_;                                               \
        }                                                                   \
    } while (0)

template <int MODE, bool PROF, bool RAGGED>
__global__ void __launch_bounds__(T2_THREADS, 2) loop_team2_kernel(WrnnTeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = (float *)smem;
    float *xb = lds + L_XB;
    float *gh1s = lds + L_GH1;
    int *misc_i = (int *)(lds + L_MISC);
    float *misc_f = lds + L_MISC;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool isC = wave < 4;
    const int wl = wave & 3;
    // the two waves of a SIMD compete for issue slots: the critical waves go first, the shadow waves fill the gaps (measured
    // A/B on one box, profiles/r03_team2_experiments.txt: 3.342 -> 3.305 us per step)
    if (isC) __builtin_amdgcn_s_setprio(3);
    const int r4 = lane >> 4, q = lane & 15;
    const WrnnDims d = a.d;
    const int NC = d.NC, HOP = d.HOP, T = a.T;
    const float4 *swl = (const float4 *)(lds + L_SW) + (tid - 256);   // S threads only

    // ---- team formation: by the XCD this workgroup actually runs on ------------
    if (tid == 0) {
        // ctl[0..7]: arrivals per physical XCC id; ctl[8]: team slots handed out; ctl[16 + xcc]: slot + 1 of that XCC.
        // Teams are numbered in order of first arrival, so any set of XCC ids (SPX, or a partition exposing a
        // subset of the XCDs) maps onto team slots 0..n_teams-1.
        const unsigned x = xcc_id2();
        misc_i[M_DEAD] = 0;
        const unsigned rank = atomicAdd(&a.ctl[x], 1u);
        unsigned slot1 = 0, arrived = 0;
        if (rank == 0) {
            slot1 = atomicAdd(&a.ctl[8], 1u) + 1u;
            __hip_atomic_store(&a.ctl[16 + x], slot1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // Co-residency, checked instead of assumed: the 32 workgroups of this XCD spin on each other for the whole launch, so
        // all of them must be running NOW.  One bounded wait (~0.1 s) for the team slot AND the arrival counter; if the counter
        // does not fill -- the GPU is shared with another process's kernel -- report WRNN_ERR_BUSY and leave instead of timing
        // out inside the loop.
        for (unsigned spins = 0; spins < WRNN_ARRIVE_POLLS; ++spins) {
            slot1 = __hip_atomic_load(&a.ctl[16 + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            arrived = __hip_atomic_load(&a.ctl[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (slot1 && arrived >= T2_WGS) break;
        }
        if (arrived < T2_WGS) { slot1 = 0; if (rank < T2_WGS) atomicCAS(a.err, 0u, WRNN_DEVERR_BUSY); }
        misc_i[0] = slot1 ? (int)slot1 - 1 : 1 << 20;   // no slot seen: treated as "not in a team" below
        misc_i[1] = (int)rank;
    }
    __syncthreads();
    const int team = __builtin_amdgcn_readfirstlane(misc_i[0]);
    const int g = __builtin_amdgcn_readfirstlane(misc_i[1]);
    __syncthreads();
    if (g >= T2_WGS || team >= a.n_teams || team >= a.n_rows) return;
    u64 *mail = a.mail + (size_t)team * WRNN_TEAM_MAIL_GRANULES;

    const int unit = 16 * g + 4 * wl + r4;          // hidden unit / fc1 / fc2 row of this quarter-wave
    const int qslot = wl * 4 + r4;
    const int c3row0 = 32 * g + 8 * wl + 2 * r4;    // first of the two fc3 rows of this C quarter-wave (even)
    const bool has_fc3 = c3row0 < NC;

    // ---- resident weights: team_w rows per (wl, lane): W_hh1 [0,96) | W_ih2 [96,192) | W_hh2 [192,288) | fc2 [288,320) | fc1 [320,352)
    float wv[160];
    {
        const float *src = a.team_w + (size_t)g * WRNN_TEAM_NWREG * WRNN_TEAM_THREADS + (wl * 64 + lane);
        if (isC) {
#pragma unroll
            for (int i = 0; i < 96; ++i) wv[i] = src[(size_t)(96 + i) * WRNN_TEAM_THREADS];     // W_ih2
#pragma unroll
            for (int i = 0; i < 64; ++i) wv[96 + i] = src[(size_t)(288 + i) * WRNN_TEAM_THREADS];  // fc2 | fc1
        } else {
#pragma unroll
            for (int i = 0; i < 96; ++i) wv[i] = src[(size_t)i * WRNN_TEAM_THREADS];            // W_hh1
#pragma unroll
            for (int i = 0; i < 64; ++i) wv[96 + i] = src[(size_t)(192 + i) * WRNN_TEAM_THREADS];  // W_hh2 gates r, z
#pragma unroll
            for (int k = 0; k < 8; ++k)                                                          // W_hh2 gate n -> LDS planes
                ((float4 *)(lds + L_SW))[k * 256 + (tid - 256)] =
                    make_float4(src[(size_t)(256 + 4 * k) * WRNN_TEAM_THREADS], src[(size_t)(257 + 4 * k) * WRNN_TEAM_THREADS],
                                src[(size_t)(258 + 4 * k) * WRNN_TEAM_THREADS], src[(size_t)(259 + 4 * k) * WRNN_TEAM_THREADS]);
        }
        const float4 *f3 = (const float4 *)(a.team_fc3 + (size_t)g * 16384);
        float4 *dst = (float4 *)(lds + L_FC3);
        for (int i = tid; i < 4096; i += T2_THREADS) dst[i] = f3[i];
        // per-unit phase-A constants, unit j = tid
        ((float4 *)(lds + L_CSTA))[tid] = make_float4(a.wI0[tid], a.u1[tid], a.u1[512 + tid], a.u1[1024 + tid]);
        if (!isC && q == 0) {
            float *cq = lds + L_CSTQ + qslot * 16;
            cq[0] = a.w[a.off.r1_bhh + unit]; cq[1] = a.w[a.off.r1_bhh + 512 + unit]; cq[2] = a.w[a.off.r1_bhh + 1024 + unit];
            cq[3] = a.w[a.off.r2_bhh + unit]; cq[4] = a.w[a.off.r2_bhh + 512 + unit]; cq[5] = a.w[a.off.r2_bhh + 1024 + unit];
            cq[6] = has_fc3 ? a.w[a.off.fc3_b + c3row0] : 0.0f;
            cq[7] = (c3row0 + 1 < NC) ? a.w[a.off.fc3_b + c3row0 + 1] : 0.0f;
        }
    }
    __syncthreads();
    float *cstQ = lds + L_CSTQ + qslot * 16;
    float *hand = lds + L_HAND + qslot * 8;
    const int pj = perm(tid), pu = perm(unit);
    const int sidx = tid - 256;   // S threads: 0..255, prepare units sidx and sidx + 256

    bool dead = false;
    unsigned epoch = 0;
    u64 prof_acc[17] = {0};
    u64 prof_last = 0;

    // Uniform batch: team t runs rows t, t + n_teams, ... for steps [seg0, seg0 + seg_len).  RAGGED (opts.frames_dev): team t runs
    // a.sched[t], a.sched[t + n_teams], ... (-1 = nothing; the longest-first rows dealt in snake order by rows_kernel, prologue.hip)
    // and every row stops at its own length.  A separate instantiation: the uniform kernel is at its register limit, one more
    // live scalar costs ~4 % of the step (measured: SGPR spills through v_writelane / v_readlane on the serial chain).
    for (int it = team; it < (RAGGED ? a.n_slots : a.n_rows); it += a.n_teams) {
        int row = it;
        if constexpr (RAGGED) {
            row = a.sched[it];
            if (row < 0) continue;
        }
        const WrnnRow rw = a.rows[row];
        int64_t seg_end = a.seg0 + a.seg_len;   // this launch runs steps [seg0, seg_end) of the row
        if constexpr (RAGGED) {
            if (a.seg0 >= rw.steps) continue;   // a shorter row: finished in an earlier segment
            if (seg_end > rw.steps) seg_end = rw.steps;
        }
        const bool resume = a.seg0 > 0;
        float *st = a.state + (size_t)row * WRNN_TEAM_STATE_FLOATS;   // [h1 | h2 | gh1 | gh2 | x]
        const float4 *CONDg = (const float4 *)a.tabCOND + ((size_t)row * a.seg_len - (size_t)a.seg0) * 512;
        const float *C2g = a.tabC2 + (size_t)rw.utt * (T + 1) * 1536;
        const float *C3g = a.tabC3 + (size_t)rw.utt * (T + 1) * 512;
        const float *C4g = a.tabC4 + (size_t)rw.utt * (T + 1) * 512;

        // first segment: h1 = h2 = 0, x = 0  (:194-196)  => gh1 = b_hh1, gh2 = b_hh2; later segments: the state the
        // previous launch left in `st`
        float h1_j = resume ? st[tid] : 0.0f;
        float xfeed = resume ? st[4096] : (a.x_init ? a.x_init[row] : 0.0f);   // x_{t-1} (:196)
        xb[XB_H2 * XB_VEC + pj] = resume ? st[512 + tid] : 0.0f;
        for (int i = tid; i < 1536; i += T2_THREADS) gh1s[i] = resume ? st[1024 + i] : a.w[a.off.r1_bhh + i];
        if (tid == 0) misc_f[M_XF] = xfeed;
        // frame of the step being prepared, tracked incrementally by the S waves (for the per-frame C constants)
        const int64_t pos0 = rw.start + a.seg0;
        int nfi = (int)(pos0 / HOP), nph = (int)(pos0 - (int64_t)nfi * HOP);
        int cst_frame = -1000000;   // frame whose c2/c3/c4 are in the C constants
        int pend_frame = -1;        // frame whose constants must be written in the next B4-B5 window
        float nzE0 = 0.f, nzE1 = 0.f, nzn0 = 0.f, nzn1 = 0.f;   // this lane's Philox block: draws for an even / odd step
        float4 cnext0 = make_float4(0.f, 0.f, 0.f, 0.f), cnext1 = cnext0;   // conditioning prefetched two steps ahead

        // S: everything of step ts that does not depend on x_{ts-1}: its conditioning (prefetched from the HBM
        // stream one call earlier) -> L_COND for units sidx / sidx+256, prefetch of step ts+1, the frame bookkeeping,
        // and the sampling noise of the paired C quarter -> hand[4 + 2*parity(ts)...]
        auto s_prepare = [&](int64_t ts, unsigned ep_of_ts) {
            const bool live = nfi < T;                  // fold padding 'after' = zero rows (:327-330)
            const int fi = live ? nfi : T;              // T = the all-zero conditioning entry
            if (++nph == HOP) { nph = 0; ++nfi; }
            pend_frame = fi;
            ((float4 *)(lds + L_COND))[sidx] = cnext0;
            ((float4 *)(lds + L_COND))[sidx + 256] = cnext1;
        };
        // S: HBM prefetch of the conditioning of step ts (consumed by s_prepare(ts) one step later).  Issued right
        // before a stretch of ALU work so that no exchange poll queues behind these ~1 us loads.
        auto s_prefetch = [&](int64_t ts) {
            if (ts < seg_end) {
                cnext0 = CONDg[(size_t)ts * 512 + sidx];
                cnext1 = CONDg[(size_t)ts * 512 + sidx + 256];
            }
        };
        // S: sampling noise of step ts for the paired C quarter -> hand[4 + 2*parity(ts) ...]
        auto s_noise = [&](int64_t ts, unsigned ep_of_ts) {
            if (MODE == WRNN_MODE_MOL && wave == 4 && lane <= NC / 3) {
                // MOL (distribution.py:106-121): the Gumbel noise of the mixture pick (lanes 0..9) and the logistic
                // noise of the sample (lane 10) do not depend on the logits: drawn one step ahead, off the critical path
                const int nr = NC / 3;
                float u;
                if (a.noise_mode == WRNN_NOISE_INJECTED)
                    u = lane < nr ? a.noise1[((size_t)ts * a.n_rows + row) * nr + lane] : a.noise2[(size_t)ts * a.n_rows + row];
                else
                    u = 1e-5f + wrnn_uniform(a.seed, (uint64_t)ts, (uint32_t)row, (uint32_t)lane) * (1.0f - 2e-5f);
                misc_f[32 + 16 * (ep_of_ts & 1u) + lane] = lane < nr ? -logf(-logf(u)) : logf(u) - logf(1.0f - u);
            }
            if (MODE == WRNN_MODE_RAW && has_fc3) {
                float nz0 = 0.f, nz1 = 0.f;
                if (a.noise_mode == WRNN_NOISE_INJECTED) {
                    const float *qp = a.noise1 + ((size_t)ts * a.n_rows + row) * NC + c3row0;
                    nz0 = -logf(qp[0]); nz1 = -logf(qp[1]);
                    if (q == 0) { hand[4 + 2 * (ep_of_ts & 1u)] = nz0; hand[5 + 2 * (ep_of_ts & 1u)] = nz1; }
                } else if (a.noise_mode == WRNN_NOISE_PHILOX) {
                    // One Philox block = classes (c3row0, c3row0+1) x steps (2s, 2s+1).  The 16 lanes of the quarter-wave
                    // evaluate 16 DIFFERENT blocks (the next 32 steps) at once instead of the same one 16 times: one
                    // Philox evaluation per lane every 32 steps; the lane that owns step ts hands its draw over.
                    if ((ts & 31) == 0) {
                        const Philox4 pz = wrnn_raw_block(a.seed, (uint64_t)ts + 2u * (unsigned)q, (uint32_t)row, (uint32_t)c3row0);
                        // -log q = -log(-log u): the inner log exactly (q is tiny for u -> 1, where v_log_f32 is not accurate
                        // relative to the result -- and such a draw tends to win the race), the outer one fast
                        nzE0 = -__logf(-logf(u01_from_bits(pz.x)));
                        nzE1 = -__logf(-logf(u01_from_bits(pz.y)));
                        nzn0 = -__logf(-logf(u01_from_bits(pz.z)));
                        nzn1 = -__logf(-logf(u01_from_bits(pz.w)));
                    }
                    if (q == (int)((ts >> 1) & 15)) {
                        const bool odd = (ts & 1) != 0;
                        hand[4 + 2 * (ep_of_ts & 1u)] = odd ? nzn0 : nzE0;
                        hand[5 + 2 * (ep_of_ts & 1u)] = odd ? nzn1 : nzE1;
                    }
                } else if (q == 0) { hand[4 + 2 * (ep_of_ts & 1u)] = 0.f; hand[5 + 2 * (ep_of_ts & 1u)] = 0.f; }
            }
        };
        // S, window B4-B5: per-frame constants of the C quarter (c2 rzn, c3, c4) once the frame changed
        auto s_frame_consts = [&]() {
            if (pend_frame >= 0 && pend_frame != cst_frame) {
                if (q == 0) {
                    const int fi = pend_frame;
                    cstQ[8] = C2g[(size_t)fi * 1536 + unit]; cstQ[9] = C2g[(size_t)fi * 1536 + 512 + unit];
                    cstQ[10] = C2g[(size_t)fi * 1536 + 1024 + unit];
                    cstQ[11] = C3g[(size_t)fi * 512 + unit];
                    cstQ[12] = C4g[(size_t)fi * 512 + unit];
                }
                cst_frame = pend_frame;
            }
        };
        if (!isC) {
            if (q == 0) {   // gh2 = b_hh2, or carried over
                hand[0] = resume ? st[2560 + unit] : cstQ[3];
                hand[1] = resume ? st[2560 + 512 + unit] : cstQ[4];
                hand[2] = resume ? st[2560 + 1024 + unit] : cstQ[5];
            }
            s_prefetch(a.seg0);
            s_prepare(a.seg0, epoch + 1);
            s_prefetch(a.seg0 + 1);
            s_noise(a.seg0, epoch + 1);
            s_frame_consts();
        }
        __syncthreads();

        int out_k = 0;
        float out_x = 0.0f;
        u64 lookb[1] = {0};   // the second look of an exchange (take_granule), retired behind the exchange's barrier
        for (int64_t t = a.seg0; t < seg_end; ++t) {
            ++epoch;
            const unsigned par = epoch & 1u;
            if (PROF) prof_last = __builtin_readcyclecounter();

            // ---- phase A (all 512 threads, unit j = tid): I + GRU1, replicated in every WG (:208-212) ----
            float x2_j;
            {
                const float4 cA = ((const float4 *)(lds + L_CSTA))[tid];
                const float4 cd = ((const float4 *)(lds + L_COND))[tid];
                const float ghr = gh1s[tid], ghz = gh1s[512 + tid], ghn = gh1s[1024 + tid];
                const float xprev = xfeed;
                const float xin = fmaf(cA.x, xprev, cd.x);
                const float rg = sigmoid_fast(fmaf(cA.y, xprev, cd.y) + ghr);
                const float zg = sigmoid_fast(fmaf(cA.z, xprev, cd.z) + ghz);
                const float ng = tanh_fast(fmaf(cA.w, xprev, cd.w) + rg * ghn);
                h1_j = (1.0f - zg) * ng + zg * h1_j;
                x2_j = xin + h1_j;
                xb[XB_H1 * XB_VEC + pj] = h1_j;
                xb[XB_X2 * XB_VEC + pj] = x2_j;
            }
            P2(0);
            __syncthreads();  // B1
            P2(1);

            if (isC) {
                // ---- phase B: GRU2 unit `unit` (:213-216); rows r,z,n of W_ih2[:, :512] . x2 ----
                const float h2o = xb[XB_H2 * XB_VEC + pu];
                const float x2u = xb[XB_X2 * XB_VEC + pu];
                const float g2r = hand[0], g2z = hand[1], g2n = hand[2];
                const float c2r = cstQ[8], c2z = cstQ[9], c2n = cstQ[10];
                float gr, gz, gn;
                dot32x3(wv, xb + XB_X2 * XB_VEC, q, gr, gz, gn);
                gr = row_sum(gr) + c2r; gz = row_sum(gz) + c2z; gn = row_sum(gn) + c2n;
                const float rg = sigmoid_fast(gr + g2r);
                const float zg = sigmoid_fast(gz + g2z);
                const float ng = tanh_fast(gn + rg * g2n);
                const float x3u = x2u + ((1.0f - zg) * ng + zg * h2o);
                if (q == 0) st_granule(mail, G_X3 + par * 512 + unit, epoch, __float_as_uint(x3u));
            } else {
                if (T2_DEFER_OUT && MODE == WRNN_MODE_RAW && g == 0 && tid == 256 && t > a.seg0) {   // outputs of step t - 1 (see the merge at the end of the step)
                    if (a.labels_out) a.labels_out[(size_t)row * a.steps + t - 1] = out_k;
                    a.samples_out[(size_t)row * a.steps + t - 1] = out_x;
                }
                // ---- S: gh1 for the next step = W_hh1 . h1' + b_hh1, published for everyone ----
                float sr, sz, sn;
                __builtin_amdgcn_s_sleep(2);   // let the critical waves' x2 reads go first
                dot32x3(wv, xb + XB_H1 * XB_VEC, q, sr, sz, sn);
                sr = row_sum(sr) + cstQ[0]; sz = row_sum(sz) + cstQ[1]; sn = row_sum(sn) + cstQ[2];
                if (q == 0) {
                    st_granule(mail, G_GH + par * 1536 + unit, epoch, __float_as_uint(sr));
                    st_granule(mail, G_GH + par * 1536 + 512 + unit, epoch, __float_as_uint(sz));
                    st_granule(mail, G_GH + par * 1536 + 1024 + unit, epoch, __float_as_uint(sn));
                }
            }
            P2(2);
            // ---- exchange 1 (all threads, granule tid): x3 = x + h2 ; h2' = x3 - x2 ----
            {
                const float x3 = __uint_as_float(take_granule(mail, G_X3 + par * 512 + tid, epoch, dead, a.err, 11u, lookb));
                xb[XB_X3 * XB_VEC + pj] = x3;
                xb[XB_H2 * XB_VEC + pj] = x3 - x2_j;
            }
            P2(3);
            __syncthreads();  // B2
            retire_look(lookb);
            P2(4);

            if (isC) {
                // ---- phase C: fc1 row `unit` (:217-218) ----
                const float s = row_sum(dot32(wv + 128, xb + XB_X3 * XB_VEC, q)) + cstQ[11];
                if (q == 0) st_granule(mail, G_F1 + par * 512 + unit, epoch, __float_as_uint(fmaxf(s, 0.0f)));
            } else {
                // ---- S: sampling noise of step t+1 (C reads the other parity slot this step) ----
                if (t + 1 < seg_end) s_noise(t + 1, epoch + 1);
            }
            P2(5);
            // ---- exchange 2: fc1 outputs ----
            {
                xb[XB_F1 * XB_VEC + pj] = __uint_as_float(take_granule(mail, G_F1 + par * 512 + tid, epoch, dead, a.err, 12u, lookb));
            }
            P2(6);
            __syncthreads();  // B3
            retire_look(lookb);
            P2(7);

            if (isC) {
                // ---- phase D: fc2 row `unit` (:220-221) ----
                const float s = row_sum(dot32(wv + 96, xb + XB_F1 * XB_VEC, q)) + cstQ[12];
                if (q == 0) st_granule(mail, G_F2 + par * 512 + unit, epoch, __float_as_uint(fmaxf(s, 0.0f)));
            } else {
                // ---- S: conditioning + noise of step t+1 ----
                if (t + 1 < seg_end) s_prepare(t + 1, epoch + 1);
            }
            P2(8);
            // ---- exchange 3: fc2 outputs ----
            {
                xb[XB_F2 * XB_VEC + pj] = __uint_as_float(take_granule(mail, G_F2 + par * 512 + tid, epoch, dead, a.err, 13u, lookb));
            }
            P2(9);
            __syncthreads();  // B4
            retire_look(lookb);
            P2(10);

            if (isC) {
                // ---- phase E: fc3 rows + race (:223, :231-235) ----
                float lg0, lg1;
                {
                    const float4 *xp = (const float4 *)(xb + XB_F2 * XB_VEC) + q;
                    const float4 *wp = (const float4 *)(lds + L_FC3) + (size_t)(wl * 2) * 8 * 64 + lane;
                    f2 pa = mk2(0.f, 0.f), pb = mk2(0.f, 0.f);
#pragma unroll
                    for (int kk = 0; kk < 8; kk += 4) {
                        float4 x[4], wa[4], wb[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { x[k] = xp[(kk + k) * (XB_PLANE / 4)]; wa[k] = wp[(kk + k) * 64]; wb[k] = wp[(8 + kk + k) * 64]; }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f2 xl = mk2(x[k].x, x[k].y), xh = mk2(x[k].z, x[k].w);
                            pa = pkfma(mk2(wa[k].x, wa[k].y), xl, pa); pb = pkfma(mk2(wb[k].x, wb[k].y), xl, pb);
                            pa = pkfma(mk2(wa[k].z, wa[k].w), xh, pa); pb = pkfma(mk2(wb[k].z, wb[k].w), xh, pb);
                        }
                    }
                    lg0 = row_sum(pa.x + pa.y) + cstQ[6];
                    lg1 = row_sum(pb.x + pb.y) + cstQ[7];
                }
                if (a.logits_out && q == 0 && has_fc3) {
                    float *lo = a.logits_out + ((size_t)t * a.n_rows + row) * NC + c3row0;
                    lo[0] = lg0;
                    if (c3row0 + 1 < NC) lo[1] = lg1;
                }
                if (MODE == WRNN_MODE_RAW) {
                    // winner of this quarter-wave's 2 classes: argmax logit_k - log q_k
                    // quarters beyond n_classes (bits < 10) own no class: they enter the race with -inf (and never read their
                    // noise slot, which the shadow wave leaves unwritten for them)
                    const float v0 = has_fc3 ? lg0 + hand[4 + 2 * par] : -INFINITY;
                    const float v1 = (c3row0 + 1 < NC) ? lg1 + hand[5 + 2 * par] : -INFINITY;
                    const bool p1 = v1 > v0;
                    if (q == 0) st_granule(mail, G_PR + par * 512 + 16 * g + qslot,
                                           (epoch << 10) | (unsigned)(p1 ? c3row0 + 1 : c3row0), __float_as_uint(p1 ? v1 : v0));
                    P2(13);
                    {
                        // ---- exchange 4: 512 {value,index} granules, 2 per lane over the 4 C waves; each wave leaves its
                        // winner in LDS, every thread merges the four after B5 (no serial tail on one wave) ----
                        __builtin_amdgcn_s_sleep(3);
                        u64 gq[2];
                        peek_n<2>(mail, G_PR + par * 512 + wl * 128 + lane * 2, 1, gq);
                        finish_n<2, 42>(mail, G_PR + par * 512 + wl * 128 + lane * 2, 1, epoch & 0x3fffffu, gq, dead, a.err, 14u);
                        P2(14);
                        const float va = __uint_as_float((unsigned)gq[0]), vb = __uint_as_float((unsigned)gq[1]);
                        const bool pb_ = vb > va;
                        const float best = pb_ ? vb : va;
                        const int besti = (int)(((pb_ ? gq[1] : gq[0]) >> 32) & 1023u);
                        const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_max(best)), 63));
                        const u64 ball = __ballot(best == mx);
                        const int src = (int)__builtin_ctzll(ball ? ball : 1ull);
                        const int k = __builtin_amdgcn_readlane(besti, src);
                        if (lane == 0) { misc_f[16 + 2 * wl] = mx; misc_i[17 + 2 * wl] = k; }
                    }
                } else {
                    // MOL (distribution.py:87-123): the 30 fc3 outputs are exchanged, wave 0 of every WG samples
                    if (q == 0 && has_fc3) {
                        st_granule(mail, G_PR + par * 512 + c3row0, epoch, __float_as_uint(lg0));
                        if (c3row0 + 1 < NC) st_granule(mail, G_PR + par * 512 + c3row0 + 1, epoch, __float_as_uint(lg1));
                    }
                    if (wave == 0) {
                        const int nr = NC / 3;
                        float mylg = 0.0f;
                        if (lane < NC) {
                            u64 gq[1];
                            peek_n<1>(mail, G_PR + par * 512 + lane, 1, gq);
                            unsigned spins = 0;
                            while (!dead && (unsigned)(gq[0] >> 32) != epoch) {
                                if (++spins > T2_SPIN_MAX) { dead = true; atomicExch(a.err, 16u); break; }
                                peek_n<1>(mail, G_PR + par * 512 + lane, 1, gq);
                            }
                            mylg = __uint_as_float((unsigned)gq[0]);
                        }
                        const float v = lane < nr ? mylg + misc_f[32 + 16 * par + lane] : -INFINITY;
                        const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_max(v)), 63));
                        const u64 ball = __ballot(v == mx);
                        const int km = (int)__builtin_ctzll(ball ? ball : 1ull);
                        // km is wave-uniform (from the ballot): v_readlane instead of a ds_bpermute round trip
                        const float mean = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mylg), nr + km));
                        const float ls = fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(mylg), 2 * nr + km)), -32.23619130191664f);
                        float xs = mean + expf(ls) * misc_f[32 + 16 * par + nr];
                        xs = fminf(fmaxf(xs, -1.0f), 1.0f);
                        if (lane == 0) {
                            float xf = xs;
                            if (a.x_forced) { float v = a.x_forced[(size_t)t * a.n_rows + row]; asm volatile("" : "+v"(v)); xf = v; }
                            misc_f[M_XF] = xf;
                            if (g == 0) {
                                if (a.labels_out) a.labels_out[(size_t)row * a.steps + t] = km;
                                a.samples_out[(size_t)row * a.steps + t] = xs;
                            }
                        }
                    }
                }
            } else {
                // ---- S, window B4-B5: per-frame constants for the C quarter; HBM prefetch of the conditioning two steps
                // ahead; the gh1 gather (published a window ago; it queues behind the prefetch, which keeps the LDS pipe
                // free for the critical waves' fc3 reads meanwhile); gh2 last, while the critical waves sit in the race
                // exchange (measured: 3.40 vs 3.45 us/step against "gh2 first") ----
                s_frame_consts();
                s_prefetch(t + 2);
                u64 gq[6];
                peek_n<6>(mail, G_GH + par * 1536 + sidx, 256, gq);
                finish_n<6, 32>(mail, G_GH + par * 1536 + sidx, 256, epoch, gq, dead, a.err, 15u);
#pragma unroll
                for (int m = 0; m < 6; ++m) gh1s[sidx + m * 256] = __uint_as_float((unsigned)gq[m]);
                P2(13);
                {
                    // gh2 for the next step = W_hh2 . h2' + b_hh2 -> hand-off slot of the paired C quarter (C read the
                    // old value back in phase B, three barriers ago)
                    float sr, sz, sn;
                    dot32x3_mixed(wv + 96, swl, xb + XB_H2 * XB_VEC, q, sr, sz, sn);
                    sr = row_sum(sr) + cstQ[3]; sz = row_sum(sz) + cstQ[4]; sn = row_sum(sn) + cstQ[5];
                    if (q == 0) { hand[0] = sr; hand[1] = sz; hand[2] = sn; }
                }
            }
            P2(11);
            __syncthreads();  // B5
            P2(12);
            if (MODE == WRNN_MODE_RAW) {
                // merge the four per-wave race winners (ties -> lower wave = lower class index range)
                const float4 m0 = *(const float4 *)(misc_f + 16), m1 = *(const float4 *)(misc_f + 20);
                float bv = m0.x; int bk = __float_as_int(m0.y);
                if (m0.z > bv) { bv = m0.z; bk = __float_as_int(m0.w); }
                if (m1.x > bv) { bv = m1.x; bk = __float_as_int(m1.y); }
                if (m1.z > bv) { bv = m1.z; bk = __float_as_int(m1.w); }
                // sample = 2 * k / (n_classes - 1.) - 1.   (:235)
                const float x_new = 2.0f * (float)bk / ((float)NC - 1.0f) - 1.0f;
                // (:237) The teacher-forced value is waited for INSIDE its branch: with `cond ? load : x_new` the compiler cannot know at
                // the top of the next step whether a load into xfeed is pending and opens EVERY step with `s_waitcnt vmcnt(0)` -- which also
                // waits for workgroup 0's output stores just below and for the S waves' conditioning prefetch of step t + 2 (round 4,
                // found in the ISA; the same in loop_batch*.hip).
                xfeed = x_new;
                if (a.x_forced) { float v = a.x_forced[(size_t)t * a.n_rows + row]; asm volatile("" : "+v"(v)); xfeed = v; }
                // T2_DEFER_OUT 1 (developer knob, NOT shipped): the outputs of step t stored one step later by a shadow wave of workgroup 0
                // instead of by thread 0 here, to keep the store acknowledgements off the critical waves.  Measured: 297 -> 293.5
                // ksamples/s -- the acknowledgements were never on the serial chain.
                out_k = bk; out_x = x_new;
#if !T2_DEFER_OUT
                if (g == 0 && tid == 0) {
                    if (a.labels_out) a.labels_out[(size_t)row * a.steps + t] = bk;
                    a.samples_out[(size_t)row * a.steps + t] = x_new;
                }
#endif
            } else {
                xfeed = misc_f[M_XF];
            }
            if ((t & 63) == 63) {   // bounded-spin bail-out, checked workgroup-wide every 64 steps
                if (dead && lane == 0) misc_i[M_DEAD] = 1;
                __syncthreads();
                if (misc_i[M_DEAD]) return;
            }
        }
        __syncthreads();
        if (T2_DEFER_OUT && MODE == WRNN_MODE_RAW && g == 0 && tid == 256 && seg_end > a.seg0) {   // the last step's outputs
            if (a.labels_out) a.labels_out[(size_t)row * a.steps + seg_end - 1] = out_k;
            a.samples_out[(size_t)row * a.steps + seg_end - 1] = out_x;
        }
        if (seg_end < (RAGGED ? (int64_t)a.rows[row].steps : a.steps)) {   // hand the recurrent state to the next segment's launch
            if (g == 0) {
                st[tid] = h1_j;
                st[512 + tid] = xb[XB_H2 * XB_VEC + pj];
                for (int i = tid; i < 1536; i += T2_THREADS) st[1024 + i] = gh1s[i];
                if (tid == 0) st[4096] = xfeed;
            }
            if (!isC && q == 0) { st[2560 + unit] = hand[0]; st[2560 + 512 + unit] = hand[1]; st[2560 + 1024 + unit] = hand[2]; }
        }
    }
    if (PROF && a.prof && lane == 0 && g == 0 && team == 0) {
        for (int i = 0; i < 17; ++i) a.prof[wave * WRNN_PROF_SLOTS + i] += prof_acc[i];
    }
}

// which instantiation (mode, prof, ragged) launches: the instrumented build exists for RAW uniform batches only
static const void *team2_fn(int mode, bool prof, bool ragged) {
    if (mode == WRNN_MODE_RAW) {
        if (ragged) return (const void *)loop_team2_kernel<WRNN_MODE_RAW, false, true>;
        return prof ? (const void *)loop_team2_kernel<WRNN_MODE_RAW, true, false> : (const void *)loop_team2_kernel<WRNN_MODE_RAW, false, false>;
    }
    return ragged ? (const void *)loop_team2_kernel<WRNN_MODE_MOL, false, true> : (const void *)loop_team2_kernel<WRNN_MODE_MOL, false, false>;
}

hipError_t wrnn_team2_occupancy(int mode, bool prof, int *blocks_per_cu, size_t *lds_bytes) {
    const size_t lds = (size_t)L_TOTAL * sizeof(float);
    *lds_bytes = lds;
    int worst = 1 << 30;
    for (int ragged = 0; ragged < 2; ++ragged) {
        const void *fn = team2_fn(mode, prof, ragged != 0);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        int blocks = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, T2_THREADS, lds);
        if (e != hipSuccess) return e;
        if (blocks < worst) worst = blocks;
    }
    *blocks_per_cu = worst;
    return hipSuccess;
}

hipError_t wrnn_launch_loop_team2(const WrnnTeamArgs &a, hipStream_t s) {
    (void)hipGetLastError();  // the runtime is shared with PyTorch: drop any stale sticky error of this thread
    const size_t lds = (size_t)L_TOTAL * sizeof(float);
    const bool ragged = a.ragged != 0, prof = a.prof != nullptr && a.d.mode == WRNN_MODE_RAW && !ragged;
    // the attribute is per device (function objects are per-device in the runtime): set it on every launch, it is a
    // host-side table write
    const void *fn = team2_fn(a.d.mode, prof, ragged);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    void *args[] = {(void *)&a};
    return hipLaunchKernel(fn, dim3(a.n_teams * 32), dim3(T2_THREADS), args, lds, s);
}
