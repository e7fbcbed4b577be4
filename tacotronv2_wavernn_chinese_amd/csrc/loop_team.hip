// WRNN_KERNEL_TEAM: the low-latency per-sample loop (fatchord_version.py:194-241).
//
// One *team* = the 32 workgroups resident on the 32 CUs of one XCD; a team runs
// one loop row (utterance or fold) at a time; the 8 XCDs run 8 rows concurrently.
// Why this shape (numbers: DESIGN.md, bench_micro/handoff.hip on MI355X):
//  * batch-1 generation is a serial chain of matvecs; per-step FLOPs are tiny
//    (4.3 MMAC) but every step must see all 13.6 MB of loop weights, so the weights
//    must be ON CHIP and the chain must not pay an HBM/L2 stream per step;
//  * 13.6 MB fp32 fits the register files of 32 CUs (32 x 512 KB): each workgroup
//    is 4 waves (one per SIMD, the full 512-register budget each) and every thread
//    keeps its 352 weights in VGPRs/AGPRs for the whole kernel; fc3's slice
//    (64 KB) lives in LDS;
//  * CUs of one XCD share an L2, so a plain store + L1-bypassing (sc1) load
//    exchanges an 8-byte {tag,value} granule in ~0.27 us; crossing XCDs costs
//    0.45-0.57 us and a chip-wide gather 2.6 us -> the team stays inside one XCD;
//  * algebra removes work from the serial chain: the I layer and W_ih1 act on
//    (x_{t-1}, conditioning) linearly, so gi1 = u * x_{t-1} + v[t] with
//    u = W_ih1 . W_I[:,0] and v[t] a per-frame table pushed through the upsampling
//    taps; W_hh1.h1, W_hh2.h2, the conditioning and the sampling noise never wait
//    on x_t and are computed in the shadow of the exchanges.
// Per step: 4 intra-XCD exchanges on the critical path (x+h2, fc1, fc2, race
// winners) + 1 off the path (W_hh1.h1), 5 workgroup barriers.
//
// Thread map (256 threads = 4 waves; lane l: quarter r4 = l>>4, q = l&15):
//   quarter-wave (w,r4) owns hidden unit / fc row  u = 16 g + 4 w + r4  of its WG g
//   and columns 32q..32q+31 of every row it owns -> a row dot product is 32 FMAs
//   per lane + a 4-step DPP reduction inside the 16-lane row; nothing leaves the
//   quarter-wave before the publish.
#include "device_util.h"
#include "wrnn_internal.h"

#define TEAM_WGS 32
#define TEAM_THREADS 256
#define TEAM_SPIN_MAX 300000u

typedef unsigned long long u64;

namespace {

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// 8-byte granule {tag (hi), payload (lo)}.  Producer: plain store (lands in the
// XCD's L2 through the write-through L1).  Consumer: sc1 load (bypasses the CU's
// L1, served by that same L2).  Valid ONLY between CUs of one XCD -- which is how
// teams are formed (by HW_REG_XCC_ID), never assumed from blockIdx.
// Addressing: uniform base (SGPR pair) + per-lane 32-bit byte offset (VGPR).
__device__ __forceinline__ void st_granule(u64 *base, unsigned idx, unsigned tag, unsigned payload) {
    const u64 v = ((u64)tag << 32) | payload;
    const unsigned off = idx * 8u;
    asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}

// First look at N granules (relaxed agent-scope load = global_load ... sc1, compiler-visible so that it can be
// issued in the middle of shadow work and waited for at first use: its ~600-cycle L2 round trip is hidden).
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
// Complete an exchange: re-read until every lane of the wave holds granules tagged `tag`.  The loop is
// WAVE-uniform (no exec-mask bookkeeping): lanes that already hold their data simply re-read it.  Bounded:
// on a timeout the (wave-uniform) dead flag is raised, the error word set, and every later poll is skipped.
template <int N, int SHIFT>
__device__ __forceinline__ void finish_n(const u64 *base, unsigned idx, unsigned stride, unsigned tag, u64 (&g)[N], bool &dead,
                                         unsigned *err, unsigned code) {
    unsigned spins = 0;
    while (!dead && !__all(tags_ok<N, SHIFT>(g, tag))) {
        if (++spins > TEAM_SPIN_MAX) { dead = true; if ((threadIdx.x & 63) == 0) atomicExch(err, code); break; }
        peek_n<N>(base, idx, stride, g);
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// Sum over the 16 lanes of each DPP row; every lane of the row gets the total.
__device__ __forceinline__ float row_sum(float v) {
    v += dpp_get<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_get<0x141>(v);  // row_half_mirror
    v += dpp_get<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ float wave_max(float v) {   // max over 64 lanes, valid in lane 63
#define WMAX_STEP(CTRL, RM) \
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, RM, 0xf, false)))
    WMAX_STEP(0xB1, 0xf); WMAX_STEP(0x4E, 0xf); WMAX_STEP(0x141, 0xf); WMAX_STEP(0x140, 0xf);
    WMAX_STEP(0x142, 0xa);  // row_bcast15 -> rows 1,3
    WMAX_STEP(0x143, 0xc);  // row_bcast31 -> rows 2,3
#undef WMAX_STEP
    return v;
}

// x vectors live in LDS in "plane" order so that the 32-float chunk of lane q is
// eight conflict-free ds_read_b128: element j -> plane p=(j>>2)&7, slot q=j>>5.
__device__ __forceinline__ int perm(int j) { return ((j >> 2) & 7) * 64 + (j >> 5) * 4 + (j & 3); }

// Shadow-phase weights (W_hh1, W_hh2) are pinned to the accumulator half of the register file
// ("a" constraint) so that the critical-path weights (W_ih2, fc1, fc2) and all temporaries own the
// 256 architectural VGPRs: an AGPR operand costs one v_accvgpr_read before use, which is paid only
// in work that runs in the shadow of an exchange.
__device__ __forceinline__ void agpr_put(float &a, float v) { asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); }
__device__ __forceinline__ float agpr_get(const float &a) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(a));
    return t;
}

// The lane's 32-float chunk of an LDS vector (8 conflict-free ds_read_b128, all issued before use)
struct X32 { float4 v[8]; };
__device__ __forceinline__ X32 load_chunk(const float *vec, int q) {
    const float4 *p = (const float4 *)vec + q;
    X32 x;
#pragma unroll
    for (int k = 0; k < 8; ++k) x.v[k] = p[k * 16];
    return x;
}
// Packed fp32 math: v_pk_fma_f32 does two FMAs per issue slot, which matters with one wave per SIMD
// (every instruction's issue + dependency latency is exposed).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// 32-term dot product with AGPR-pinned weights ah[0..31]: reads batched 8 at a time
__device__ __forceinline__ float dot32_agpr(const float *ah, const X32 &x) {
    f2 s0 = mk2(0.f, 0.f), s1 = mk2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = agpr_get(ah[4 * k + e]);
        s0 = pkfma(mk2(w[0], w[1]), mk2(x.v[k].x, x.v[k].y), s0);
        s1 = pkfma(mk2(w[2], w[3]), mk2(x.v[k].z, x.v[k].w), s1);
        s0 = pkfma(mk2(w[4], w[5]), mk2(x.v[k + 1].x, x.v[k + 1].y), s0);
        s1 = pkfma(mk2(w[6], w[7]), mk2(x.v[k + 1].z, x.v[k + 1].w), s1);
    }
    return (s0.x + s0.y) + (s1.x + s1.y);
}
// three rows sharing one chunk (GRU gates r, z, n), VGPR-resident weights w[0..95]
__device__ __forceinline__ void dot32x3(const float *w, const X32 &x, float &o0, float &o1, float &o2) {
    f2 a = mk2(0.f, 0.f), b = mk2(0.f, 0.f), c = mk2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f2 xl = mk2(x.v[k].x, x.v[k].y), xh = mk2(x.v[k].z, x.v[k].w);
        a = pkfma(mk2(w[4 * k + 0], w[4 * k + 1]), xl, a);
        b = pkfma(mk2(w[32 + 4 * k + 0], w[32 + 4 * k + 1]), xl, b);
        c = pkfma(mk2(w[64 + 4 * k + 0], w[64 + 4 * k + 1]), xl, c);
        a = pkfma(mk2(w[4 * k + 2], w[4 * k + 3]), xh, a);
        b = pkfma(mk2(w[32 + 4 * k + 2], w[32 + 4 * k + 3]), xh, b);
        c = pkfma(mk2(w[64 + 4 * k + 2], w[64 + 4 * k + 3]), xh, c);
    }
    o0 = a.x + a.y; o1 = b.x + b.y; o2 = c.x + c.y;
}
// the same with AGPR-pinned weights ah[0..95]: 12 reads then 6 packed FMAs per plane; planes [K0, K1) only,
// accumulating into the pair accumulators so a caller can interleave other work between the two halves
template <int K0, int K1>
__device__ __forceinline__ void dot32x3_agpr(const float *ah, const X32 &x, f2 &a, f2 &b, f2 &c) {
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        float w[12];
#pragma unroll
        for (int e = 0; e < 4; ++e) { w[e] = agpr_get(ah[4 * k + e]); w[4 + e] = agpr_get(ah[32 + 4 * k + e]); w[8 + e] = agpr_get(ah[64 + 4 * k + e]); }
        const f2 xl = mk2(x.v[k].x, x.v[k].y), xh = mk2(x.v[k].z, x.v[k].w);
        a = pkfma(mk2(w[0], w[1]), xl, a); b = pkfma(mk2(w[4], w[5]), xl, b); c = pkfma(mk2(w[8], w[9]), xl, c);
        a = pkfma(mk2(w[2], w[3]), xh, a); b = pkfma(mk2(w[6], w[7]), xh, b); c = pkfma(mk2(w[10], w[11]), xh, c);
    }
}
// fast gate non-linearities: v_exp_f32 + v_rcp_f32 (each ~1 ulp)
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// LDS carve-up (floats).  Everything addressed as "per-thread base + constant" sits in the first
// 64 KB so the constant folds into the 16-bit DS offset field (one address VGPR per access family
// instead of one per array); the 64 KB fc3 image goes last.
constexpr int REC_F = 28;                   // floats per conditioning record (pack_records_kernel)
constexpr int L_REC = 0;                    // [512 units][28]  conditioning records of the current frame
constexpr int L_KT = L_REC + 512 * REC_F;   // [HOP <= 275][8] composite upsampling taps, 5 used per phase
constexpr int L_GH1 = L_KT + 275 * 8 + 8;          // [3][512]
constexpr int L_MISC = L_GH1 + 1536;        // 64 floats of scratch
constexpr int L_CST = L_MISC + 64;          // per-thread constants: [8][256] phase-A (wI0,u_r,u_z,u_n x2), [16][16] per quarter-wave
constexpr int L_LUT = L_CST + 8 * 256 + 16 * 16;   // [1024] label -> fed-back sample value (RAW)
constexpr int L_XB = L_LUT + 1024;          // 6 vectors x 512 (plane order)
constexpr int XB_H1 = 0, XB_X2 = 1, XB_X3 = 2, XB_H2 = 3, XB_F1 = 4, XB_F2 = 5;
constexpr int L_FC3 = L_XB + 6 * 512;       // [4 waves][2 rows][8 planes][64 lanes][4]  = 16384
constexpr int L_TOTAL = L_FC3 + 16384;

}  // namespace

// PROF: accumulate s_memtime deltas per phase (developer instrumentation, tools/quick_check.py --prof)
#define PROF_MARK(i)                                                        \
    do {                                                                    \
        if (PROF) {                                                         \
            const u64 now_ = __builtin_readcyclecounter();                  \
            prof_acc[i] += now_ - prof_last;                                \
            prof_last = now_;                                               \
        }                                                                   \
    } while (0)

template <int MODE, bool PROF>
__global__ void __launch_bounds__(TEAM_THREADS, 1) loop_team_kernel(WrnnTeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = (float *)smem;
    float *xb = lds + L_XB;
    float *gh1s = lds + L_GH1;
    int *misc_i = (int *)(lds + L_MISC);
    float *misc_f = lds + L_MISC;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r4 = lane >> 4, q = lane & 15;
    const WrnnDims d = a.d;
    const int NC = d.NC, HOP = d.HOP, T = a.T;

    // ---- team formation: by the XCD this workgroup actually runs on ------------
    if (tid == 0) {
        // ctl[0..7]: arrivals per physical XCC id; ctl[8]: team slots handed out; ctl[16 + xcc]: slot + 1 of that XCC.
        // Teams are numbered in order of first arrival, so any set of XCC ids (SPX, or a partition exposing a
        // subset of the XCDs) maps onto team slots 0..n_teams-1.
        const unsigned x = xcc_id();
        misc_i[10] = 0;
        const unsigned rank = atomicAdd(&a.ctl[x], 1u);
        unsigned slot1 = 0;
        if (rank == 0) {
            slot1 = atomicAdd(&a.ctl[8], 1u) + 1u;
            __hip_atomic_store(&a.ctl[16 + x], slot1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (unsigned spins = 0; spins < 4000000u; ++spins) {
                slot1 = __hip_atomic_load(&a.ctl[16 + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (slot1) break;
            }
        }
        misc_i[0] = slot1 ? (int)slot1 - 1 : 1 << 20;   // no slot seen: treated as "not in a team" below
        misc_i[1] = (int)rank;
    }
    __syncthreads();
    const int team = __builtin_amdgcn_readfirstlane(misc_i[0]);
    const int g = __builtin_amdgcn_readfirstlane(misc_i[1]);
    __syncthreads();
    if (g >= TEAM_WGS || team >= a.n_teams || team >= a.n_rows) return;

    u64 *mail = a.mail + (size_t)team * WRNN_TEAM_MAIL_GRANULES;
    u64 *mX3 = mail, *mF1 = mail + 2 * 512, *mF2 = mail + 4 * 512, *mPR = mail + 6 * 512, *mGH = mail + 8 * 512;

    // ---- resident weights ---------------------------------------------------------
    // team_w order per thread: W_hh1 [0,96) | W_ih2 [96,192) | W_hh2 [192,288) | fc2 [288,320) | fc1 [320,352)
    float wv[96];    // VGPRs: W_ih2 (the largest critical-path matrix)
    float ah[256];   // AGPRs: W_hh1 [0,96) | W_hh2 [96,192) (shadow work) | fc2 [192,224) | fc1 [224,256)
    {
        const float *src = a.team_w + (size_t)g * WRNN_TEAM_NWREG * TEAM_THREADS + tid;
#pragma unroll
        for (int i = 0; i < 96; ++i) agpr_put(ah[i], src[(size_t)i * TEAM_THREADS]);
#pragma unroll
        for (int i = 0; i < 96; ++i) agpr_put(ah[96 + i], src[(size_t)(192 + i) * TEAM_THREADS]);
#pragma unroll
        for (int i = 0; i < 96; ++i) wv[i] = src[(size_t)(96 + i) * TEAM_THREADS];
#pragma unroll
        for (int i = 0; i < 64; ++i) agpr_put(ah[192 + i], src[(size_t)(288 + i) * TEAM_THREADS]);
        const float4 *f3 = (const float4 *)(a.team_fc3 + (size_t)g * 16384);
        float4 *dst = (float4 *)(lds + L_FC3);
        for (int i = tid; i < 4096; i += TEAM_THREADS) dst[i] = f3[i];
        for (int i = tid; i < HOP * 8; i += TEAM_THREADS) lds[L_KT + i] = (i & 7) < 5 ? a.w[a.off.ktab + (i >> 3) * 5 + (i & 7)] : 0.0f;
        // sample = 2 * k / (n_classes - 1.) - 1.  (:235), IEEE fp32 like the reference evaluates it
        for (int i = tid; i < 1024; i += TEAM_THREADS) lds[L_LUT + i] = 2.0f * (float)i / ((float)NC - 1.0f) - 1.0f;
    }
    const float *W_IH2 = wv;
    const float *A_HH1 = ah, *A_HH2 = ah + 96, *A_FC2 = ah + 192, *A_FC1 = ah + 224;
    const int unit = 16 * g + 4 * wave + r4;          // hidden unit / fc1 / fc2 row of this quarter-wave
    const int c3row0 = 32 * g + 8 * wave + 2 * r4;    // first of the two fc3 rows of this quarter-wave (even)
    const bool has_fc3 = c3row0 < NC;
    // constants live in LDS, not registers (the VGPR budget belongs to W_ih2 and the chunk reads):
    //   phase A, units j0 = tid and j1 = tid + 256:  [k][tid], k = wI0_0, wI0_1, u_r0, u_r1, u_z0, u_z1, u_n0, u_n1
    //   per quarter-wave:                            [qid][8] = b_hh1 r,z,n | b_hh2 r,z,n | b3 of its two fc3 rows
    const int j0 = tid, j1 = tid + 256;
    {
        float *cst = lds + L_CST;
        cst[0 * 256 + tid] = a.wI0[j0]; cst[1 * 256 + tid] = a.wI0[j1];
        cst[2 * 256 + tid] = a.u1[j0]; cst[3 * 256 + tid] = a.u1[j1];
        cst[4 * 256 + tid] = a.u1[512 + j0]; cst[5 * 256 + tid] = a.u1[512 + j1];
        cst[6 * 256 + tid] = a.u1[1024 + j0]; cst[7 * 256 + tid] = a.u1[1024 + j1];
        if (q == 0) {
            float *cq = lds + L_CST + 8 * 256 + (wave * 4 + r4) * 16;
            cq[0] = a.w[a.off.r1_bhh + unit]; cq[1] = a.w[a.off.r1_bhh + 512 + unit]; cq[2] = a.w[a.off.r1_bhh + 1024 + unit];
            cq[3] = a.w[a.off.r2_bhh + unit]; cq[4] = a.w[a.off.r2_bhh + 512 + unit]; cq[5] = a.w[a.off.r2_bhh + 1024 + unit];
            cq[6] = has_fc3 ? a.w[a.off.fc3_b + c3row0] : 0.0f;
            cq[7] = (c3row0 + 1 < NC) ? a.w[a.off.fc3_b + c3row0 + 1] : 0.0f;
        }
    }
    __syncthreads();
    const float *cstA = lds + L_CST + tid;
    float *cstQ = lds + L_CST + 8 * 256 + (wave * 4 + r4) * 16;
#define wI0_0 cstA[0 * 256]
#define wI0_1 cstA[1 * 256]
#define ur0 cstA[2 * 256]
#define ur1 cstA[3 * 256]
#define uz0 cstA[4 * 256]
#define uz1 cstA[5 * 256]
#define un0 cstA[6 * 256]
#define un1 cstA[7 * 256]
#define bhh1_r cstQ[0]
#define bhh1_z cstQ[1]
#define bhh1_n cstQ[2]
#define bhh2_r cstQ[3]
#define bhh2_z cstQ[4]
#define bhh2_n cstQ[5]
#define b3_0 cstQ[6]
#define b3_1 cstQ[7]
#define c2_r cstQ[8]
#define c2_z cstQ[9]
#define c2_n cstQ[10]
#define c3v cstQ[11]
#define c4v cstQ[12]
    const int pj0 = perm(j0), pj1 = perm(j1), pu = perm(unit);

    bool dead = false;
    unsigned epoch = 0;
    u64 prof_acc[17] = {0};
    u64 prof_last = 0;

    for (int row = team; row < a.n_rows; row += a.n_teams) {
        const WrnnRow rw = a.rows[row];
        const float *RECg = a.tabREC + (size_t)rw.utt * (T + 1) * 512 * REC_F;
        const float *C2g = a.tabC2 + (size_t)rw.utt * (T + 1) * 1536;
        const float *C3g = a.tabC3 + (size_t)rw.utt * (T + 1) * 512;
        const float *C4g = a.tabC4 + (size_t)rw.utt * (T + 1) * 512;

        // h1 = h2 = 0, x = 0  (:194-196)  => gh1 = b_hh1, gh2 = b_hh2
        float h1_0 = 0.0f, h1_1 = 0.0f;
        float xprev = 0.0f;
        xb[XB_H2 * 512 + pj0] = 0.0f;
        xb[XB_H2 * 512 + pj1] = 0.0f;
        for (int i = tid; i < 1536; i += TEAM_THREADS) gh1s[i] = a.w[a.off.r1_bhh + i];
        float gh2_r = bhh2_r, gh2_z = bhh2_z, gh2_n = bhh2_n;
        int cur_frame = -1000000;
        // conditioning / noise of the step about to run (software-pipelined: computed one step ahead,
        // in the shadow of the last exchange)
        float cI0 = 0.f, cI1 = 0.f, vr0 = 0.f, vz0 = 0.f, vn0 = 0.f, vr1 = 0.f, vz1 = 0.f, vn1 = 0.f;
        float nz0 = 0.f, nz1 = 0.f, nzn0 = 0.f, nzn1 = 0.f, xforce = 0.f;

        // Everything about step `ts` that does not depend on x_{ts-1}.  Executed by all waves at a
        // workgroup-uniform point (it may hit the per-frame table reload, which has barriers).
        // position of the step being prepared, tracked incrementally (no 64-bit division per step)
        int nfi = (int)(rw.start / HOP), nph = (int)(rw.start - (int64_t)nfi * HOP);
        auto prepare_step = [&](int64_t ts) {
            const bool live = nfi < T;                  // fold padding 'after' = zero rows (:327-330)
            const int fi = live ? nfi : T;              // frame index; T = the all-zero conditioning entry
            const int ph = live ? nph : 0;
            if (++nph == HOP) { nph = 0; ++nfi; }
            if (fi != cur_frame) {
                // conditioning records of frame fi -> LDS (once per hop_length steps): a straight 56 KB copy
                __syncthreads();
                const float4 *src = (const float4 *)(RECg + (size_t)fi * 512 * REC_F);
                float4 *dst = (float4 *)(lds + L_REC);
#pragma unroll
                for (int i = 0; i < 512 * REC_F / 4 / TEAM_THREADS; ++i) dst[i * TEAM_THREADS + tid] = src[i * TEAM_THREADS + tid];
                if (q == 0) {
                    cstQ[8] = C2g[(size_t)fi * 1536 + unit]; cstQ[9] = C2g[(size_t)fi * 1536 + 512 + unit];
                    cstQ[10] = C2g[(size_t)fi * 1536 + 1024 + unit];
                    cstQ[11] = C3g[(size_t)fi * 512 + unit];
                    cstQ[12] = C4g[(size_t)fi * 512 + unit];
                }
                cur_frame = fi;
                __syncthreads();
            }
            {
                // all reads first (12 + 2 ds_read_b128), then 40 FMAs
                const float4 *r0 = (const float4 *)(lds + L_REC + j0 * REC_F), *r1 = (const float4 *)(lds + L_REC + j1 * REC_F);
                const float4 *kp = (const float4 *)(lds + L_KT + ph * 8);
                const float4 a0 = r0[0], a1 = r0[1], a2 = r0[2], a3 = r0[3], a4 = r0[4], a5 = r0[5];
                const float4 b0 = r1[0], b1 = r1[1], b2 = r1[2], b3 = r1[3], b4 = r1[4], b5 = r1[5];
                const float4 k03 = kp[0];
                const float k4 = lds[L_KT + ph * 8 + 4];
                // record: {CA, VAr, VAz, VAn | CM0..3 | CM4, VM0r, VM0z, VM0n | VM1r, VM1z, VM1n, VM2r | VM2z, VM2n, VM3r, VM3z | VM3n, VM4r, VM4z, VM4n}
                cI0 = fmaf(k4, a2.x, fmaf(k03.w, a1.w, fmaf(k03.z, a1.z, fmaf(k03.y, a1.y, fmaf(k03.x, a1.x, a0.x)))));
                vr0 = fmaf(k4, a5.y, fmaf(k03.w, a4.z, fmaf(k03.z, a3.w, fmaf(k03.y, a3.x, fmaf(k03.x, a2.y, a0.y)))));
                vz0 = fmaf(k4, a5.z, fmaf(k03.w, a4.w, fmaf(k03.z, a4.x, fmaf(k03.y, a3.y, fmaf(k03.x, a2.z, a0.z)))));
                vn0 = fmaf(k4, a5.w, fmaf(k03.w, a5.x, fmaf(k03.z, a4.y, fmaf(k03.y, a3.z, fmaf(k03.x, a2.w, a0.w)))));
                cI1 = fmaf(k4, b2.x, fmaf(k03.w, b1.w, fmaf(k03.z, b1.z, fmaf(k03.y, b1.y, fmaf(k03.x, b1.x, b0.x)))));
                vr1 = fmaf(k4, b5.y, fmaf(k03.w, b4.z, fmaf(k03.z, b3.w, fmaf(k03.y, b3.x, fmaf(k03.x, b2.y, b0.y)))));
                vz1 = fmaf(k4, b5.z, fmaf(k03.w, b4.w, fmaf(k03.z, b4.x, fmaf(k03.y, b3.y, fmaf(k03.x, b2.z, b0.z)))));
                vn1 = fmaf(k4, b5.w, fmaf(k03.w, b5.x, fmaf(k03.z, b4.y, fmaf(k03.y, b3.z, fmaf(k03.x, b2.w, b0.w)))));
            }
        };
        // sampling noise of step `ts` for this quarter-wave's two classes: -log q  (Gumbel when q = -log u)
        auto prepare_noise = [&](int64_t ts) {
            if (MODE == WRNN_MODE_RAW && has_fc3) {
                if (a.noise_mode == WRNN_NOISE_INJECTED) {
                    const float *qp = a.noise1 + ((size_t)ts * a.n_rows + row) * NC + c3row0;
                    nz0 = -logf(qp[0]); nz1 = -logf(qp[1]);
                } else if (a.noise_mode == WRNN_NOISE_PHILOX) {
                    // one Philox block = classes (c3row0, c3row0+1) x steps (2s, 2s+1): evaluated on even steps
                    if ((ts & 1) == 0) {
                        const Philox4 pz = wrnn_raw_block(a.seed, (uint64_t)ts, (uint32_t)row, (uint32_t)c3row0);
                        nz0 = -__logf(-__logf(u01_from_bits(pz.x)));
                        nz1 = -__logf(-__logf(u01_from_bits(pz.y)));
                        nzn0 = -__logf(-__logf(u01_from_bits(pz.z)));
                        nzn1 = -__logf(-__logf(u01_from_bits(pz.w)));
                    } else { nz0 = nzn0; nz1 = nzn1; }
                } else { nz0 = 0.f; nz1 = 0.f; }
            }
            xforce = a.x_forced ? a.x_forced[(size_t)ts * a.n_rows + row] : 0.0f;
        };
        __syncthreads();
        prepare_step(0);
        prepare_noise(0);

        for (int64_t t = 0; t < a.steps; ++t) {
            ++epoch;
            const int par = (int)(epoch & 1u);
            if (PROF) prof_last = __builtin_readcyclecounter();

            // ---- phase A: I + GRU1 for units j0, j1, replicated in every WG (:208-212) ----
            float x2_0, x2_1;
            {
                const float xin0 = fmaf(wI0_0, xprev, cI0), xin1 = fmaf(wI0_1, xprev, cI1);
                const float rg0 = sigmoid_fast(fmaf(ur0, xprev, vr0) + gh1s[j0]);
                const float rg1 = sigmoid_fast(fmaf(ur1, xprev, vr1) + gh1s[j1]);
                const float zg0 = sigmoid_fast(fmaf(uz0, xprev, vz0) + gh1s[512 + j0]);
                const float zg1 = sigmoid_fast(fmaf(uz1, xprev, vz1) + gh1s[512 + j1]);
                const float ng0 = tanh_fast(fmaf(un0, xprev, vn0) + rg0 * gh1s[1024 + j0]);
                const float ng1 = tanh_fast(fmaf(un1, xprev, vn1) + rg1 * gh1s[1024 + j1]);
                h1_0 = (1.0f - zg0) * ng0 + zg0 * h1_0;
                h1_1 = (1.0f - zg1) * ng1 + zg1 * h1_1;
                x2_0 = xin0 + h1_0; x2_1 = xin1 + h1_1;
                xb[XB_H1 * 512 + pj0] = h1_0; xb[XB_H1 * 512 + pj1] = h1_1;
                xb[XB_X2 * 512 + pj0] = x2_0; xb[XB_X2 * 512 + pj1] = x2_1;
            }
            PROF_MARK(1);
            __syncthreads();  // B1
            PROF_MARK(2);

            // ---- phase B: GRU2 unit `unit` (:213-216); rows r,z,n of W_ih2[:, :512] . x2 ----
            {
                const X32 xc = load_chunk(xb + XB_X2 * 512, q);
                const float h2o = xb[XB_H2 * 512 + pu];
                const float x2u = xb[XB_X2 * 512 + pu];
                float gr, gz, gn;
                dot32x3(W_IH2, xc, gr, gz, gn);
                gr = row_sum(gr) + c2_r; gz = row_sum(gz) + c2_z; gn = row_sum(gn) + c2_n;
                const float rg = sigmoid_fast(gr + gh2_r);
                const float zg = sigmoid_fast(gz + gh2_z);
                const float ng = tanh_fast(gn + rg * gh2_n);
                const float h2n = (1.0f - zg) * ng + zg * h2o;
                const float x3u = x2u + h2n;
                if (q == 0) st_granule(mX3, par * 512 + unit, epoch, __float_as_uint(x3u));
            }
            PROF_MARK(3);
            u64 gx[2];
            // shadow work: gh1 for the next step = W_hh1 . h1' + b_hh1, published for everyone
            {
                const X32 hc = load_chunk(xb + XB_H1 * 512, q);
                f2 pr = mk2(0.f, 0.f), pz = mk2(0.f, 0.f), pn = mk2(0.f, 0.f);
                dot32x3_agpr<0, 3>(A_HH1, hc, pr, pz, pn);
                peek_n<2>(mX3, par * 512 + tid, 256, gx);     // first look at exchange 1, RTT hidden below
                dot32x3_agpr<3, 8>(A_HH1, hc, pr, pz, pn);
                const float sr = row_sum(pr.x + pr.y) + bhh1_r, sz = row_sum(pz.x + pz.y) + bhh1_z, sn = row_sum(pn.x + pn.y) + bhh1_n;
                if (q == 0) {
                    st_granule(mGH, par * 1536 + unit, epoch, __float_as_uint(sr));
                    st_granule(mGH, par * 1536 + 512 + unit, epoch, __float_as_uint(sz));
                    st_granule(mGH, par * 1536 + 1024 + unit, epoch, __float_as_uint(sn));
                }
            }
            PROF_MARK(4);
            // ---- exchange 1: x3 = x + h2 for all units; h2' = x3 - x2 ---------------------
            {
                u64 (&gq)[2] = gx;
                finish_n<2, 32>(mX3, par * 512 + tid, 256, epoch, gq, dead, a.err, 11u);
                const float x3_0 = __uint_as_float((unsigned)gq[0]), x3_1 = __uint_as_float((unsigned)gq[1]);
                xb[XB_X3 * 512 + pj0] = x3_0; xb[XB_X3 * 512 + pj1] = x3_1;
                xb[XB_H2 * 512 + pj0] = x3_0 - x2_0; xb[XB_H2 * 512 + pj1] = x3_1 - x2_1;
            }
            PROF_MARK(5);
            __syncthreads();  // B2
            PROF_MARK(6);

            // ---- phase C: fc1 row `unit` (:217-218) ---------------------------------------
            {
                const X32 xc = load_chunk(xb + XB_X3 * 512, q);
                const float s = row_sum(dot32_agpr(A_FC1, xc)) + c3v;
                if (q == 0) st_granule(mF1, par * 512 + unit, epoch, __float_as_uint(fmaxf(s, 0.0f)));
            }
            // shadow work: gh2 for the next step = W_hh2 . h2' + b_hh2 (stays in this quarter-wave)
            u64 gf[2];
            {
                const X32 hc = load_chunk(xb + XB_H2 * 512, q);
                f2 pr = mk2(0.f, 0.f), pz = mk2(0.f, 0.f), pn = mk2(0.f, 0.f);
                dot32x3_agpr<0, 3>(A_HH2, hc, pr, pz, pn);
                peek_n<2>(mF1, par * 512 + tid, 256, gf);     // first look at exchange 2
                dot32x3_agpr<3, 8>(A_HH2, hc, pr, pz, pn);
                gh2_r = row_sum(pr.x + pr.y) + bhh2_r; gh2_z = row_sum(pz.x + pz.y) + bhh2_z; gh2_n = row_sum(pn.x + pn.y) + bhh2_n;
            }
            PROF_MARK(7);
            // ---- exchange 2: fc1 outputs ------------------------------------------------------
            {
                u64 (&gq)[2] = gf;
                finish_n<2, 32>(mF1, par * 512 + tid, 256, epoch, gq, dead, a.err, 12u);
                xb[XB_F1 * 512 + pj0] = __uint_as_float((unsigned)gq[0]);
                xb[XB_F1 * 512 + pj1] = __uint_as_float((unsigned)gq[1]);
            }
            PROF_MARK(8);
            __syncthreads();  // B3
            PROF_MARK(9);

            // ---- phase D: fc2 row `unit` (:220-221) ---------------------------------------
            {
                const X32 xc = load_chunk(xb + XB_F1 * 512, q);
                const float s = row_sum(dot32_agpr(A_FC2, xc)) + c4v;
                if (q == 0) st_granule(mF2, par * 512 + unit, epoch, __float_as_uint(fmaxf(s, 0.0f)));
            }
            PROF_MARK(10);
            // space the first look ~250 cycles after the publish (the other workgroups' stores need that
            // long to reach L2; an immediate look would miss and cost a second round trip)
            __builtin_amdgcn_s_sleep(3);
            // ---- exchange 3: fc2 outputs ------------------------------------------------------
            {
                u64 gq[2];
                peek_n<2>(mF2, par * 512 + tid, 256, gq);
                finish_n<2, 32>(mF2, par * 512 + tid, 256, epoch, gq, dead, a.err, 13u);
                xb[XB_F2 * 512 + pj0] = __uint_as_float((unsigned)gq[0]);
                xb[XB_F2 * 512 + pj1] = __uint_as_float((unsigned)gq[1]);
            }
            PROF_MARK(11);
            __syncthreads();  // B4
            PROF_MARK(12);

            // ---- phase E: fc3 rows + race (:223, :231-235) ------------------------------------
            float lg0, lg1;
            {
                const X32 xc = load_chunk(xb + XB_F2 * 512, q);
                const float4 *wp = (const float4 *)(lds + L_FC3) + (size_t)(wave * 2) * 8 * 64 + lane;
                f2 pa = mk2(0.f, 0.f), pb = mk2(0.f, 0.f);
#pragma unroll
                for (int kk = 0; kk < 8; kk += 4) {
                    float4 wa[4], wb[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { wa[k] = wp[(kk + k) * 64]; wb[k] = wp[(8 + kk + k) * 64]; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f2 xl = mk2(xc.v[kk + k].x, xc.v[kk + k].y), xh = mk2(xc.v[kk + k].z, xc.v[kk + k].w);
                        pa = pkfma(mk2(wa[k].x, wa[k].y), xl, pa); pb = pkfma(mk2(wb[k].x, wb[k].y), xl, pb);
                        pa = pkfma(mk2(wa[k].z, wa[k].w), xh, pa); pb = pkfma(mk2(wb[k].z, wb[k].w), xh, pb);
                    }
                }
                lg0 = row_sum(pa.x + pa.y) + b3_0;
                lg1 = row_sum(pb.x + pb.y) + b3_1;
            }
            if (a.logits_out && q == 0 && has_fc3) {
                float *lo = a.logits_out + ((size_t)t * a.n_rows + row) * NC + c3row0;
                lo[0] = lg0;
                if (c3row0 + 1 < NC) lo[1] = lg1;
            }
            PROF_MARK(13);
            const float xforce_now = xforce;
            float x_new = 0.0f;
            if (MODE == WRNN_MODE_RAW) {
                // winner of this quarter-wave's 2 classes: argmax logit_k - log q_k
                const float v0 = lg0 + nz0, v1 = lg1 + nz1;
                const bool p1 = v1 > v0;
                if (q == 0) st_granule(mPR, par * 512 + 16 * g + 4 * wave + r4,
                                       (epoch << 10) | (unsigned)(p1 ? c3row0 + 1 : c3row0), __float_as_uint(p1 ? v1 : v0));
                // shadow work: everything of step t+1 that does not need x_t; the first look at exchange 4
                // (wave 0: race winners, waves 1-3: gh1 published back in phase B) rides under the Philox part
                if (t + 1 < a.steps) prepare_step(t + 1);
                u64 gq[8];
                const int ghbase = tid - 64;
                if (wave != 0) peek_n<8>(mGH, par * 1536 + ghbase, 192, gq);
                else peek_n<8>(mPR, par * 512 + lane * 8, 1, gq);
                if (t + 1 < a.steps) prepare_noise(t + 1);
                PROF_MARK(0);
                // ---- exchange 4 (wave 0) + gh1 collection (waves 1-3) ------------------------
                if (wave == 0) {
                    finish_n<8, 42>(mPR, par * 512 + lane * 8, 1, epoch & 0x3fffffu, gq, dead, a.err, 14u);
                    float best = -INFINITY; int besti = 0;
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const float vv = __uint_as_float((unsigned)gq[m]);
                        const int ii = (int)((gq[m] >> 32) & 1023u);
                        if (vv > best || (vv == best && ii < besti)) { best = vv; besti = ii; }
                    }
                    const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_max(best)), 63));
                    const u64 ball = __ballot(best == mx);
                    const int src = (int)__builtin_ctzll(ball ? ball : 1ull);
                    const int k = __builtin_amdgcn_readlane(besti, src);
                    if (lane == 0) misc_i[8] = k;
                } else {
                    // gh1 for the next step (published during phase B of this step): 192 threads x 8
                    const int base = ghbase;
                    finish_n<8, 32>(mGH, par * 1536 + base, 192, epoch, gq, dead, a.err, 15u);
#pragma unroll
                    for (int m = 0; m < 8; ++m) gh1s[base + m * 192] = __uint_as_float((unsigned)gq[m]);
                }
                PROF_MARK(14);
                __syncthreads();  // B5
                PROF_MARK(15);
                const int k = misc_i[8];
                x_new = lds[L_LUT + k];   // 2 * k / (n_classes - 1.) - 1.   (:235)
                if (g == 0 && tid == 0) {
                    if (a.labels_out) a.labels_out[(size_t)row * a.steps + t] = k;
                    a.samples_out[(size_t)row * a.steps + t] = x_new;
                }
            } else {
                // MOL (distribution.py:87-123): the 30 fc3 outputs are exchanged, every WG samples redundantly
                if (q == 0 && has_fc3) {
                    st_granule(mPR, par * 512 + c3row0, epoch, __float_as_uint(lg0));
                    if (c3row0 + 1 < NC) st_granule(mPR, par * 512 + c3row0 + 1, epoch, __float_as_uint(lg1));
                }
                const int64_t tnow = t;
                if (t + 1 < a.steps) { prepare_step(t + 1); prepare_noise(t + 1); }
                if (wave == 0) {
                    const int nr = NC / 3;
                    float mylg = 0.0f;
                    if (lane < NC) {
                        u64 gq[1];
                        peek_n<1>(mPR, par * 512 + lane, 1, gq);
                        unsigned spins = 0;
                        while (!dead && (unsigned)(gq[0] >> 32) != epoch) {
                            if (++spins > TEAM_SPIN_MAX) { dead = true; atomicExch(a.err, 16u); break; }
                            peek_n<1>(mPR, par * 512 + lane, 1, gq);
                        }
                        mylg = __uint_as_float((unsigned)gq[0]);
                    }
                    float v = -INFINITY;
                    if (lane < nr) {
                        float u1;
                        if (a.noise_mode == WRNN_NOISE_INJECTED) u1 = a.noise1[((size_t)tnow * a.n_rows + row) * nr + lane];
                        else u1 = 1e-5f + wrnn_uniform(a.seed, (uint64_t)tnow, (uint32_t)row, (uint32_t)lane) * (1.0f - 2e-5f);
                        v = mylg - logf(-logf(u1));
                    }
                    const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_max(v)), 63));
                    const u64 ball = __ballot(v == mx);
                    const int km = (int)__builtin_ctzll(ball ? ball : 1ull);
                    const float mean = __shfl(mylg, nr + km, 64);
                    const float ls = fmaxf(__shfl(mylg, 2 * nr + km, 64), -32.23619130191664f);
                    float u2;
                    if (a.noise_mode == WRNN_NOISE_INJECTED) u2 = a.noise2[(size_t)tnow * a.n_rows + row];
                    else u2 = 1e-5f + wrnn_uniform(a.seed, (uint64_t)tnow, (uint32_t)row, 10u) * (1.0f - 2e-5f);
                    float xs = mean + expf(ls) * (logf(u2) - logf(1.0f - u2));
                    xs = fminf(fmaxf(xs, -1.0f), 1.0f);
                    if (lane == 0) { misc_f[9] = xs; misc_i[8] = km; }
                } else {
                    u64 gq[8];
                    const int base = tid - 64;
                    peek_n<8>(mGH, par * 1536 + base, 192, gq);
                    finish_n<8, 32>(mGH, par * 1536 + base, 192, epoch, gq, dead, a.err, 15u);
#pragma unroll
                    for (int m = 0; m < 8; ++m) gh1s[base + m * 192] = __uint_as_float((unsigned)gq[m]);
                }
                __syncthreads();  // B5
                x_new = misc_f[9];
                if (g == 0 && tid == 0) {
                    if (a.labels_out) a.labels_out[(size_t)row * a.steps + t] = misc_i[8];
                    a.samples_out[(size_t)row * a.steps + t] = x_new;
                }
            }
            xprev = a.x_forced ? xforce_now : x_new;   // (:228, :237)
            PROF_MARK(16);
            if ((t & 63) == 63) {   // bounded-spin bail-out, checked workgroup-wide every 64 steps
                if (dead && lane == 0) misc_i[10] = 1;
                __syncthreads();
                if (misc_i[10]) return;
            }
        }
        __syncthreads();
    }
    if (PROF && a.prof && lane == 0 && (g == 0 || g == 31)) {
        for (int i = 0; i < 17; ++i) a.prof[((g ? 4 : 0) + wave) * 17 + i] = prof_acc[i];
    }
}

hipError_t wrnn_launch_loop_team(const WrnnTeamArgs &a, hipStream_t s) {
    (void)hipGetLastError();  // the runtime is shared with PyTorch: drop any stale sticky error of this thread
    static bool attr_set = false;
    const size_t lds = (size_t)L_TOTAL * sizeof(float);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)loop_team_kernel<WRNN_MODE_RAW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void *)loop_team_kernel<WRNN_MODE_MOL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (a.prof && a.d.mode == WRNN_MODE_RAW) {
        hipError_t e = hipFuncSetAttribute((const void *)loop_team_kernel<WRNN_MODE_RAW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((loop_team_kernel<WRNN_MODE_RAW, true>), dim3(a.n_teams * 32), dim3(TEAM_THREADS), lds, s, a);
    } else if (a.d.mode == WRNN_MODE_RAW)
        hipLaunchKernelGGL((loop_team_kernel<WRNN_MODE_RAW, false>), dim3(a.n_teams * 32), dim3(TEAM_THREADS), lds, s, a);
    else
        hipLaunchKernelGGL((loop_team_kernel<WRNN_MODE_MOL, false>), dim3(a.n_teams * 32), dim3(TEAM_THREADS), lds, s, a);
    return hipGetLastError();
}
