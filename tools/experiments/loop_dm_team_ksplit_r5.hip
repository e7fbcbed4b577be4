// ARCHIVED EXPERIMENT -- not built, not part of the product: the K-split O2 / O4 variant of the dual-softmax team kernel (DMT_KSPLIT; needs the matching image fill in loop_deepmind.hip and WRNN_DM_MAIL_GRANULES = 40 960): parity-green, 11.2 us per sample against 4.75.
// Measurements: profiles/r05_batch_cs_experiments.txt; why it is kept: DESIGN.md 3.3c / 3.4.  To build it, copy it over the csrc/ file of the same base name.
//
// Team kernel for the dual-softmax (coarse/fine) WaveRNN of wavernn/models/deepmind_version.py, generate() :75-165
// (SURVEY.md section 8a row A12).  Same machinery as loop_team2.hip: one team = the 32 workgroups of one XCD, fp32
// weights resident on chip, 8-byte {tag,value} granules exchanged through the XCD's L2, double-buffered by sample
// parity.  One sample = 6 exchanges (DMT_KSPLIT 0) or 4 (DMT_KSPLIT 1, round 5: O2 / O4 split over K -- every workgroup turns its own U relu(O1) values into partial logits of
// ALL classes, the partials are all-gathered ([32 producers][class]: 32 eight-byte loads per thread, every wave-load 512 contiguous bytes) and summed in producer order by everybody; bias + noise ride on the partial of the
// workgroup that owns the class; the t1 gathers disappear):
//
//   R.h rows of the own hidden units (registers) -> coarse gates -> [h_c] -> O1 rows -> [t1] -> O2 rows + noise ->
//   [256 class values] -> argmax = coarse -> fine gates (need coarse) -> [h_f] -> O3 rows -> [t1] -> O4 rows + noise
//   -> [256 class values] -> argmax = fine
//
// Workgroup g owns hidden units {gU..gU+U-1} of each half (U = S/32; 14 for hidden_size 896), rows gU.. of O1/O3 and
// classes g*Q/32.. of O2/O4.  A quarter-wave (16 lanes) owns one hidden unit: its three R rows (u, r, e; H columns,
// H/16 per lane) live in VGPRs; the O slices live in LDS.  R(hidden) is evaluated once per sample with the hidden
// state of the previous sample, for both halves (:116-119), so the fine rows wait in registers until coarse is known.
// Sampling: Categorical(softmax(l)).sample() == argmax_k l_k - log q_k, q ~ Exp(1)  (as on the main path).
#include "device_util.h"
#include "dm_internal.h"
#include "wrnn_internal.h"

#define DMT_THREADS 512
#define DMT_SPIN_MAX 300000u

typedef unsigned long long u64;

namespace {

__device__ __forceinline__ unsigned dmt_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ void dmt_st(u64 *base, unsigned idx, unsigned tag, float payload) {
    const u64 v = ((u64)tag << 32) | __float_as_uint(payload);
    const unsigned off = idx * 8u;
    asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ u64 dmt_peek(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one granule per lane; wave-uniform completion; two staggered first looks (see loop_team2.hip)
__device__ __forceinline__ float dmt_take(const u64 *base, unsigned idx, unsigned tag, bool &dead, unsigned *err, unsigned code) {
    u64 ga = dmt_peek(base + idx);
    __builtin_amdgcn_s_sleep(3);
    u64 gb = dmt_peek(base + idx);
    if (__all((unsigned)(ga >> 32) == tag)) return __uint_as_float((unsigned)ga);
    unsigned spins = 0;
    while (!dead && !__all((unsigned)(gb >> 32) == tag)) {
        if (++spins > DMT_SPIN_MAX) { dead = true; if ((threadIdx.x & 63) == 0) atomicExch(err, code); break; }
        gb = dmt_peek(base + idx);
    }
    return __uint_as_float((unsigned)gb);
}
template <int CTRL>
__device__ __forceinline__ float dmt_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float dmt_row_sum(float v) {   // sum over the 16 lanes of a DPP row, in every lane
    v += dmt_dpp<0xB1>(v);
    v += dmt_dpp<0x4E>(v);
    v += dmt_dpp<0x141>(v);
    v += dmt_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float dmt_wave_max(float v) {   // max over 64 lanes, valid in lane 63
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

__device__ __forceinline__ float dmt_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float dmt_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

constexpr unsigned G_HC = 0, G_T1C = 1024, G_HF = 2048, G_T1F = 3072, G_C = 4096, G_F = 4608;   // x2 parities each
constexpr unsigned G_PC = 8192, G_PF = 8192 + 2 * 8192;   // DMT_KSPLIT: partial logits [parity][class][32 producers], 8192 granules per parity (Q <= 256)
typedef unsigned u2v_ __attribute__((ext_vector_type(2)));

// A vector of N = 64*P floats in LDS, chunked for 16 lanes of P float4 each: element j (chunk q = j / (4P), float4 k
// inside the chunk, component e) -> plane k: [k][16 lanes][4].  Lane q reads its P float4 at stride 64 floats
// (conflict-free ds_read_b128); element j is written with one ds_write_b32.
template <int P>
__device__ __forceinline__ int chunk_idx(int j) { return ((j % (4 * P)) >> 2) * 64 + (j / (4 * P)) * 4 + (j & 3); }

// CPL = float4 per lane of an H-vector (H = 64 CPL); an S-vector (S = H/2) has CPL/2 per lane
template <int CPL>
__global__ void __launch_bounds__(DMT_THREADS, 2) dm_team_kernel(WrnnDmTeamArgs ta) {
    constexpr int H = 64 * CPL, S = H / 2, U = S / 32, PS = CPL / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = (float *)smem;
    const WrnnDmArgs &a = ta.base;
    const int Q = a.Q, QW = Q / 32;
    // LDS carve-up (floats)
    float *hR = lds;                  // [2 parities][H]  hidden state in H-chunk order (input of R)
    float *hcS = hR + 2 * H;          // [S] new coarse half in S-chunk order (input of O1)
    float *hfS = hcS + S;             // [S] new fine half (input of O3)
    float *t1 = hfS + S;              // [S] relu(O1 / O3 output) (input of O2 / O4)
    float *misc = t1 + S;             // [128]: 0-2 team/rank/bail-out, 16-23 race partials, 32-95 sampling noise [parity][coarse 16 | fine 16]
    int *misc_i = (int *)misc;
    float *imgO1 = misc + 128, *imgO3 = imgO1 + U * S, *imgO2 = imgO3 + U * S, *imgO4 = imgO2 + QW * S;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qw = tid >> 4, q = tid & 15;

    // ---- team formation (as loop_team2.hip): the first XCD to arrive is the team ----
    if (tid == 0) {
        const unsigned x = dmt_xcc_id();
        const unsigned rank = atomicAdd(&ta.ctl[x], 1u);
        unsigned slot1 = 0;
        if (rank == 0) {
            slot1 = atomicAdd(&ta.ctl[8], 1u) + 1u;
            __hip_atomic_store(&ta.ctl[16 + x], slot1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (unsigned spins = 0; spins < 4000000u; ++spins) {
                slot1 = __hip_atomic_load(&ta.ctl[16 + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (slot1) break;
            }
        }
        // co-residency checked, not assumed (see loop_team2.hip): the 32 workgroups of the team's XCD must all have arrived
        if (slot1 == 1u && rank < 32u) {
            unsigned arrived = 0;
            for (unsigned spins = 0; spins < WRNN_ARRIVE_POLLS; ++spins) {
                arrived = __hip_atomic_load(&ta.ctl[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (arrived >= 32u) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (arrived < 32u) { atomicCAS(ta.err, 0u, WRNN_DEVERR_BUSY); slot1 = 0; }
        }
        misc_i[0] = slot1 ? (int)slot1 - 1 : 1 << 20;
        misc_i[1] = (int)rank;
        misc_i[2] = 0;   // bail-out flag
    }
    __syncthreads();
    const int team = __builtin_amdgcn_readfirstlane(misc_i[0]);
    const int g = __builtin_amdgcn_readfirstlane(misc_i[1]);
    __syncthreads();
    if (team != 0 || g >= 32) return;
    u64 *mail = ta.mail;
    const float *w = a.w;

    // ---- roles ----
    const bool isCq = qw < U, isFq = qw >= U && qw < 2 * U;
    const int hi = isCq ? g * U + qw : S + g * U + (qw - U);   // hidden index of this quarter-wave (valid if qw < 2U)
    const int orow = g * U + qw;                                // O1 / O3 row (valid if qw < U)
    const int cls = g * QW + qw;                                // O2 / O4 class (valid if qw < QW)

    // ---- resident weights ----
    float wR[3 * CPL * 4];
    {
        const float *src = ta.team_w + (size_t)g * (3 * CPL * 4) * DMT_THREADS + tid;
#pragma unroll
        for (int i = 0; i < 3 * CPL * 4; ++i) wR[i] = src[(size_t)i * DMT_THREADS];
        const int nimg = 2 * U * S + 2 * QW * S;
        const float4 *img = (const float4 *)(ta.team_lds + (size_t)g * nimg);
        float4 *dst = (float4 *)imgO1;
        for (int i = tid; i < nimg / 4; i += DMT_THREADS) dst[i] = img[i];
        for (int i = tid; i < 2 * H; i += DMT_THREADS) hR[i] = 0.0f;   // get_initial_hidden :168-170
    }
    float bu = 0.f, br = 0.f, be = 0.f, iw[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (qw < 2 * U) {
        bu = w[a.obu + hi]; br = w[a.obr + hi]; be = w[a.obe + hi];
        if (isCq) {
            const float *Ic = w + a.oIc;
            const int j = hi;
            iw[0] = Ic[j * 2]; iw[1] = Ic[j * 2 + 1]; iw[3] = Ic[(S + j) * 2]; iw[4] = Ic[(S + j) * 2 + 1];
            iw[6] = Ic[(2 * S + j) * 2]; iw[7] = Ic[(2 * S + j) * 2 + 1];
        } else {
            const float *If = w + a.oIf;
            const int j = hi - S;
#pragma unroll
            for (int c = 0; c < 3; ++c) { iw[c] = If[j * 3 + c]; iw[3 + c] = If[(S + j) * 3 + c]; iw[6 + c] = If[(2 * S + j) * 3 + c]; }
        }
    }
    const float bO1 = isCq ? w[a.oO1b + orow] : 0.f, bO3 = isCq ? w[a.oO3b + orow] : 0.f;
    const float bO2 = qw < QW ? w[a.oO2b + cls] : 0.f, bO4 = qw < QW ? w[a.oO4b + cls] : 0.f;
    (void)bO2; (void)bO4; (void)orow;   // DMT_KSPLIT: the class biases ride on the owner's partial logits instead
    __syncthreads();

    // -log q for the workgroup's QW classes of both softmaxes of sample ts, drawn one sample ahead by the otherwise
    // idle last wave (lanes 0..QW-1 coarse, 16..16+QW-1 fine) -> misc[32 + 32 * parity(ts) + lane]
    auto draw = [&](long ts) {
        if (wave != 7 || ts >= a.seq_len) return;
        const unsigned which = (unsigned)(lane >> 4) & 1u;
        const int c = lane & 15;
        if (lane >= 32 || c >= QW) return;
        const int k = g * QW + c;
        float nz = 0.f;
        if (a.noise_mode == WRNN_NOISE_INJECTED) nz = -logf(a.noise[((size_t)ts * 2 + which) * Q + k]);
        else if (a.noise_mode == WRNN_NOISE_PHILOX) nz = -logf(-logf(wrnn_uniform(a.seed, (uint64_t)ts, which, (uint32_t)k)));
        misc[32 + 32 * (int)((ts + 1) & 1) + lane] = nz;
    };
    // dot of an LDS-resident row (image [P planes][16 lanes] float4 per quarter-wave row) with an S-vector
    auto dotS = [&](const float *img, int row_in_wg, const float *vec) -> float {
        const float4 *wp = (const float4 *)img + (size_t)row_in_wg * PS * 16 + q;
        const float4 *xp = (const float4 *)vec + q;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < PS; ++k) {
            const float4 ww = wp[k * 16], xx = xp[k * 16];
            s0 = fmaf(ww.x, xx.x, s0); s1 = fmaf(ww.y, xx.y, s1);
            s0 = fmaf(ww.z, xx.z, s0); s1 = fmaf(ww.w, xx.w, s1);
        }
        return dmt_row_sum(s0 + s1);
    };
    // argmax over the Q published class values (one granule per thread of the first Q/64 waves), ties -> lowest index
    // DMT_KSPLIT: the summed score of class (tid - 256) is handed in by the threads of waves 4 .. 4 + Q/64 - 1
    auto race_k = [&](float v) -> int {
        if (wave >= 4 && wave < 4 + Q / 64) {
            const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dmt_wave_max(v)), 63));
            const u64 ball = __ballot(v == mx);
            const int src = (int)__builtin_ctzll(ball ? ball : 1ull);
            if (lane == 0) { misc[16 + 2 * (wave - 4)] = mx; misc_i[17 + 2 * (wave - 4)] = (wave - 4) * 64 + src; }
        }
        __syncthreads();
        float bv = misc[16];
        int bi = misc_i[17];
        for (int i = 1; i < Q / 64; ++i) {
            const float v2 = misc[16 + 2 * i];
            if (v2 > bv) { bv = v2; bi = misc_i[17 + 2 * i]; }
        }
        return bi;
    };
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void *)mail, 0, (int)(WRNN_DM_MAIL_GRANULES * 8u), 0x00020000);
    // DMT_KSPLIT: one softmax from the workgroup's own t1 values (misc[96 + u], written by the C quarter-waves in front of a barrier): partials of all Q classes -> mailbox,
    // all-gather + sum in producer order -> the class's score in the threads of waves 4 .. (class = tid - 256); 0 elsewhere
    auto ksoftmax = [&](const float *imgk, size_t obias, int noise_off, unsigned region, unsigned par, unsigned epoch, bool &dead) -> float {
        float tot = 0.0f;
        if (tid >= 256 && tid < 256 + Q) {
            const int c = tid - 256;
            float p = 0.0f;
#pragma unroll
            for (int u = 0; u < U; ++u) p = fmaf(imgk[u * Q + c], misc[96 + u], p);
            if (c / QW == g) p += w[obias + c] + misc[32 + 32 * par + noise_off + (c - g * QW)];   // the class's bias and -log q ride on its owner's partial
            // mailbox order [producer][class]: a producer's Q partials are 2 KB of contiguous stores, a consumer wave's load of one producer row is 512 contiguous bytes
            // (first version: [class][producer] with 16-byte loads -- 32 producers in every cache line: 17 us per sample instead of 4.75)
            dmt_st(mail, region + par * 8192 + (unsigned)g * 256u + (unsigned)c, epoch, p);
            const unsigned base = (region + par * 8192 + (unsigned)c) * 8u;
            unsigned spins = 0;
            for (;;) {   // sentinel: every lane watches another producer's partial of its class
                const u2v_ sv = __builtin_amdgcn_raw_buffer_load_b64(mrs, base + (unsigned)(lane & 31) * 2048u, 0, 16);
                if (__all(sv.y == epoch) || dead) break;
                if (++spins > DMT_SPIN_MAX) { dead = true; if (lane == 0) atomicExch(ta.err, 22u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            u2v_ gg[32];
            for (;;) {
#pragma unroll
                for (int i = 0; i < 32; ++i) gg[i] = __builtin_amdgcn_raw_buffer_load_b64(mrs, base + (unsigned)i * 2048u, 0, 16);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 32; ++i) ok = ok && gg[i].y == epoch;
                if (__all(ok) || dead) break;
                if (++spins > DMT_SPIN_MAX) { dead = true; if (lane == 0) atomicExch(ta.err, 23u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) tot += __uint_as_float(gg[i].x);   // producer order 0 .. 31
        }
        return tot;
    };
    auto race = [&](unsigned region, unsigned par, unsigned epoch, bool &dead) -> int {
        if (wave < Q / 64) {
            const float v = dmt_take(mail, region + par * 256 + tid, epoch, dead, ta.err, 21u);
            const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dmt_wave_max(v)), 63));
            const u64 ball = __ballot(v == mx);
            const int src = (int)__builtin_ctzll(ball ? ball : 1ull);
            if (lane == 0) { misc[16 + 2 * wave] = mx; misc_i[17 + 2 * wave] = wave * 64 + src; }
        }
        __syncthreads();
        float bv = misc[16];
        int bi = misc_i[17];
        for (int i = 1; i < Q / 64; ++i) {
            const float v = misc[16 + 2 * i];
            if (v > bv) { bv = v; bi = misc_i[17 + 2 * i]; }
        }
        return bi;
    };

    // R(hidden) rows u, r, e of the own hidden unit (:116-119): evaluated as soon as the hidden state of a sample is
    // complete (after the fine half was exchanged), i.e. under the O3 / O4 phases, for the NEXT sample
    float ru = 0.f, rr = 0.f, re = 0.f;
    auto r_rows = [&](const float *hin) {
        const float4 *xp = (const float4 *)hin + q;
        float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f, c0 = 0.f, c1 = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const float4 x = xp[k * 16];
            a0 = fmaf(wR[4 * k + 0], x.x, a0); a1 = fmaf(wR[4 * k + 1], x.y, a1);
            b0 = fmaf(wR[CPL * 4 + 4 * k + 0], x.x, b0); b1 = fmaf(wR[CPL * 4 + 4 * k + 1], x.y, b1);
            c0 = fmaf(wR[CPL * 8 + 4 * k + 0], x.x, c0); c1 = fmaf(wR[CPL * 8 + 4 * k + 1], x.y, c1);
            a0 = fmaf(wR[4 * k + 2], x.z, a0); a1 = fmaf(wR[4 * k + 3], x.w, a1);
            b0 = fmaf(wR[CPL * 4 + 4 * k + 2], x.z, b0); b1 = fmaf(wR[CPL * 4 + 4 * k + 3], x.w, b1);
            c0 = fmaf(wR[CPL * 8 + 4 * k + 2], x.z, c0); c1 = fmaf(wR[CPL * 8 + 4 * k + 3], x.w, c1);
        }
        ru = dmt_row_sum(a0 + a1); rr = dmt_row_sum(b0 + b1); re = dmt_row_sum(c0 + c1);
    };

    bool dead = false;
    int oc = 0, of = 0;                      // out_coarse = out_fine = 0 :90-91
    float hown = 0.0f;                       // this quarter-wave's hidden unit
    draw(0);
    __syncthreads();
    for (long t = 0; t < a.seq_len; ++t) {
        const unsigned epoch = (unsigned)t + 1u, par = epoch & 1u;
        float *hout = hR + par * H;
        const float pc = (float)oc / 127.5f - 1.0f, pf = (float)of / 127.5f - 1.0f;   // :106-107
        draw(t + 1);
        if (DMT_KSPLIT && isFq && t > 0) r_rows(hR + (par ^ 1u) * H);   // R.h rows of the fine units from the previous sample's hidden state, while the coarse half is evaluated and exchanged

        // ---- coarse gates :111-125 ----
        if (isCq) {
            const float Iu = iw[0] * pc + iw[1] * pf, Ir = iw[3] * pc + iw[4] * pf, Ie = iw[6] * pc + iw[7] * pf;
            const float u = dmt_sigmoid(ru + Iu + bu);
            const float r = dmt_sigmoid(rr + Ir + br);
            const float e = dmt_tanh(r * re + Ie + be);
            hown = u * hown + (1.0f - u) * e;
            if (q == 0) dmt_st(mail, G_HC + par * 512 + hi, epoch, hown);
        }
        // ---- exchange 1: new coarse half ----
        if (wave < S / 64) {
            const float v = dmt_take(mail, G_HC + par * 512 + tid, epoch, dead, ta.err, 11u);
            hout[chunk_idx<CPL>(tid)] = v;
            hcS[chunk_idx<PS>(tid)] = v;
        }
        __syncthreads();
        // ---- out_coarse = O2(relu(O1(hidden_coarse))) :128 ----
#if DMT_KSPLIT
        if (isCq) {
            const float s = dotS(imgO1, qw, hcS) + bO1;
            if (q == 0) misc[96 + qw] = fmaxf(s, 0.0f);
        }
        __syncthreads();
        oc = race_k(ksoftmax(imgO2, a.oO2b, 0, G_PC, par, epoch, dead));   // Categorical(...).sample() :130-131
#else
        if (isCq) {
            const float s = dotS(imgO1, qw, hcS) + bO1;
            if (q == 0) dmt_st(mail, G_T1C + par * 512 + orow, epoch, fmaxf(s, 0.0f));
        }
        if (wave < S / 64) t1[chunk_idx<PS>(tid)] = dmt_take(mail, G_T1C + par * 512 + tid, epoch, dead, ta.err, 12u);
        __syncthreads();
        if (qw < QW) {
            const float s = dotS(imgO2, qw, t1) + bO2 + misc[32 + 32 * par + qw];
            if (q == 0) dmt_st(mail, G_C + par * 256 + cls, epoch, s);
        }
        oc = race(G_C, par, epoch, dead);                                   // Categorical(...).sample() :130-131
#endif
        (void)race; (void)race_k; (void)ksoftmax;
        if (g == 0 && tid == 0) a.coarse[t] = oc;
        const float cp = (float)oc / 127.5f - 1.0f;                         // :135
        // ---- fine gates :136-145 ----
        if (isFq) {
            const float Iu = iw[0] * pc + iw[1] * pf + iw[2] * cp;
            const float Ir = iw[3] * pc + iw[4] * pf + iw[5] * cp;
            const float Ie = iw[6] * pc + iw[7] * pf + iw[8] * cp;
            const float u = dmt_sigmoid(ru + Iu + bu);
            const float r = dmt_sigmoid(rr + Ir + br);
            const float e = dmt_tanh(r * re + Ie + be);
            hown = u * hown + (1.0f - u) * e;
            if (q == 0) dmt_st(mail, G_HF + par * 512 + (hi - S), epoch, hown);
        }
        if (wave < S / 64) {
            const float v = dmt_take(mail, G_HF + par * 512 + tid, epoch, dead, ta.err, 13u);
            hout[chunk_idx<CPL>(S + tid)] = v;
            hfS[chunk_idx<PS>(tid)] = v;
        }
        __syncthreads();
#if DMT_KSPLIT
        // ---- out_fine = O4(relu(O3(hidden_fine))) :148 ----
        if (isCq) {
            const float s = dotS(imgO3, qw, hfS) + bO3;
            if (q == 0) misc[96 + qw] = fmaxf(s, 0.0f);
        }
        __syncthreads();
        if (isCq) r_rows(hout);   // the C quarter-waves (waves 0-3) idle from here on: R.h of the next sample under the partials exchange of waves 4-7;
                                  // the F quarter-waves sit in those waves: their R.h rows wait for the start of the next sample (under its coarse gates + h_c exchange)
        of = race_k(ksoftmax(imgO4, a.oO4b, 16, G_PF, par, epoch, dead));   // :150-151
#else
        if (isFq) r_rows(hout);   // idle from here on: R.h of the next sample now
        // ---- out_fine = O4(relu(O3(hidden_fine))) :148 ----
        if (isCq) {
            const float s = dotS(imgO3, qw, hfS) + bO3;
            if (q == 0) dmt_st(mail, G_T1F + par * 512 + orow, epoch, fmaxf(s, 0.0f));
            r_rows(hout);         // under the t1 exchange
        }
        if (wave < S / 64) t1[chunk_idx<PS>(tid)] = dmt_take(mail, G_T1F + par * 512 + tid, epoch, dead, ta.err, 14u);
        __syncthreads();
        if (qw < QW) {
            const float s = dotS(imgO4, qw, t1) + bO4 + misc[32 + 32 * par + 16 + qw];
            if (q == 0) dmt_st(mail, G_F + par * 256 + cls, epoch, s);
        }
        of = race(G_F, par, epoch, dead);                                   // :150-151
#endif
        if (g == 0 && tid == 0) a.fine[t] = of;
        if ((t & 63) == 63) {   // bounded-spin bail-out, checked workgroup-wide every 64 samples
            if (dead && lane == 0) misc_i[2] = 1;
            __syncthreads();
            if (misc_i[2]) return;
        }
    }
}

template <int CPL>
hipError_t launch_cpl(const WrnnDmTeamArgs &a, hipStream_t s) {
    const int H = 64 * CPL, S = H / 2, U = S / 32, QW = a.base.Q / 32;
    const size_t lds = (size_t)(2 * H + 3 * S + 128 + 2 * U * S + 2 * QW * S) * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void *)dm_team_kernel<CPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((dm_team_kernel<CPL>), dim3(256), dim3(DMT_THREADS), lds, s, a);
    return hipGetLastError();
}

}  // namespace

// hidden_size in {512, 640, 768, 896}: H/128 integral (float4 chunks of both vector kinds), 3*H/16 <= 168 weight
// registers per lane; quantisation a multiple of 64 up to 256 (one class value per thread of the first Q/64 waves,
// Q/32 class rows per workgroup)
bool wrnn_dm_team_supported(int H, int Q) {
    return (H == 512 || H == 640 || H == 768 || H == 896) && Q >= 64 && Q <= 256 && Q % 64 == 0;
}

hipError_t wrnn_launch_dm_team(const WrnnDmTeamArgs &a, hipStream_t s) {
    (void)hipGetLastError();
    switch (a.base.H) {
        case 512: return launch_cpl<8>(a, s);
        case 640: return launch_cpl<10>(a, s);
        case 768: return launch_cpl<12>(a, s);
        case 896: return launch_cpl<14>(a, s);
        default: return hipErrorInvalidValue;
    }
}
