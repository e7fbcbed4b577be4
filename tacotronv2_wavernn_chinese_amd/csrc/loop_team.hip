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

// Poll N granules p[0], p[stride], ... until all carry `tag`; all loads of one
// round are in flight together.  (g >> SHIFT) is compared with tag.
template <int N, int SHIFT>
__device__ __forceinline__ void poll_n(const u64 *base, unsigned idx, unsigned stride, unsigned tag, u64 (&g)[N], bool &dead,
                                       unsigned *err, unsigned code) {
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] = 0;
    if (dead) return;
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const unsigned off = (idx + (unsigned)i * stride) * 8u;
            asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=&v"(g[i]) : "v"(off), "s"(base) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            asm volatile("" : "+v"(g[i]));
            ok = ok && ((unsigned)(g[i] >> SHIFT) == tag);
        }
        if (ok) return;
        if (++spins > TEAM_SPIN_MAX) { dead = true; atomicExch(err, code); return; }
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

// 32-term dot product: register weights w[0..31] x the lane's chunk of an LDS vector
__device__ __forceinline__ float dot32(const float *w, const float *vec, int q) {
    const float4 *p = (const float4 *)vec + q;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float4 x = p[k * 16];
        s0 = fmaf(w[4 * k + 0], x.x, s0); s1 = fmaf(w[4 * k + 1], x.y, s1);
        s2 = fmaf(w[4 * k + 2], x.z, s2); s3 = fmaf(w[4 * k + 3], x.w, s3);
    }
    return (s0 + s1) + (s2 + s3);
}
// three rows sharing one chunk read (GRU gates r, z, n)
__device__ __forceinline__ void dot32x3(const float *w, const float *vec, int q, float &o0, float &o1, float &o2) {
    const float4 *p = (const float4 *)vec + q;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f, c0 = 0.f, c1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float4 x = p[k * 16];
        a0 = fmaf(w[4 * k + 0], x.x, a0); a1 = fmaf(w[4 * k + 1], x.y, a1);
        a0 = fmaf(w[4 * k + 2], x.z, a0); a1 = fmaf(w[4 * k + 3], x.w, a1);
        b0 = fmaf(w[32 + 4 * k + 0], x.x, b0); b1 = fmaf(w[32 + 4 * k + 1], x.y, b1);
        b0 = fmaf(w[32 + 4 * k + 2], x.z, b0); b1 = fmaf(w[32 + 4 * k + 3], x.w, b1);
        c0 = fmaf(w[64 + 4 * k + 0], x.x, c0); c1 = fmaf(w[64 + 4 * k + 1], x.y, c1);
        c0 = fmaf(w[64 + 4 * k + 2], x.z, c0); c1 = fmaf(w[64 + 4 * k + 3], x.w, c1);
    }
    o0 = a0 + a1; o1 = b0 + b1; o2 = c0 + c1;
}

// LDS carve-up (floats)
constexpr int L_FC3 = 0;                    // [4 waves][2 rows][8 planes][64 lanes][4]  = 16384
constexpr int L_XB = L_FC3 + 16384;         // 6 vectors x 512 (plane order)
constexpr int XB_H1 = 0, XB_X2 = 1, XB_X3 = 2, XB_H2 = 3, XB_F1 = 4, XB_F2 = 5;
constexpr int L_GH1 = L_XB + 6 * 512;       // [3][512]
constexpr int L_CM = L_GH1 + 1536;          // [ND<=5][512]
constexpr int L_CA = L_CM + 5 * 512;        // [512]
constexpr int L_VM = L_CA + 512;            // [5][1536]
constexpr int L_VA = L_VM + 5 * 1536;       // [1536]
constexpr int L_MISC = L_VA + 1536;         // 64 floats of scratch
constexpr int L_TOTAL = L_MISC + 64;

}  // namespace

template <int MODE>
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
    const int NC = d.NC, HOP = d.HOP, ND = d.ND, T = a.T;

    // ---- team formation: by the XCD this workgroup actually runs on ------------
    if (tid == 0) {
        const unsigned x = xcc_id();
        misc_i[0] = (int)x;
        misc_i[1] = (int)atomicAdd(&a.ctl[x], 1u);
    }
    __syncthreads();
    const int team = __builtin_amdgcn_readfirstlane(misc_i[0]);
    const int g = __builtin_amdgcn_readfirstlane(misc_i[1]);
    __syncthreads();
    if (g >= TEAM_WGS || team >= a.n_teams || team >= a.n_rows) return;

    u64 *mail = a.mail + (size_t)team * WRNN_TEAM_MAIL_GRANULES;
    u64 *mX3 = mail, *mF1 = mail + 2 * 512, *mF2 = mail + 4 * 512, *mPR = mail + 6 * 512, *mGH = mail + 8 * 512;

    // ---- resident weights ---------------------------------------------------------
    float wr[WRNN_TEAM_NWREG];
    {
        const float *src = a.team_w + (size_t)g * WRNN_TEAM_NWREG * TEAM_THREADS + tid;
#pragma unroll
        for (int i = 0; i < WRNN_TEAM_NWREG; ++i) wr[i] = src[(size_t)i * TEAM_THREADS];
        const float4 *f3 = (const float4 *)(a.team_fc3 + (size_t)g * 16384);
        float4 *dst = (float4 *)(lds + L_FC3);
        for (int i = tid; i < 4096; i += TEAM_THREADS) dst[i] = f3[i];
    }
    const float *W_HH1 = wr, *W_IH2 = wr + 96, *W_HH2 = wr + 192, *W_FC2 = wr + 288, *W_FC1 = wr + 320;
    const int unit = 16 * g + 4 * wave + r4;          // hidden unit / fc1 / fc2 row of this quarter-wave
    const int c3row0 = 32 * g + 8 * wave + 2 * r4;    // first of the two fc3 rows of this quarter-wave
    const bool has_fc3 = c3row0 < NC;
    // phase-A constants of units j0 = tid, j1 = tid + 256
    const int j0 = tid, j1 = tid + 256;
    const float wI0_0 = a.wI0[j0], wI0_1 = a.wI0[j1];
    const float ur0 = a.u1[j0], uz0 = a.u1[512 + j0], un0 = a.u1[1024 + j0];
    const float ur1 = a.u1[j1], uz1 = a.u1[512 + j1], un1 = a.u1[1024 + j1];
    // per-quarter-wave biases
    const float bhh1_r = a.w[a.off.r1_bhh + unit], bhh1_z = a.w[a.off.r1_bhh + 512 + unit], bhh1_n = a.w[a.off.r1_bhh + 1024 + unit];
    const float bhh2_r = a.w[a.off.r2_bhh + unit], bhh2_z = a.w[a.off.r2_bhh + 512 + unit], bhh2_n = a.w[a.off.r2_bhh + 1024 + unit];
    const float b3_0 = has_fc3 ? a.w[a.off.fc3_b + c3row0] : 0.0f;
    const float b3_1 = (c3row0 + 1 < NC) ? a.w[a.off.fc3_b + c3row0 + 1] : 0.0f;
    const float *ktab = a.w + a.off.ktab;
    const int pj0 = perm(j0), pj1 = perm(j1), pu = perm(unit);

    bool dead = false;
    unsigned epoch = 0;

    for (int row = team; row < a.n_rows; row += a.n_teams) {
        const WrnnRow rw = a.rows[row];
        const float *CMg = a.tabCM + (size_t)rw.utt * (T + 2 * d.P) * 512;
        const float *CAg = a.tabCA + (size_t)rw.utt * (T + 1) * 512;
        const float *VMg = a.tabVM + (size_t)rw.utt * (T + 2 * d.P) * 1536;
        const float *VAg = a.tabVA + (size_t)rw.utt * (T + 1) * 1536;
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
        float c2_r = 0.f, c2_z = 0.f, c2_n = 0.f, c3v = 0.f, c4v = 0.f;
        int cur_frame = -1000000;
        __syncthreads();

        for (int64_t t = 0; t < a.steps; ++t) {
            ++epoch;
            const int par = (int)(epoch & 1u);
            // ---- conditioning for this step (independent of x_{t-1}) ---------------
            const int64_t pos = rw.start + t;
            const bool live = pos < a.total_len;       // fold padding 'after' = zero rows (:327-330)
            const int fi = live ? (int)(pos / HOP) : T; // frame index; T = the all-zero conditioning entry
            const int ph = live ? (int)(pos - (int64_t)fi * HOP) : 0;
            if (fi != cur_frame) {
                // per-frame tables -> LDS (once per hop_length steps)
                __syncthreads();
                for (int i = tid; i < ND * 512; i += TEAM_THREADS) {
                    const int dd = i >> 9, j = i & 511;
                    lds[L_CM + i] = live ? CMg[(size_t)(fi + dd) * 512 + j] : 0.0f;
                }
                for (int i = tid; i < 512; i += TEAM_THREADS) lds[L_CA + i] = CAg[(size_t)fi * 512 + i];
                for (int i = tid; i < ND * 1536; i += TEAM_THREADS) {
                    const int dd = i / 1536, j = i - dd * 1536;
                    lds[L_VM + i] = live ? VMg[(size_t)(fi + dd) * 1536 + j] : 0.0f;
                }
                for (int i = tid; i < 1536; i += TEAM_THREADS) lds[L_VA + i] = VAg[(size_t)fi * 1536 + i];
                c2_r = C2g[(size_t)fi * 1536 + unit]; c2_z = C2g[(size_t)fi * 1536 + 512 + unit]; c2_n = C2g[(size_t)fi * 1536 + 1024 + unit];
                c3v = C3g[(size_t)fi * 512 + unit];
                c4v = C4g[(size_t)fi * 512 + unit];
                cur_frame = fi;
                __syncthreads();
            }
            float cI0 = lds[L_CA + j0], cI1 = lds[L_CA + j1];
            float vr0 = lds[L_VA + j0], vz0 = lds[L_VA + 512 + j0], vn0 = lds[L_VA + 1024 + j0];
            float vr1 = lds[L_VA + j1], vz1 = lds[L_VA + 512 + j1], vn1 = lds[L_VA + 1024 + j1];
            for (int dd = 0; dd < ND; ++dd) {
                const float kk = live ? ktab[ph * ND + dd] : 0.0f;
                cI0 = fmaf(kk, lds[L_CM + dd * 512 + j0], cI0);
                cI1 = fmaf(kk, lds[L_CM + dd * 512 + j1], cI1);
                const float *vm = lds + L_VM + dd * 1536;
                vr0 = fmaf(kk, vm[j0], vr0); vz0 = fmaf(kk, vm[512 + j0], vz0); vn0 = fmaf(kk, vm[1024 + j0], vn0);
                vr1 = fmaf(kk, vm[j1], vr1); vz1 = fmaf(kk, vm[512 + j1], vz1); vn1 = fmaf(kk, vm[1024 + j1], vn1);
            }
            // sampling noise for this quarter-wave's two classes: -log q  (Gumbel when q = -log u)
            float nz0 = 0.0f, nz1 = 0.0f;
            if (MODE == WRNN_MODE_RAW && has_fc3) {
                if (a.noise_mode == WRNN_NOISE_INJECTED) {
                    const float *qp = a.noise1 + ((size_t)t * a.n_rows + row) * NC + c3row0;
                    nz0 = -logf(qp[0]); nz1 = -logf(qp[1]);
                } else if (a.noise_mode == WRNN_NOISE_PHILOX) {
                    nz0 = -logf(-logf(wrnn_uniform(a.seed, (uint64_t)t, (uint32_t)row, (uint32_t)c3row0)));
                    nz1 = -logf(-logf(wrnn_uniform(a.seed, (uint64_t)t, (uint32_t)row, (uint32_t)c3row0 + 1u)));
                }
            }
            const float xforce = a.x_forced ? a.x_forced[(size_t)t * a.n_rows + row] : 0.0f;

            // ---- phase A: I + GRU1 for units j0, j1, replicated in every WG (:208-212) ----
            float x2_0, x2_1;
            {
                const float xin0 = fmaf(wI0_0, xprev, cI0), xin1 = fmaf(wI0_1, xprev, cI1);
                const float rg0 = sigmoid_f(fmaf(ur0, xprev, vr0) + gh1s[j0]);
                const float rg1 = sigmoid_f(fmaf(ur1, xprev, vr1) + gh1s[j1]);
                const float zg0 = sigmoid_f(fmaf(uz0, xprev, vz0) + gh1s[512 + j0]);
                const float zg1 = sigmoid_f(fmaf(uz1, xprev, vz1) + gh1s[512 + j1]);
                const float ng0 = tanh_f(fmaf(un0, xprev, vn0) + rg0 * gh1s[1024 + j0]);
                const float ng1 = tanh_f(fmaf(un1, xprev, vn1) + rg1 * gh1s[1024 + j1]);
                h1_0 = (1.0f - zg0) * ng0 + zg0 * h1_0;
                h1_1 = (1.0f - zg1) * ng1 + zg1 * h1_1;
                x2_0 = xin0 + h1_0; x2_1 = xin1 + h1_1;
                xb[XB_H1 * 512 + pj0] = h1_0; xb[XB_H1 * 512 + pj1] = h1_1;
                xb[XB_X2 * 512 + pj0] = x2_0; xb[XB_X2 * 512 + pj1] = x2_1;
            }
            __syncthreads();  // B1

            // ---- phase B: GRU2 unit `unit` (:213-216); rows r,z,n of W_ih2[:, :512] . x2 ----
            {
                float gr, gz, gn;
                dot32x3(W_IH2, xb + XB_X2 * 512, q, gr, gz, gn);
                gr = row_sum(gr) + c2_r; gz = row_sum(gz) + c2_z; gn = row_sum(gn) + c2_n;
                const float h2o = xb[XB_H2 * 512 + pu];
                const float x2u = xb[XB_X2 * 512 + pu];
                const float rg = sigmoid_f(gr + gh2_r);
                const float zg = sigmoid_f(gz + gh2_z);
                const float ng = tanh_f(gn + rg * gh2_n);
                const float h2n = (1.0f - zg) * ng + zg * h2o;
                const float x3u = x2u + h2n;
                if (q == 0) st_granule(mX3, par * 512 + unit, epoch, __float_as_uint(x3u));
            }
            // shadow work: gh1 for the next step = W_hh1 . h1' + b_hh1, published for everyone
            {
                float sr, sz, sn;
                dot32x3(W_HH1, xb + XB_H1 * 512, q, sr, sz, sn);
                sr = row_sum(sr) + bhh1_r; sz = row_sum(sz) + bhh1_z; sn = row_sum(sn) + bhh1_n;
                if (q == 0) {
                    st_granule(mGH, par * 1536 + unit, epoch, __float_as_uint(sr));
                    st_granule(mGH, par * 1536 + 512 + unit, epoch, __float_as_uint(sz));
                    st_granule(mGH, par * 1536 + 1024 + unit, epoch, __float_as_uint(sn));
                }
            }
            // ---- exchange 1: x3 = x + h2 for all units; h2' = x3 - x2 ---------------------
            {
                u64 gq[2];
                poll_n<2, 32>(mX3, par * 512 + tid, 256, epoch, gq, dead, a.err, 11u);
                const float x3_0 = __uint_as_float((unsigned)gq[0]), x3_1 = __uint_as_float((unsigned)gq[1]);
                xb[XB_X3 * 512 + pj0] = x3_0; xb[XB_X3 * 512 + pj1] = x3_1;
                xb[XB_H2 * 512 + pj0] = x3_0 - x2_0; xb[XB_H2 * 512 + pj1] = x3_1 - x2_1;
            }
            __syncthreads();  // B2

            // ---- phase C: fc1 row `unit` (:217-218) ---------------------------------------
            {
                const float s = row_sum(dot32(W_FC1, xb + XB_X3 * 512, q)) + c3v;
                if (q == 0) st_granule(mF1, par * 512 + unit, epoch, __float_as_uint(fmaxf(s, 0.0f)));
            }
            // shadow work: gh2 for the next step = W_hh2 . h2' + b_hh2 (stays in this quarter-wave)
            {
                float sr, sz, sn;
                dot32x3(W_HH2, xb + XB_H2 * 512, q, sr, sz, sn);
                gh2_r = row_sum(sr) + bhh2_r; gh2_z = row_sum(sz) + bhh2_z; gh2_n = row_sum(sn) + bhh2_n;
            }
            // ---- exchange 2: fc1 outputs ------------------------------------------------------
            {
                u64 gq[2];
                poll_n<2, 32>(mF1, par * 512 + tid, 256, epoch, gq, dead, a.err, 12u);
                xb[XB_F1 * 512 + pj0] = __uint_as_float((unsigned)gq[0]);
                xb[XB_F1 * 512 + pj1] = __uint_as_float((unsigned)gq[1]);
            }
            __syncthreads();  // B3

            // ---- phase D: fc2 row `unit` (:220-221) ---------------------------------------
            {
                const float s = row_sum(dot32(W_FC2, xb + XB_F1 * 512, q)) + c4v;
                if (q == 0) st_granule(mF2, par * 512 + unit, epoch, __float_as_uint(fmaxf(s, 0.0f)));
            }
            // ---- exchange 3: fc2 outputs ------------------------------------------------------
            {
                u64 gq[2];
                poll_n<2, 32>(mF2, par * 512 + tid, 256, epoch, gq, dead, a.err, 13u);
                xb[XB_F2 * 512 + pj0] = __uint_as_float((unsigned)gq[0]);
                xb[XB_F2 * 512 + pj1] = __uint_as_float((unsigned)gq[1]);
            }
            __syncthreads();  // B4

            // ---- phase E: fc3 rows + race (:223, :231-235) ------------------------------------
            float lg0, lg1;
            {
                const float4 *xp = (const float4 *)(xb + XB_F2 * 512) + q;
                const float4 *wp = (const float4 *)(lds + L_FC3) + (size_t)(wave * 2) * 8 * 64 + lane;
                float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float4 x = xp[k * 16];
                    const float4 wa = wp[k * 64], wb = wp[(8 + k) * 64];
                    a0 = fmaf(wa.x, x.x, a0); a1 = fmaf(wa.y, x.y, a1); a0 = fmaf(wa.z, x.z, a0); a1 = fmaf(wa.w, x.w, a1);
                    b0 = fmaf(wb.x, x.x, b0); b1 = fmaf(wb.y, x.y, b1); b0 = fmaf(wb.z, x.z, b0); b1 = fmaf(wb.w, x.w, b1);
                }
                lg0 = row_sum(a0 + a1) + b3_0;
                lg1 = row_sum(b0 + b1) + b3_1;
            }
            if (a.logits_out && q == 0 && has_fc3) {
                float *lo = a.logits_out + ((size_t)t * a.n_rows + row) * NC + c3row0;
                lo[0] = lg0;
                if (c3row0 + 1 < NC) lo[1] = lg1;
            }
            float x_new = 0.0f;
            if (MODE == WRNN_MODE_RAW) {
                // winner of this quarter-wave's 2 classes: argmax logit_k - log q_k
                const float v0 = lg0 + nz0, v1 = lg1 + nz1;
                const bool p1 = v1 > v0;
                if (q == 0) st_granule(mPR, par * 512 + 16 * g + 4 * wave + r4,
                                       (epoch << 10) | (unsigned)(p1 ? c3row0 + 1 : c3row0), __float_as_uint(p1 ? v1 : v0));
                // ---- exchange 4 (wave 0) + gh1 collection (waves 1-3) ------------------------
                if (wave == 0) {
                    u64 gq[8];
                    poll_n<8, 42>(mPR, par * 512 + lane * 8, 1, epoch & 0x3fffffu, gq, dead, a.err, 14u);
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
                    u64 gq[8];
                    const int base = tid - 64;
                    poll_n<8, 32>(mGH, par * 1536 + base, 192, epoch, gq, dead, a.err, 15u);
#pragma unroll
                    for (int m = 0; m < 8; ++m) gh1s[base + m * 192] = __uint_as_float((unsigned)gq[m]);
                }
                __syncthreads();  // B5
                const int k = misc_i[8];
                // sample = 2 * k / (n_classes - 1.) - 1.   (:235)
                x_new = 2.0f * (float)k / ((float)NC - 1.0f) - 1.0f;
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
                if (wave == 0) {
                    const int nr = NC / 3;
                    float mylg = 0.0f;
                    if (lane < NC) {
                        u64 gq[1];
                        poll_n<1, 32>(mPR, par * 512 + lane, 1, epoch, gq, dead, a.err, 16u);
                        mylg = __uint_as_float((unsigned)gq[0]);
                    }
                    float v = -INFINITY;
                    if (lane < nr) {
                        float u1;
                        if (a.noise_mode == WRNN_NOISE_INJECTED) u1 = a.noise1[((size_t)t * a.n_rows + row) * nr + lane];
                        else u1 = 1e-5f + wrnn_uniform(a.seed, (uint64_t)t, (uint32_t)row, (uint32_t)lane) * (1.0f - 2e-5f);
                        v = mylg - logf(-logf(u1));
                    }
                    const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_max(v)), 63));
                    const u64 ball = __ballot(v == mx);
                    const int km = (int)__builtin_ctzll(ball ? ball : 1ull);
                    const float mean = __shfl(mylg, nr + km, 64);
                    const float ls = fmaxf(__shfl(mylg, 2 * nr + km, 64), -32.23619130191664f);
                    float u2;
                    if (a.noise_mode == WRNN_NOISE_INJECTED) u2 = a.noise2[(size_t)t * a.n_rows + row];
                    else u2 = 1e-5f + wrnn_uniform(a.seed, (uint64_t)t, (uint32_t)row, 10u) * (1.0f - 2e-5f);
                    float xs = mean + expf(ls) * (logf(u2) - logf(1.0f - u2));
                    xs = fminf(fmaxf(xs, -1.0f), 1.0f);
                    if (lane == 0) { misc_f[9] = xs; misc_i[8] = km; }
                } else {
                    u64 gq[8];
                    const int base = tid - 64;
                    poll_n<8, 32>(mGH, par * 1536 + base, 192, epoch, gq, dead, a.err, 15u);
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
            xprev = a.x_forced ? xforce : x_new;   // (:228, :237)
            if ((t & 63) == 63 && __syncthreads_or(dead ? 1 : 0)) return;
        }
        __syncthreads();
    }
}

hipError_t wrnn_launch_loop_team(const WrnnTeamArgs &a, hipStream_t s) {
    static bool attr_set = false;
    const size_t lds = (size_t)L_TOTAL * sizeof(float);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)loop_team_kernel<WRNN_MODE_RAW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void *)loop_team_kernel<WRNN_MODE_MOL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (a.d.mode == WRNN_MODE_RAW)
        hipLaunchKernelGGL(loop_team_kernel<WRNN_MODE_RAW>, dim3(256), dim3(TEAM_THREADS), lds, s, a);
    else
        hipLaunchKernelGGL(loop_team_kernel<WRNN_MODE_MOL>, dim3(256), dim3(TEAM_THREADS), lds, s, a);
    return hipGetLastError();
}
