// The two GRU recurrences of wrnn_train_step (train.hip) as PERSISTENT team kernels: one launch per recurrence instead of one per
// time step.  A step kernel starts with cold caches (the previous step wrote its results back at its end) and costs 6.6 us (forward) /
// 13.9 us (backward) for ~1 us of arithmetic; here the recurrent weights stay in LDS for the whole sequence and the state travels
// through the XCD's L2 with the team kernels' protocol (loop_batch.hip): team = the 32 workgroups of one XCD (HW_REG_XCC_ID), 8-byte
// {tag, value} granules, plain store + sc1 load, the data is the flag, parity double-buffering, bounded spins, arrival check.
//
// Work split (rnn_dims = 512 only; other dims use the step kernels): a team runs R = 4 or 8 batch rows, workgroup g owns hidden units
// 16 g .. 16 g + 15 (48 gate rows of W_hh), wave wl units 16 g + 4 wl .. + 3.
//   forward   gh[R x 48] = h_{t-1}[R x 512] . W_own^T          (output-split, like phase B of the batch kernel)
//             v_mfma_f32_4x4x1: A = weight image in LDS (k phases x 4 units), B = h_{t-1} gathered into LDS in B-operand order;
//             gates by the (unit, row) threads; h_t published, all-gathered (R x 512 granules), 1 barrier per step.
//   backward  carry[R x 512] needs dGH_t[R x 1536] . W_hh: an output-split would have every workgroup gather all 1536 gate
//             derivatives (3x the forward's exchange).  K-SPLIT instead: workgroup g multiplies ITS 48 gate derivatives (computed
//             locally by its (unit, row) threads -- no gather) with its 48 rows of W_hh: a partial carry for ALL 512 units; the 32
//             partials of a unit are summed by the unit's owner (reduce-scatter through the mailbox: R x 512 granules in, R x 512 out
//             per workgroup -- the forward's volume).  v_mfma_f32_4x4x1 with block = 4 output units, i = batch row: one issue =
//             4 rows x 64 outputs x one k.  2 barriers per step.
//
// Order of a step's memory traffic (measured, profiles/r03_train_step_kernel_stats.txt: 2.25 / 2.67 -> 1.70 / 1.84 us per step):
//   * a poll opens with a SENTINEL (one 16-byte slice per lane) and reads everything once that carries the step's tag -- a full look that
//     comes back stale is R x 4 KB of L2 reads per workgroup for nothing, and the publishes of the other workgroups queue behind them;
//   * global loads are issued BEHIND the poll (vmcnt retires in order: a load from HBM in front of a poll holds it up), two steps ahead,
//     into the register set the step has just consumed; the loop is unrolled by two so that the sets rotate by name;
//   * global stores go between the publish and the first look (acknowledged by the L2, off the serial chain).
#include "batch_common.h"

// developer build (-DTT_PROF): cycles per phase of a step, summed over the launch by thread 0 of workgroup 0 of team 0, printed at the end
#ifdef TT_PROF
#define TP_DECL unsigned long long tp_last = __builtin_readcyclecounter(), tp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); tp_acc[i] += n_ - tp_last; tp_last = n_; } while (0)
#define TP_PRINT(name, steps) do { if (team == 0 && g == 0 && tid == 0) printf("%s: cycles per step:  %llu %llu %llu %llu %llu %llu %llu %llu\n", name, \
    tp_acc[0] / (steps), tp_acc[1] / (steps), tp_acc[2] / (steps), tp_acc[3] / (steps), tp_acc[4] / (steps), tp_acc[5] / (steps), tp_acc[6] / (steps), tp_acc[7] / (steps)); } while (0)
#else
#define TP_DECL
#define TP(i)
#define TP_PRINT(name, steps)
#endif

namespace {

constexpr int TT_H = 512;

// team formation + co-residency check, shared by both kernels (see loop_team2.hip); returns false when this workgroup is not in a team
__device__ __forceinline__ bool join_team(unsigned *ctl, unsigned *err, int *misc_i, int n_teams, int &team, int &g) {
    if (threadIdx.x == 0) {
        const unsigned x = xcc_idb();
        misc_i[M_DEAD] = 0;
        const unsigned rank = atomicAdd(&ctl[x], 1u);
        unsigned slot1 = 0, arrived = 0;
        if (rank == 0) {
            slot1 = atomicAdd(&ctl[8], 1u) + 1u;
            __hip_atomic_store(&ctl[16 + x], slot1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (unsigned spins = 0; spins < WRNN_ARRIVE_POLLS; ++spins) {
            slot1 = __hip_atomic_load(&ctl[16 + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            arrived = __hip_atomic_load(&ctl[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (slot1 && arrived >= TB_WGS) break;
        }
        if (arrived < TB_WGS) { slot1 = 0; if (rank < TB_WGS) atomicCAS(err, 0u, WRNN_DEVERR_BUSY); }
        misc_i[M_TEAM] = slot1 ? (int)slot1 - 1 : 1 << 20;
        misc_i[M_RANK] = (int)rank;
    }
    __syncthreads();
    team = __builtin_amdgcn_readfirstlane(misc_i[M_TEAM]);
    g = __builtin_amdgcn_readfirstlane(misc_i[M_RANK]);
    __syncthreads();
    return g < TB_WGS && team < n_teams;
}

// ---------------------------------------------------------------------------------------------------------------- forward
// LDS (floats): weight image [4 waves][3 gates][8 S][64 lanes][4 e] (96 KB) | h vector [rq][S][kp][j][e] (R x 512) | misc
template <int NQ>
struct FL {
    static constexpr int R = 4 * NQ, VEC = R * 512;
    static constexpr int L_W = 0, L_H = 24576, L_MISC = L_H + VEC, L_TOTAL = L_MISC + 16;
    static constexpr unsigned RG = (unsigned)VEC, MAIL = 2 * RG;
};

template <int NQ>
__global__ void __launch_bounds__(TB_THREADS) gru_team_fwd_kernel(WrnnGruTeamArgs a) {
    typedef FL<NQ> F;
    constexpr int R = F::R, NM = R, H = TT_H, G3 = 3 * TT_H;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = (float *)smem;
    int *misc_i = (int *)(lds + F::L_MISC);
    const int tid = threadIdx.x, lane = tid & 63, wl = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 3, kp2 = (lane >> 2) & 3, rho = lane >> 4;
    const int iu = ((rho & 1) << 1) | (rho >> 1);
    const int my_rq = kp2 % NQ;
    const bool primary = kp2 < NQ;
    const int rb = 4 * my_rq + j;
    int team, g;
    if (!join_team(a.ctl, a.err, misc_i, a.n_teams, team, g)) return;
    const int n_batches = (a.B + a.rpb - 1) / a.rpb;
    if (team >= n_batches) return;
    u64 *mail = a.mail + (size_t)team * F::MAIL;
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void *)mail, 0, (int)(F::MAIL * 8u), 0x00020000);
    const int unit = 16 * g + 4 * wl + iu;
    const unsigned mb_own = ((((unsigned)my_rq * 4u + (unsigned)wl) * 8u + (unsigned)(g >> 2)) * 4u + (unsigned)iu) * 16u + (unsigned)j * 4u + (unsigned)(g & 3);
    const unsigned gvoff = (unsigned)tid * 16u;
    {   // this workgroup's slice of the image: [g][4 waves][3][8][64][4]
        const float4 *src = (const float4 *)(a.img + (size_t)g * 24576);
        float4 *dst = (float4 *)(lds + F::L_W);
        for (int i = tid; i < 6144; i += TB_THREADS) dst[i] = src[i];
    }
    const float bh_r = a.bhh[unit], bh_z = a.bhh[H + unit], bh_n = a.bhh[2 * H + unit];
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const lds_cf4p vH = (lds_cf4p)(size_t)launder(smem_base + (unsigned)F::L_H * 4u + (unsigned)lane * 16u);
    const lds_cf4p wim = (lds_cf4p)(size_t)launder(smem_base + (unsigned)F::L_W * 4u + ((unsigned)(wl * 3) * 8u * 64u + (unsigned)lane) * 16u);
    const lds_f2p gdst = (lds_f2p)(size_t)launder(smem_base + (unsigned)F::L_H * 4u + ((unsigned)(tid >> 5) * 256u + 2u * (unsigned)(tid & 31)) * 4u);
    bool dead = false;
    unsigned epoch = 0;
    TP_DECL;
    for (int batch = team; batch < n_batches; batch += a.n_teams) {
        const int brow = batch * a.rpb + rb;
        const bool row_ok = primary && rb < a.rpb && brow < a.B;
        const size_t rbase = (size_t)(row_ok ? brow : 0) * a.L;
        for (int i = tid; i < F::VEC; i += TB_THREADS) lds[F::L_H + i] = 0.0f;   // h_{-1} = 0 (:141-142)
        float hprev = 0.0f;
        // the input parts of steps t (ga) and t + 1 (gb): a step consumes its set at the gates and refills it for step t + 2 right BEHIND
        // its gather, so the loads never sit in front of a poll (vmcnt retires in order: a poll issued behind an HBM load waits for it too)
        // and have two steps to land; the loop is unrolled by two so that the sets rotate by name, not by copies (a copy of a register that
        // is being loaded is a wait for the load)
        float ga[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f};
        if (row_ok) {
            const float *gi = a.GI + rbase * G3 + unit;
            ga[0] = gi[0]; ga[1] = gi[H]; ga[2] = gi[2 * H];
            if (a.L > 1) { gb[0] = gi[G3]; gb[1] = gi[G3 + H]; gb[2] = gi[G3 + 2 * H]; }
        }
        __syncthreads();
        auto step = [&](const int64_t t, float (&gv)[3]) __attribute__((always_inline)) {
            ++epoch;
            const unsigned par = epoch & 1u;
            TP(7);
            f4 acc[3][NQ];
#pragma unroll
            for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[gt][q] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < 8; ++S) {
                f4 b[NQ], w[3];
#pragma unroll
                for (int q = 0; q < NQ; ++q) b[q] = vH[(q * 8 + S) * 64];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt) w[gt] = wim[(gt * 8 + S) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) acc[gt][q] = mfma4(w[gt][e], b[q][e], acc[gt][q]);
            }
            TP(0);   // MFMA phase
            float tr = 0.f, tz = 0.f, tn = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float fr = fold_kp(acc[0][q]), fz = fold_kp(acc[1][q]), fn = fold_kp(acc[2][q]);
                if (q == 0 || my_rq == q) { tr = fr; tz = fz; tn = fn; }
            }
            const float ghr = tr + bh_r, ghz = tz + bh_z, ghn = tn + bh_n;
            const float r = 1.0f / (1.0f + expf(-(gv[0] + ghr))), z = 1.0f / (1.0f + expf(-(gv[1] + ghz)));
            const float n = tanhf(gv[2] + r * ghn);
            const float h = (1.0f - z) * n + z * hprev;
            TP(1);   // folds + gates
            if (primary) st_granule(mail, par * F::RG + mb_own, epoch, __float_as_uint(h));
            hprev = h;
            // the step's results go out between the publish and the first look (stores are acknowledged by the L2, they do not hold
            // the poll up like a load from HBM would)
            if (row_ok) {
                const size_t o = (rbase + t) * H + unit;
                a.Hs[o] = h; a.Rs[o] = r; a.Zs[o] = z; a.Ns[o] = n; a.GHN[o] = ghn;
                if (t + 1 < a.L) a.HP[o + H] = h;
            }
            TP(2);   // publish + stores
            {   // also after the LAST step, whose result nobody needs: the gather is what keeps the 32 workgroups within one publish
                // of each other -- without it a fast workgroup starts the team's next batch and overwrites the parity region a slow one
                // is still reading for step L - 2 (found with B = 70: 9 batches on 8 teams)
                const unsigned so = par * F::RG * 8u;
                u4v gx[NM];
                unsigned spins = 0;
                // SENTINEL FIRST: one slice per lane until it carries this step's tag, then everything once.  A full look that comes back
                // stale is R x 4 KB of L2 reads per workgroup for nothing, and the publishes of the other workgroups queue behind those
                // reads: with the full look first a gather took 2.6-3.0 k cycles, like this 1.3 k
                for (;;) {
                    const u4v sv = ld_pair(mrs, gvoff, so + (NM - 1) * 4096u);
                    if (__all(sv.y == epoch && sv.w == epoch) || dead) break;
                    if (++spins > TB_SPIN_MAX) { dead = true; if (lane == 0) atomicExch(a.err, 31u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int m = 0; m < NM; ++m) gx[m] = ld_pair(mrs, gvoff, so + m * 4096u);
                for (;;) {   // per-granule retry: only what came back stale is looked at again
                    bool ok = true;
#pragma unroll
                    for (int m = 0; m < NM; ++m) ok = ok && gx[m].y == epoch && gx[m].w == epoch;
                    if (__all(ok) || dead) break;
                    if (++spins > TB_SPIN_MAX) { dead = true; if (lane == 0) atomicExch(a.err, 31u); break; }
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int m = 0; m < NM; ++m)
                        if (!(gx[m].y == epoch && gx[m].w == epoch)) gx[m] = ld_pair(mrs, gvoff, so + m * 4096u);
                }
                TP(3);   // gather
#pragma unroll
                for (int m = 0; m < NM; ++m) gdst[((m >> 2) * 2048 + (m & 3) * 64) / 2] = (f2v){__uint_as_float(gx[m].x), __uint_as_float(gx[m].z)};
            }
            if (row_ok && t + 2 < a.L) {   // refill the consumed set for step t + 2: behind the poll, never in front of it (measured: in front,
                                           // the HBM round trip of ~1 500 cycles holds the first look up: 1.29 -> 1.90 k cycles per gather)
                const float *gi = a.GI + (rbase + t + 2) * G3 + unit;
                gv[0] = gi[0]; gv[1] = gi[H]; gv[2] = gi[2 * H];
            }
            TP(4);   // LDS write + loads issued
            __syncthreads();
            TP(5);   // barrier
        };
        for (int64_t t = 0; t < a.L; t += 2) {
            step(t, ga);
            if (t + 1 < a.L) step(t + 1, gb);
            if ((t & 62) == 62) {
                if (dead && lane == 0) misc_i[M_DEAD] = 1;
                __syncthreads();
                if (misc_i[M_DEAD]) return;
            }
        }
        __syncthreads();
    }
    TP_PRINT("gru_team_fwd", (unsigned long long)a.L);
}

// ---------------------------------------------------------------------------------------------------------------- backward
// LDS (floats): W image [12 k4][8 c][64 lanes][4] (96 KB: rows k = gate * 16 + u16 of W_hh, as B operands: lane (blk, j) of chunk c
// = output unit 64 c + 4 blk + j) | G [R rows][48 k] this step's gate derivatives of the own units | RED [source groups][R][16] | misc
template <int NQ>
struct BL {
    static constexpr int R = 4 * NQ;
    static constexpr int L_W = 0, L_G = 24576, L_RED = L_G + R * 48, L_MISC = L_RED + 512, L_TOTAL = L_MISC + 16;
    // mailbox per team and parity: [dest workgroup 32][row R][source workgroup 32][unit 16]
    static constexpr unsigned RG = 32u * (unsigned)R * 32u * 16u, MAIL = 2 * RG;
};

template <int NQ>
__global__ void __launch_bounds__(TB_THREADS) gru_team_bwd_kernel(WrnnGruTeamArgs a) {
    typedef BL<NQ> Bk;
    constexpr int R = Bk::R, H = TT_H, G3 = 3 * TT_H;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = (float *)smem;
    int *misc_i = (int *)(lds + Bk::L_MISC);
    float *Gs = lds + Bk::L_G, *red = lds + Bk::L_RED;
    const int tid = threadIdx.x, lane = tid & 63, wl = __builtin_amdgcn_readfirstlane(tid >> 6);
    int team, g;
    if (!join_team(a.ctl, a.err, misc_i, a.n_teams, team, g)) return;
    const int n_batches = (a.B + a.rpb - 1) / a.rpb;
    if (team >= n_batches) return;
    u64 *mail = a.mail + (size_t)team * Bk::MAIL;
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void *)mail, 0, (int)(Bk::MAIL * 8u), 0x00020000);
    {
        const float4 *src = (const float4 *)(a.img + (size_t)g * 24576);
        float4 *dst = (float4 *)(lds + Bk::L_W);
        for (int i = tid; i < 6144; i += TB_THREADS) dst[i] = src[i];
    }
    // (unit, row) threads of the gate-derivative phase: tid < 16 * R, u16 = tid & 15, row = tid >> 4
    const int pu = tid & 15, prow = tid >> 4;
    const bool pair = tid < 16 * R;
    const int punit = 16 * g + pu;
    // MFMA roles: lane (blk = lane >> 2, ij = lane & 3): A index i = batch row of the quad, B index j = output unit 64 c + 4 blk + j
    const int blk = lane >> 2, ij = lane & 3;
    // consumer of the reduce-scatter: 16-byte loads = 2 adjacent units; R = 8: row = (tid >> 3) & 7, sources 8 wl .. 8 wl + 7;
    // R = 4: row = (tid >> 3) & 3, sources 4 (tid >> 5) .. + 3
    const int cup = tid & 7, crow = (tid >> 3) & (R - 1);
    constexpr int NSRC = NQ == 2 ? 8 : 4;
    const int csrc0 = NQ == 2 ? 8 * wl : 4 * (tid >> 5);
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const lds_cf4p wim = (lds_cf4p)(size_t)launder(smem_base + (unsigned)Bk::L_W * 4u + (unsigned)lane * 16u);
    bool dead = false;
    unsigned epoch = 0;
    TP_DECL;
    for (int batch = team; batch < n_batches; batch += a.n_teams) {
        const int brow = batch * a.rpb + prow;
        const bool row_ok = pair && prow < a.rpb && brow < a.B;
        const size_t rbase = (size_t)(row_ok ? brow : 0) * a.L;
        float carry = 0.0f, cd = 0.0f;
        // saved activations {dHext, r, z, n, gh_n, h_prev} of steps t (ea) and t - 1 (eb): consumed by the gate derivatives at the top of
        // a step and refilled for step t - 2 BEHIND the step's poll (see the forward kernel); unrolled by two, the sets rotate by name
        float ea[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, eb[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (row_ok) {
            const size_t o = (rbase + a.L - 1) * H + punit;
            ea[0] = a.dHext[o]; ea[1] = a.Rs[o]; ea[2] = a.Zs[o]; ea[3] = a.Ns[o]; ea[4] = a.GHN[o]; ea[5] = a.HP[o];
            if (a.L > 1) { const size_t o2 = o - H; eb[0] = a.dHext[o2]; eb[1] = a.Rs[o2]; eb[2] = a.Zs[o2]; eb[3] = a.Ns[o2]; eb[4] = a.GHN[o2]; eb[5] = a.HP[o2]; }
        }
        auto step = [&](const int64_t t, float (&ev)[6]) __attribute__((always_inline)) {
            ++epoch;
            const unsigned par = epoch & 1u;
            TP(7);
            // ---- gate derivatives of the own (unit, row) pairs (see gru_bwd_step_kernel, train.hip)
            float dpr = 0.f, dpz = 0.f, dpn = 0.f, dghn = 0.f;
            if (pair) {
                if (row_ok) {
                    const float e_dh = ev[0], e_r = ev[1], e_z = ev[2], e_n = ev[3], e_ghn = ev[4], e_hp = ev[5];
                    const float dH = e_dh + carry + cd;
                    const float dn = dH * (1.0f - e_z), dz = dH * (e_hp - e_n);
                    dpn = dn * (1.0f - e_n * e_n);
                    dpr = (dpn * e_ghn) * e_r * (1.0f - e_r);
                    dpz = dz * e_z * (1.0f - e_z);
                    dghn = dpn * e_r;
                    cd = dH * e_z;
                }
                Gs[prow * 48 + pu] = dpr; Gs[prow * 48 + 16 + pu] = dpz; Gs[prow * 48 + 32 + pu] = dghn;
            }
            TP(0);   // gate derivatives, stores
            __syncthreads();   // (t == 0 included: its carry is not needed, its exchange keeps the team in lock-step across batches)
            TP(1);   // barrier
            // ---- partial carry of ALL 512 units from the own 48 gate rows: P[row][out] = sum_k G[row][k] W[k][out]
            f4 acc[2][NQ];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[cc][q] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k4 = 0; k4 < 12; ++k4) {
                f4 wb[2], ga[NQ];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) wb[cc] = wim[(k4 * 8 + 2 * wl + cc) * 64];
#pragma unroll
                for (int q = 0; q < NQ; ++q) ga[q] = *(const f4 *)(Gs + (4 * q + ij) * 48 + 4 * k4);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) acc[cc][q] = mfma4(ga[q][kk], wb[cc][kk], acc[cc][q]);
            }
            TP(2);   // MFMA phase
            // ---- publish: D[i] of lane (blk, j) = P[row 4 q + i][out = 64 c + 4 blk + j] -> owner workgroup out >> 4
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int out = 64 * (2 * wl + cc) + 4 * blk + ij;
                const unsigned base = par * Bk::RG + (unsigned)(out >> 4) * (unsigned)(R * 512) + (unsigned)g * 16u + (unsigned)(out & 15);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        st_granule(mail, base + (unsigned)(4 * q + i) * 512u, epoch, __float_as_uint(acc[cc][q][i]));
            }
            // ---- behind the publish, in front of the first look: this step's results out (stores are acknowledged by the L2)
            if (row_ok) {
                const size_t o = (rbase + t) * G3 + punit;
                a.dGI[o] = dpr; a.dGI[o + H] = dpz; a.dGI[o + 2 * H] = dpn;
                a.dGH[o] = dpr; a.dGH[o + H] = dpz; a.dGH[o + 2 * H] = dghn;
            }
            TP(3);   // publish, stores
            // ---- reduce-scatter: the partials of the own units from all 32 workgroups
            {
                const unsigned voff = (((unsigned)crow * 32u) * 16u + 2u * (unsigned)cup) * 8u;
                const unsigned soff = (par * Bk::RG + (unsigned)g * (unsigned)(R * 512)) * 8u;
                u4v gq[NSRC];
                unsigned spins = 0;
                // sentinel first (see the forward kernel)
                for (;;) {
                    const u4v sv = ld_pair(mrs, voff + (unsigned)(csrc0 + NSRC - 1) * 128u, soff);
                    if (__all(sv.y == epoch && sv.w == epoch) || dead) break;
                    if (++spins > TB_SPIN_MAX) { dead = true; if (lane == 0) atomicExch(a.err, 32u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int m = 0; m < NSRC; ++m) gq[m] = ld_pair(mrs, voff + (unsigned)(csrc0 + m) * 128u, soff);
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int m = 0; m < NSRC; ++m) ok = ok && gq[m].y == epoch && gq[m].w == epoch;
                    if (__all(ok) || dead) break;
                    if (++spins > TB_SPIN_MAX) { dead = true; if (lane == 0) atomicExch(a.err, 32u); break; }
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int m = 0; m < NSRC; ++m)
                        if (!(gq[m].y == epoch && gq[m].w == epoch)) gq[m] = ld_pair(mrs, voff + (unsigned)(csrc0 + m) * 128u, soff);
                }
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int m = 0; m < NSRC; ++m) { s0 += __uint_as_float(gq[m].x); s1 += __uint_as_float(gq[m].z); }
                // partial sums of this thread's source group -> RED [group][row][unit]
                const int grp = NQ == 2 ? wl : (tid >> 5);
                red[(grp * R + crow) * 16 + 2 * cup] = s0;
                red[(grp * R + crow) * 16 + 2 * cup + 1] = s1;
            }
            TP(4);   // reduce-scatter poll + partial sums
            if (row_ok && t > 1) {   // the inputs of step t - 2: behind the poll
                const size_t o2 = (rbase + t - 2) * H + punit;
                ev[0] = a.dHext[o2]; ev[1] = a.Rs[o2]; ev[2] = a.Zs[o2]; ev[3] = a.Ns[o2]; ev[4] = a.GHN[o2]; ev[5] = a.HP[o2];
            }
            __syncthreads();
            TP(5);   // loads issued + barrier
            if (pair) {
                constexpr int NG = NQ == 2 ? 4 : 8;
                float acc_c = 0.0f;
#pragma unroll
                for (int gq2 = 0; gq2 < NG; ++gq2) acc_c += red[(gq2 * R + prow) * 16 + pu];
                carry = acc_c;
            }
        };
        for (int64_t t = a.L - 1; t >= 0; t -= 2) {
            step(t, ea);
            if (t >= 1) step(t - 1, eb);
            if ((epoch & 62u) == 62u) {
                if (dead && lane == 0) misc_i[M_DEAD] = 1;
                __syncthreads();
                if (misc_i[M_DEAD]) return;
            }
        }
        __syncthreads();
    }
    TP_PRINT("gru_team_bwd", (unsigned long long)a.L);
}

// ---- weight images (device-side repack; the parameters change every optimiser step) -----------------------------------------------
// forward: img[g][wl][gate][S][lane][e] = W_hh[gate * 512 + 16 g + 4 wl + (lane & 3)][64 S + 16 e + (lane >> 2)]
__global__ void __launch_bounds__(256) pack_fwd_image_kernel(const float *__restrict__ Whh, float *__restrict__ img) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // 32 * 4 * 3 * 8 * 64 * 4 = 786 432
    if (idx >= 786432) return;
    const int e = idx & 3, lane = (idx >> 2) & 63, S = (idx >> 8) & 7, rest = idx >> 11;
    const int gate = rest % 3, wl = (rest / 3) & 3, g = rest / 12;
    img[idx] = Whh[(size_t)(gate * 512 + 16 * g + 4 * wl + (lane & 3)) * 512 + 64 * S + 16 * e + (lane >> 2)];
}
// backward: img[g][k4][c][lane][kk] = W_hh[(gate = k / 16) * 512 + 16 g + (k % 16)][64 c + lane], k = 4 k4 + kk
__global__ void __launch_bounds__(256) pack_bwd_image_kernel(const float *__restrict__ Whh, float *__restrict__ img) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // 32 * 12 * 8 * 64 * 4
    if (idx >= 786432) return;
    const int kk = idx & 3, lane = (idx >> 2) & 63, c = (idx >> 8) & 7, rest = idx >> 11;
    const int k4 = rest % 12, g = rest / 12;
    const int k = 4 * k4 + kk;
    img[idx] = Whh[(size_t)((k >> 4) * 512 + 16 * g + (k & 15)) * 512 + 64 * c + lane];
}

}  // namespace

size_t wrnn_gru_team_mail_granules(int nq, bool bwd) { return bwd ? (nq == 2 ? BL<2>::MAIL : BL<1>::MAIL) : (nq == 2 ? FL<2>::MAIL : FL<1>::MAIL); }

hipError_t wrnn_gru_team_pack(const float *Whh, float *img, bool bwd, hipStream_t s) {
    (void)hipGetLastError();
    if (bwd) hipLaunchKernelGGL(pack_bwd_image_kernel, dim3(3072), dim3(256), 0, s, Whh, img);
    else hipLaunchKernelGGL(pack_fwd_image_kernel, dim3(3072), dim3(256), 0, s, Whh, img);
    return hipGetLastError();
}

template <class K>
static hipError_t launch_team(K kern, size_t lds, const WrnnGruTeamArgs &a, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.n_teams * TB_WGS), dim3(TB_THREADS), lds, s, a);
    return hipGetLastError();
}

hipError_t wrnn_gru_team_launch(const WrnnGruTeamArgs &a, int nq, bool bwd, hipStream_t s) {
    (void)hipGetLastError();
    if (bwd) return nq == 2 ? launch_team(gru_team_bwd_kernel<2>, (size_t)BL<2>::L_TOTAL * 4, a, s) : launch_team(gru_team_bwd_kernel<1>, (size_t)BL<1>::L_TOTAL * 4, a, s);
    return nq == 2 ? launch_team(gru_team_fwd_kernel<2>, (size_t)FL<2>::L_TOTAL * 4, a, s) : launch_team(gru_team_fwd_kernel<1>, (size_t)FL<1>::L_TOTAL * 4, a, s);
}

// co-residency facts for wrnn_train_step's check
hipError_t wrnn_gru_team_occupancy(int nq, bool bwd, int *blocks_per_cu) {
    const void *fn;
    size_t lds;
    if (bwd) { fn = nq == 2 ? (const void *)gru_team_bwd_kernel<2> : (const void *)gru_team_bwd_kernel<1>; lds = (size_t)(nq == 2 ? BL<2>::L_TOTAL : BL<1>::L_TOTAL) * 4; }
    else { fn = nq == 2 ? (const void *)gru_team_fwd_kernel<2> : (const void *)gru_team_fwd_kernel<1>; lds = (size_t)(nq == 2 ? FL<2>::L_TOTAL : FL<1>::L_TOTAL) * 4; }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, TB_THREADS, lds);
}
