// WRNN_KERNEL_BATCH_CS: the batch kernel (loop_batch.hip: R = 4*NQ rows per XCD team in lock-step on v_mfma_f32_4x4x1, the
// reference's "all B rows advance together", fatchord_version.py:194-237) with WAVE SPECIALISATION -- two waves per SIMD.
//
// Why: loop_batch.hip keeps all 320 weight registers of a SIMD lane in ONE wave (512 registers), so the MFMA phases of the serial
// chain, the shadow MFMAs (W_hh1.h1', W_hh2.h2'), noise, conditioning and the six exchange waits run one after the other in
// one instruction stream: 832 MFMAs issue in 6 800 of a 22 300-cycle step (R = 8).  Here a SIMD holds two waves of 256 registers with
// different jobs, as in loop_team2.hip -- NOT for matrix throughput (round 5, bench_micro/mfma4_chip: one MFMA-only wave per SIMD delivers
// 148 TFLOP/s chip-wide, two deliver 154: the pipe is one resource per SIMD) but so that the shadow MFMAs run while the critical wave
// waits for memory (the sentinel round trips of its exchanges):
//   waves 0-3 "C" (critical): W_ih2 (96) + fc1 (32) + fc2 (32) weights in VGPRs, the fc3 slice in LDS: phase A (I + GRU1),
//                             phase B (GRU2), fc1, fc2, fc3 + the race; the four gathers of the serial chain (x2, x3, fc1, fc2)
//                             and the winners.  Nothing else: between a publish and its gather a C wave only polls.
//   waves 4-7 "S" (shadow):   W_hh1 (96) + W_hh2 gates r,z (64) in VGPRs, gate n in LDS: the h1' gather (off the serial chain
//                             now: C waits for x2 only), gh1' = W_hh1.h1', gh2' = W_hh2.(x3 - x2), the sampling noise and the
//                             conditioning of the NEXT step.  Results go to the C wave of the same SIMD -- same lane = same
//                             (unit, batch row) -- through small LDS slots, separated by the step's 5 workgroup barriers.
// No AGPR parking: hipcc splits a 256-register wave 128 : 128 between VGPRs and AGPRs as soon as a kernel touches an AGPR, so this
// file is compiled with -mllvm -amdgpu-mfma-vgpr-form (MFMA results in VGPRs; see the Makefile) and all 160 weights are plain floats.
//
// Team, residency, mailbox regions, granule protocol, thread <-> (unit, row) map, B-operand order in LDS, the K-phase fold and
// the software-pipelined MFMA loops are those of loop_batch.hip (batch_common.h).  Per step: 5 exchanges, 5 barriers (all 8 waves).
//   window 1: C phase A, publish x2 | h1', gather x2 -> P          S gather h1' -> H1
//   window 2: C phase B (W_ih2.x2), publish x3, gather -> Q         S W_hh1.h1' -> gh1 slots
//   window 3: C fc1, publish, gather -> H1                          S W_hh2.(Q - P) -> gh2 slots            (RAW at 8 rows, CS_SPREAD: in window 4, and the
//   window 4: C fc2, publish, gather -> P                           S (MOL: conditioning of step t+1)         fc2 outputs go to H1 instead of P)
//   window 5: C fc3 + race, publish, candidates, winners -> xn      S conditioning of step t+1 -> cd / frame-constant slots
#include "batch_common.h"

#define CS_THREADS 512
// Developer knobs (tools/build_variant.sh NAME loop_batch_cs -DCS_...).  What round 4 measured and rejected -- slice rotation, a throttled shadow product,
// every polling variant except "sentinel slice first, then everything", yielding shadow waves, a v_min3 tag check, the wrong-result timing diagnostics of
// the shadow products' operands -- is recorded in profiles/r04_batch_cs_experiments.txt and no longer lives in this file.  Round 5 rebuilt the exchange
// (untagged 4-byte words with an "empty" bit pattern as the flag, {x2, h1'} as one 8-byte pair so that the S waves never look, W_hh1 from B1 on, gate n of
// W_hh2 on the C waves, RAW fc3 split between the waves of a SIMD, shadow waves yielding through an LDS token): parity-green and 9-19 % SLOWER in every
// combination (profiles/r05_batch_cs_experiments.txt; that kernel: git show e3e6332:tools/experiments/loop_batch_cs_words_r5.hip).  The two waves of a SIMD share ONE
// matrix pipe and ONE VALU issue port: a shadow MFMA is free only beside a memory wait of the critical wave, which is where this schedule has them.
#ifndef CS_PRIO
#define CS_PRIO 1        // the C waves run at s_setprio 3: the two waves of a SIMD compete for issue slots, the serial chain goes first
#endif
#ifndef CS_SPLIT_WIN
#define CS_SPLIT_WIN 1   // RAW, 8 rows per team: the winner of batch row wl + 4 is reduced by the S wave (window 5, behind its conditioning work)
#endif
#ifndef CS_SPRIO
#define CS_SPRIO 1       // the S waves' h1' gather + meeting point run at the C waves' priority (+1 % at R = 4, nothing at R = 8)
#endif
#ifndef CS_FC3_SPLIT
#define CS_FC3_SPLIT 1   // MOL: fc3 shared between the C and the S wave of a SIMD
#endif
#ifndef CS_COND_W4
#define CS_COND_W4 (MODE == WRNN_MODE_MOL)   // MOL: the conditioning of the next step in window 4 (in window 5 the C waves waited for it at B4b: 780 cycles)
#endif
#ifndef CS_PROF_SPLIT
#define CS_PROF_SPLIT 0  // instrumented build: the C waves' exchanges are reported in two parts (sentinel wait: markers 17 / 7 / 12 / 16, the rest under the usual marker)
#endif
#if CS_PROF_SPLIT
#define GSF_DECL unsigned ts_ = 0
#define GSF_TS , PROF ? &ts_ : nullptr
#define GSF_ACC(i) PBS(i, ts_)
#else
#define GSF_DECL
#define GSF_TS
#define GSF_ACC(i)
#endif
#ifndef CS_PROF_WG
#define CS_PROF_WG 0     // instrumented build: the workgroup (arrival rank inside its team) whose wave 0 / wave 4 are reported
#endif
#ifndef CS_DIAG
#define CS_DIAG 0        // TIMING DIAGNOSTIC ONLY (wrong results): bit 0 = the S waves skip the W_hh1 MFMAs, bit 1 = the W_hh2 MFMAs
#endif
#ifndef CS_EARLY_LOOK
#define CS_EARLY_LOOK 8  // the full look goes out once this many of the sentinel slice's 64 lanes carry the tag (0 = all of them)
#endif
#ifndef CS_SPREAD
#define CS_SPREAD (MODE == WRNN_MODE_RAW && NQ == 2)   // RAW at 8 rows per team (+1.5 % with CS_SCHEDBAR; MOL at 4 rows: -7 %, its window 4 holds the conditioning; round 5 sessions 9-11):
                         // W_hh2 . (x3 - x2) runs in window 4, where the S waves idle, instead of window 3, where the C waves' fc1 look waited for it.  The fc2 outputs are then gathered
                         // into H1 (dead after fc2) instead of P, so that x2 (P) and x3 (Q) stay intact until the next step; fc3 reads H1.
#endif
#ifndef CS_SPREAD_MEET
#define CS_SPREAD_MEET 1   // CS_SPREAD only: the C waves meet (an LDS counter read under the fc2 look) before they land fc2 outputs in H1 -- a CORRECTNESS requirement
                           // (see window 4); 0 exists for timing A/Bs only
#endif
#ifndef CS_SCHEDBAR
#define CS_SCHEDBAR (MODE == WRNN_MODE_RAW && NQ == 2)   // a scheduling barrier at every phase boundary of the step (where the instrumented build has its time stamps): hipcc otherwise moves
                         // instructions across the boundaries; RAW at 8 rows +0.7 %, MOL at 4 rows -2.8 % (session 10)
#endif
#ifndef CS_MINCHK
#define CS_MINCHK (NM > 4)   // (8 rows per team: +0.3 ... +0.8 %, 4 rows: -0.3 %; round 5 session 8) the tags of a look are checked with ONE v_min3_u32 per 16-byte load, in the order the loads return (a tag is never AHEAD of the step: nobody
                         // can publish step e + 2 into a parity while somebody still looks for step e, so "all fresh" <=> min == tag), instead of two compares + two scalar ANDs
#endif
// PRECONDITION of CS_MINCHK (round-5 advisor): no granule of a region may carry a tag GREATER than the epoch being looked for.  Holds because (1) api.hip
// zeroes the whole mailbox inside the team gate before every launch, (2) `epoch` only grows inside a launch (one counter across the passes of a
// launch, never reset), (3) a parity is re-published only two steps later, behind the barriers that end the looks at the older step, and (4) 2^32 steps
// (~5 h of one launch at 4.8 us per step) are never reached: wrnn_generate's steps are bounded by the caller's clip.  A future segment-resume or mailbox-
// reuse path that restarts `epoch` must either zero the mailbox again or fall back to the equality check (CS_MINCHK 0).
#ifndef CS_PUT2
#define CS_PUT2 (NQ == 1)    // (4 rows per team: +1.4 ... +2 %, 8 rows: -0.4 %; round 5 session 8, round 6 session 3) a gathered vector goes to LDS as ds_write2_b32 from the
                             // registers the load filled, instead of one ds_write_b64 per load behind a v_mov per value.  Its two words are adjacent, so 32 lanes hit 16 of the
                             // 32 write banks twice (SQ_LDS_BANK_CONFLICT 14.9 % of the LDS-active cycles) -- which costs nothing that can be measured: round 6 built the
                             // conflict-free form (producers place the pair a lane loads 32 words apart; conflicts back to 2.2 %) and it was 1.7 % SLOWER at its best
                             // early-look threshold, 10 % slower at the shipped one -- the placement decides whose publish a sentinel lane waits for, and that is what
                             // the step time follows (profiles/r06_batch_cs_experiments.txt)
#endif
#ifndef CS_FLAG_POLL_SLEEP
#define CS_FLAG_POLL_SLEEP 1   // s_sleep units between two looks at the S waves' LDS meeting flags (0 = a tight ds_read loop at the C waves' priority beside their
                               // phase-B MFMAs: -3.5 % / -4.5 %, round 5 session 5)
#endif
#ifndef CS_MAX_NQ
#define CS_MAX_NQ 2      // row quads per team this file is built for
#endif

namespace {

template <int NQ>
struct LayCS {
    static constexpr int R = 4 * NQ;
    static constexpr int VEC = R * 512;
    static constexpr int SL = 64 * NQ;              // one hand-over / constant slot: [4 waves][4 units][NQ quads][4 rows]
    static constexpr int NH = 19;
    static constexpr int L_FC3 = 0;                 // as loop_batch.hip
    static constexpr int L_WN = 16384;
    static constexpr int L_CST = L_WN + 8192;       // [12][SL]
    static constexpr int L_HAND = L_CST + 12 * SL;  // [NH][SL]  S -> C
    static constexpr int L_P = L_HAND + NH * SL;    // x2, later fc2 outputs
    static constexpr int L_Q = L_P + VEC;           // x3
    static constexpr int L_H1 = L_Q + VEC;          // h1', later fc1 outputs
    static constexpr int L_XN = L_H1 + VEC;
    static constexpr int L_LG = L_H1;               // [R][32] MOL: the 30 fc3 outputs of every batch row, in window 5 (H1 is dead from B4 to the next h1' gather)
    static constexpr int L_MISC = L_XN + 16;
    static constexpr int L_PROF = L_MISC + 16;      // [2 roles][24]: phase cycles of wave 0 (C) and wave 4 (S), instrumented build only
    static constexpr int L_TOTAL = L_PROF + 48;
    static_assert(L_TOTAL * 4 <= 163840, "LDS budget");
    static_assert((L_P % 4) == 0, "B operands are read as 16-byte vectors");
};
// hand-over slots
constexpr int H_GH1R = 0, H_GH1Z = 1, H_GH1N = 2, H_CDX = 3, H_CDY = 4, H_CDZ = 5, H_CDW = 6, H_GH2R = 7, H_GH2Z = 8, H_GH2N = 9,
              H_C2R = 10, H_C2Z = 11, H_C2N = 12, H_C3 = 13, H_C4 = 14, H_NZ = 15;   // H_NZ: [parity][2]

// sentinel first, then everything: a C / S wave has nothing to do between its publish and this gather, and a poll that opens with a
// full look re-reads R x 4 KB per workgroup while the producers' stores queue behind those reads.  The full look
// goes out as soon as CS_EARLY_LOOK of the sentinel slice's 64 lanes carry the step's tag: the stragglers' granules land while it is
// in flight, so the sentinel round trip and the data round trip overlap (round 4: +2.1 % / +4.4 %).
typedef volatile unsigned __attribute__((address_space(3))) *lds_vup;
// `count` (optional): one LDS word read UNDER the data look -- issued behind the look's loads, so that the DS round trip is covered by the
// L2 round trip -- and returned in *count_out for the caller to check (the C waves' meeting point of CS_SPREAD, see window 4).
template <int NM>
__device__ __forceinline__ void gather_sf(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned tag, u4v (&g)[1][NM], bool &dead,
                                          unsigned *err, unsigned code, unsigned *sent_cyc = nullptr, lds_vup count = nullptr, unsigned *count_out = nullptr) {
    unsigned spins = 0;
    for (;;) {
        const u4v sv = ld_pair(rs, voff, soff + (NM - 1) * 4096u);
#if CS_EARLY_LOOK
        if (__builtin_popcountll(__ballot(sv.y == tag && sv.w == tag)) >= CS_EARLY_LOOK || dead) break;
#else
        if (__all(sv.y == tag && sv.w == tag) || dead) break;
#endif
        if (++spins > TB_SPIN_MAX) { dead = true; if ((threadIdx.x & 63) == 0) atomicExch(err, code); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    if (sent_cyc) *sent_cyc = (unsigned)__builtin_readcyclecounter();   // instrumented build: the sentinel wait ends here, the data look starts
    if (CS_MINCHK) {
      for (;;) {
#pragma unroll
        for (int m = 0; m < NM; ++m) g[0][m] = ld_pair(rs, voff, soff + m * 4096u);
        if (count) *count_out = *count;
        unsigned mn = 0xffffffffu;
#pragma unroll
        for (int m = 0; m < NM; ++m) { const unsigned a = mn < g[0][m].y ? mn : g[0][m].y; mn = a < g[0][m].w ? a : g[0][m].w; }   // v_min3_u32 mn, mn, y, w
        if (__all(mn == tag) || dead) break;
        for (;;) {
            if (++spins > TB_SPIN_MAX) { dead = true; if ((threadIdx.x & 63) == 0) atomicExch(err, code); break; }
            __builtin_amdgcn_s_sleep(1);
            const u4v sv = ld_pair(rs, voff, soff + (NM - 1) * 4096u);
            if (__all(sv.y == tag && sv.w == tag)) break;
        }
        if (dead) break;
      }
    } else {
        const unsigned offs[1] = {soff};
        gather_vecs<NM, 1, false>(rs, voff, offs, tag, g, dead, err, code);
        if (count) *count_out = *count;
    }
}

// one set of 4 fc3 rows (A-operand image `w3s` in LDS: [8 slabs][64 lanes] f4) times the gathered fc2 outputs: the thread's folded logit
template <int NQ, int D3>
__device__ __forceinline__ float fc3_one_set(lds_cf4p w3s, lds_cf4p xv, int my_rq) {
    constexpr int NP = NQ == 1 ? 2 : 1;
    f4 acc[NP][NQ];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[p][q] = (f4){0.f, 0.f, 0.f, 0.f};
    f4 ring[D3][NQ], rwa[D3];
#pragma unroll
    for (int dd = 0; dd < D3; ++dd) {
        rwa[dd] = w3s[dd * 64];
#pragma unroll
        for (int q = 0; q < NQ; ++q) ring[dd][q] = xv[(q * 8 + dd) * 64];
    }
#pragma unroll
    for (int S = 0; S < 8; ++S) {
        f4 b[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) b[q] = ring[S % D3][q];
        const f4 wa = rwa[S % D3];
        if (S + D3 < 8) {
            rwa[S % D3] = w3s[(S + D3) * 64];
#pragma unroll
            for (int q = 0; q < NQ; ++q) ring[S % D3][q] = xv[(q * 8 + S + D3) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[e % NP][q] = mfma4(wa[e], b[q][e], acc[e % NP][q]);
        __builtin_amdgcn_sched_barrier(0);
    }
    float lg = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        f4 s0 = acc[0][q];
#pragma unroll
        for (int p = 1; p < NP; ++p) s0 += acc[p][q];
        const float f0 = fold_kp(s0);
        if (q == 0 || my_rq == q) lg = f0;
    }
    return lg;
}

}  // namespace

#define PBW(i)                                                                 \
    do {                                                                       \
        if (CS_SCHEDBAR && !PROF) __builtin_amdgcn_sched_barrier(0);           \
        if (PROF) {                                                            \
            __builtin_amdgcn_sched_barrier(0);                                 \
            const unsigned now_ = (unsigned)__builtin_readcyclecounter();      \
            __builtin_amdgcn_sched_barrier(0);                                 \
            if (lane == 0 && wl == 0) prof_lds[(wave >> 2) * 24 + (i)] += now_ - prof_last; \
            prof_last = now_;                                                  \
        }                                                                      \
    } while (0)

// instrumented build: the part of an exchange up to the stamp `ts` (taken inside gather_sf, behind the sentinel wait) goes to marker i
#define PBS(i, ts)                                                             \
    do {                                                                       \
        if (PROF) {                                                            \
            if (lane == 0 && wl == 0) prof_lds[(wave >> 2) * 24 + (i)] += (ts) - prof_last; \
            prof_last = (ts);                                                  \
        }                                                                      \
    } while (0)

template <int MODE, int NQ, bool PROF>
__global__ void __launch_bounds__(CS_THREADS) loop_batch_cs_kernel(WrnnBatchArgs a) {
    typedef Lay<NQ> LM;       // mailbox regions (shared with loop_batch.hip)
    typedef LayCS<NQ> L;
    constexpr int R = L::R, NM = LM::NM, SL = L::SL;
    constexpr int DG = NQ == 1 ? 2 : 1, DS = NQ == 1 ? 4 : 2, D3 = NQ == 1 ? 2 : 1;
    // rows whose race a C wave finishes itself: RAW at 8 rows per team hands the second one (batch row wl + 4) to the S wave of its SIMD
    constexpr bool FC3_SPLIT = MODE == WRNN_MODE_MOL && CS_FC3_SPLIT;
    constexpr int NBC = (MODE == WRNN_MODE_RAW && NQ == 2 && CS_SPLIT_WIN) ? 1 : NQ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = (float *)smem;
    int *misc_i = (int *)(lds + L::L_MISC);
    float *xn = lds + L::L_XN;
    float *lgt = lds + (CS_SPREAD ? L::L_P : L::L_LG);   // CS_SPREAD: H1 holds the fc2 outputs in window 5, P (x2) is dead there (W_hh2 ends in front of B4)
    float *molnz = lds + L::L_HAND + H_NZ * SL;   // MOL: [parity][R][16] noise of the rows' samplers (the RAW nz slots are unused there)
    static_assert(2 * R * 16 <= 4 * SL, "MOL noise fits the nz slots");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool isC = wave < 4;
    const int wl = wave & 3;
    const int tl = tid & 255;                              // thread inside its role
    if (CS_PRIO && isC) __builtin_amdgcn_s_setprio(3);                // the two waves of a SIMD compete for issue slots: the serial chain goes first
    const int j = lane & 3, kp2 = (lane >> 2) & 3, rho = lane >> 4;
    const int iu = ((rho & 1) << 1) | (rho >> 1);
    const int my_rq = kp2 % NQ;
    const bool primary = kp2 < NQ;
    const int rb = 4 * my_rq + j;
    const WrnnDims d = a.d;
    const int NC = d.NC, HOP = d.HOP, T = a.T;

    // ---- team formation (loop_batch.hip) ------------
    if (tid == 0) {
        const unsigned x = xcc_idb();
        misc_i[M_DEAD] = 0;
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
        if (slot1 && rank < TB_WGS) {
            unsigned arrived = 0;
            for (unsigned spins = 0; spins < WRNN_ARRIVE_POLLS; ++spins) {
                arrived = __hip_atomic_load(&a.ctl[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (arrived >= TB_WGS) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (arrived < TB_WGS) { atomicCAS(a.err, 0u, WRNN_DEVERR_BUSY); slot1 = 0; }
        }
        misc_i[M_TEAM] = slot1 ? (int)slot1 - 1 : 1 << 20;
        misc_i[M_RANK] = (int)rank;
    }
    __syncthreads();
    const int team = __builtin_amdgcn_readfirstlane(misc_i[M_TEAM]);
    const int g = __builtin_amdgcn_readfirstlane(misc_i[M_RANK]);
    __syncthreads();
    const int n_batches = (a.n_rows + a.rpb - 1) / a.rpb;
    if (g >= TB_WGS || team >= a.n_teams || team >= n_batches) return;
    u64 *mail = a.mail + (size_t)team * WRNN_BATCH_MAIL_GRANULES;
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void *)mail, 0, (int)(WRNN_BATCH_MAIL_GRANULES * 8u), 0x00020000);

    const int unit = 16 * g + 4 * wl + iu;
    // RAW: workgroup g owns classes 32 g .. 32 g + 31 (fc3 is output-split like every other layer, the race needs an exchange).
    // MOL: fc3 has 30 rows -- every workgroup holds ALL of them (the 64 KB LDS image the RAW slice would occupy) and evaluates them
    // redundantly from the gathered fc2 outputs: no fifth exchange (round 3 had workgroup 0 evaluate them while 31 others waited for
    // its 30 granules: one more L2 round trip on the serial chain).
    const int cls0 = (MODE == WRNN_MODE_MOL ? 0 : 32 * g) + 8 * wl + iu;
    const bool wg_has_fc3 = MODE == WRNN_MODE_MOL || 32 * g < NC;
    const unsigned mb_own = ((((unsigned)my_rq * 4u + (unsigned)wl) * 8u + (unsigned)(g >> 2)) * 4u + (unsigned)iu) * 16u + (unsigned)j * 4u + (unsigned)(g & 3);
    const unsigned gvoff = (unsigned)tl * 16u;
    // compact slot index of this thread's (unit, row): duplicates (kp2 >= NQ) read their primary lane's entry (an LDS broadcast)
    const int ci = (wl * 4 + rho) * (4 * NQ) + my_rq * 4 + j;

    // ---- resident weights (the image of loop_batch.hip): C: W_ih2 r,z,n [0,96) | fc1 [256,288) | fc2 [288,320);  S: W_hh1 r,z,n
    //      [96,192) | W_hh2 r,z [192,256).  Gate n of W_hh2 and the fc3 slice are A-operand images in LDS.
    float wv[160];
    {
        // buffer loads: the wave-uniform part of every address (the weight's index) is the instruction's scalar / immediate offset.  As
        // `src[i * 64]` global loads the 160 offsets became 160 64-bit scalar constants that hipcc kept live and spilled into VGPR lanes:
        // ~590 SGPR spills = 10 of the wave's 256 VGPRs reserved as spill space for the whole kernel (round-5 ISA census).
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.batch_w + (((size_t)g * 4 + wl) * 320) * 64), 0, 320 * 64 * 4, 0x00020000);
        const unsigned wvo = (unsigned)lane * 4u;
        if (isC) {
#pragma unroll
            for (int i = 0; i < 96; ++i) wv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, wvo, (unsigned)i * 256u, 0));
#pragma unroll
            for (int i = 0; i < 64; ++i) wv[96 + i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, wvo, (unsigned)(256 + i) * 256u, 0));
        } else {
#pragma unroll
            for (int i = 0; i < 160; ++i) wv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, wvo, (unsigned)(96 + i) * 256u, 0));
        }
        const float4 *f3 = (const float4 *)(a.batch_fc3 + (size_t)(MODE == WRNN_MODE_MOL ? 0 : g) * 16384);
        float4 *dst = (float4 *)(lds + L::L_FC3);
        for (int i = tid; i < 4096; i += CS_THREADS) dst[i] = f3[i];
        const float4 *wn = (const float4 *)(a.batch_wn + (size_t)g * 8192);
        dst = (float4 *)(lds + L::L_WN);
        for (int i = tid; i < 2048; i += CS_THREADS) dst[i] = wn[i];
        for (int i = tid; i < L::L_TOTAL - L::L_HAND; i += CS_THREADS) lds[L::L_HAND + i] = 0.0f;
        if (isC && primary) {
            float *cs = lds + L::L_CST + ci;
            cs[C_A0 * SL] = a.wI0[unit]; cs[C_A1 * SL] = a.u1[unit]; cs[C_A2 * SL] = a.u1[512 + unit]; cs[C_A3 * SL] = a.u1[1024 + unit];
            cs[C_B30 * SL] = cls0 < NC ? a.w[a.off.fc3_b + cls0] : 0.0f;
            cs[C_B31 * SL] = cls0 + 4 < NC ? a.w[a.off.fc3_b + cls0 + 4] : 0.0f;
            cs[C_H1R * SL] = a.w[a.off.r1_bhh + unit]; cs[C_H1Z * SL] = a.w[a.off.r1_bhh + 512 + unit]; cs[C_H1N * SL] = a.w[a.off.r1_bhh + 1024 + unit];
            cs[C_H2R * SL] = a.w[a.off.r2_bhh + unit]; cs[C_H2Z * SL] = a.w[a.off.r2_bhh + 512 + unit]; cs[C_H2N * SL] = a.w[a.off.r2_bhh + 1024 + unit];
        }
    }
    __syncthreads();
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const lds_cf4p vP = (lds_cf4p)(size_t)launder(smem_base + (unsigned)L::L_P * 4u + (unsigned)lane * 16u);
    const lds_cf4p vQ = vP + L::VEC / 4, vH1 = vP + 2 * (L::VEC / 4);
    const lds_cf4p vF2 = CS_SPREAD ? vH1 : vP;   // where the gathered fc2 outputs are (fc3's B operand)
    const lds_cf4p w3 = (lds_cf4p)(size_t)launder(smem_base + (unsigned)L::L_FC3 * 4u + ((unsigned)(wl * 2) * 8u * 64u + (unsigned)lane) * 16u);
    const lds_cf4p wnl = (lds_cf4p)(size_t)launder(smem_base + (unsigned)L::L_WN * 4u + ((unsigned)wl * 8u * 64u + (unsigned)lane) * 16u);
    const lds_cfp cst = (lds_cfp)(size_t)launder(smem_base + (unsigned)L::L_CST * 4u + (unsigned)ci * 4u);
    typedef float __attribute__((address_space(3))) *lds_fp;
    const lds_fp hand = (lds_fp)(size_t)launder(smem_base + (unsigned)L::L_HAND * 4u + (unsigned)ci * 4u);
    const lds_f2p gdst = (lds_f2p)(size_t)launder(smem_base + (unsigned)L::L_P * 4u + ((unsigned)(tl >> 5) * 256u + 2u * (unsigned)(tl & 31)) * 4u);
    // the two values of a 16-byte load ({x, tag, z, tag}) -> two adjacent LDS words.  CS_PUT2 1: as two 4-byte stores, which hipcc merges into one
    // ds_write2_b32 that takes x and z from where the load left them; as an 8-byte vector store every value costs a v_mov into a register pair first, and a
    // VALU instruction of a C wave takes ~30 cycles while the S wave of its SIMD multiplies (round-4 probe) -- which is when these run
    typedef float __attribute__((address_space(3))) *lds_fp0;
    const unsigned gdst_addr = (unsigned)(size_t)gdst;
    // four loads of a gathered vector (one row quad: slices wl = 0..3) -> LDS, from ONE opaque base + immediate offsets
    auto put8 = [&](int fidx, const u4v *g4) {
        if (CS_PUT2) {
            const lds_fp0 q = (lds_fp0)(size_t)launder(gdst_addr + (unsigned)fidx * 4u);
#pragma unroll
            for (int k = 0; k < 4; ++k) { q[k * 64] = __uint_as_float(g4[k].x); q[k * 64 + 1] = __uint_as_float(g4[k].z); }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) gdst[(fidx + k * 64) / 2] = (f2v){__uint_as_float(g4[k].x), __uint_as_float(g4[k].z)};
        }
    };

    // S-wave meeting point (epoch of the H1 each S wave has written): LDS-address-space pointers, so that the flag is stored and polled with DS
    // instructions (round-4 advisor: through generic `volatile` pointers hipcc emitted flat_store / flat_load, which the memory model does not
    // order against the ds_write of the data).  A wave's DS instructions execute in order, and the LDS serves the waves' instructions one after
    // the other: a wave that has read the flag with a DS read reads the data behind it.
    typedef int i4v __attribute__((ext_vector_type(4)));
    typedef volatile int __attribute__((address_space(3))) *lds_vip;
    typedef volatile i4v __attribute__((address_space(3))) *lds_vi4p;
    const lds_vip sflag = (lds_vip)(size_t)(smem_base + (unsigned)(L::L_MISC + 8) * 4u);
    // C-wave meeting point of CS_SPREAD (window 4): number of (C wave, step) pairs that have finished READING H1 as fc2's B operand
    // (zeroed with the rest of the LDS tail above; 4 per step, `epoch` counts this workgroup's steps from 1)
    const lds_vup ccount = (lds_vup)(size_t)(smem_base + (unsigned)(L::L_MISC + 12) * 4u);
    bool dead = false;
    unsigned epoch = 0;
    unsigned *prof_lds = (unsigned *)(lds + L::L_PROF);
    unsigned prof_last = 0;

    for (int pass = 0; pass * a.n_teams < n_batches; ++pass) {
        const int batch = pass * a.n_teams + ((a.snake && (pass & 1)) ? a.n_teams - 1 - team : team);
        if (batch >= n_batches) continue;
        const int slot_raw = batch * a.rpb + rb;
        const bool row_ok = rb < a.rpb && slot_raw < a.n_rows;
        const int row = a.order[row_ok ? slot_raw : a.n_rows - 1];
        const WrnnRow rw = a.rows[row];
        const int64_t bsteps = a.rows[a.order[batch * a.rpb]].steps;
        if (tid < R) {
            const int s0 = batch * a.rpb + tid;
            xn[tid] = (a.x_init && tid < a.rpb && s0 < a.n_rows) ? a.x_init[a.order[s0]] : 0.0f;
        }

        if (isC) {
            // =========================================== C: the serial chain ===========================================
            float h1 = 0.0f, h2 = 0.0f, x2own = 0.0f;
            int frow[NQ];
            int fsteps[NQ];
#pragma unroll
            for (int bi = 0; bi < NQ; ++bi) {
                const int brow = wl + 4 * bi, s0 = batch * a.rpb + brow;
                const bool rok = brow < a.rpb && s0 < a.n_rows;
                frow[bi] = a.order[rok ? s0 : a.n_rows - 1];
                fsteps[bi] = rok ? a.rows[frow[bi]].steps : 0;
            }
            __syncthreads();   // S has filled the hand-over slots of step 0

            for (int64_t t = 0; t < bsteps; ++t) {
                ++epoch;
                const unsigned par = epoch & 1u;
                if (PROF) prof_last = (unsigned)__builtin_readcyclecounter();

                // ---------------- window 1: phase A (:208-212) | publish x2, h1' | gather x2 ----------------
                {
                    const float xprev = xn[rb];
                    const float xin = fmaf(cst[C_A0 * SL], xprev, hand[H_CDX * SL]);
                    const float rg = sigmoid_fast(fmaf(cst[C_A1 * SL], xprev, hand[H_CDY * SL]) + hand[H_GH1R * SL]);
                    const float zg = sigmoid_fast(fmaf(cst[C_A2 * SL], xprev, hand[H_CDZ * SL]) + hand[H_GH1Z * SL]);
                    const float ng = tanh_fast(fmaf(cst[C_A3 * SL], xprev, hand[H_CDW * SL]) + rg * hand[H_GH1N * SL]);
                    h1 = (1.0f - zg) * ng + zg * h1;
                    x2own = xin + h1;
                    if (primary) {
                        st_granule(mail, LM::G_X2 + par * LM::RG + mb_own, epoch, __float_as_uint(x2own));
                        st_granule(mail, LM::G_H1 + par * LM::RG + mb_own, epoch, __float_as_uint(h1));
                    }
                }
                PBW(0);
                {
                    u4v gx[1][NM];
                    GSF_DECL;
                    gather_sf<NM>(mrs, gvoff, (LM::G_X2 + par * LM::RG) * 8u, epoch, gx, dead, a.err, 21u GSF_TS);
                    GSF_ACC(17);
                    PBW(1);
#pragma unroll
                    for (int h = 0; h < NM / 4; ++h) put8(0 * L::VEC + h * 2048, &gx[0][4 * h]);
                }
                PBW(2);
                __syncthreads();   // B1
                PBW(3);

                // ---------------- window 2: phase B (GRU2, :213-216) | publish x3 | gather x3 ----------------
                {
                    f4 acc[3][NQ];
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) acc[gt][q] = (f4){0.f, 0.f, 0.f, 0.f};
                    mfma_gates<NQ, 3, false, DG>(wv, vP, acc, NoMid());
                    PBW(4);
                    float tr = 0.f, tz = 0.f, tn = 0.f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float fr = fold_kp(acc[0][q]), fz = fold_kp(acc[1][q]), fn = fold_kp(acc[2][q]);
                        if (q == 0 || my_rq == q) { tr = fr; tz = fz; tn = fn; }
                    }
                    PBW(5);
                    const float rg = sigmoid_fast((tr + hand[H_C2R * SL]) + hand[H_GH2R * SL]);
                    const float zg = sigmoid_fast((tz + hand[H_C2Z * SL]) + hand[H_GH2Z * SL]);
                    const float ng = tanh_fast((tn + hand[H_C2N * SL]) + rg * hand[H_GH2N * SL]);
                    h2 = (1.0f - zg) * ng + zg * h2;
                    const float x3 = x2own + h2;
                    if (primary) st_granule(mail, LM::G_X3 + par * LM::RG + mb_own, epoch, __float_as_uint(x3));
                }
                PBW(6);
                {
                    u4v gx[1][NM];
                    GSF_DECL;
                    gather_sf<NM>(mrs, gvoff, (LM::G_X3 + par * LM::RG) * 8u, epoch, gx, dead, a.err, 23u GSF_TS);
                    GSF_ACC(7);
#pragma unroll
                    for (int h = 0; h < NM / 4; ++h) put8(1 * L::VEC + h * 2048, &gx[0][4 * h]);
                }
                PBW(9);
                __syncthreads();   // B2
                PBW(10);

                // ---------------- window 3: fc1 (:217-218) | publish | gather ----------------
                {
                    f4 sum[NQ];
                    mfma_single<NQ, (NQ == 1 ? 4 : 2), false, DS>(wv + 96, vQ, sum);
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float f = fold_kp(sum[q]);
                        if (q == 0 || my_rq == q) s = f;
                    }
                    if (primary) st_granule(mail, LM::G_F1 + par * LM::RG + mb_own, epoch, __float_as_uint(fmaxf(s + hand[H_C3 * SL], 0.0f)));
                }
                PBW(11);
                {
                    u4v gx[1][NM];
                    GSF_DECL;
                    gather_sf<NM>(mrs, gvoff, (LM::G_F1 + par * LM::RG) * 8u, epoch, gx, dead, a.err, 24u GSF_TS);
                    GSF_ACC(12);
#pragma unroll
                    for (int h = 0; h < NM / 4; ++h) put8(2 * L::VEC + h * 2048, &gx[0][4 * h]);
                }
                PBW(13);
                __syncthreads();   // B3
                PBW(14);

                // ---------------- window 4: fc2 (:220-221) | publish | gather ----------------
                {
                    f4 sum[NQ];
                    mfma_single<NQ, (NQ == 1 ? 4 : 2), false, DS>(wv + 128, vH1, sum);
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float f = fold_kp(sum[q]);
                        if (q == 0 || my_rq == q) s = f;
                    }
                    if (primary) st_granule(mail, LM::G_F2 + par * LM::RG + mb_own, epoch, __float_as_uint(fmaxf(s + hand[H_C4 * SL], 0.0f)));
                    // CS_SPREAD gathers the fc2 outputs INTO H1, the vector all four C waves have just streamed as fc2's B operand.  "H1 is dead after
                    // fc2" holds per wave, not per workgroup, and no barrier separates a sibling's reads from this wave's landing: a wave's look
                    // covers the granules of workgroups 8 w .. 8 w + 7 only, so for three of the four waves the fresh tags say nothing about the
                    // siblings in their OWN workgroup (round-5 advisor: the 7 M-step parity run passed on timing -- an exchange takes ~2 us, the
                    // inter-wave skew is a fraction of that).  INVARIANT: a C wave adds 1 to `ccount` (ds_add) behind its fc2 publish -- its B-operand
                    // reads of H1 fed the MFMAs whose fold it has just published, DS instructions of a wave execute in order, and the st_granule
                    // asm is a compiler barrier -- and lands fc2 outputs in H1 only after reading ccount == 4 * epoch (no sibling can be a step
                    // ahead: barriers B4 / B5 are in between).
                    if (CS_SPREAD && CS_SPREAD_MEET && lane == 0)
                        __hip_atomic_fetch_add((unsigned __attribute__((address_space(3))) *)ccount, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                PBW(15);
                {
                    u4v gx[1][NM];
                    GSF_DECL;
                    if (CS_SPREAD && CS_SPREAD_MEET) {
                        unsigned cc = 0;
#if CS_PROF_SPLIT
                        gather_sf<NM>(mrs, gvoff, (LM::G_F2 + par * LM::RG) * 8u, epoch, gx, dead, a.err, 25u, PROF ? &ts_ : nullptr, ccount, &cc);
#else
                        gather_sf<NM>(mrs, gvoff, (LM::G_F2 + par * LM::RG) * 8u, epoch, gx, dead, a.err, 25u, nullptr, ccount, &cc);
#endif
                        unsigned sp = 0;
                        while (__builtin_amdgcn_readfirstlane(cc) != 4u * epoch && !dead) {   // normally false at once: the counter was read under the data look
                            if (++sp > 200000u) { dead = true; if (lane == 0) atomicExch(a.err, 30u); break; }
                            __builtin_amdgcn_s_sleep(1);
                            cc = *ccount;
                        }
                        asm volatile("" ::: "memory");
                    } else {
                        gather_sf<NM>(mrs, gvoff, (LM::G_F2 + par * LM::RG) * 8u, epoch, gx, dead, a.err, 25u GSF_TS);
                    }
                    GSF_ACC(16);
#pragma unroll
                    for (int h = 0; h < NM / 4; ++h) put8((CS_SPREAD ? 2 : 0) * L::VEC + h * 2048, &gx[0][4 * h]);
                }
                PBW(18);
                __syncthreads();   // B4
                PBW(19);

                // ---------------- window 5: fc3 (:223) + sampler (:225-237) ----------------
                {
                    float lg0 = 0.f, lg1 = 0.f;
                    if (FC3_SPLIT) {
                        // MOL: the 8 sets of four fc3 rows are shared between the two waves of a SIMD -- this wave evaluates classes 8 wl + iu,
                        // the S wave 8 wl + iu + 4 (the fc3 image is in LDS, so either wave can): 32 MFMAs each, side by side, instead of 64 here
                        lg0 = fc3_one_set<NQ, D3>(w3, vF2, my_rq) + cst[C_B30 * SL];
                        if (a.logits_out && primary && row_ok && t < rw.steps && g == 0 && cls0 < NC)
                            a.logits_out[((size_t)t * a.n_rows + row) * NC + cls0] = lg0;
                    } else if (wg_has_fc3) {
                        constexpr int NP = NQ == 1 ? 2 : 1;
                        f4 acc[2][NP][NQ];
#pragma unroll
                        for (int st = 0; st < 2; ++st)
#pragma unroll
                            for (int p = 0; p < NP; ++p)
#pragma unroll
                                for (int q = 0; q < NQ; ++q) acc[st][p][q] = (f4){0.f, 0.f, 0.f, 0.f};
                        f4 ring[D3][NQ], rwa[D3], rwb[D3];
#pragma unroll
                        for (int dd = 0; dd < D3; ++dd) {
                            rwa[dd] = w3[dd * 64]; rwb[dd] = w3[(8 + dd) * 64];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) ring[dd][q] = vF2[(q * 8 + dd) * 64];
                        }
#pragma unroll
                        for (int S = 0; S < 8; ++S) {
                            f4 b[NQ];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) b[q] = ring[S % D3][q];
                            const f4 wa = rwa[S % D3], wb = rwb[S % D3];
                            if (S + D3 < 8) {
                                rwa[S % D3] = w3[(S + D3) * 64]; rwb[S % D3] = w3[(8 + S + D3) * 64];
#pragma unroll
                                for (int q = 0; q < NQ; ++q) ring[S % D3][q] = vF2[(q * 8 + S + D3) * 64];
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
#pragma unroll
                                for (int q = 0; q < NQ; ++q) {
                                    acc[0][e % NP][q] = mfma4(wa[e], b[q][e], acc[0][e % NP][q]);
                                    acc[1][e % NP][q] = mfma4(wb[e], b[q][e], acc[1][e % NP][q]);
                                }
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            f4 s0 = acc[0][0][q], s1 = acc[1][0][q];
#pragma unroll
                            for (int p = 1; p < NP; ++p) { s0 += acc[0][p][q]; s1 += acc[1][p][q]; }
                            const float f0 = fold_kp(s0), f1 = fold_kp(s1);
                            if (q == 0 || my_rq == q) { lg0 = f0; lg1 = f1; }
                        }
                        lg0 += cst[C_B30 * SL]; lg1 += cst[C_B31 * SL];
                        if (a.logits_out && primary && row_ok && t < rw.steps && (MODE != WRNN_MODE_MOL || g == 0)) {
                            float *lo = a.logits_out + ((size_t)t * a.n_rows + row) * NC;
                            if (cls0 < NC) lo[cls0] = lg0;
                            if (cls0 + 4 < NC) lo[cls0 + 4] = lg1;
                        }
                    }
                    if (MODE == WRNN_MODE_RAW) {
                        const lds_fp hz = hand + (H_NZ + 2 * par) * SL;
                        const float nz0 = hz[0], nz1 = hz[SL];
                        float v = cls0 < NC ? lg0 + nz0 : -INFINITY;
                        int k = cls0;
                        const float v1 = cls0 + 4 < NC ? lg1 + nz1 : -INFINITY;
                        if (v1 > v) { v = v1; k = cls0 + 4; }
                        {
                            const u2v pv = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                            const u2v pk = __builtin_amdgcn_permlane32_swap((unsigned)k, (unsigned)k, false, false);
                            const float va = __uint_as_float(pv.x), vb = __uint_as_float(pv.y);
                            const int ka = (int)pk.x, kb = (int)pk.y;
                            const bool tb = vb > va || (vb == va && kb < ka);
                            v = tb ? vb : va; k = tb ? kb : ka;
                        }
                        {
                            const u2v pv = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                            const u2v pk = __builtin_amdgcn_permlane16_swap((unsigned)k, (unsigned)k, false, false);
                            const float va = __uint_as_float(pv.x), vb = __uint_as_float(pv.y);
                            const int ka = (int)pk.x, kb = (int)pk.y;
                            const bool tb = vb > va || (vb == va && kb < ka);
                            v = tb ? vb : va; k = tb ? kb : ka;
                        }
                        if (primary && rho == 0)
                            st_granule(mail, LM::G_PR + par * LM::PRG + (unsigned)rb * 128u + (unsigned)(g * 4 + wl),
                                       (epoch << 10) | (unsigned)(k & 1023), __float_as_uint(v));
                    } else {
                        // MOL: the wave's 8 fc3 outputs of every batch row -> LDS; wave w samples batch rows w, w + 4 behind the barrier
                        if (primary) { lgt[rb * 32 + cls0] = lg0; if (!FC3_SPLIT) lgt[rb * 32 + cls0 + 4] = lg1; }
                    }
                }
                PBW(20);
                if (MODE == WRNN_MODE_MOL) __syncthreads();   // B4b
                u4v gqa[NBC];
                if (MODE == WRNN_MODE_RAW) {
                    const unsigned tg = epoch & 0x3fffffu;
                    unsigned spins = 0;
                    for (;;) {
#pragma unroll
                        for (int i = 0; i < NBC; ++i) gqa[i] = ld_pair(mrs, (unsigned)lane * 16u, (LM::G_PR + par * LM::PRG + (unsigned)(wl + 4 * i) * 128u) * 8u);
                        bool ok = true;
#pragma unroll
                        for (int i = 0; i < NBC; ++i) ok = ok && (gqa[i].y >> 10) == tg && (gqa[i].w >> 10) == tg;
                        if (__all(ok) || dead) break;
                        if (++spins > TB_SPIN_MAX) { dead = true; if (lane == 0) atomicExch(a.err, 26u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                PBW(21);
                // The teacher-forced value (forward(): x_forced) is fetched HERE, behind the poll's own wait, and the output stores of
                // workgroup 0 go out after the LAST row's value is in LDS: with the load and the stores inside the per-row code the
                // compiler had to put an `s_waitcnt vmcnt(0)` in front of the second row -- workgroup 0, the one every other workgroup
                // waits for at the next exchange, sat there until the first row's global stores were acknowledged (~500 cycles per row).
                float xfv[NBC], xnv[NBC];
                int labv[NBC];
#pragma unroll
                for (int bi = 0; bi < NBC; ++bi) xfv[bi] = a.x_forced ? a.x_forced[(size_t)t * a.n_rows + frow[bi]] : 0.0f;
#pragma unroll
                for (int bi = 0; bi < NBC; ++bi) asm volatile("" : "+v"(xfv[bi]));   // waited for here, once
#pragma unroll
                for (int bi = 0; bi < NBC; ++bi) {
                    const int brow = wl + 4 * bi;
                    float x_new;
                    int lab;
                    if (MODE == WRNN_MODE_RAW) {
                        const u4v gq = gqa[bi];
                        const float va = __uint_as_float(gq.x), vb = __uint_as_float(gq.z);
                        const bool pb = vb > va;
                        const float best = pb ? vb : va;
                        const int besti = (int)((pb ? gq.w : gq.y) & 1023u);
                        const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_max_b(best)), 63));
                        const u64 ball = __ballot(best == mx);
                        const int src = (int)__builtin_ctzll(ball ? ball : 1ull);
                        lab = __builtin_amdgcn_readlane(besti, src);
                        x_new = 2.0f * (float)lab / ((float)NC - 1.0f) - 1.0f;   // (:235)
                    } else {
                        // sample_from_discretized_mix_logistic (distribution.py:87-123) for batch row `brow`
                        const int nr = NC / 3;
                        // the Gumbel / logistic noise of (step, row) was prepared by the S wave of this SIMD (noise_step)
                        const float nzv = lane <= nr ? molnz[((int)par * R + brow) * 16 + lane] : 0.0f;
                        const float mylg = lgt[brow * 32 + (lane < NC ? lane : 0)];
                        const float v = lane < nr ? mylg + nzv : -INFINITY;
                        const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_max_b(v)), 63));
                        const u64 ball = __ballot(v == mx);
                        const int km = (int)__builtin_ctzll(ball ? ball : 1ull);
                        const float mean = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mylg), nr + km));
                        const float ls = fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(mylg), 2 * nr + km)), -32.23619130191664f);
                        const float nlog = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nzv), nr));
                        x_new = fminf(fmaxf(mean + expf(ls) * nlog, -1.0f), 1.0f);
                        lab = km;
                    }
                    xnv[bi] = x_new; labv[bi] = lab;
                    if (lane == 0) xn[brow] = a.x_forced ? xfv[bi] : x_new;   // (:237)
                }
                if (lane == 0 && g == 0) {
#pragma unroll
                    for (int bi = 0; bi < NBC; ++bi) {
                        if (t < fsteps[bi]) {   // a real row that has not reached its own length (ragged batch)
                            if (a.labels_out) a.labels_out[(size_t)frow[bi] * a.steps + t] = labv[bi];
                            a.samples_out[(size_t)frow[bi] * a.steps + t] = xnv[bi];
                        }
                    }
                }
                PBW(22);
                __syncthreads();   // B5
                PBW(23);
                if ((t & 63) == 63) {
                    if (dead && lane == 0) misc_i[M_DEAD] = 1;
                    __syncthreads();
                    if (misc_i[M_DEAD]) return;
                }
            }
        } else {
            // =========================================== S: everything that never waits on x_t ===========================================
            const float *recb = a.tabREC32 + (size_t)rw.utt * (T + 1) * 512 * 32 + (size_t)unit * 32;
            const float *ktab = a.w + a.off.ktab;
            float pz0 = 0.f, pz1 = 0.f;                 // raw bits of the odd step of the sampler's Philox block
            int nfi = (int)(rw.start / HOP), nph = (int)(rw.start - (int64_t)nfi * HOP);
            int cst_frame = -1000000;
            // conditioning {cI, v_r, v_z, v_n} of step ts for (unit, row) -> cd slots; per-frame constants -> their slots when the frame
            // changed.  The record is read where it is used (L1-resident: 128 bytes per (frame, unit)); nothing is carried in registers.
            // per-frame constants (c2 r, z, n, c3, c4) of a new frame: read with the conditioning, written to their slots by frame_flush() -- at
            // once, or (CS_COND_W4: the conditioning runs in window 4, where C still reads this frame's c4) behind barrier B4
            float4 pc2 = make_float4(0.f, 0.f, 0.f, 0.f);
            float pc4 = 0.0f;
            bool frame_pending = false;
            auto frame_flush = [&]() {
                if (frame_pending) {
                    if (primary) { hand[H_C2R * SL] = pc2.x; hand[H_C2Z * SL] = pc2.y; hand[H_C2N * SL] = pc2.z; hand[H_C3 * SL] = pc2.w; hand[H_C4 * SL] = pc4; }
                    frame_pending = false;
                }
            };
            auto cond_step = [&](int64_t ts) {
                const int64_t pos = rw.start + ts;
                const bool live = pos < a.total_len;
                const int fi = live ? nfi : T;
                const int ph = live ? nph : 0;
                if (++nph == HOP) { nph = 0; ++nfi; }
                const float4 *r = (const float4 *)(recb + (size_t)fi * 512 * 32);
                const float4 ra0 = r[0], ra1 = r[1], ra2 = r[2], ra3 = r[3], ra4 = r[4], ra5 = r[5];
                const float *kt = ktab + ph * 5;
                const float rk0 = kt[0], rk1 = kt[1], rk2 = kt[2], rk3 = kt[3], rk4 = kt[4];
                const float cx = fmaf(rk4, ra2.x, fmaf(rk3, ra1.w, fmaf(rk2, ra1.z, fmaf(rk1, ra1.y, fmaf(rk0, ra1.x, ra0.x)))));
                const float cy = fmaf(rk4, ra5.y, fmaf(rk3, ra4.z, fmaf(rk2, ra3.w, fmaf(rk1, ra3.x, fmaf(rk0, ra2.y, ra0.y)))));
                const float cz = fmaf(rk4, ra5.z, fmaf(rk3, ra4.w, fmaf(rk2, ra4.x, fmaf(rk1, ra3.y, fmaf(rk0, ra2.z, ra0.z)))));
                const float cw = fmaf(rk4, ra5.w, fmaf(rk3, ra5.x, fmaf(rk2, ra4.y, fmaf(rk1, ra3.z, fmaf(rk0, ra2.w, ra0.w)))));
                if (primary) { hand[H_CDX * SL] = cx; hand[H_CDY * SL] = cy; hand[H_CDZ * SL] = cz; hand[H_CDW * SL] = cw; }
                if (fi != cst_frame) {
                    pc2 = r[6];
                    pc4 = recb[(size_t)fi * 512 * 32 + 28];
                    cst_frame = fi;
                    frame_pending = true;
                    if (!CS_COND_W4) frame_flush();
                }
            };
            // -log q of this thread's two classes for step ts (RAW) -> nz slots of parity `np` (see loop_batch.hip for the Philox block)
            int frowS[NQ], fstepsS[NQ];   // rows whose sampler the C wave of this SIMD runs (MOL: their noise is prepared here; RAW R = 8: row wl + 4 finished here)
#pragma unroll
            for (int bi = 0; bi < NQ; ++bi) {
                const int brow = wl + 4 * bi, s0 = batch * a.rpb + brow;
                const bool rok = brow < a.rpb && s0 < a.n_rows;
                frowS[bi] = a.order[rok ? s0 : a.n_rows - 1];
                fstepsS[bi] = rok ? a.rows[frowS[bi]].steps : 0;
            }
            auto noise_step = [&](int64_t ts, unsigned np) {
                if (MODE != WRNN_MODE_RAW) {
                    // sample_from_discretized_mix_logistic (distribution.py:106-121): 10 Gumbel draws (mixture pick) + 1 logistic draw per row
                    const int nr = NC / 3;
#pragma unroll
                    for (int bi = 0; bi < NQ; ++bi) {
                        const int brow = wl + 4 * bi, rrow = frowS[bi];
                        if (lane <= nr) {
                            float u;
                            if (a.noise_mode == WRNN_NOISE_INJECTED)
                                u = lane < nr ? a.noise1[((size_t)ts * a.n_rows + rrow) * nr + lane] : a.noise2[(size_t)ts * a.n_rows + rrow];
                            else
                                u = 1e-5f + wrnn_uniform(a.seed, (uint64_t)ts, (uint32_t)rrow, (uint32_t)lane) * (1.0f - 2e-5f);
                            molnz[((int)np * R + brow) * 16 + lane] = lane < nr ? -logf(-logf(u)) : logf(u) - logf(1.0f - u);
                        }
                    }
                    return;
                }
                float nz0, nz1;
                if (a.noise_mode == WRNN_NOISE_INJECTED) {
                    const float *qp = a.noise1 + ((size_t)ts * a.n_rows + row) * NC;
                    nz0 = cls0 < NC ? -logf(qp[cls0]) : 0.0f;
                    nz1 = cls0 + 4 < NC ? -logf(qp[cls0 + 4]) : 0.0f;
                } else if (a.noise_mode == WRNN_NOISE_PHILOX) {
                    const bool upper = lane >= 32;
                    unsigned ba, bb;
                    if ((ts & 1) == 0) {
                        const Philox4 pb = wrnn_raw_block(a.seed, (uint64_t)ts, (uint32_t)row, (uint32_t)(upper ? cls0 + 4 : cls0));
                        ba = pb.x; bb = pb.y;
                        pz0 = __uint_as_float(pb.z); pz1 = __uint_as_float(pb.w);
                    } else { ba = __float_as_uint(pz0); bb = __float_as_uint(pz1); }
                    const float ge = -__logf(-logf(u01_from_bits(ba))), go = -__logf(-logf(u01_from_bits(bb)));
                    const float mine_e = upper ? go : ge, give_e = upper ? ge : go;
                    const u2v se = __builtin_amdgcn_permlane32_swap(__float_as_uint(give_e), __float_as_uint(give_e), false, false);
                    const float recv_e = __uint_as_float(upper ? se.x : se.y);
                    nz0 = upper ? recv_e : mine_e; nz1 = upper ? mine_e : recv_e;
                } else { nz0 = 0.f; nz1 = 0.f; }
                if (primary) { hand[(H_NZ + 2 * np) * SL] = nz0; hand[(H_NZ + 2 * np + 1) * SL] = nz1; }
            };
            // step 0: h1 = h2 = 0 (:194-196) => gh1 = b_hh1, gh2 = b_hh2
            if (primary) {
                hand[H_GH1R * SL] = cst[C_H1R * SL]; hand[H_GH1Z * SL] = cst[C_H1Z * SL]; hand[H_GH1N * SL] = cst[C_H1N * SL];
                hand[H_GH2R * SL] = cst[C_H2R * SL]; hand[H_GH2Z * SL] = cst[C_H2Z * SL]; hand[H_GH2N * SL] = cst[C_H2N * SL];
            }
            cond_step(0);
            frame_flush();
            __syncthreads();

            for (int64_t t = 0; t < bsteps; ++t) {
                ++epoch;
                const unsigned par = epoch & 1u;
                if (PROF) prof_last = (unsigned)__builtin_readcyclecounter();

                // ---------------- window 1: the sampler's noise of THIS step, while the C waves wait for x2 (an S wave has nothing else to do
                // before B1; in window 4, behind the W_hh2 fold, it made the S waves late at B4) ----------------
                noise_step(t, par);
                PBW(2);
                __syncthreads();   // B1
                PBW(3);
                {
                    if (CS_SPRIO) __builtin_amdgcn_s_setprio(3);   // the gather + meeting point at the C waves' priority
                    // h1' is not needed before this wave's own W_hh1 product: fetched HERE, behind B1, the S waves are never the last to reach B1
                    // (they were: their 32 KB look ran beside the C waves' x2 look through the same 64 B/clk port) and the x2 look has the port
                    // to itself.  The data was published a whole window ago: one look, no sentinel.  The four S waves then meet through LDS
                    // flags (s_barrier would need the C waves).
                    u4v gx[1][NM];
                    const unsigned offs[1] = {(LM::G_H1 + par * LM::RG) * 8u};
                    gather_vecs<NM, 1, false>(mrs, gvoff, offs, epoch, gx, dead, a.err, 22u);
                    PBW(1);
#pragma unroll
                    for (int h = 0; h < NM / 4; ++h) put8(2 * L::VEC + h * 2048, &gx[0][4 * h]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's part of H1 has left the DS queue
                    if (lane == 0) sflag[wl] = (int)epoch;
                    PBW(5);   // h1' written to LDS
                    unsigned sp = 0;
                    for (;;) {
                        const i4v f = *(lds_vi4p)sflag;
                        if ((f.x == (int)epoch && f.y == (int)epoch && f.z == (int)epoch && f.w == (int)epoch) || dead) break;
                        if (++sp > 200000u) { dead = true; if (lane == 0) atomicExch(a.err, 29u); break; }   // a lost S wave: reported, not multiplied through
                        if (CS_FLAG_POLL_SLEEP) __builtin_amdgcn_s_sleep(CS_FLAG_POLL_SLEEP);
                    }
                    asm volatile("" ::: "memory");
                    PBW(6);   // the four S waves have met
                    if (CS_SPRIO) __builtin_amdgcn_s_setprio(0);
                }

                // ---------------- windows 2 - 4: gh1' = W_hh1 . h1' + b_hh1 and gh2' = W_hh2 . (x3 - x2) + b_hh2 of the next step ----------------
                // The MFMAs of a product run in "its" window (they read H1 resp. Q / P, which the C waves overwrite at the end of the next
                // one); its FOLD -- pure register work, ~700 cycles at 8 rows -- runs behind the barrier, at the start of the next window:
                // with the fold in front of B2 the S waves were the last to arrive there (B2 wait of the C waves 390 cycles + a stretched
                // x3 exchange), and the same at B3.  The hand-over slots are read by C a whole step later.
                f4 acc1[3][NQ], acc2[3][NQ];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) { acc1[gt][q] = (f4){0.f, 0.f, 0.f, 0.f}; acc2[gt][q] = (f4){0.f, 0.f, 0.f, 0.f}; }
                auto fold1 = [&]() {
                    float gr = 0.f, gz = 0.f, gn = 0.f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float fr = fold_kp(acc1[0][q]), fz = fold_kp(acc1[1][q]), fn = fold_kp(acc1[2][q]);
                        if (q == 0 || my_rq == q) { gr = fr + cst[C_H1R * SL]; gz = fz + cst[C_H1Z * SL]; gn = fn + cst[C_H1N * SL]; }
                    }
                    if (primary) { hand[H_GH1R * SL] = gr; hand[H_GH1Z * SL] = gz; hand[H_GH1N * SL] = gn; }
                };
                auto fold2 = [&]() {
                    float gr = 0.f, gz = 0.f, gn = 0.f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float fr = fold_kp(acc2[0][q]), fz = fold_kp(acc2[1][q]), fn = fold_kp(acc2[2][q]);
                        if (q == 0 || my_rq == q) { gr = fr + cst[C_H2R * SL]; gz = fz + cst[C_H2Z * SL]; gn = fn + cst[C_H2N * SL]; }
                    }
                    if (primary) { hand[H_GH2R * SL] = gr; hand[H_GH2Z * SL] = gz; hand[H_GH2N * SL] = gn; }
                };
                auto whh2 = [&]() {
                    if (CS_DIAG & 2) return;
                    f4 xq[NQ], xp[NQ], wn = wnl[0];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) { xq[q] = vQ[(q * 8) * 64]; xp[q] = vP[(q * 8) * 64]; }
#pragma unroll
                    for (int S = 0; S < 8; ++S) {
                        f4 b[NQ];
#pragma unroll
                        for (int q = 0; q < NQ; ++q) b[q] = xq[q] - xp[q];
                        const f4 wcur = wn;
                        if (S < 7) {
                            wn = wnl[(S + 1) * 64];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) { xq[q] = vQ[(q * 8 + S + 1) * 64]; xp[q] = vP[(q * 8 + S + 1) * 64]; }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float wr = wv[96 + 4 * S + e], wz = wv[128 + 4 * S + e];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) {
                                acc2[0][q] = mfma4(wr, b[q][e], acc2[0][q]);
                                acc2[1][q] = mfma4(wz, b[q][e], acc2[1][q]);
                                acc2[2][q] = mfma4(wcur[e], b[q][e], acc2[2][q]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if (!(CS_DIAG & 1)) mfma_gates<NQ, 3, false, (NQ == 1 ? 2 : 1)>(wv, vH1, acc1, NoMid());
                PBW(7);
                PBW(8);
                __syncthreads();   // B2
                PBW(10);
                fold1();
                if (!CS_SPREAD) whh2();
                PBW(12);
                __syncthreads();   // B3
                PBW(14);
                if (!CS_SPREAD) fold2();

                // ---------------- window 4 (MOL: the conditioning of the next step) ----------------
                if (CS_SPREAD) { whh2(); fold2(); }   // x2 (P) and x3 (Q) are intact: the fc2 outputs of this window go to H1
                if (CS_COND_W4 && t + 1 < bsteps) cond_step(t + 1);   // the cd slots are read in phase A of the next step only
                PBW(16);
                __syncthreads();   // B4
                PBW(19);

                // ---------------- window 5: conditioning of the next step ----------------
                // (MOL: in front of B4b -- behind it the C waves only sample, 800 cycles, and then waited 775 at B5 for this)
                if (CS_COND_W4) frame_flush(); else if (t + 1 < bsteps) cond_step(t + 1);
                if (FC3_SPLIT) {   // this wave's half of fc3 (see the C waves' window 5)
                    const float lg1 = fc3_one_set<NQ, D3>(w3 + 8 * 64, vF2, my_rq) + cst[C_B31 * SL];
                    if (primary) lgt[rb * 32 + cls0 + 4] = lg1;
                    if (a.logits_out && primary && row_ok && t < rw.steps && g == 0 && cls0 + 4 < NC)
                        a.logits_out[((size_t)t * a.n_rows + row) * NC + cls0 + 4] = lg1;
                }
                PBW(17);
                if (MODE == WRNN_MODE_MOL) __syncthreads();   // B4b: C's fc3 outputs of all rows are in LDS
                if (NBC < NQ) {
                    // exchange 5 for batch row wl + 4 (RAW, 8 rows per team): the C wave of this SIMD finishes row wl meanwhile.  One row per
                    // wave instead of two one after the other in the four C waves (1 130 -> ~600 cycles at the end of the serial chain).
                    const int brow = wl + 4;
                    const unsigned tg = epoch & 0x3fffffu;
                    u4v gq;
                    unsigned spins = 0;
                    for (;;) {
                        gq = ld_pair(mrs, (unsigned)lane * 16u, (LM::G_PR + par * LM::PRG + (unsigned)brow * 128u) * 8u);
                        if (__all((gq.y >> 10) == tg && (gq.w >> 10) == tg) || dead) break;
                        if (++spins > TB_SPIN_MAX) { dead = true; if (lane == 0) atomicExch(a.err, 28u); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    float xf = a.x_forced ? a.x_forced[(size_t)t * a.n_rows + frowS[1]] : 0.0f;
                    asm volatile("" : "+v"(xf));
                    const float va = __uint_as_float(gq.x), vb = __uint_as_float(gq.z);
                    const bool pb = vb > va;   // equal scores: the lower slot = the lower class range wins
                    const float best = pb ? vb : va;
                    const int besti = (int)((pb ? gq.w : gq.y) & 1023u);
                    const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_max_b(best)), 63));
                    const u64 ball = __ballot(best == mx);
                    const int src = (int)__builtin_ctzll(ball ? ball : 1ull);
                    const int lab = __builtin_amdgcn_readlane(besti, src);
                    const float x_new = 2.0f * (float)lab / ((float)NC - 1.0f) - 1.0f;   // (:235)
                    if (lane == 0) {
                        xn[brow] = a.x_forced ? xf : x_new;   // (:237)
                        if (g == 0 && t < fstepsS[1]) {
                            if (a.labels_out) a.labels_out[(size_t)frowS[1] * a.steps + t] = lab;
                            a.samples_out[(size_t)frowS[1] * a.steps + t] = x_new;
                        }
                    }
                }
                __syncthreads();   // B5
                PBW(23);
                if ((t & 63) == 63) {
                    if (dead && lane == 0) misc_i[M_DEAD] = 1;
                    __syncthreads();
                    if (misc_i[M_DEAD]) return;
                }
            }
        }
        __syncthreads();
    }
    if (PROF && a.prof && lane == 0 && wl == 0 && g == CS_PROF_WG && team == 0) {   // reported as "wave 0" (C) and "wave 4" (S)
        for (int i = 0; i < 24; ++i) a.prof[wave * WRNN_PROF_SLOTS + i] += prof_lds[(wave >> 2) * 24 + i];
    }
}

template <int MODE, int NQ>
static hipError_t launch_cs(const WrnnBatchArgs &a, hipStream_t s) {
    const size_t lds = (size_t)LayCS<NQ>::L_TOTAL * sizeof(float);
    hipError_t e;
    if (a.prof) {
        e = hipFuncSetAttribute((const void *)loop_batch_cs_kernel<MODE, NQ, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((loop_batch_cs_kernel<MODE, NQ, true>), dim3(a.n_teams * TB_WGS), dim3(CS_THREADS), lds, s, a);
    } else {
        e = hipFuncSetAttribute((const void *)loop_batch_cs_kernel<MODE, NQ, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((loop_batch_cs_kernel<MODE, NQ, false>), dim3(a.n_teams * TB_WGS), dim3(CS_THREADS), lds, s, a);
    }
    return hipGetLastError();
}

// row quads per team this file is built for (a.nq = 1: 4 rows per team)
int wrnn_batch_cs_max_nq(int mode) { (void)mode; return CS_MAX_NQ; }

hipError_t wrnn_launch_loop_batch_cs(const WrnnBatchArgs &a, hipStream_t s) {
    (void)hipGetLastError();
    if (a.nq < 1 || a.nq > CS_MAX_NQ) return hipErrorInvalidValue;
#if CS_MAX_NQ >= 2
    if (a.nq == 2) return a.d.mode == WRNN_MODE_RAW ? launch_cs<WRNN_MODE_RAW, 2>(a, s) : launch_cs<WRNN_MODE_MOL, 2>(a, s);
#endif
    if (a.d.mode == WRNN_MODE_RAW) return launch_cs<WRNN_MODE_RAW, 1>(a, s);
    return launch_cs<WRNN_MODE_MOL, 1>(a, s);
}

template <int MODE, int NQ>
static hipError_t occ_cs(bool prof, int *blocks_per_cu, size_t *lds_bytes) {
    const size_t lds = (size_t)LayCS<NQ>::L_TOTAL * sizeof(float);
    *lds_bytes = lds;
    const void *fn = prof ? (const void *)loop_batch_cs_kernel<MODE, NQ, true> : (const void *)loop_batch_cs_kernel<MODE, NQ, false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, CS_THREADS, lds);
}
hipError_t wrnn_batch_cs_occupancy(int mode, int nq, bool prof, int *blocks_per_cu, size_t *lds_bytes) {
    if (nq < 1 || nq > CS_MAX_NQ) return hipErrorInvalidValue;
#if CS_MAX_NQ >= 2
    if (nq == 2) return mode == WRNN_MODE_RAW ? occ_cs<WRNN_MODE_RAW, 2>(prof, blocks_per_cu, lds_bytes) : occ_cs<WRNN_MODE_MOL, 2>(prof, blocks_per_cu, lds_bytes);
#endif
    if (mode == WRNN_MODE_RAW) return occ_cs<WRNN_MODE_RAW, 1>(prof, blocks_per_cu, lds_bytes);
    return occ_cs<WRNN_MODE_MOL, 1>(prof, blocks_per_cu, lds_bytes);
}
