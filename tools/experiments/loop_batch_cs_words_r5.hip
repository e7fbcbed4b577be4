// ARCHIVED EXPERIMENT -- not built, not part of the product: the round-5 exchange experiment of the batch kernel (untagged words, {x2, h1'} pairs, yield tokens, S-side emptying): parity-green, 9-19 % slower than the shipped kernel.
// Measurements: profiles/r05_batch_cs_experiments.txt; why it is kept: DESIGN.md 3.3c / 3.4.  To build it, copy it over the csrc/ file of the same base name.
//
// WRNN_KERNEL_BATCH_CS: the batch kernel (loop_batch.hip: R = 4*NQ rows per XCD team in lock-step on v_mfma_f32_4x4x1, the
// reference's "all B rows advance together", fatchord_version.py:194-237) with WAVE SPECIALISATION -- two waves per SIMD.
//
// Why: loop_batch.hip keeps all 320 weight registers of a SIMD lane in ONE wave (512 registers), so the MFMA phases of the serial
// chain, the shadow MFMAs (W_hh1.h1', W_hh2.h2'), noise, conditioning and the six exchange waits run one after the other in
// one instruction stream: 832 MFMAs issue in 6 800 of a 22 300-cycle step (R = 8), and bench_micro/mfma4_probe shows that one
// wave cannot saturate the matrix pipe anyway (8.1 cycles per 4x4x1 MFMA per wave, 4.07 per SIMD with two waves).  Here a SIMD
// holds two waves of 256 registers with different jobs, as in loop_team2.hip:
//   waves 0-3 "C" (critical): W_ih2 (96) + fc1 (32) + fc2 (32) weights in VGPRs, the fc3 slice in LDS: phase A (I + GRU1),
//                             phase B (GRU2), fc1, fc2, half of fc3 + the race; the four gathers of the serial chain and the winners.
//   waves 4-7 "S" (shadow):   W_hh1 (96) + W_hh2 gates r,z (64) in VGPRs: gh1' = W_hh1.h1', gh2' = W_hh2.(x3 - x2), the sampling
//                             noise, the conditioning of the NEXT step, the other half of fc3.  Results go to the C wave of the same
//                             SIMD -- same lane = same (unit, batch row) -- through small LDS slots, separated by the step's 5
//                             workgroup barriers.  An S wave never looks at the mailbox (round 5).
// No AGPR parking: hipcc splits a 256-register wave 128 : 128 between VGPRs and AGPRs as soon as a kernel touches an AGPR, so this
// file is compiled with -mllvm -amdgpu-mfma-vgpr-form (MFMA results in VGPRs; see the Makefile) and all 160 weights are plain floats.
//
// Team, residency, thread <-> (unit, row) map, B-operand order in LDS, the K-phase fold and the software-pipelined MFMA loops are those
// of loop_batch.hip (batch_common.h).  Per step: 5 exchanges (RAW; MOL 4), 5 barriers (all 8 waves):
//   window 1: C phase A, publish {x2, h1'}, gather -> P and H1                         S noise of this step -> nz slots
//   window 2: C phase B (W_ih2.x2), publish x3, gather -> Q                            S W_hh1.h1' -> gh1 slots (starts AT B1, beside phase B)
//   window 3: C fc1, publish, gate n of W_hh2.(Q - P) while the data travels, gather -> H1      S W_hh2 r,z.(Q - P) -> gh2 slots
//   window 4: C fc2, publish, gather -> P                                              S conditioning of step t+1 -> cd slots
//   window 5: C / S half of fc3 each (+ race, RAW), publish, candidates, winners -> xn         S frame constants; RAW 8 rows: the winner of row wl + 4
//
// THE EXCHANGE PROTOCOL (round 5; rounds 2-4 published 8-byte {tag = step, value} granules and the S waves gathered h1' themselves):
//   * a published vector is R x 512 plain 4-byte WORDS in the order [rq][wl][S][iu][j][e] (every 2 KB holds words of all 32 producers;
//     a 16-byte load returns the four e of one (rq, S, kp, j) = one ds_write_b128 in B-operand order).  An empty mailbox word holds
//     CS_EMPTY = 0xffffffff -- a NaN bit pattern no fp32 operation of this kernel produces (the hardware's NaN is 0x7fc00000) -- so THE
//     DATA IS THE FLAG without a tag: half the bytes of a look (the tags were half of every look: 32 KB per workgroup and exchange at
//     8 rows through a 64 B/clk port; round-4 diagnostic: a look of half the bytes is worth +4.7 % / +2.2 %).
//   * x2 and h1' of a (unit, row) travel as ONE 8-byte pair (one store): the C waves' x2 look brings h1' with it -- the bytes of the old
//     {tag, x2} look -- and writes both P and H1.  The S waves' own 32 KB look (2 280 cycles at R = 8, stretched by the C waves' phase-B
//     MFMAs), their LDS meeting point and its flags are gone; W_hh1.h1' starts at B1 and ends before the C waves' x3 look goes out
//     (profiles/r04_batch_cs_experiments.txt: a look beside a multiplying neighbour wave returns only when the neighbour's MFMA stream ends).
//   * who empties a word: its producer's workgroup.  Step e uses parity p = e & 1.  When a workgroup's C waves have gathered x2 of step e
//     from all 32 workgroups (barrier B1), every workgroup has finished every read of step e - 1 (x2 is published after the previous
//     step's last gather), so the S lane of the same (unit, row) stores CS_EMPTY into the workgroup's words of parity 1 - p right behind B1
//     -- off the serial chain.  The stores must have reached the L2 before anybody polls parity 1 - p again (step e + 1): the S wave waits
//     for them (s_waitcnt vmcnt(0), free by then) in front of B3, the C waves publish fc2 of step e behind B3, and nobody enters step
//     e + 1 without having gathered that.  The next store to such a word is the C lane's publish of step e + 1: behind B3..B5 as well.
//     A pass (batch) boundary changes nothing: the epoch keeps counting, the first gather of the next batch empties the last step's words.
//   * api.hip fills the mailbox with 0xff bytes before the launch.  A model that produces the CS_EMPTY pattern itself (only possible with
//     such NaN payloads in its inputs) runs into the bounded spin and reports WRNN_ERR_TIMEOUT -- never a silently wrong sample.
//   * the race candidates (RAW) keep {tag | class, score} granules: one per wave, row and step.
#include "batch_common.h"

#define CS_THREADS 512
// Developer knobs (tools/build_variant.sh NAME loop_batch_cs -DCS_...).  What round 4 measured and rejected -- slice rotation, a throttled shadow product,
// every polling variant except "sentinel slice first, then everything", yielding shadow waves, a v_min3 tag check, the wrong-result timing diagnostics of
// the shadow products' operands -- is recorded in profiles/r04_batch_cs_experiments.txt and no longer lives in this file.
#ifndef CS_PRIO
#define CS_PRIO 1        // the C waves run at s_setprio 3: the two waves of a SIMD compete for issue slots, the serial chain goes first
#endif
#ifndef CS_SPLIT_WIN
#define CS_SPLIT_WIN 1   // RAW, 8 rows per team: the winner of batch row wl + 4 is reduced by the S wave (window 5)
#endif
#ifndef CS_FC3_SPLIT
#define CS_FC3_SPLIT 1   // MOL: fc3 shared between the C and the S wave of a SIMD (each wave 4 of the SIMD's 8 rows)
#endif
#ifndef CS_FC3_SPLIT_RAW
#define CS_FC3_SPLIT_RAW 0   // the same for RAW (round 5, session 1: the S waves' half ends later than the C waves', the candidates wait doubles: -2 %)
#endif
#ifndef CS_PAIR
#define CS_PAIR 1        // 1: x2 and h1' travel as one 8-byte pair, the C waves' look writes P and H1, the S waves never look;  0: two word vectors, the S waves
                         // gather h1' themselves behind B1 and meet through LDS flags (the round-4 schedule on the round-5 protocol)
#endif
#ifndef CS_LATE_FOLD
#define CS_LATE_FOLD (!CS_PAIR)   // the fold of a shadow product runs behind the barrier that ends its window (round 4: with their own h1' look in window 2 the S
                                  // waves were the last at B2 / B3); with CS_PAIR they have ~3 000 cycles of slack there and fold at once
#endif
#ifndef CS_SMPRIO
#define CS_SMPRIO 0      // s_setprio of an S wave while it issues the MFMAs of a shadow product (0 = stays below the C wave's 3)
#endif
#ifndef CS_YIELD
#define CS_YIELD 0       // an S wave issues MFMAs of a shadow product only while the C wave of its SIMD multiplies or waits for a sentinel (token in LDS, looked
                         // at in front of every slab): 1 = not beside C's fold / gates / publish, 2 = not beside C's data look + LDS write either, 3 = W_hh1 not before C's x3 publish
                         // (the round-4 timing of that product without the S waves' own look)
#endif
#ifndef CS_WN_ON_C
#define CS_WN_ON_C 1     // gate n of W_hh2.(x3 - x2) (A operands in LDS, 64 MFMAs at 8 rows) by the C wave between its fc1 publish and the fc1 look: the
                         // data needs >= 600 cycles to arrive anyway, and the S waves' product (what the fc1 look waits for) shrinks by a third
#endif
#ifndef CS_COND_W4_RAW
#define CS_COND_W4_RAW 0
#endif
#ifndef CS_COND_W4
#define CS_COND_W4 (MODE == WRNN_MODE_MOL || CS_COND_W4_RAW)   // the conditioning of the next step in window 4 (MOL: in window 5 the C waves waited for it at B4b)
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
#define CS_EARLY_LOOK 8  // the full look goes out once this many of the sentinel slice's 64 lanes hold data (0 = all of them)
#endif
#ifndef CS_MAX_NQ
#define CS_MAX_NQ 2      // row quads per team this file is built for
#endif

namespace {

constexpr unsigned CS_EMPTY = 0xffffffffu;   // a mailbox word nobody has published yet (see the protocol above)

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
    static constexpr int L_LG = L_H1;               // [R][32] MOL: the 30 fc3 outputs of every batch row, in window 5 (H1 is dead from B4 to the next pair gather)
    static constexpr int L_MISC = L_XN + 16;
    static constexpr int L_PROF = L_MISC + 16;      // [2 roles][24]: phase cycles of wave 0 (C) and wave 4 (S), instrumented build only
    static constexpr int L_TOTAL = L_PROF + 48;
    static_assert(L_TOTAL * 4 <= 163840, "LDS budget");
    static_assert((L_P % 4) == 0, "B operands are read as 16-byte vectors");
    // mailbox regions per team, in 4-byte words; every region is double-buffered by step parity
    static constexpr unsigned RGD = (unsigned)VEC;                       // one published vector of words
    static constexpr unsigned M_XH = 0;                                  // {x2, h1'} pairs: 2 parities x 2 RGD
    static constexpr unsigned M_X3 = 4 * RGD, M_F1 = 6 * RGD, M_F2 = 8 * RGD;   // 2 parities x RGD each
    static constexpr unsigned M_PR = 10 * RGD;                           // race candidates: 2 parities x PRG granules of 8 bytes
    static constexpr unsigned PRG = (unsigned)R * 256u;                  // [row][32 workgroups][8 waves] (CS_FC3_SPLIT_RAW; else [row][32][4 C waves], half of it used)
    static constexpr unsigned MAIL_WORDS = M_PR + 2 * 2 * PRG;
    static_assert(MAIL_WORDS <= 2 * WRNN_BATCH_MAIL_GRANULES, "mailbox budget");
    static constexpr int NMP = R;                   // 16-byte loads per thread (256 threads) of a pair vector
    static constexpr int NMW = R / 2;               //                                          of a word vector
};
// hand-over slots
constexpr int H_GH1R = 0, H_GH1Z = 1, H_GH1N = 2, H_CDX = 3, H_CDY = 4, H_CDZ = 5, H_CDW = 6, H_GH2R = 7, H_GH2Z = 8, H_GH2N = 9,
              H_C2R = 10, H_C2Z = 11, H_C2N = 12, H_C3 = 13, H_C4 = 14, H_NZ = 15;   // H_NZ: [parity][2]

__device__ __forceinline__ void st_word(unsigned *base, unsigned idx, unsigned v) {
    const unsigned off = idx * 4u;
    asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void st_pair(unsigned *base, unsigned idx, unsigned lo, unsigned hi) {   // idx: word index of the pair (even)
    const u64 v = ((u64)hi << 32) | lo;
    const unsigned off = idx * 4u;
    asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}
// has every word of a 16-byte load arrived?  (a pair arrives whole: one 8-byte store -- its first word stands for both)
template <bool PAIR>
__device__ __forceinline__ bool arrived(const u4v &v) {
    return PAIR ? (v.x != CS_EMPTY && v.z != CS_EMPTY) : (v.x != CS_EMPTY && v.y != CS_EMPTY && v.z != CS_EMPTY && v.w != CS_EMPTY);
}
// All-gather of one published vector: NL 16-byte sc1 loads per thread (voff = thread * 16, soff = the vector's byte offset).
// Sentinel first, then everything: a C wave has nothing (or little) to do between its publish and this gather, and a poll that opens
// with a full look re-reads the whole vector per workgroup while the producers' stores queue behind those reads.
// The sentinel is the vector's last 4 KB (words of all 32 producers); the full look goes out as soon as CS_EARLY_LOOK of its 64 lanes hold
// data: the stragglers' words land while it is in flight, so the sentinel round trip and the data round trip overlap (round 4: +2.1 % /
// +4.4 %).  A look that still comes back incomplete polls the sentinel until it is complete and then fetches everything again: no
// per-slice state (the retry masks of the first version cost ~16 SGPR pairs).  Every spin is bounded.
typedef volatile int __attribute__((address_space(3))) *lds_vip;
template <int NL, bool PAIR, bool SENTINEL = true>
__device__ __forceinline__ void gather_words(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, u4v (&g)[NL], bool &dead, unsigned *err, unsigned code,
                                             unsigned *sent_cyc = nullptr, lds_vip hold = nullptr) {
    unsigned spins = 0;
    while (SENTINEL) {
        const u4v sv = ld_pair(rs, voff, soff + (NL - 1) * 4096u);
#if CS_EARLY_LOOK
        if (__builtin_popcountll(__ballot(arrived<PAIR>(sv))) >= CS_EARLY_LOOK || dead) break;
#else
        if (__all(arrived<PAIR>(sv)) || dead) break;
#endif
        if (++spins > TB_SPIN_MAX) { dead = true; if ((threadIdx.x & 63) == 0) atomicExch(err, code); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    if (sent_cyc) *sent_cyc = (unsigned)__builtin_readcyclecounter();   // instrumented build: the sentinel wait ends here, the data look starts
    if (hold && (threadIdx.x & 63) == 0) *hold = 0;                      // CS_YIELD 2: the S wave of this SIMD holds its MFMAs while the look is checked and written
    for (;;) {
#pragma unroll
        for (int m = 0; m < NL; ++m) g[m] = ld_pair(rs, voff, soff + m * 4096u);
        bool ok = true;
#pragma unroll
        for (int m = 0; m < NL; ++m) ok = ok && arrived<PAIR>(g[m]);
        if (__all(ok) || dead) break;
        for (;;) {
            if (++spins > TB_SPIN_MAX) { dead = true; if ((threadIdx.x & 63) == 0) atomicExch(err, code); break; }
            __builtin_amdgcn_s_sleep(1);
            const u4v sv = ld_pair(rs, voff, soff + (NL - 1) * 4096u);
            if (__all(arrived<PAIR>(sv))) break;
        }
        if (dead) break;
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

// RAW sampler, in-wave part: (v, k) = score and class of this lane's candidate for its batch row -> the best of the wave's classes for that row
// (the four unit groups rho = lane >> 4), ties -> the lower class.  Valid in the lanes with rho == 0.
__device__ __forceinline__ void race_fold(float &v, int &k) {
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
}
// ... and the winner among a row's candidates {tag | class, score}: lane l holds slots 2 l, 2 l + 1 (one 16-byte load: 128 candidates, one per C wave) or
// 4 l .. 4 l + 3 (two loads: 256, one per wave, CS_FC3_SPLIT_RAW); slots are in class order (workgroup, wave pair, C before S), so "the first of equal
// scores" is the lowest class, as torch's argmax of p / q picks it
template <bool TWO>
__device__ __forceinline__ int race_winner(const u4v &ga, const u4v &gb) {
    float best = __uint_as_float(ga.x);
    unsigned bk = ga.y;
    { const float v = __uint_as_float(ga.z); if (v > best) { best = v; bk = ga.w; } }
    if (TWO) {
        { const float v = __uint_as_float(gb.x); if (v > best) { best = v; bk = gb.y; } }
        { const float v = __uint_as_float(gb.z); if (v > best) { best = v; bk = gb.w; } }
    }
    const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_max_b(best)), 63));
    const u64 ball = __ballot(best == mx);
    const int src = (int)__builtin_ctzll(ball ? ball : 1ull);
    return __builtin_amdgcn_readlane((int)(bk & 1023u), src);
}

}  // namespace

#define PBW(i)                                                                 \
    do {                                                                       \
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
    typedef LayCS<NQ> L;
    constexpr int R = L::R, NMP = L::NMP, NMW = L::NMW, SL = L::SL;
    constexpr int DG = NQ == 1 ? 2 : 1, DS = NQ == 1 ? 4 : 2, D3 = NQ == 1 ? 2 : 1;
    // fc3: the 8 sets of four rows a SIMD owns are shared between its two waves -- the C wave evaluates classes 8 wl + iu, the S wave 8 wl + iu + 4
    // (the fc3 image is in LDS, so either wave can): half the MFMAs each, side by side on the matrix pipe, instead of all of them in the C wave
    constexpr bool FC3_SPLIT = MODE == WRNN_MODE_MOL ? CS_FC3_SPLIT != 0 : CS_FC3_SPLIT_RAW != 0;
    // rows whose race a C wave finishes itself: RAW at 8 rows per team hands the second one (batch row wl + 4) to the S wave of its SIMD
    constexpr int NBC = (MODE == WRNN_MODE_RAW && NQ == 2 && CS_SPLIT_WIN) ? 1 : NQ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = (float *)smem;
    int *misc_i = (int *)(lds + L::L_MISC);
    float *xn = lds + L::L_XN;
    float *lgt = lds + L::L_LG;
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
    unsigned *mailw = (unsigned *)mail;   // the same bytes in words (pairs and words: LayCS::M_*; the race candidates stay 8-byte granules)
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
    const lds_cf4p w3 = (lds_cf4p)(size_t)launder(smem_base + (unsigned)L::L_FC3 * 4u + ((unsigned)(wl * 2) * 8u * 64u + (unsigned)lane) * 16u);
    const lds_cf4p wnl = (lds_cf4p)(size_t)launder(smem_base + (unsigned)L::L_WN * 4u + ((unsigned)wl * 8u * 64u + (unsigned)lane) * 16u);
    const lds_cfp cst = (lds_cfp)(size_t)launder(smem_base + (unsigned)L::L_CST * 4u + (unsigned)ci * 4u);
    typedef float __attribute__((address_space(3))) *lds_fp;
    const lds_fp hand = (lds_fp)(size_t)launder(smem_base + (unsigned)L::L_HAND * 4u + (unsigned)ci * 4u);
    // where a thread's loads of a gathered vector go (B-operand order [rq][S][kp][j][e]; thread tl, 16-byte load m):
    //   pairs (8-byte entries, mailbox order [rq][wl][S][iu][j][e]): load m = (rq, wl) = (m >> 2, m & 3), the thread's two entries are e, e + 1 -> one 8-byte write
    //   words: load m covers (rq, wl) = (m >> 1, 2 (m & 1) + (tl >> 7)), the thread's four words are e = 0..3 -> one 16-byte write
    const lds_f2p gdst = (lds_f2p)(size_t)launder(smem_base + (unsigned)L::L_P * 4u + ((unsigned)(tl >> 5) * 256u + 2u * (unsigned)(tl & 31)) * 4u);
    typedef f4 __attribute__((address_space(3))) *lds_f4p;
    const lds_f4p gd4 = (lds_f4p)(size_t)launder(smem_base + (unsigned)L::L_P * 4u + ((unsigned)((tl >> 4) & 7) * 64u + (unsigned)(tl >> 7) * 16u + (unsigned)(tl & 15)) * 16u);

    // LDS words two waves talk through, addressed in the LDS address space so that they are stored and polled with DS instructions (round-4
    // advisor: through generic `volatile` pointers hipcc emitted flat_store / flat_load, which the memory model does not order against the
    // ds_write of the data; a wave's DS instructions execute in order and the LDS serves the waves' instructions one after the other):
    //   tok[wl]   CS_YIELD: C wave wl -> S wave wl of the same SIMD: 1 = "I multiply or wait for a sentinel: the matrix pipe is yours too", 0 = "hold"
    //   sflag[4]  CS_PAIR 0: meeting point of the four S waves (epoch of the H1 each of them has written)
    typedef int i4v __attribute__((ext_vector_type(4)));
    typedef volatile i4v __attribute__((address_space(3))) *lds_vi4p;
    const lds_vip tok = (lds_vip)(size_t)(smem_base + (unsigned)(L::L_MISC + 4 + wl) * 4u);
    const lds_vip sflag = (lds_vip)(size_t)(smem_base + (unsigned)(L::L_MISC + 8) * 4u);
    auto tok_set = [&](int v) { if (CS_YIELD && lane == 0) *tok = v; };
    const lds_vip hold2 = CS_YIELD == 2 ? tok : nullptr;
    if (CS_YIELD && !isC && lane == 0) *tok = 1;
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
            float gh2n = cst[C_H2N * SL];   // CS_WN_ON_C: gate n of W_hh2.h2' + b_hh2, evaluated by this wave a step ahead (step 0: h2 = 0, :194-196)
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
                        if (CS_PAIR) st_pair(mailw, L::M_XH + par * 2u * L::RGD + 2u * mb_own, __float_as_uint(x2own), __float_as_uint(h1));
                        else {   // two word vectors: x2 for the C waves' look, h1' for the S waves' (behind B1)
                            st_word(mailw, L::M_XH + par * 2u * L::RGD + mb_own, __float_as_uint(x2own));
                            st_word(mailw, L::M_XH + par * 2u * L::RGD + L::RGD + mb_own, __float_as_uint(h1));
                        }
                    }
                }
                PBW(0);
                if (CS_PAIR) {
                    u4v gx[NMP];
                    GSF_DECL;
                    gather_words<NMP, true>(mrs, gvoff, (L::M_XH + par * 2u * L::RGD) * 4u, gx, dead, a.err, 21u GSF_TS);
                    GSF_ACC(17);
                    PBW(1);
#pragma unroll
                    for (int m = 0; m < NMP; ++m) {   // x2 -> P, h1' -> H1 (the S waves multiply it from B1 on)
                        gdst[(0 * L::VEC + (m >> 2) * 2048 + (m & 3) * 64) / 2] = (f2v){__uint_as_float(gx[m].x), __uint_as_float(gx[m].z)};
                        gdst[(2 * L::VEC + (m >> 2) * 2048 + (m & 3) * 64) / 2] = (f2v){__uint_as_float(gx[m].y), __uint_as_float(gx[m].w)};
                    }
                } else {
                    u4v gx[NMW];
                    GSF_DECL;
                    gather_words<NMW, false>(mrs, gvoff, (L::M_XH + par * 2u * L::RGD) * 4u, gx, dead, a.err, 21u GSF_TS);
                    GSF_ACC(17);
                    PBW(1);
#pragma unroll
                    for (int m = 0; m < NMW; ++m) gd4[0 * (L::VEC / 4) + (m >> 1) * 512 + (m & 1) * 32] = __builtin_bit_cast(f4, gx[m]);
                }
                tok_set(CS_YIELD == 3 ? 0 : 1);
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
                    if (CS_YIELD != 3) tok_set(0);   // the fold and the gates are a dependent VALU chain: beside an MFMA stream they take three times as long (session 1)
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
                    const float ng = tanh_fast((tn + hand[H_C2N * SL]) + rg * (CS_WN_ON_C ? gh2n : hand[H_GH2N * SL]));
                    h2 = (1.0f - zg) * ng + zg * h2;
                    const float x3 = x2own + h2;
                    if (primary) st_word(mailw, L::M_X3 + par * L::RGD + mb_own, __float_as_uint(x3));
                    tok_set(1);   // from here the wave waits for a sentinel
                }
                PBW(6);
                {
                    u4v gx[NMW];
                    GSF_DECL;
#if CS_PROF_SPLIT
                    gather_words<NMW, false>(mrs, gvoff, (L::M_X3 + par * L::RGD) * 4u, gx, dead, a.err, 23u GSF_TS, hold2);
#else
                    gather_words<NMW, false>(mrs, gvoff, (L::M_X3 + par * L::RGD) * 4u, gx, dead, a.err, 23u, nullptr, hold2);
#endif
                    GSF_ACC(7);
#pragma unroll
                    for (int m = 0; m < NMW; ++m) gd4[1 * (L::VEC / 4) + (m >> 1) * 512 + (m & 1) * 32] = __builtin_bit_cast(f4, gx[m]);
                }
                tok_set(1);
                PBW(9);
                __syncthreads();   // B2
                PBW(10);

                // ---------------- window 3: fc1 (:217-218) | publish | gather ----------------
                {
                    f4 sum[NQ];
                    mfma_single<NQ, (NQ == 1 ? 4 : 2), false, DS>(wv + 96, vQ, sum);
                    if (CS_YIELD != 3) tok_set(0);
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float f = fold_kp(sum[q]);
                        if (q == 0 || my_rq == q) s = f;
                    }
                    if (primary) st_word(mailw, L::M_F1 + par * L::RGD + mb_own, __float_as_uint(fmaxf(s + hand[H_C3 * SL], 0.0f)));
                    tok_set(1);
                }
                PBW(11);
                if (CS_WN_ON_C) {
                    // gate n of gh2' = W_hh2.(x3 - x2) + b_hh2 for the NEXT step (A operands: the LDS image of the S waves' round-4 product), while the
                    // fc1 words travel: a publish is visible everywhere ~600-800 cycles after the store at the earliest, these 32 NQ MFMAs + fold take
                    // about that.  Q (x3) and P (x2) are both intact in window 3.
                    __builtin_amdgcn_sched_barrier(0);
                    f4 accn[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) accn[q] = (f4){0.f, 0.f, 0.f, 0.f};
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
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int q = 0; q < NQ; ++q) accn[q] = mfma4(wcur[e], b[q][e], accn[q]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float fn = fold_kp(accn[q]);
                        if (q == 0 || my_rq == q) gh2n = fn + cst[C_H2N * SL];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
                    u4v gx[NMW];
                    GSF_DECL;
#if CS_PROF_SPLIT
                    gather_words<NMW, false>(mrs, gvoff, (L::M_F1 + par * L::RGD) * 4u, gx, dead, a.err, 24u GSF_TS, hold2);
#else
                    gather_words<NMW, false>(mrs, gvoff, (L::M_F1 + par * L::RGD) * 4u, gx, dead, a.err, 24u, nullptr, hold2);
#endif
                    GSF_ACC(12);
#pragma unroll
                    for (int m = 0; m < NMW; ++m) gd4[2 * (L::VEC / 4) + (m >> 1) * 512 + (m & 1) * 32] = __builtin_bit_cast(f4, gx[m]);
                }
                tok_set(1);
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
                    if (primary) st_word(mailw, L::M_F2 + par * L::RGD + mb_own, __float_as_uint(fmaxf(s + hand[H_C4 * SL], 0.0f)));
                }
                PBW(15);
                {
                    u4v gx[NMW];
                    GSF_DECL;
                    gather_words<NMW, false>(mrs, gvoff, (L::M_F2 + par * L::RGD) * 4u, gx, dead, a.err, 25u GSF_TS);
                    GSF_ACC(16);
#pragma unroll
                    for (int m = 0; m < NMW; ++m) gd4[0 * (L::VEC / 4) + (m >> 1) * 512 + (m & 1) * 32] = __builtin_bit_cast(f4, gx[m]);
                }
                PBW(18);
                __syncthreads();   // B4
                PBW(19);

                // ---------------- window 5: fc3 (:223) + sampler (:225-237) ----------------
                {
                    float lg0 = 0.f, lg1 = 0.f;
                    if (FC3_SPLIT) {
                        // this wave's half: classes 8 wl + iu of the workgroup's slice (RAW) / of all 30 rows (MOL); the S wave of the SIMD: + 4
                        if (wg_has_fc3) lg0 = fc3_one_set<NQ, D3>(w3, vP, my_rq) + cst[C_B30 * SL];
                        if (a.logits_out && primary && row_ok && t < rw.steps && (MODE != WRNN_MODE_MOL || g == 0) && cls0 < NC)
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
                            for (int q = 0; q < NQ; ++q) ring[dd][q] = vP[(q * 8 + dd) * 64];
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
                                for (int q = 0; q < NQ; ++q) ring[S % D3][q] = vP[(q * 8 + S + D3) * 64];
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
                        float v = cls0 < NC ? lg0 + hz[0] : -INFINITY;
                        int k = cls0;
                        if (!FC3_SPLIT) {
                            const float v1 = cls0 + 4 < NC ? lg1 + hz[SL] : -INFINITY;
                            if (v1 > v) { v = v1; k = cls0 + 4; }
                        }
                        race_fold(v, k);
                        if (primary && rho == 0) {
                            const unsigned slot = L::M_PR / 2u + par * L::PRG + (unsigned)rb * 256u + (FC3_SPLIT ? (unsigned)(g * 8 + wl * 2) : (unsigned)(g * 4 + wl));
                            st_granule(mail, slot, (epoch << 10) | (unsigned)(k & 1023), __float_as_uint(v));
                        }
                    } else {
                        // MOL: the wave's 8 fc3 outputs of every batch row -> LDS; wave w samples batch rows w, w + 4 behind the barrier
                        if (primary) { lgt[rb * 32 + cls0] = lg0; if (!FC3_SPLIT) lgt[rb * 32 + cls0 + 4] = lg1; }
                    }
                }
                PBW(20);
                if (MODE == WRNN_MODE_MOL) __syncthreads();   // B4b
                u4v gqa[NBC][2];
                if (MODE == WRNN_MODE_RAW) {
                    const unsigned tg = epoch & 0x3fffffu;
                    unsigned spins = 0;
                    for (;;) {
#pragma unroll
                        for (int i = 0; i < NBC; ++i) {
                            const unsigned cb = (L::M_PR / 2u + par * L::PRG + (unsigned)(wl + 4 * i) * 256u) * 8u;
                            gqa[i][0] = ld_pair(mrs, (unsigned)lane * (FC3_SPLIT ? 32u : 16u), cb);
                            gqa[i][1] = FC3_SPLIT ? ld_pair(mrs, (unsigned)lane * 32u, cb + 16u) : gqa[i][0];
                        }
                        bool ok = true;
#pragma unroll
                        for (int i = 0; i < NBC; ++i)
                            ok = ok && (gqa[i][0].y >> 10) == tg && (gqa[i][0].w >> 10) == tg && (gqa[i][1].y >> 10) == tg && (gqa[i][1].w >> 10) == tg;
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
                        lab = race_winner<FC3_SPLIT>(gqa[bi][0], gqa[bi][1]);
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

                // ---------------- window 1: the sampler's noise of THIS step, while the C waves wait for the {x2, h1'} pairs (an S wave has nothing
                // else to do before B1; in window 4 of the step before it made the S waves late at B4) ----------------
                noise_step(t, par);
                PBW(2);
                __syncthreads();   // B1: the C waves have written x2 -> P and h1' -> H1
                PBW(3);
                // The C waves have gathered this step's x2 from all 32 workgroups: everybody has finished every read of the PREVIOUS step.  Its words (other
                // parity) are emptied by their producer's S lane -- same lane = same (unit, row) = same word as the C lane that published it -- off the
                // serial chain (session 2: issued by the C waves with their x3 publish, the five stores cost the x3 exchange ~550 cycles; session 3: in
                // front of the S waves' own h1' look they cost THAT look ~800 cycles -- a wave's loads are counted behind its stores -- so behind it).
                auto empty_previous = [&]() {
                    if (primary) {
                        const unsigned op = par ^ 1u;
                        if (CS_PAIR) st_pair(mailw, L::M_XH + op * 2u * L::RGD + 2u * mb_own, CS_EMPTY, CS_EMPTY);
                        else { st_word(mailw, L::M_XH + op * 2u * L::RGD + mb_own, CS_EMPTY); st_word(mailw, L::M_XH + op * 2u * L::RGD + L::RGD + mb_own, CS_EMPTY); }
                        st_word(mailw, L::M_X3 + op * L::RGD + mb_own, CS_EMPTY);
                        st_word(mailw, L::M_F1 + op * L::RGD + mb_own, CS_EMPTY);
                        st_word(mailw, L::M_F2 + op * L::RGD + mb_own, CS_EMPTY);
                    }
                };
                if (CS_PAIR) empty_previous();
                // an S wave looks at the token of the C wave it shares the SIMD with in front of every slab (CS_YIELD)
                auto yield = [&]() {
                    if (CS_YIELD) { for (unsigned sp = 0; sp < 20000u && *tok == 0; ++sp) __builtin_amdgcn_s_sleep(1); }
                };
                if (!CS_PAIR) {
                    // h1' gathered by the S waves themselves, BEHIND B1 (round 4: their look no longer runs beside the C waves' x2 look through the same
                    // 64 B/clk port, and they are never the last at B1).  The words were published a whole window ago: one look, no sentinel.  The four S
                    // waves then meet through LDS flags (s_barrier would need the C waves).
                    __builtin_amdgcn_s_setprio(3);
                    u4v gx[NMW];
                    gather_words<NMW, false, false>(mrs, gvoff, (L::M_XH + par * 2u * L::RGD + L::RGD) * 4u, gx, dead, a.err, 22u);
                    PBW(1);
#pragma unroll
                    for (int m = 0; m < NMW; ++m) gd4[2 * (L::VEC / 4) + (m >> 1) * 512 + (m & 1) * 32] = __builtin_bit_cast(f4, gx[m]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's part of H1 has left the DS queue
                    if (lane == 0) sflag[wl] = (int)epoch;
                    PBW(5);
                    unsigned sp = 0;
                    for (;;) {
                        const i4v f = *(lds_vi4p)sflag;
                        if ((f.x == (int)epoch && f.y == (int)epoch && f.z == (int)epoch && f.w == (int)epoch) || dead) break;
                        if (++sp > 200000u) { dead = true; if (lane == 0) atomicExch(a.err, 29u); break; }   // a lost S wave: reported, not multiplied through
                    }
                    asm volatile("" ::: "memory");
                    PBW(6);
                    __builtin_amdgcn_s_setprio(0);
                    empty_previous();
                }
                constexpr int NG2 = CS_WN_ON_C ? 2 : 3;
                f4 acc1[3][NQ], acc2[NG2][NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) acc1[gt][q] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int gt = 0; gt < NG2; ++gt) acc2[gt][q] = (f4){0.f, 0.f, 0.f, 0.f};
                }
                auto fold1 = [&]() {
                    float gr = 0.f, gz = 0.f, gn = 0.f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float fr = fold_kp(acc1[0][q]), fz = fold_kp(acc1[1][q]), fn = fold_kp(acc1[2][q]);
                        if (q == 0 || my_rq == q) { gr = fr + cst[C_H1R * SL]; gz = fz + cst[C_H1Z * SL]; gn = fn + cst[C_H1N * SL]; }
                    }
                    if (primary) { hand[H_GH1R * SL] = gr; hand[H_GH1Z * SL] = gz; hand[H_GH1N * SL] = gn; }   // read by C in phase A of the next step
                };
                auto fold2 = [&]() {
                    float gr = 0.f, gz = 0.f, gn = 0.f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float fr = fold_kp(acc2[0][q]), fz = fold_kp(acc2[1][q]);
                        if (q == 0 || my_rq == q) { gr = fr + cst[C_H2R * SL]; gz = fz + cst[C_H2Z * SL]; }
                        if (!CS_WN_ON_C) { const float fn = fold_kp(acc2[NG2 - 1][q]); if (q == 0 || my_rq == q) gn = fn + cst[C_H2N * SL]; }
                    }
                    if (primary) { hand[H_GH2R * SL] = gr; hand[H_GH2Z * SL] = gz; if (!CS_WN_ON_C) hand[H_GH2N * SL] = gn; }   // read by C in window 2 of the next step
                };
                if (CS_SMPRIO) __builtin_amdgcn_s_setprio(CS_SMPRIO);
                if (!(CS_DIAG & 1)) mfma_gates<NQ, 3, false, (NQ == 1 ? 2 : 1), decltype(yield), 0, (CS_YIELD != 0)>(wv, vH1, acc1, yield);
                if (CS_SMPRIO) __builtin_amdgcn_s_setprio(0);
                PBW(7);
                if (!CS_LATE_FOLD) fold1();
                PBW(8);
                __syncthreads();   // B2: x3 -> Q
                PBW(10);
                if (CS_LATE_FOLD) fold1();

                // ---------------- window 3: gh2' = W_hh2 . (x3 - x2) + b_hh2 of the next step: gates r, z (the weights in this wave's registers);
                // gate n (A operands in LDS) is the C wave's, between its fc1 publish and its fc1 look (CS_WN_ON_C) ----------------
                {
                    if (CS_SMPRIO) __builtin_amdgcn_s_setprio(CS_SMPRIO);
                    if (!(CS_DIAG & 2)) {
                        f4 xq[NQ], xp[NQ], wn = (f4){0.f, 0.f, 0.f, 0.f};
                        if (!CS_WN_ON_C) wn = wnl[0];
#pragma unroll
                        for (int q = 0; q < NQ; ++q) { xq[q] = vQ[(q * 8) * 64]; xp[q] = vP[(q * 8) * 64]; }
#pragma unroll
                        for (int S = 0; S < 8; ++S) {
                            yield();
                            f4 b[NQ];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) b[q] = xq[q] - xp[q];
                            const f4 wcur = wn;
                            if (S < 7) {
                                if (!CS_WN_ON_C) wn = wnl[(S + 1) * 64];
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
                                    if (!CS_WN_ON_C) acc2[NG2 - 1][q] = mfma4(wcur[e], b[q][e], acc2[NG2 - 1][q]);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (CS_SMPRIO) __builtin_amdgcn_s_setprio(0);
                    PBW(12);
                    if (!CS_LATE_FOLD) fold2();
                }
                // the CS_EMPTY stores of window 2 must have reached the L2 before anybody polls their parity again (next step).  They have, thousands of
                // cycles ago -- made certain HERE: behind B3 the C waves publish fc2, and nobody enters the next step without having gathered that.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PBW(13);
                __syncthreads();   // B3
                PBW(14);
                if (CS_LATE_FOLD) fold2();

                // ---------------- window 4: the conditioning of the next step (the cd slots are read in phase A of the next step only; the frame
                // constants wait for B4: C still reads this frame's c4 in this window) ----------------
                if (CS_COND_W4 && t + 1 < bsteps) cond_step(t + 1);
                PBW(16);
                __syncthreads();   // B4: fc2 outputs -> P
                PBW(19);

                // ---------------- window 5: frame constants of a new frame; this wave's half of fc3 (+ its candidate of the race, RAW) ----------------
                if (CS_COND_W4) frame_flush(); else if (t + 1 < bsteps) cond_step(t + 1);
                if (FC3_SPLIT) {   // classes 8 wl + iu + 4 (see the C waves' window 5)
                    float lg1 = 0.f;
                    if (wg_has_fc3) lg1 = fc3_one_set<NQ, D3>(w3 + 8 * 64, vP, my_rq) + cst[C_B31 * SL];
                    if (MODE == WRNN_MODE_MOL && primary) lgt[rb * 32 + cls0 + 4] = lg1;
                    if (a.logits_out && primary && row_ok && t < rw.steps && (MODE != WRNN_MODE_MOL || g == 0) && cls0 + 4 < NC)
                        a.logits_out[((size_t)t * a.n_rows + row) * NC + cls0 + 4] = lg1;
                    if (MODE == WRNN_MODE_RAW) {
                        float v = cls0 + 4 < NC ? lg1 + hand[(H_NZ + 2 * par + 1) * SL] : -INFINITY;
                        int k = cls0 + 4;
                        race_fold(v, k);
                        if (primary && rho == 0)
                            st_granule(mail, L::M_PR / 2u + par * L::PRG + (unsigned)rb * 256u + (unsigned)(g * 8 + wl * 2 + 1),
                                       (epoch << 10) | (unsigned)(k & 1023), __float_as_uint(v));
                    }
                }
                PBW(17);
                if (MODE == WRNN_MODE_MOL) __syncthreads();   // B4b: the fc3 outputs of all rows are in LDS
                if (NBC < NQ) {
                    // exchange 5 for batch row wl + 4 (RAW, 8 rows per team): the C wave of this SIMD finishes row wl meanwhile.  One row per
                    // wave instead of two one after the other in the four C waves (1 130 -> ~600 cycles at the end of the serial chain).
                    const int brow = wl + 4;
                    const unsigned tg = epoch & 0x3fffffu;
                    const unsigned cb = (L::M_PR / 2u + par * L::PRG + (unsigned)brow * 256u) * 8u;
                    u4v ga, gb;
                    unsigned spins = 0;
                    for (;;) {
                        ga = ld_pair(mrs, (unsigned)lane * (FC3_SPLIT ? 32u : 16u), cb);
                        gb = FC3_SPLIT ? ld_pair(mrs, (unsigned)lane * 32u, cb + 16u) : ga;
                        if (__all((ga.y >> 10) == tg && (ga.w >> 10) == tg && (gb.y >> 10) == tg && (gb.w >> 10) == tg) || dead) break;
                        if (++spins > TB_SPIN_MAX) { dead = true; if (lane == 0) atomicExch(a.err, 28u); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    float xf = a.x_forced ? a.x_forced[(size_t)t * a.n_rows + frowS[1]] : 0.0f;
                    asm volatile("" : "+v"(xf));
                    const int lab = race_winner<FC3_SPLIT>(ga, gb);
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
