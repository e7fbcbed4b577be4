// Internal definitions shared by the HIP translation units of libwavernn_amd.so.
// Product code (gfx950 only).  Reference line citations are relative to
// /root/reference/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/wavernn_amd.h"

#define WRNN_MAX_UP 4
#define WRNN_PROF_SLOTS 32   // phase-cycle counters per wave (WRNN_TEAM_PROF=1)
#define WRNN_KTAB_MAXD 8

// Dimensions derived from wrnn_config (WaveRNN.__init__, fatchord_version.py:93-129).
struct WrnnDims {
    int H;        // rnn_dims
    int FC;       // fc_dims
    int F;        // feat_dims
    int A;        // aux_dims = res_out_dims / 4   (:109)
    int C;        // compute_dims
    int R;        // res_out_dims
    int NBLK;     // res_blocks
    int P;        // pad
    int KS;       // conv_in kernel = 2*pad+1      (:33)
    int HOP;      // prod(upsample_factors) (== hop_length)
    int NC;       // n_classes                      (:98-101)
    int mode;     // WRNN_MODE_*
    int ND;       // frames of support of the composite upsampling FIR
};

// Offsets (in floats) into the single packed device allocation.
struct WrnnPacked {
    // prologue, BatchNorm(eval) folded into the adjacent conv (eps 1e-5)
    size_t conv_in_t;   // [F*KS][C]
    size_t conv_in_b;   // [C]
    size_t res_w1_t;    // [NBLK][C(in)][C(out)]
    size_t res_b1;      // [NBLK][C]
    size_t res_w2_t;    // [NBLK][C][C]
    size_t res_b2;      // [NBLK][C]
    size_t conv_out_t;  // [C][R]
    size_t conv_out_b;  // [R]
    size_t ktab;        // [HOP][ND] composite (5,5,11) stretch+FIR taps
    // loop parameters, transposed to [in][out] so that consecutive threads
    // (= consecutive output rows) read consecutive addresses
    size_t I_t, I_b;              // [1+F+A][H], [H]
    size_t r1_wih_t, r1_whh_t;    // [H][3H], [H][3H]
    size_t r1_bih, r1_bhh;        // [3H]
    size_t r2_wih_t, r2_whh_t;    // [H+A][3H], [H][3H]
    size_t r2_bih, r2_bhh;
    size_t fc1_t, fc1_b;          // [H+A][FC], [FC]
    size_t fc2_t, fc2_b;          // [FC+A][FC], [FC]
    size_t fc3_t, fc3_b;          // [FC][NC], [NC]
    size_t total;
};

// One loop row = one utterance (unbatched) or one fold (fold_with_overlap :293-340).
struct WrnnRow {
    int32_t utt;      // index into the mel batch
    int32_t steps;    // loop steps of this row: the call's `steps`, or frames[utt] * hop in a ragged batch (opts.frames_dev)
    int64_t start;    // first upsampled position of this row
};
// Device error word codes (first come, first kept): 3 = a team kernel's workgroups did not all become resident (WRNN_ERR_BUSY),
// everything else = a bounded exchange spin gave up (WRNN_ERR_TIMEOUT)
#define WRNN_DEVERR_BUSY 3u
// polls a workgroup waits at the start of a team kernel for the other 31 of its XCD (~1.5 ms; a resident launch needs ~10 us)
#define WRNN_ARRIVE_POLLS 200000u

struct WrnnTrainState;   // train.hip: workspace + captured step graphs of wrnn_train_step
void wrnn_train_state_free(WrnnTrainState *st);

struct wrnn_handle {
    wrnn_config cfg;
    WrnnDims d;
    WrnnPacked off;
    float *wdev = nullptr;        // packed weights on device
    bool loaded = false;
    int64_t loop_weight_bytes = 0;
    // scratch (grown on demand)
    float *aux_frames = nullptr;  // (B, T, R)
    size_t aux_cap = 0;
    WrnnRow *rows_dev = nullptr;
    int32_t *order_dev = nullptr; // [rows] rows by length, longest first, in a ragged batch; identity otherwise (BATCH kernel)
    int32_t *sched_dev = nullptr; // [rows rounded up to n_teams] per-team row lists of the TEAM2 kernel (see WrnnTeamArgs)
    size_t rows_cap = 0;
    unsigned *err_dev = nullptr;  // device error word (bounded spins)
    // team kernel state
    float *team_w = nullptr, *team_fc3 = nullptr, *wI0 = nullptr, *u1 = nullptr;
    float *batch_w = nullptr, *batch_fc3 = nullptr, *batch_wn = nullptr;   // batch kernel images of the same weights
    bool team_dims = false;       // the constructor dims are the reference hparams the team kernels are built for
    bool team_ok = false;         // the 32-workgroup team kernels can be co-resident on this device (checked at create)
    std::string team_why;         // why not, when team_ok is false
    bool force_no_teams = false;  // wrnn_debug_force_no_teams: AUTO behaves as if residency had failed (tests of the slow-path warning)
    bool cs_ok = false;           // ... and loop_batch_cs_kernel in particular (AUTO falls back to WRNN_KERNEL_BATCH without it)
    float *tab = nullptr;         // CM|CA|VM|VA|C2|C3|C4 for the current batch
    size_t tab_cap = 0;
    float *cond = nullptr;        // conditioning stream of the current segment (TEAM2)
    size_t cond_cap = 0;
    double *epi_tab = nullptr;    // epilogue tables [dec NC | fade_in | fade_out | tail] for epi_overlap
    long epi_overlap = -1;
    float *team_state = nullptr;  // recurrent state of every row between segment launches (TEAM2)
    size_t team_state_cap = 0;
    unsigned long long *mail = nullptr;
    unsigned *ctl = nullptr;
    int n_teams = 8;              // XCDs (32-CU teams) of this device
    unsigned long long *prof = nullptr;   // phase-cycle counters, allocated by wrnn_phase_profile(h, 1)
    bool prof_on = false;
    double prof_div = 0;
    double *loss_partial = nullptr;   // per-block partial sums of wrnn_loss
    size_t loss_cap = 0;
    WrnnTrainState *train = nullptr;
    bool train_force_steps = false;   // wrnn_train_force_step_kernels
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    bool timing_valid = false;
    wrnn_timing last{};
    std::string err;
};

// Arguments of the per-sample loop kernels.
struct WrnnLoopArgs {
    const float *w;           // packed weights
    WrnnPacked off;
    WrnnDims d;
    const float *mels;        // (B, F, mel_T)
    int32_t mel_T, mel_off;   // frames per mel row, index of frame 0 (see wrnn_launch_resnet)
    const float *aux_frames;  // (B, T, R)
    const WrnnRow *rows;
    int32_t n_rows;
    int32_t T;                // mel frames per utterance
    int64_t total_len;        // T * HOP
    int64_t steps;            // loop length per row (the longest row's in a ragged batch; row r runs rows[r].steps)
    int32_t noise_mode;
    uint64_t seed;
    const float *noise1;      // RAW (L, rows, NC) | MOL (L, rows, 10)
    const float *noise2;      // MOL (L, rows)
    const float *x_forced;    // (L, rows) or null
    const float *x_init;      // (rows) value fed to step 0 instead of 0, or null (teacher-forced forward())
    float *logits_out;        // (L, rows, NC) or null
    int32_t *labels_out;      // (rows, L) or null
    float *samples_out;       // (rows, L)
    unsigned *err;
};

// Team kernel (loop_team.hip): mailbox size per team in 8-byte granules:
// x3, fc1, fc2, race winners (2 x 512 each), gh1 (2 x 1536)
#define WRNN_TEAM_MAIL_GRANULES (8 * 512 + 2 * 1536)
#define WRNN_TEAM_NWREG 352
#define WRNN_TEAM_STATE_FLOATS (8 * 512 + 64)
#define WRNN_TEAM_THREADS 256

struct WrnnTeamArgs {
    const float *w;           // packed weights (biases, ktab)
    WrnnPacked off;
    WrnnDims d;
    const float *team_w;      // [32 WGs][352][256 threads] register-resident weights
    const float *team_fc3;    // [32 WGs][16384] LDS image of the fc3 slices
    const float *wI0;         // [H]   W_I[:,0]
    const float *u1;          // [3H]  W_ih1 . W_I[:,0]
    // per-utterance conditioning pushed through the linear layers it feeds:
    //   CM (T+2P, H) = W_I[:,1:1+F] . melpad[f]     CA (T+1, H) = W_I[:,1+F:] . a1[i] + b_I (entry T: zeros in)
    //   VM (T+2P,3H) = W_ih1 . CM[f]                VA (T+1,3H) = W_ih1 . CA[i] + b_ih1
    // packed per frame and hidden unit, see pack_records_kernel (prologue.hip)
    const float *tabREC;      // (B, T+1, H, 28)
    const float *tabCOND;     // (rows, seg_len, H, 4) phase-A conditioning stream of steps [seg0, seg0+seg_len) (TEAM2) or null
    const float *tabC2;       // (B, T+1, 3H)   W_ih2[:,H:] . a2[i] + b_ih2
    const float *tabC3;       // (B, T+1, FC)   fc1.W[:,H:] . a3[i] + b1
    const float *tabC4;       // (B, T+1, FC)   fc2.W[:,FC:] . a4[i] + b2
    const WrnnRow *rows;
    const int32_t *sched;     // [n_slots] team t runs rows sched[t], sched[t + n_teams], ... (-1 = none): identity for a uniform batch,
    int32_t n_slots;          // longest-first rows dealt in snake order (0..n-1, n-1..0, ...) for a ragged one; n_slots = n_rows rounded
    int32_t ragged;           // up to a multiple of n_teams.  ragged != 0: use sched and the rows' own steps (opts.frames_dev)
    int32_t n_rows;
    int32_t n_teams;          // teams = XCDs in use; the launch grid is n_teams * 32 workgroups
    int32_t T;
    int64_t total_len;
    int64_t steps;
    // TEAM2 runs a row in segments (one launch each) so that the conditioning stream of a segment is still cache
    // resident when it is read; the recurrent state crosses launches through `state`:
    // per row [h1 H | h2 H | gh1 3H | gh2 3H | x 1 | pad] floats (WRNN_TEAM_STATE_FLOATS)
    int64_t seg0, seg_len;
    float *state;
    int32_t noise_mode;
    uint64_t seed;
    const float *noise1;
    const float *noise2;
    const float *x_forced;
    const float *x_init;
    float *logits_out;
    int32_t *labels_out;
    float *samples_out;
    unsigned long long *mail;  // [n_teams][WRNN_TEAM_MAIL_GRANULES]
    unsigned *ctl;             // [16] per-XCD arrival counters
    unsigned *err;
    unsigned long long *prof;  // [8][WRNN_PROF_SLOTS] phase cycle counters (developer instrumentation) or null
};

// Batch kernel (loop_batch.hip): R = 4 * nq rows per team in lock-step on the matrix cores.
// mailbox per team: 5 gathered vectors (x2, h1', x3, fc1, fc2) x 2 parities x R*512 granules + race 2 x R*128
#define WRNN_BATCH_MAIL_GRANULES (10 * 8 * 512 + 2 * 8 * 128)
#define WRNN_BATCH_MAX_ROWS 8
#define WRNN_MAIL_GRANULES_MAX (WRNN_BATCH_MAIL_GRANULES > WRNN_TEAM_MAIL_GRANULES ? WRNN_BATCH_MAIL_GRANULES : WRNN_TEAM_MAIL_GRANULES)

struct WrnnBatchArgs {
    const float *w;           // packed weights (biases, ktab)
    WrnnPacked off;
    WrnnDims d;
    const float *batch_w;     // [32 WGs][4 waves][320][64 lanes] MFMA A-operand images of the register-resident weights
    const float *batch_fc3;   // [32 WGs][4 waves][2 sets][8][64 lanes][4] A-operand image of the fc3 slice (LDS)
    const float *batch_wn;    // [32 WGs][4 waves][8][64 lanes][4] A-operand image of gate n of W_hh2 (LDS)
    const float *wI0;         // [H]   W_I[:,0]
    const float *u1;          // [3H]  W_ih1 . W_I[:,0]
    const float *tabREC32;    // (B, T+1, H, 32): the 24 phase-A record floats (pack_records_kernel) | c2 r,z,n | c3 | c4 | pad
    const WrnnRow *rows;
    const int32_t *order;     // schedule slot -> row (longest first in a ragged batch, identity otherwise)
    int32_t snake;            // != 0: batches are dealt to the teams in snake order (ragged batch), else round-robin
    int32_t n_rows;
    int32_t n_teams;
    int32_t nq;               // row quads per team: 1 (4 rows) or 2 (8 rows)
    int32_t rpb;              // rows actually placed in one batch (<= 4 * nq): batch b = slots [b * rpb, (b + 1) * rpb); it runs
                              // for the steps of its first (longest) row
    int32_t T;
    int64_t total_len;
    int64_t steps;
    int32_t noise_mode;
    uint64_t seed;
    const float *noise1;
    const float *noise2;
    const float *x_forced;
    const float *x_init;
    float *logits_out;
    int32_t *labels_out;
    float *samples_out;
    unsigned long long *mail;  // [n_teams][WRNN_BATCH_MAIL_GRANULES]
    unsigned *ctl;
    unsigned *err;
    unsigned long long *prof;
};

// Persistent team kernels of the two GRU recurrences of wrnn_train_step (train_team.hip)
struct WrnnGruTeamArgs {
    const float *img;          // weight image of this recurrence for the kernel at hand (wrnn_gru_team_pack)
    const float *bhh;          // [3H] (forward)
    const float *GI;           // (B, L, 3H) forward: input part of the gates
    float *Hs, *HP, *Rs, *Zs, *Ns, *GHN;   // (B, L, H) saved by the forward, read by the backward
    const float *dHext;        // (B, L, H) backward: gradient flowing into h_t from the layers above
    float *dGI, *dGH;          // (B, L, 3H) backward outputs
    int32_t B;
    int64_t L;
    int32_t n_teams, rpb;      // rows per team batch (<= 4 * nq)
    unsigned long long *mail;  // [n_teams][wrnn_gru_team_mail_granules(nq, bwd)]
    unsigned *ctl, *err;
};
size_t wrnn_gru_team_mail_granules(int nq, bool bwd);
hipError_t wrnn_gru_team_pack(const float *Whh, float *img, bool bwd, hipStream_t s);
hipError_t wrnn_gru_team_launch(const WrnnGruTeamArgs &a, int nq, bool bwd, hipStream_t s);
hipError_t wrnn_gru_team_occupancy(int nq, bool bwd, int *blocks_per_cu);

// kernels / launchers (defined in the .hip files)
// mel_T = frames per row of `mels`, mel_off = index of frame 0 in it: (T, 0) for generate()'s unpadded mels (zero
// padding applied on the fly, :183), (T + 2 pad, pad) for mels already padded like WaveRNN.forward receives them (:143)
hipError_t wrnn_launch_resnet(const wrnn_handle *h, const float *mels, int B, int T, int mel_T, int mel_off, float *aux_frames,
                              hipStream_t s);
hipError_t wrnn_launch_materialize(const wrnn_handle *h, const float *mels, const float *aux_frames, int B,
                                   int T, int mel_T, int mel_off, float *up, float *aux_up, hipStream_t s);
hipError_t wrnn_launch_loop_simple(const WrnnLoopArgs &a, hipStream_t s);
size_t wrnn_simple_lds_bytes(const WrnnDims &d);
hipError_t wrnn_launch_loop_batch(const WrnnBatchArgs &a, hipStream_t s);
hipError_t wrnn_batch_occupancy(int mode, int nq, bool prof, int *blocks_per_cu, size_t *lds_bytes);
// loop_batch_cs.hip: the same step with critical / shadow wave roles (two waves per SIMD); nq as built (wrnn_batch_cs_max_nq)
hipError_t wrnn_launch_loop_batch_cs(const WrnnBatchArgs &a, hipStream_t s);
hipError_t wrnn_batch_cs_occupancy(int mode, int nq, bool prof, int *blocks_per_cu, size_t *lds_bytes);
int wrnn_batch_cs_max_nq(int mode);
hipError_t wrnn_team2_occupancy(int mode, bool prof, int *blocks_per_cu, size_t *lds_bytes);
// Per-device ordering of team-kernel launches inside this process (api.hip): enter() makes `s` wait for the previous team
// kernel launched on `device` by any handle / stream and takes the device's launch lock, leave() records the new tail and
// releases the lock.  Nothing blocks on the GPU's progress; only the launching threads are serialised.
hipError_t wrnn_team_gate_enter(int device, hipStream_t s);
hipError_t wrnn_team_gate_leave(int device, hipStream_t s);
hipError_t wrnn_launch_loss(int mode, const float *y_hat, const void *y, int NC, long n_rows, double *partial, int *bad, float *out,
                            hipStream_t s);
// rows[r] = {utt, steps, start}; order[] = rows sorted by steps, longest first (stable), when frames != null, else identity;
// sched[] (n_rows rounded up to n_teams entries) = the same order dealt to n_teams teams in snake order, -1 where empty
hipError_t wrnn_launch_rows(WrnnRow *rows, int32_t *order, int32_t *sched, int n_rows, int n_teams, int batched, long stride, long steps,
                            const int32_t *frames, int T, int hop, hipStream_t s);
hipError_t wrnn_launch_pack_records32(const float *CM, const float *CA, const float *VM, const float *VA, const float *C2,
                                      const float *C3, const float *C4, float *rec, int B, int T, int P, hipStream_t s);
hipError_t wrnn_launch_loop_team2(const WrnnTeamArgs &a, hipStream_t s);
hipError_t wrnn_launch_cond_stream(const float *rec, const float *ktab, const WrnnRow *rows, float *cond, int n_rows, int T,
                                   int HOP, long total_len, long seg0, long seg_len, hipStream_t s);
hipError_t wrnn_launch_pack_records(const float *CM, const float *CA, const float *VM, const float *VA, float *rec, int B,
                                    int T, int P, hipStream_t s);
// out[b][f][n] = bias[n] + sum_k in(b,f,k) * Wt[k*ldw + n]; mode 0: row-major src (rows >= valid read as 0),
// mode 1: src = mels (B,F,T) read as zero-padded frames melpad[f] = mel[:, f - P]
hipError_t wrnn_launch_frame_linear(int mode, const float *src, size_t src_bstride, int ld, int valid, const float *Wt,
                                    int ldw, const float *bias, float *out, size_t out_bstride, int frames, int K,
                                    int N, int B, int T, int P, hipStream_t s);
