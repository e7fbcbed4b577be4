/*
 * wavernn_amd.h -- C-ABI of the MI355X-native WaveRNN vocoder (mel -> wav).
 *
 * The reference (lturing/tacotronv2_wavernn_chinese) has NO native/FFI layer:
 * its boundary for this path is the Python class
 *   wavernn/models/fatchord_version.py:92-129  WaveRNN.__init__
 *   wavernn/models/fatchord_version.py:169-264 WaveRNN.generate
 *   wavernn/models/fatchord_version.py:414-417 WaveRNN.load  (flat state_dict)
 * and the script wavernn_gen.py:13-43 (gen_from_file).  This header is the
 * C-ABI a maintainer would bind from that class (ctypes stub in
 * INTEGRATION.md); every entry point cites the reference lines it replaces.
 *
 * Conventions: plain pointers and sizes, no torch/C++ types; integer status
 * returns (0 = ok, <0 = error, message via wrnn_last_error); no exceptions
 * cross the ABI; caller allocates outputs; the library owns packed weights and
 * scratch; one handle per device; a handle is not thread-safe, distinct
 * handles are independent.  All `*_dev` pointers are device (HBM) pointers on
 * the handle's device; everything is enqueued on the caller's `stream`
 * (a hipStream_t passed as void*) and is asynchronous unless stated
 * (wrnn_last_timing and wrnn_dm_sync_status wait; nothing else does).
 *
 * The TEAM2 / BATCH kernels keep n_teams * 32 workgroups spinning on each other inside one launch, so all of them
 * must be resident at once (one per CU).  Three layers make that a checked fact instead of a rule for the caller:
 *   1. wrnn_create asks the runtime's occupancy query about the instantiations this handle can launch (mode, profile
 *      build) and reads the CU count; where a team kernel cannot be resident (LDS / registers, fewer than 32 CUs) AUTO
 *      uses the SIMPLE kernel and an explicit request for a team kernel fails with WRNN_ERR_INVALID.
 *   2. Inside one process every team-kernel launch on a device -- any handle, any stream, wrnn_generate and
 *      wrnn_dm_generate alike -- is ordered behind the previous one with a per-device event (stream wait, nothing blocks
 *      on the host): two handles (RAW + MOL, two threads) share a GPU safely.
 *   3. Across processes nothing can be ordered; a team kernel whose workgroups do not all become resident within a short
 *      bounded wait at its start (another process holds CUs) gives up at once and the call reports WRNN_ERR_BUSY through
 *      wrnn_last_timing / wrnn_dm_sync_status -- retry, or use WRNN_KERNEL_SIMPLE.  The occupancy query cannot see other
 *      processes; this run-time check is what covers a shared GPU.
 */
#ifndef WAVERNN_AMD_H
#define WAVERNN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WRNN_ABI_VERSION 6

/* mode: fatchord_version.py:98-103 */
#define WRNN_MODE_RAW 0 /* softmax over 2**bits classes */
#define WRNN_MODE_MOL 1 /* 10-component mixture of logistics, 30 outputs */

/* where the sampler's randomness comes from (fatchord_version.py:225-237,
 * wavernn/utils/distribution.py:87-123) */
#define WRNN_NOISE_PHILOX 0   /* device counter RNG keyed by (seed, step, row, class) */
#define WRNN_NOISE_INJECTED 1 /* caller-supplied draws: parity protocol with the reference */
#define WRNN_NOISE_ARGMAX 2   /* RAW only: greedy (q == 1) */

/* which device implementation runs the per-sample loop */
#define WRNN_KERNEL_AUTO 0   /* rows <= XCD teams: TEAM2 (latency); more rows: BATCH_CS (throughput); SIMPLE when the team
                              * kernels cannot run on this device / configuration */
#define WRNN_KERNEL_SIMPLE 1 /* one workgroup per row, weights streamed from L2/HBM; any shape */
/* 2 was the 4-wave team kernel of ABI 2 (retired) */
#define WRNN_KERNEL_TEAM2 3  /* one XCD-resident 32-workgroup team per row, weights on chip, critical / shadow wave roles */
#define WRNN_KERNEL_BATCH 4  /* one team per 4 or 8 rows in lock-step on the matrix cores (v_mfma_f32_4x4x1) */
#define WRNN_KERNEL_BATCH_CS 5 /* ABI 5: the same batch step with two waves per SIMD -- critical / shadow wave roles, the shadow
                                * matrix products, noise and conditioning run beside the serial chain instead of inside it */

/* tensor dtypes accepted by wrnn_load_weights */
#define WRNN_DTYPE_F32 0
#define WRNN_DTYPE_I64 1

#define WRNN_OK 0
#define WRNN_ERR_INVALID -1     /* bad argument / unsupported configuration */
#define WRNN_ERR_HIP -2         /* a HIP runtime call failed */
#define WRNN_ERR_STATE -3       /* e.g. generate before load_weights */
#define WRNN_ERR_MISSING_KEY -4 /* state_dict key absent (strict load) */
#define WRNN_ERR_TIMEOUT -5     /* a bounded device spin gave up */
#define WRNN_ERR_BUSY -6        /* a team kernel could not get all its workgroups resident (GPU shared with another process) */

typedef struct wrnn_handle wrnn_handle;

/* Constructor arguments of WaveRNN (fatchord_version.py:93-95) + device. */
typedef struct wrnn_config {
    int32_t rnn_dims;            /* 512  */
    int32_t fc_dims;             /* 512  */
    int32_t bits;                /* 10 (RAW) */
    int32_t pad;                 /* 2    */
    int32_t n_upsample;          /* 3    */
    int32_t upsample_factors[4]; /* 5,5,11 */
    int32_t feat_dims;           /* 80   */
    int32_t compute_dims;        /* 128  */
    int32_t res_out_dims;        /* 128  */
    int32_t res_blocks;          /* 10   */
    int32_t hop_length;          /* 275  */
    int32_t sample_rate;         /* 22050 */
    int32_t mode;                /* WRNN_MODE_* */
    int32_t device;              /* HIP device ordinal */
} wrnn_config;

/* One entry of the flat state_dict (fatchord_version.py:414-417; key names and
 * shapes listed in SURVEY.md section 8a).  `data` is a HOST pointer to a
 * contiguous row-major tensor; the library copies/repacks, the caller keeps
 * ownership. */
typedef struct wrnn_tensor_desc {
    const char *name;
    int32_t dtype; /* WRNN_DTYPE_* */
    int32_t ndim;
    int64_t shape[4];
    const void *data;
} wrnn_tensor_desc;

typedef struct wrnn_sample_opts {
    /* = sizeof(wrnn_sample_opts) of the CALLER.  The library refuses a size it does not know (WRNN_ERR_INVALID): a caller
     * built against another ABI revision fails loudly instead of having its fields read at shifted offsets. */
    uint32_t struct_size;
    int32_t noise_mode; /* WRNN_NOISE_* */
    int32_t kernel;     /* WRNN_KERNEL_* */
    /* != 0: mels_dev is (B, feat, T + 2*pad), already padded by `pad` frames on both sides with real context like the
     * training collate does and WaveRNN.forward receives it (:143); 0: (B, feat, T), zero padding applied on the fly
     * like generate() (:183-185).  T is the unpadded frame count either way. */
    int32_t mels_padded;
    uint64_t seed;      /* WRNN_NOISE_PHILOX */
    /* WRNN_NOISE_INJECTED, device pointers, step-major like the reference's
     * RNG consumption (one sampler call per step, batch inside):
     *   RAW: noise1 = Exp(1) draws (L, rows, n_classes)   [torch.multinomial]
     *   MOL: noise1 = u_mix (L, rows, 10), noise2 = u_log (L, rows)          */
    const float *noise1_dev;
    const float *noise2_dev;
    /* teacher forcing: value fed back as x_t instead of the drawn sample,
     * (L, rows) device pointer or NULL */
    const float *x_forced_dev;
    /* optional dump of the fc3 outputs, (L, rows, n_classes) device or NULL */
    float *logits_out_dev;
    /* value fed to step 0 of every row instead of 0 (:196), (rows) device pointer or NULL.  With x_forced_dev this is
     * the teacher-forced pass of WaveRNN.forward (fatchord_version.py:131-167): input sequence x[0..L) = x_init,
     * x_forced[0..L-1), logits in logits_out_dev. */
    const float *x_init_dev;
    /* Ragged batch (unbatched mode only), device pointer to B int32 or NULL: utterance b has frames_dev[b] valid mel
     * frames (1 <= frames_dev[b] <= T; mels_dev stays (B, feat, T), zero beyond an utterance's own frames -- the padding
     * generate() itself applies, :183).  Row b then runs frames_dev[b] * hop steps instead of T * hop: its first
     * frames_dev[b] * hop outputs are exactly what a call on that clip alone produces (rows are independent, :194-196;
     * the noise is keyed by (row, step)), the rest of the row is left unwritten.  The library orders the rows by length
     * on the device (longest first), fills team batches with rows of similar length and deals the batches to the teams
     * in snake order, so that no team idles behind a long clip. */
    const int32_t *frames_dev;
    /* tuning, 0 = the library's choice.  batch_rows: rows per team batch of WRNN_KERNEL_BATCH, 1..8 (default
     * ceil(rows / teams), at most 8).  team2_segment: steps per launch of WRNN_KERNEL_TEAM2 (a row is generated in
     * segments so that the conditioning stream of one segment stays cache resident; rounded down to a multiple of 32). */
    int32_t batch_rows;
    int32_t team2_segment;
} wrnn_sample_opts;

typedef struct wrnn_timing {
    float prologue_ms; /* conditioning kernels of the last wrnn_generate */
    float loop_ms;     /* the per-sample loop kernel of the last wrnn_generate */
    int32_t kernel;    /* WRNN_KERNEL_* that actually ran */
    int32_t rows;      /* rows the loop processed (B or num_folds) */
    int64_t steps;     /* loop length per row */
    int32_t launches;  /* loop-kernel launches the call was split into (segments) */
    int32_t reserved_;
} wrnn_timing;

/* replaces WaveRNN.__init__ (fatchord_version.py:93-129) */
int wrnn_create(const wrnn_config *cfg, wrnn_handle **out);

/* replaces WaveRNN.load / load_state_dict (fatchord_version.py:414-417).  Every parameter the path reads must be
 * present with the reference's dtype and shape (WRNN_ERR_MISSING_KEY / WRNN_ERR_INVALID otherwise); keys the path does
 * not read (step, num_batches_tracked, optimizer leftovers) are ignored.  The reference's strict=False tolerance for
 * missing keys lives on the host side of the binding (nn.Module keeps its initial value for them). */
int wrnn_load_weights(wrnn_handle *h, const wrnn_tensor_desc *tensors, int32_t n);

/* Conditioning exactly as generate() builds it (fatchord_version.py:183-186:
 * pad_tensor 'both' + UpsampleNetwork.forward :82-89), materialised:
 *   mels_dev (B, feat, T) -> up_dev (B, T*hop, feat), aux_dev (B, T*hop, res_out)
 * mels_padded != 0: mels_dev is (B, feat, T + 2*pad) as WaveRNN.forward receives it (see wrnn_sample_opts).
 * Either output may be NULL.  Used by parity tests of the prologue; the loop
 * itself never materialises these tensors. */
int wrnn_conditioning(wrnn_handle *h, const float *mels_dev, int32_t B, int32_t T, int32_t mels_padded, float *up_dev,
                      float *aux_dev, void *stream);

/* Number of loop rows / steps generate() will run for (B, T, batched, target,
 * overlap): rows = B (unbatched) or num_folds (fold_with_overlap :293-340,
 * requires B == 1); steps = T*hop or target + 2*overlap. */
int wrnn_plan(wrnn_handle *h, int32_t B, int32_t T, int32_t batched, int32_t target, int32_t overlap,
              int32_t *rows_out, int64_t *steps_out);

/* replaces the device part of WaveRNN.generate (fatchord_version.py:183-241):
 * prologue + per-sample loop for all rows.
 *   labels_out_dev  (rows, steps) int32: RAW class index | MOL mixture index (may be NULL)
 *   samples_out_dev (rows, steps) fp32: the value appended to `output` (:227,:236)
 * The float64 epilogue (:243-258) and the wav write (:260) stay on the host
 * side of the binding. */
int wrnn_generate(wrnn_handle *h, const float *mels_dev, int32_t B, int32_t T, int32_t batched,
                  int32_t target, int32_t overlap, const wrnn_sample_opts *opts, int32_t *labels_out_dev,
                  float *samples_out_dev, void *stream);

/* replaces the float64 tail of generate() (fatchord_version.py:243-258): decode_mu_law (wavernn/utils/dsp.py:98-103)
 * when mu_law != 0 on a RAW model (needs labels_dev), xfade_and_unfold (:342-405) when batched != 0, the trim to
 * wave_len and the 20-hop linear fade-out (:255-258).  samples_dev / labels_dev: (rows, steps) as written by
 * wrnn_generate; wave_out_dev: wave_len doubles.  Unbatched calls use row 0 only, like :253.  wave_len shorter than
 * 20 hops is WRNN_ERR_INVALID (the reference raises ValueError for T < 21).  Asynchronous on `stream`. */
int wrnn_epilogue(wrnn_handle *h, const float *samples_dev, const int32_t *labels_dev, int32_t rows, int64_t steps,
                  int32_t batched, int32_t target, int32_t overlap, int32_t mu_law, int64_t wave_len,
                  double *wave_out_dev, void *stream);

/* The same tail for a batch of INDEPENDENT utterances (throughput mode, generate_many): every row r is finished like an
 * unbatched call finishes row 0 -- decode_mu_law (when mu_law != 0 on a RAW model), trim, 20-hop fade-out -- in ONE launch:
 *   wave_out_dev[r * out_stride + n], n < wave_len_r;  wave_len_r = wave_len, or (frames_dev[r] - 1) * hop when frames_dev
 *   (device, rows int32, the array given to wrnn_generate) is not NULL; out_stride >= wave_len doubles per row, entries
 *   [wave_len_r, out_stride) of a row are set to 0.  A row shorter than the 20-hop fade-out (the reference raises
 *   ValueError for T < 21, :256-258) is all zeros: the host side of the binding rejects such clips before the call. */
int wrnn_epilogue_rows(wrnn_handle *h, const float *samples_dev, const int32_t *labels_dev, int32_t rows, int64_t steps,
                       int32_t mu_law, int64_t wave_len, const int32_t *frames_dev, double *wave_out_dev, int64_t out_stride,
                       void *stream);

/* Host-only helper (no device): the float64 tables wrnn_epilogue gathers from, built in NumPy's evaluation order --
 * dec[n_classes] (decode_mu_law of 2k/(n_classes-1)-1, dsp.py:98-103), fade_in/fade_out[overlap] (:374-385, may be
 * NULL when overlap == 0), tail[20*hop] (np.linspace(1, 0, 20*hop_length), :256).  Caller-allocated. */
int wrnn_epilogue_tables(int32_t n_classes, int32_t overlap, int32_t hop, double *dec, double *fade_in, double *fade_out,
                         double *tail);

/* The loss the reference's training script applies to WaveRNN.forward's output (wavernn_train.py:82,112-121), forward value
 * only.  y_hat_dev (n_rows, n_classes) = the fc3 outputs of n_rows = B*L (batch, step) pairs, row-major.
 *   RAW model: F.cross_entropy -- y_dev = int32 class labels (n_rows); a label outside [0, n_classes) gives NaN.
 *   MOL model: discretized_mix_logistic_loss (wavernn/utils/distribution.py:16-84; num_classes 65536,
 *              log_scale_min log(1e-14), reduce=True) -- y_dev = float32 targets in [-1, 1] (n_rows).
 * loss_out_dev: one float32 on the device.  Asynchronous on `stream`. */
int wrnn_loss(wrnn_handle *h, const float *y_hat_dev, const void *y_dev, int64_t n_rows, float *loss_out_dev, void *stream);

/* ---- training step of the loop layers -------------------------------------------------------------------
 * WaveRNN.forward (fatchord_version.py:131-167) from the upsampled conditioning on, the training script's loss
 * (wavernn_train.py:82,112-121) and the backward pass of `loss.backward()` through I, rnn1, rnn2, fc1, fc2, fc3.
 * Every pointer is a DEVICE pointer to a contiguous float32 tensor in the reference's own layout (nn.Linear / nn.GRU:
 * weight (out, in), gate rows r | z | n) -- the parameters are used where torch keeps them, nothing is repacked, so an
 * optimizer step between two calls costs nothing here. */
typedef struct wrnn_loop_params {
    float *I_w, *I_b;                                   /* (rnn, 1 + feat + aux), (rnn)                 :115 */
    float *rnn1_w_ih, *rnn1_w_hh, *rnn1_b_ih, *rnn1_b_hh; /* (3 rnn, rnn) x2, (3 rnn) x2                  :117 */
    float *rnn2_w_ih, *rnn2_w_hh, *rnn2_b_ih, *rnn2_b_hh; /* (3 rnn, rnn + aux), (3 rnn, rnn), (3 rnn) x2 :118 */
    float *fc1_w, *fc1_b, *fc2_w, *fc2_b, *fc3_w, *fc3_b; /* (fc, rnn + aux), (fc, fc + aux), (n_classes, fc) :121-123 */
} wrnn_loop_params;

/*   w            parameters (read)
 *   g            gradients of the MEAN loss, same shapes (overwritten, not accumulated), or NULL: forward + loss only
 *   x_dev        (B, L) input samples                     mels_up_dev (B, L, feat), aux_dev (B, L, res_out): what
 *                self.upsample(mels) returns (:143), computed by the caller (the upsample network trains through the
 *                framework's autograd: BatchNorm in training mode needs batch statistics)
 *   y_dev        (B, L) targets: int32 class labels (RAW) / float32 in [-1, 1] (MOL); may be NULL when g and loss_out_dev are
 *   loss_out_dev one float32 (F.cross_entropy / discretized_mix_logistic_loss, as wrnn_loss) or NULL
 *   logits_out_dev (B, L, n_classes) fc3 outputs = forward()'s return value, or NULL
 *   d_mels_up_dev, d_aux_dev  gradients w.r.t. the conditioning (same shapes; written when g != NULL), or NULL
 * The handle supplies dims / mode and owns the workspace (grown on demand, ~86 KB per (batch, step) pair at the default
 * dims) and the captured step graphs; it needs no loaded weights.  Asynchronous on `stream`. */
int wrnn_train_step(wrnn_handle *h, const wrnn_loop_params *w, const wrnn_loop_params *g, const float *x_dev,
                    const float *mels_up_dev, const float *aux_dev, const void *y_dev, int32_t B, int64_t L,
                    float *loss_out_dev, float *logits_out_dev, float *d_mels_up_dev, float *d_aux_dev, void *stream);

/* The same computation split where autograd splits it -- so that the reference's own training loop runs unchanged
 * (`y_hat = model(x, m); loss = loss_func(y_hat, y); loss.backward()`, wavernn_train.py:103-122, with ANY loss on y_hat):
 *   wrnn_train_forward   forward() from the conditioning on (:145-167): logits_out_dev (B, L, n_classes); every activation stays in
 *                        the handle's workspace
 *   wrnn_train_backward  given d_logits_dev (B, L, n_classes) = d loss / d y_hat: the 16 parameter gradients g (overwritten) and the
 *                        gradients w.r.t. the conditioning (either may be NULL).  Must follow a wrnn_train_forward on the same handle
 *                        with the same B, L, inputs and parameters, with no other training call in between (WRNN_ERR_STATE otherwise). */
int wrnn_train_forward(wrnn_handle *h, const wrnn_loop_params *w, const float *x_dev, const float *mels_up_dev, const float *aux_dev,
                       int32_t B, int64_t L, float *logits_out_dev, void *stream);
int wrnn_train_backward(wrnn_handle *h, const wrnn_loop_params *w, const wrnn_loop_params *g, const float *d_logits_dev,
                        const float *x_dev, const float *mels_up_dev, const float *aux_dev, int32_t B, int64_t L,
                        float *d_mels_up_dev, float *d_aux_dev, void *stream);

/* Waits for `stream`, then reports the device-side error word of the team kernels wrnn_train_step launched (WRNN_ERR_BUSY,
 * WRNN_ERR_TIMEOUT) -- what wrnn_last_timing does for wrnn_generate. */
int wrnn_sync_status(wrnn_handle *h, void *stream);
/* The two GRU recurrences of wrnn_train_step run as one persistent XCD-team kernel each where rnn_dims is 512 and the device has
 * 32-CU teams, else as one kernel per time step replayed from a hipGraph.  on != 0 forces the per-step kernels (tests compare the two). */
int wrnn_train_force_step_kernels(wrnn_handle *h, int32_t on);

/* Blocks until the last wrnn_generate on this handle finished, then reports
 * HIP-event timings and any device-side error (WRNN_ERR_TIMEOUT). */
int wrnn_last_timing(wrnn_handle *h, wrnn_timing *out);

/* Developer instrumentation (replaces the WRNN_TEAM_PROF environment variable of ABI 3): enable != 0 makes the following
 * TEAM2 / BATCH calls of this handle run the instrumented instantiation of the loop kernel (s_memtime stamps between the
 * phases of a step; ~1 % slower).  wrnn_phase_cycles waits for the last call and returns, for workgroup 0 of team 0,
 * cycles per step of every phase marker: out[wave * 32 + marker], 8 waves x 32 markers (unused entries 0). */
int wrnn_phase_profile(wrnn_handle *h, int32_t enable);
int wrnn_phase_cycles(wrnn_handle *h, double *out /* [8 * 32] */);

/* n_classes (fatchord_version.py:98-101) and loop-parameter bytes (roofline) */
int32_t wrnn_n_classes(const wrnn_handle *h);
int64_t wrnn_loop_weight_bytes(const wrnn_handle *h);

/* ABI 6.  Can the XCD-team kernels (TEAM2 / BATCH / BATCH_CS: weights resident on chip) run for this handle on its device?  Returns 1 / 0;
 * *n_teams_out = number of 32-CU teams (8 on an MI355X in SPX mode; what the host side sizes fold counts and batches for),
 * *why_not_out = "" or the reason (points into the handle; valid until wrnn_destroy).  Either pointer may be NULL.  When the answer is 0,
 * WRNN_KERNEL_AUTO runs WRNN_KERNEL_SIMPLE -- reference-ordered, any dims, ~1 ms per step: slower than the reference on CPU cores -- and the
 * host side of the binding warns about it (the reference has no equivalent: this protects wavernn_gen.py:126's one call). */
int32_t wrnn_team_info(const wrnn_handle *h, int32_t *n_teams_out, const char **why_not_out);
/* Test hook: on != 0 makes this handle behave as if the residency check of wrnn_create had failed (AUTO -> SIMPLE, an explicit team kernel
 * -> WRNN_ERR_INVALID), so that the slow-path warning can be exercised on a healthy device. */
int wrnn_debug_force_no_teams(wrnn_handle *h, int32_t on);

const char *wrnn_last_error(const wrnn_handle *h);
int32_t wrnn_abi_version(void);
void wrnn_destroy(wrnn_handle *h);

/* ---- secondary model: wavernn/models/deepmind_version.py (unconditioned dual-softmax coarse/fine WaveRNN; no
 * reference script imports it).  Entry points mirror WaveRNN(hidden_size, quantisation) :9-31, load_state_dict and
 * generate(seq_len) :75-165.  noise_dev (WRNN_NOISE_INJECTED): Exp(1) draws (seq_len, 2, quantisation), [t][0] for the
 * coarse Categorical.sample() (:131), [t][1] for the fine one (:151).  Outputs: int32 (seq_len,) each; the signal is
 * coarse * 256 + fine - 2**15 (wavernn/utils/dsp.py:33-34), combined on the host side of the binding. */
typedef struct wrnn_dm_handle wrnn_dm_handle;
int wrnn_dm_create(int32_t hidden_size, int32_t quantisation, int32_t device, wrnn_dm_handle **out);
int wrnn_dm_load_weights(wrnn_dm_handle *h, const wrnn_tensor_desc *tensors, int32_t n);
int wrnn_dm_generate(wrnn_dm_handle *h, int64_t seq_len, int32_t noise_mode, uint64_t seed, const float *noise_dev,
                     int32_t *coarse_out_dev, int32_t *fine_out_dev, void *stream);
/* kernel: 0 auto (the 32-workgroup team kernel when hidden_size is 512/640/768/896 and quantisation a multiple of
 * 64, else the single-workgroup kernel), 1 single workgroup (reference-ordered sums), 2 team */
int wrnn_dm_set_kernel(wrnn_dm_handle *h, int32_t kernel);
/* synchronises `stream` and reports the device-side error word of the team kernel (WRNN_ERR_TIMEOUT) */
int wrnn_dm_sync_status(wrnn_dm_handle *h, void *stream);
const char *wrnn_dm_last_error(const wrnn_dm_handle *h);
void wrnn_dm_destroy(wrnn_dm_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* WAVERNN_AMD_H */
