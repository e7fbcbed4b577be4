// Row A12 of SURVEY.md section 8a (secondary): the unconditioned dual-softmax (coarse/fine) WaveRNN of
// wavernn/models/deepmind_version.py, generate(seq_len) :75-165.  No reference script ever imports that model, so
// this is a straightforward, reference-ordered kernel (one persistent workgroup, weights streamed [in][out] from
// L2 each step) with the same sampler as the main path: Categorical(softmax(l)).sample() == argmax_k l_k - log q_k.
#include "device_util.h"
#include "wrnn_internal.h"

#define DM_THREADS 1024

#include "dm_internal.h"

namespace {

// argmax over Q <= 1024 candidates held one per thread (threads >= Q pass -inf); result broadcast through LDS
__device__ int block_argmax(float v, int idx, float *redv, int *redi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    wave_argmax(v, idx);
    if (lane == 0) { redv[wave] = v; redi[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bv = redv[0]; int bi = redi[0];
        for (int i = 1; i < DM_THREADS / 64; ++i)
            if (redv[i] > bv || (redv[i] == bv && redi[i] < bi)) { bv = redv[i]; bi = redi[i]; }
        redi[31] = bi;
    }
    __syncthreads();
    return redi[31];
}

__global__ void __launch_bounds__(DM_THREADS) dm_loop_kernel(WrnnDmArgs a) {
    __shared__ float h[1024], Rh[3072], t1[512], lg[256];
    __shared__ float redv[16];
    __shared__ int redi[32];
    const int H = a.H, S = H / 2, Q = a.Q, j = threadIdx.x;
    const float *w = a.w;
    if (j < H) h[j] = 0.0f;                                  // get_initial_hidden :168-170
    int oc = 0, of = 0;                                      // out_coarse = out_fine = 0 :90-91
    __syncthreads();
    for (long t = 0; t < a.seq_len; ++t) {
        const float pc = (float)oc / 127.5f - 1.0f, pf = (float)of / 127.5f - 1.0f;   // :106-107
        // R(hidden), no bias :116
        for (int r = j; r < 3 * H; r += DM_THREADS) {
            float acc = 0.0f;
            const float *col = w + a.oRT + r;
#pragma unroll 8
            for (int k = 0; k < H; ++k) acc = fmaf(col[(size_t)k * 3 * H], h[k], acc);
            Rh[r] = acc;
        }
        __syncthreads();
        // coarse gates :111-125   (R_hidden = [u_c | u_f | r_c | r_f | e_c | e_f])
        if (j < S) {
            const float *Ic = w + a.oIc;
            const float Iu = Ic[j * 2] * pc + Ic[j * 2 + 1] * pf;
            const float Ir = Ic[(S + j) * 2] * pc + Ic[(S + j) * 2 + 1] * pf;
            const float Ie = Ic[(2 * S + j) * 2] * pc + Ic[(2 * S + j) * 2 + 1] * pf;
            const float u = 1.0f / (1.0f + expf(-(Rh[j] + Iu + w[a.obu + j])));
            const float r = 1.0f / (1.0f + expf(-(Rh[H + j] + Ir + w[a.obr + j])));
            const float e = tanhf(r * Rh[2 * H + j] + Ie + w[a.obe + j]);
            h[j] = u * h[j] + (1.0f - u) * e;
        }
        __syncthreads();
        // out_coarse = O2(relu(O1(hidden_coarse))) :128
        if (j < S) {
            float acc = 0.0f;
            for (int k = 0; k < S; ++k) acc = fmaf(w[a.oO1T + (size_t)k * S + j], h[k], acc);
            t1[j] = fmaxf(acc + w[a.oO1b + j], 0.0f);
        }
        __syncthreads();
        float v = -INFINITY;
        if (j < Q) {
            float acc = 0.0f;
            for (int k = 0; k < S; ++k) acc = fmaf(w[a.oO2T + (size_t)k * Q + j], t1[k], acc);
            v = acc + w[a.oO2b + j];
            if (a.noise_mode == WRNN_NOISE_INJECTED) v -= logf(a.noise[((size_t)t * 2 + 0) * Q + j]);
            else if (a.noise_mode == WRNN_NOISE_PHILOX) v -= logf(-logf(wrnn_uniform(a.seed, (uint64_t)t, 0u, (uint32_t)j)));
        }
        oc = block_argmax(v, j < Q ? j : 0x7fffffff, redv, redi);          // Categorical(...).sample() :130-131
        if (j == 0) a.coarse[t] = oc;
        const float cp = (float)oc / 127.5f - 1.0f;                       // :135
        // fine gates :136-145
        if (j < S) {
            const float *If = w + a.oIf;
            const float *wu = If + (size_t)j * 3, *wr = If + (size_t)(S + j) * 3, *we = If + (size_t)(2 * S + j) * 3;
            const float Iu = wu[0] * pc + wu[1] * pf + wu[2] * cp;
            const float Ir = wr[0] * pc + wr[1] * pf + wr[2] * cp;
            const float Ie = we[0] * pc + we[1] * pf + we[2] * cp;
            const float u = 1.0f / (1.0f + expf(-(Rh[S + j] + Iu + w[a.obu + S + j])));
            const float r = 1.0f / (1.0f + expf(-(Rh[H + S + j] + Ir + w[a.obr + S + j])));
            const float e = tanhf(r * Rh[2 * H + S + j] + Ie + w[a.obe + S + j]);
            h[S + j] = u * h[S + j] + (1.0f - u) * e;
        }
        __syncthreads();
        // out_fine = O4(relu(O3(hidden_fine))) :148
        if (j < S) {
            float acc = 0.0f;
            for (int k = 0; k < S; ++k) acc = fmaf(w[a.oO3T + (size_t)k * S + j], h[S + k], acc);
            t1[j] = fmaxf(acc + w[a.oO3b + j], 0.0f);
        }
        __syncthreads();
        v = -INFINITY;
        if (j < Q) {
            float acc = 0.0f;
            for (int k = 0; k < S; ++k) acc = fmaf(w[a.oO4T + (size_t)k * Q + j], t1[k], acc);
            v = acc + w[a.oO4b + j];
            if (a.noise_mode == WRNN_NOISE_INJECTED) v -= logf(a.noise[((size_t)t * 2 + 1) * Q + j]);
            else if (a.noise_mode == WRNN_NOISE_PHILOX) v -= logf(-logf(wrnn_uniform(a.seed, (uint64_t)t, 1u, (uint32_t)j)));
        }
        of = block_argmax(v, j < Q ? j : 0x7fffffff, redv, redi);          // :150-151
        if (j == 0) a.fine[t] = of;
        (void)lg;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------- C-ABI
struct wrnn_dm_handle {
    int H = 0, Q = 0, device = 0;
    float *wdev = nullptr;
    WrnnDmArgs args{};
    bool loaded = false;
    int kernel = 0;               // 0 auto (team kernel when the sizes allow), 1 single-workgroup kernel, 2 team kernel
    float *team_w = nullptr, *team_lds = nullptr;
    unsigned long long *mail = nullptr;
    unsigned *ctl = nullptr;      // [32] team counters + [32] error word
    std::string err;
};

#include <cstdarg>
#include <map>
static int dm_fail(wrnn_dm_handle *h, int code, const char *fmt, ...) {
    char buf[400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}
#define DM_TRY(h, expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) return dm_fail((h), WRNN_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

extern "C" {

int wrnn_dm_create(int32_t hidden_size, int32_t quantisation, int32_t device, wrnn_dm_handle **out) {
    if (!out) return WRNN_ERR_INVALID;
    wrnn_dm_handle *h = new wrnn_dm_handle();
    *out = h;
    h->H = hidden_size; h->Q = quantisation; h->device = device;
    if (hidden_size < 2 || hidden_size > 1024 || (hidden_size & 1) || quantisation < 2 || quantisation > 256)
        return dm_fail(h, WRNN_ERR_INVALID, "unsupported sizes: hidden_size even and <= 1024, quantisation <= 256");
    return WRNN_OK;
}

void wrnn_dm_destroy(wrnn_dm_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->wdev) (void)hipFree(h->wdev);
    if (h->team_w) (void)hipFree(h->team_w);
    if (h->team_lds) (void)hipFree(h->team_lds);
    if (h->mail) (void)hipFree(h->mail);
    if (h->ctl) (void)hipFree(h->ctl);
    delete h;
}

const char *wrnn_dm_last_error(const wrnn_dm_handle *h) { return h ? h->err.c_str() : "null handle"; }

int wrnn_dm_load_weights(wrnn_dm_handle *h, const wrnn_tensor_desc *tensors, int32_t n) {
    if (!h || !tensors) return WRNN_ERR_INVALID;
    const int H = h->H, S = H / 2, Q = h->Q;
    std::map<std::string, const wrnn_tensor_desc *> tv;
    for (int i = 0; i < n; ++i)
        if (tensors[i].name && tensors[i].data) tv[tensors[i].name] = &tensors[i];
    auto get = [&](const char *name, int64_t numel, const float **out) -> int {
        auto it = tv.find(name);
        if (it == tv.end()) return dm_fail(h, WRNN_ERR_MISSING_KEY, "state_dict key missing: %s", name);
        int64_t ne = 1;
        for (int i = 0; i < it->second->ndim; ++i) ne *= it->second->shape[i];
        if (it->second->dtype != WRNN_DTYPE_F32 || ne != numel) return dm_fail(h, WRNN_ERR_INVALID, "%s: bad dtype/shape", name);
        *out = (const float *)it->second->data;
        return WRNN_OK;
    };
    WrnnDmArgs &a = h->args;
    size_t cur = 0;
    auto take = [&](size_t nfl) { size_t at = cur; cur += (nfl + 63) & ~(size_t)63; return at; };
    a.oRT = take((size_t)H * 3 * H);
    a.oO1T = take((size_t)S * S); a.oO1b = take(S); a.oO2T = take((size_t)S * Q); a.oO2b = take(Q);
    a.oO3T = take((size_t)S * S); a.oO3b = take(S); a.oO4T = take((size_t)S * Q); a.oO4b = take(Q);
    a.oIc = take((size_t)3 * S * 2); a.oIf = take((size_t)3 * S * 3);
    a.obu = take(H); a.obr = take(H); a.obe = take(H);
    std::vector<float> pk(cur, 0.0f);
    int rc;
    const float *src;
    auto put_t = [&](const char *name, size_t at, int rows, int cols) -> int {   // (rows, cols) -> [cols][rows]
        if ((rc = get(name, (int64_t)rows * cols, &src))) return rc;
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) pk[at + (size_t)c * rows + r] = src[(size_t)r * cols + c];
        return WRNN_OK;
    };
    auto put_v = [&](const char *name, size_t at, int nel) -> int {
        if ((rc = get(name, nel, &src))) return rc;
        std::copy(src, src + nel, pk.begin() + at);
        return WRNN_OK;
    };
    if ((rc = put_t("R.weight", a.oRT, 3 * H, H)) || (rc = put_t("O1.weight", a.oO1T, S, S)) || (rc = put_v("O1.bias", a.oO1b, S)) ||
        (rc = put_t("O2.weight", a.oO2T, Q, S)) || (rc = put_v("O2.bias", a.oO2b, Q)) || (rc = put_t("O3.weight", a.oO3T, S, S)) ||
        (rc = put_v("O3.bias", a.oO3b, S)) || (rc = put_t("O4.weight", a.oO4T, Q, S)) || (rc = put_v("O4.bias", a.oO4b, Q)) ||
        (rc = put_v("I_coarse.weight", a.oIc, 3 * S * 2)) || (rc = put_v("I_fine.weight", a.oIf, 3 * S * 3)) ||
        (rc = put_v("bias_u", a.obu, H)) || (rc = put_v("bias_r", a.obr, H)) || (rc = put_v("bias_e", a.obe, H)))
        return rc;
    DM_TRY(h, hipSetDevice(h->device));
    if (h->wdev) { (void)hipFree(h->wdev); h->wdev = nullptr; }
    DM_TRY(h, hipMalloc(&h->wdev, cur * sizeof(float)));
    DM_TRY(h, hipMemcpy(h->wdev, pk.data(), cur * sizeof(float), hipMemcpyHostToDevice));
    // ---- team kernel layouts (loop_dm_team.hip) ----
    if (wrnn_dm_team_supported(H, Q)) {
        const int U = S / 32, QW = Q / 32, CPL = H / 64, PS = CPL / 2, NR = 3 * CPL * 4;
        const float *R, *O1, *O2, *O3, *O4;
        if ((rc = get("R.weight", (int64_t)3 * H * H, &R)) || (rc = get("O1.weight", (int64_t)S * S, &O1)) ||
            (rc = get("O2.weight", (int64_t)Q * S, &O2)) || (rc = get("O3.weight", (int64_t)S * S, &O3)) ||
            (rc = get("O4.weight", (int64_t)Q * S, &O4)))
            return rc;
        std::vector<float> tw((size_t)32 * NR * 512, 0.0f);
        const size_t nimg = (size_t)2 * U * S + (size_t)2 * QW * S;
        std::vector<float> tl((size_t)32 * nimg, 0.0f);
        for (int g = 0; g < 32; ++g) {
            for (int tid = 0; tid < 512; ++tid) {
                const int qw = tid >> 4, q = tid & 15;
                if (qw >= 2 * U) continue;
                const int hi = qw < U ? g * U + qw : S + g * U + (qw - U);
                for (int gate = 0; gate < 3; ++gate)
                    for (int c = 0; c < 4 * CPL; ++c)
                        tw[((size_t)g * NR + gate * 4 * CPL + c) * 512 + tid] = R[(size_t)(gate * H + hi) * H + q * 4 * CPL + c];
            }
            // LDS images: [row of the workgroup][plane k][lane q][4] <- W[row][q * 4 PS + 4 k + e]
            float *img = tl.data() + (size_t)g * nimg;
            auto fill = [&](float *dst, const float *W, int row0, int nrows) {
                for (int r = 0; r < nrows; ++r)
                    for (int k = 0; k < PS; ++k)
                        for (int q = 0; q < 16; ++q)
                            for (int e = 0; e < 4; ++e)
                                dst[(((size_t)r * PS + k) * 16 + q) * 4 + e] = W[(size_t)(row0 + r) * S + q * 4 * PS + 4 * k + e];
            };
            fill(img, O1, g * U, U);
            fill(img + (size_t)U * S, O3, g * U, U);
            fill(img + (size_t)2 * U * S, O2, g * QW, QW);
            fill(img + (size_t)2 * U * S + (size_t)QW * S, O4, g * QW, QW);
        }
        auto upload = [&](float *&dst, const std::vector<float> &src) -> int {
            if (dst) { (void)hipFree(dst); dst = nullptr; }
            DM_TRY(h, hipMalloc(&dst, src.size() * sizeof(float)));
            DM_TRY(h, hipMemcpy(dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
            return WRNN_OK;
        };
        if ((rc = upload(h->team_w, tw)) || (rc = upload(h->team_lds, tl))) return rc;
        if (!h->mail) {
            DM_TRY(h, hipMalloc(&h->mail, (size_t)WRNN_DM_MAIL_GRANULES * sizeof(unsigned long long)));
            DM_TRY(h, hipMalloc(&h->ctl, 256));
        }
    }
    h->loaded = true;
    return WRNN_OK;
}

int wrnn_dm_set_kernel(wrnn_dm_handle *h, int32_t kernel) {
    if (!h) return WRNN_ERR_INVALID;
    if (kernel < 0 || kernel > 2) return dm_fail(h, WRNN_ERR_INVALID, "kernel must be 0 (auto), 1 (single workgroup) or 2 (team)");
    if (kernel == 2 && !wrnn_dm_team_supported(h->H, h->Q))
        return dm_fail(h, WRNN_ERR_INVALID, "team kernel needs hidden_size in {512,640,768,896} and quantisation in {64,128,192,256}");
    h->kernel = kernel;
    return WRNN_OK;
}

int wrnn_dm_generate(wrnn_dm_handle *h, int64_t seq_len, int32_t noise_mode, uint64_t seed, const float *noise_dev,
                     int32_t *coarse_out_dev, int32_t *fine_out_dev, void *stream) {
    if (!h || seq_len < 0 || !coarse_out_dev || !fine_out_dev) return dm_fail(h, WRNN_ERR_INVALID, "wrnn_dm_generate: bad arguments");
    if (!h->loaded) return dm_fail(h, WRNN_ERR_STATE, "weights not loaded");
    if (noise_mode == WRNN_NOISE_INJECTED && !noise_dev) return dm_fail(h, WRNN_ERR_INVALID, "WRNN_NOISE_INJECTED needs a noise pointer");
    if (noise_mode < 0 || noise_mode > 2) return dm_fail(h, WRNN_ERR_INVALID, "bad noise_mode");
    if (seq_len == 0) return WRNN_OK;
    DM_TRY(h, hipSetDevice(h->device));
    WrnnDmArgs a = h->args;
    a.w = h->wdev; a.H = h->H; a.Q = h->Q; a.seq_len = seq_len; a.noise_mode = noise_mode; a.seed = seed; a.noise = noise_dev;
    a.coarse = coarse_out_dev; a.fine = fine_out_dev;
    const bool team = h->kernel == 2 || (h->kernel == 0 && wrnn_dm_team_supported(h->H, h->Q));
    if (team) {
        hipStream_t s = (hipStream_t)stream;
        DM_TRY(h, hipMemsetAsync(h->mail, 0, (size_t)WRNN_DM_MAIL_GRANULES * sizeof(unsigned long long), s));
        DM_TRY(h, hipMemsetAsync(h->ctl, 0, 256, s));
        WrnnDmTeamArgs ta{};
        ta.base = a; ta.team_w = h->team_w; ta.team_lds = h->team_lds; ta.mail = h->mail; ta.ctl = h->ctl; ta.err = h->ctl + 32;
        // team kernels of one device are ordered behind each other, whatever handle / stream launches them (wavernn_amd.h)
        DM_TRY(h, wrnn_team_gate_enter(h->device, s));
        const hipError_t le = wrnn_launch_dm_team(ta, s);
        const hipError_t ge = wrnn_team_gate_leave(h->device, s);
        DM_TRY(h, le);
        DM_TRY(h, ge);
        return WRNN_OK;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(dm_loop_kernel, dim3(1), dim3(DM_THREADS), 0, (hipStream_t)stream, a);
    DM_TRY(h, hipGetLastError());
    return WRNN_OK;
}

/* device-side error word of the last team-kernel call (0 = ok); blocks until the stream work is done */
int wrnn_dm_sync_status(wrnn_dm_handle *h, void *stream) {
    if (!h) return WRNN_ERR_INVALID;
    DM_TRY(h, hipSetDevice(h->device));
    DM_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    if (!h->ctl) return WRNN_OK;
    unsigned errw = 0;
    DM_TRY(h, hipMemcpy(&errw, h->ctl + 32, sizeof(errw), hipMemcpyDeviceToHost));
    if (errw == WRNN_DEVERR_BUSY)
        return dm_fail(h, WRNN_ERR_BUSY, "the team kernel's 32 workgroups did not all become resident: the GPU is shared with another kernel (retry, or wrnn_dm_set_kernel(h, 1))");
    if (errw) return dm_fail(h, WRNN_ERR_TIMEOUT, "device-side bounded spin gave up (code %u)", errw);
    return WRNN_OK;
}

}  // extern "C"
