// C-ABI entry points of libwavernn_amd.so (see include/wavernn_amd.h).
// Host-side work here is limited to: validating the configuration, repacking
// the reference state_dict into the device layouts the kernels want, and
// enqueueing kernels on the caller's stream.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "wrnn_internal.h"

namespace {

int fail(wrnn_handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                           \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) return fail((h), WRNN_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// Composite taps of the stretch+conv chain (UpsampleNetwork, fatchord_version.py:73-88),
// computed in fp64 by pushing one impulse frame through the exact chain.
std::vector<float> build_ktab(const WrnnDims &d, const int *scales, int n_up, const std::vector<std::vector<double>> &taps) {
    const int P = d.P, HOP = d.HOP, ND = d.ND;
    const int f0 = P + 2, TS = 2 * P + 5;
    std::vector<double> a(TS, 0.0), c;
    a[f0] = 1.0;
    for (int li = 0; li < n_up; ++li) {
        const int s = scales[li], K = 2 * s + 1;
        const size_t n2 = a.size() * (size_t)s;
        c.assign(n2, 0.0);
        for (size_t i = 0; i < a.size(); ++i)
            for (int r = 0; r < s; ++r) c[i * s + r] = a[i];
        a.assign(n2, 0.0);
        for (size_t i = 0; i < n2; ++i) {
            double acc = 0.0;
            for (int k = 0; k < K; ++k) {
                const long jx = (long)i + k - s;
                if (jx >= 0 && (size_t)jx < n2) acc += taps[li][k] * c[jx];
            }
            a[i] = acc;
        }
    }
    std::vector<float> kt((size_t)HOP * ND);
    for (int r = 0; r < HOP; ++r)
        for (int dd = 0; dd < ND; ++dd) kt[(size_t)r * ND + dd] = (float)a[(size_t)HOP * (P + f0 - dd) + r];
    return kt;
}

struct TensorView {
    const wrnn_tensor_desc *t;
    int64_t numel() const {
        int64_t n = 1;
        for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
        return n;
    }
    const float *f() const { return (const float *)t->data; }
};

// Per-device ordering of team-kernel launches (wrnn_internal.h).  One slot per HIP device ordinal: the launch lock and an
// event recorded behind the last team kernel; the next launch (any handle, any stream) waits on that event on the device.
struct TeamGate {
    std::mutex mu;
    hipEvent_t tail = nullptr;
    bool recorded = false;
};
TeamGate g_team_gate[64];

// What keeps the generate loop off the XCD-team kernels for this handle (nullptr: nothing): the co-residency check of wrnn_create, the
// 5-frame upsampling support the team kernels are built for (pad = 2, hop <= 275), or the test hook.
const char *loop_team_obstacle(const wrnn_handle *h) {
    if (h->force_no_teams) return "team kernels disabled by wrnn_debug_force_no_teams (test hook)";
    if (!h->team_ok) return h->team_why.c_str();
    if (h->d.ND != 5 || h->d.HOP > 275) return "team kernels are built for pad=2 (5-frame upsampling support), hop <= 275";
    return nullptr;
}

}  // namespace

hipError_t wrnn_team_gate_enter(int device, hipStream_t s) {
    TeamGate &g = g_team_gate[(unsigned)device % 64u];
    g.mu.lock();
    hipError_t e = hipSuccess;
    if (!g.tail) e = hipEventCreateWithFlags(&g.tail, hipEventDisableTiming);
    if (e == hipSuccess && g.recorded) e = hipStreamWaitEvent(s, g.tail, 0);
    if (e != hipSuccess) g.mu.unlock();
    return e;
}
hipError_t wrnn_team_gate_leave(int device, hipStream_t s) {
    TeamGate &g = g_team_gate[(unsigned)device % 64u];
    hipError_t e = g.tail ? hipEventRecord(g.tail, s) : hipSuccess;
    if (e == hipSuccess) g.recorded = true;
    g.mu.unlock();
    return e;
}

extern "C" {

int32_t wrnn_abi_version(void) { return WRNN_ABI_VERSION; }

int wrnn_create(const wrnn_config *cfg, wrnn_handle **out) {
    if (!cfg || !out) return WRNN_ERR_INVALID;
    *out = nullptr;
    wrnn_handle *h = new wrnn_handle();
    h->cfg = *cfg;
    WrnnDims &d = h->d;
    d.H = cfg->rnn_dims; d.FC = cfg->fc_dims; d.F = cfg->feat_dims; d.C = cfg->compute_dims;
    d.R = cfg->res_out_dims; d.A = cfg->res_out_dims / 4; d.NBLK = cfg->res_blocks; d.P = cfg->pad;
    d.KS = 2 * cfg->pad + 1; d.mode = cfg->mode;
    int hop = 1, reach = 0;
    if (cfg->n_upsample < 1 || cfg->n_upsample > WRNN_MAX_UP) { delete h; return WRNN_ERR_INVALID; }
    for (int i = 0; i < cfg->n_upsample; ++i) hop *= cfg->upsample_factors[i];
    {
        int later = hop;
        for (int i = 0; i < cfg->n_upsample; ++i) { later /= cfg->upsample_factors[i]; reach += cfg->upsample_factors[i] * later; }
    }
    d.HOP = hop;
    d.ND = 2 * cfg->pad + 1;
    if (cfg->mode == WRNN_MODE_RAW) d.NC = 1 << cfg->bits;       // fatchord_version.py:98-99
    else if (cfg->mode == WRNN_MODE_MOL) d.NC = 30;             // :100-101
    else { delete h; return WRNN_ERR_INVALID; }
    *out = h;  // from here on errors are reported through the handle
    // Any constructor dims (fatchord_version.py:93-129) run on the SIMPLE kernel as long as its activation vectors fit a
    // CU's LDS; the team kernels (TEAM2, BATCH) are built for the reference hparams (wavernn_hparams.py:18-57).
    if (d.H < 1 || d.FC < 1 || d.F < 1 || d.C < 1 || d.R < 4 || d.NBLK < 0 || d.P < 0 || cfg->bits < 1 || cfg->bits > 16)
        return fail(h, WRNN_ERR_INVALID, "bad dims");
    if (cfg->res_out_dims % 4 != 0) return fail(h, WRNN_ERR_INVALID, "res_out_dims must be a multiple of 4 (aux split, :109)");
    if (d.H > 1024 || d.C > 1024 || d.R > 1024)
        return fail(h, WRNN_ERR_INVALID, "unsupported dims: rnn_dims, compute_dims, res_out_dims up to 1024");
    h->team_dims = d.H == 512 && d.FC == 512 && d.F == 80 && d.R == 128 && d.C == 128 && d.A == 32;
    if (hop != cfg->hop_length) return fail(h, WRNN_ERR_INVALID, "prod(upsample_factors)=%d != hop_length=%d", hop, cfg->hop_length);
    if (reach > cfg->pad * hop || d.ND > WRNN_KTAB_MAXD)
        return fail(h, WRNN_ERR_INVALID, "upsample edge reach %d exceeds indent %d: composite FIR not shift-invariant", reach, cfg->pad * hop);
    if (wrnn_simple_lds_bytes(d) > 160u * 1024u || ((size_t)(8 + d.KS - 1) * d.F + 16u * d.C) * sizeof(float) > 160u * 1024u)
        return fail(h, WRNN_ERR_INVALID, "dims too large: the activation vectors of one row (%zu bytes) must fit 160 KB of LDS", wrnn_simple_lds_bytes(d));
    HIP_TRY(h, hipSetDevice(cfg->device));
    {
        // team kernels: one team per XCD = 32 CUs (SPX: 256 CUs = 8 teams; a CPX/DPX partition exposes fewer)
        hipDeviceProp_t prop;
        HIP_TRY(h, hipGetDeviceProperties(&prop, cfg->device));
        h->n_teams = prop.multiProcessorCount / 32;
        if (h->n_teams > 8) h->n_teams = 8;
    }
    {
        // The team kernels spin on each other inside one launch: all n_teams * 32 workgroups must be resident at once,
        // one per CU (they take most of a CU's LDS).  Establish that here, loudly, instead of discovering it as a
        // bounded-spin timeout: the runtime must admit >= 1 workgroup per CU for every team kernel.
        h->team_ok = true;
        if (!h->team_dims || d.NC > 1024) { h->team_ok = false; h->team_why = "the team kernels are built for rnn=fc=512, feat=80, compute=res_out=128, n_classes <= 1024"; }
        else if (h->n_teams < 1) { h->team_ok = false; h->team_why = "fewer than 32 CUs visible (one team = the 32 CUs of an XCD)"; }
        // the instantiations THIS handle launches (its mode; the instrumented builds are checked by wrnn_phase_profile)
        int blocks = 0;
        size_t lds = 0;
        if (h->team_ok) {
            hipError_t e = wrnn_team2_occupancy(cfg->mode, false, &blocks, &lds);
            if (e != hipSuccess || blocks < 1) { h->team_ok = false; h->team_why = "loop_team2_kernel cannot be resident (LDS/registers)"; }
        }
        for (int nq = 1; nq <= 2 && h->team_ok; ++nq) {
            hipError_t e = wrnn_batch_occupancy(cfg->mode, nq, false, &blocks, &lds);
            if (e != hipSuccess || blocks < 1) { h->team_ok = false; h->team_why = "loop_batch_kernel cannot be resident (LDS/registers)"; }
        }
        // its own flag (round-4 advisor): a toolchain that allocates this kernel's registers differently must not take TEAM2 and BATCH down with
        // it -- AUTO then runs the one-wave-per-SIMD batch kernel, only an explicit WRNN_KERNEL_BATCH_CS is an error
        h->cs_ok = h->team_ok;
        for (int nq = 1; nq <= wrnn_batch_cs_max_nq(cfg->mode) && h->cs_ok; ++nq) {
            hipError_t e = wrnn_batch_cs_occupancy(cfg->mode, nq, false, &blocks, &lds);
            if (e != hipSuccess || blocks < 1) h->cs_ok = false;
        }
        (void)hipGetLastError();
    }
    for (int i = 0; i < 3; ++i) HIP_TRY(h, hipEventCreate(&h->ev[i]));
    HIP_TRY(h, hipMalloc(&h->err_dev, 64));
    HIP_TRY(h, hipMemset(h->err_dev, 0, 64));
    return WRNN_OK;
}

void wrnn_destroy(wrnn_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->wdev) (void)hipFree(h->wdev);
    if (h->aux_frames) (void)hipFree(h->aux_frames);
    if (h->rows_dev) (void)hipFree(h->rows_dev);
    if (h->order_dev) (void)hipFree(h->order_dev);
    if (h->sched_dev) (void)hipFree(h->sched_dev);
    if (h->prof) (void)hipFree(h->prof);
    if (h->err_dev) (void)hipFree(h->err_dev);
    if (h->team_w) (void)hipFree(h->team_w);
    if (h->team_fc3) (void)hipFree(h->team_fc3);
    if (h->batch_w) (void)hipFree(h->batch_w);
    if (h->batch_fc3) (void)hipFree(h->batch_fc3);
    if (h->batch_wn) (void)hipFree(h->batch_wn);
    if (h->wI0) (void)hipFree(h->wI0);
    if (h->u1) (void)hipFree(h->u1);
    if (h->tab) (void)hipFree(h->tab);
    if (h->cond) (void)hipFree(h->cond);
    if (h->team_state) (void)hipFree(h->team_state);
    if (h->epi_tab) (void)hipFree(h->epi_tab);
    if (h->loss_partial) (void)hipFree(h->loss_partial);
    if (h->train) wrnn_train_state_free(h->train);
    if (h->mail) (void)hipFree(h->mail);
    if (h->ctl) (void)hipFree(h->ctl);
    for (int i = 0; i < 3; ++i)
        if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    delete h;
}

const char *wrnn_last_error(const wrnn_handle *h) { return h ? h->err.c_str() : "null handle"; }

int32_t wrnn_team_info(const wrnn_handle *h, int32_t *n_teams_out, const char **why_not_out) {
    if (!h) return 0;
    const char *no = loop_team_obstacle(h);
    if (n_teams_out) *n_teams_out = h->n_teams;
    if (why_not_out) *why_not_out = no ? no : "";
    return no ? 0 : 1;
}

int wrnn_debug_force_no_teams(wrnn_handle *h, int32_t on) {
    if (!h) return WRNN_ERR_INVALID;
    h->force_no_teams = on != 0;
    return WRNN_OK;
}
int32_t wrnn_n_classes(const wrnn_handle *h) { return h ? h->d.NC : 0; }
int64_t wrnn_loop_weight_bytes(const wrnn_handle *h) { return h ? h->loop_weight_bytes : 0; }

int wrnn_load_weights(wrnn_handle *h, const wrnn_tensor_desc *tensors, int32_t n) {
    if (!h || !tensors) return WRNN_ERR_INVALID;
    const WrnnDims &d = h->d;
    std::map<std::string, TensorView> tv;
    for (int i = 0; i < n; ++i)
        if (tensors[i].name && tensors[i].data) tv[tensors[i].name] = TensorView{&tensors[i]};
    auto need = [&](const std::string &name, std::initializer_list<int64_t> shape, const float **out) -> int {
        auto it = tv.find(name);
        if (it == tv.end()) return fail(h, WRNN_ERR_MISSING_KEY, "state_dict key missing: %s", name.c_str());
        const wrnn_tensor_desc *t = it->second.t;
        if (t->dtype != WRNN_DTYPE_F32) return fail(h, WRNN_ERR_INVALID, "%s: expected float32", name.c_str());
        if ((size_t)t->ndim != shape.size()) return fail(h, WRNN_ERR_INVALID, "%s: expected %d dimensions, got %d", name.c_str(), (int)shape.size(), (int)t->ndim);
        int di = 0;
        for (auto s : shape) {
            if (t->shape[di] != s) return fail(h, WRNN_ERR_INVALID, "%s: dimension %d is %lld, expected %lld", name.c_str(), di, (long long)t->shape[di], (long long)s);
            ++di;
        }
        *out = it->second.f();
        return WRNN_OK;
    };
    const int H = d.H, FC = d.FC, F = d.F, A = d.A, C = d.C, R = d.R, KS = d.KS, NC = d.NC, NB = d.NBLK;
    const int IN_I = 1 + F + A;
#define NEED(name, out, ...) do { int rc__ = need(name, {__VA_ARGS__}, &out); if (rc__) return rc__; } while (0)

    // ---- layout ------------------------------------------------------------
    WrnnPacked &o = h->off;
    size_t cur = 0;
    auto take = [&](size_t nfl) { size_t at = cur; cur += (nfl + 63) & ~(size_t)63; return at; };  // 256-B aligned blocks
    o.conv_in_t = take((size_t)F * KS * C); o.conv_in_b = take(C);
    o.res_w1_t = take((size_t)NB * C * C); o.res_b1 = take((size_t)NB * C);
    o.res_w2_t = take((size_t)NB * C * C); o.res_b2 = take((size_t)NB * C);
    o.conv_out_t = take((size_t)C * R); o.conv_out_b = take(R);
    o.ktab = take((size_t)d.HOP * d.ND);
    o.I_t = take((size_t)IN_I * H); o.I_b = take(H);
    o.r1_wih_t = take((size_t)H * 3 * H); o.r1_whh_t = take((size_t)H * 3 * H);
    o.r1_bih = take(3 * H); o.r1_bhh = take(3 * H);
    o.r2_wih_t = take((size_t)(H + A) * 3 * H); o.r2_whh_t = take((size_t)H * 3 * H);
    o.r2_bih = take(3 * H); o.r2_bhh = take(3 * H);
    o.fc1_t = take((size_t)(H + A) * FC); o.fc1_b = take(FC);
    o.fc2_t = take((size_t)(FC + A) * FC); o.fc2_b = take(FC);
    o.fc3_t = take((size_t)FC * NC); o.fc3_b = take(NC);
    o.total = cur;
    std::vector<float> pk(o.total, 0.0f);

    // ---- prologue: fold BatchNorm1d(eval, eps=1e-5) into the preceding conv ---
    auto bn_fold = [&](const std::string &prefix, std::vector<double> &scale, std::vector<double> &shift) -> int {
        const float *g, *b, *m, *v;
        NEED(prefix + ".weight", g, C); NEED(prefix + ".bias", b, C);
        NEED(prefix + ".running_mean", m, C); NEED(prefix + ".running_var", v, C);
        scale.resize(C); shift.resize(C);
        for (int c = 0; c < C; ++c) {
            const double inv = 1.0 / std::sqrt((double)v[c] + 1e-5);
            scale[c] = (double)g[c] * inv;
            shift[c] = (double)b[c] - (double)m[c] * scale[c];
        }
        return WRNN_OK;
    };
    std::vector<double> sc, sh;
    {
        const float *wci;
        NEED("upsample.resnet.conv_in.weight", wci, C, F, KS);
        if (int rc = bn_fold("upsample.resnet.batch_norm", sc, sh)) return rc;
        for (int c = 0; c < C; ++c) {
            for (int f = 0; f < F; ++f)
                for (int k = 0; k < KS; ++k)
                    pk[o.conv_in_t + (size_t)(f * KS + k) * C + c] = (float)((double)wci[((size_t)c * F + f) * KS + k] * sc[c]);
            pk[o.conv_in_b + c] = (float)sh[c];
        }
    }
    for (int l = 0; l < NB; ++l) {
        const std::string p = "upsample.resnet.layers." + std::to_string(l);
        const float *w1, *w2;
        NEED(p + ".conv1.weight", w1, C, C, 1); NEED(p + ".conv2.weight", w2, C, C, 1);
        if (int rc = bn_fold(p + ".batch_norm1", sc, sh)) return rc;
        for (int co = 0; co < C; ++co) {
            for (int ci = 0; ci < C; ++ci) pk[o.res_w1_t + ((size_t)l * C + ci) * C + co] = (float)((double)w1[(size_t)co * C + ci] * sc[co]);
            pk[o.res_b1 + (size_t)l * C + co] = (float)sh[co];
        }
        if (int rc = bn_fold(p + ".batch_norm2", sc, sh)) return rc;
        for (int co = 0; co < C; ++co) {
            for (int ci = 0; ci < C; ++ci) pk[o.res_w2_t + ((size_t)l * C + ci) * C + co] = (float)((double)w2[(size_t)co * C + ci] * sc[co]);
            pk[o.res_b2 + (size_t)l * C + co] = (float)sh[co];
        }
    }
    {
        const float *wo, *bo;
        NEED("upsample.resnet.conv_out.weight", wo, R, C, 1); NEED("upsample.resnet.conv_out.bias", bo, R);
        for (int r = 0; r < R; ++r) {
            for (int c = 0; c < C; ++c) pk[o.conv_out_t + (size_t)c * R + r] = wo[(size_t)r * C + c];
            pk[o.conv_out_b + r] = bo[r];
        }
    }
    {
        std::vector<std::vector<double>> taps(h->cfg.n_upsample);
        for (int li = 0; li < h->cfg.n_upsample; ++li) {
            const int s = h->cfg.upsample_factors[li];
            const float *tw;
            NEED("upsample.up_layers." + std::to_string(2 * li + 1) + ".weight", tw, 1, 1, 1, 2 * s + 1);
            taps[li].assign(tw, tw + 2 * s + 1);
        }
        std::vector<float> kt = build_ktab(d, h->cfg.upsample_factors, h->cfg.n_upsample, taps);
        std::memcpy(&pk[o.ktab], kt.data(), kt.size() * sizeof(float));
    }
    // ---- loop parameters: transpose to [in][out] ---------------------------------
    auto put_t = [&](const std::string &name, size_t at, int rows, int cols) -> int {
        const float *src;
        NEED(name, src, rows, cols);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) pk[at + (size_t)c * rows + r] = src[(size_t)r * cols + c];
        return WRNN_OK;
    };
    auto put_v = [&](const std::string &name, size_t at, int nel) -> int {
        const float *src;
        NEED(name, src, nel);
        std::memcpy(&pk[at], src, (size_t)nel * sizeof(float));
        return WRNN_OK;
    };
    int rc;
    if ((rc = put_t("I.weight", o.I_t, H, IN_I)) || (rc = put_v("I.bias", o.I_b, H)) ||
        (rc = put_t("rnn1.weight_ih_l0", o.r1_wih_t, 3 * H, H)) || (rc = put_t("rnn1.weight_hh_l0", o.r1_whh_t, 3 * H, H)) ||
        (rc = put_v("rnn1.bias_ih_l0", o.r1_bih, 3 * H)) || (rc = put_v("rnn1.bias_hh_l0", o.r1_bhh, 3 * H)) ||
        (rc = put_t("rnn2.weight_ih_l0", o.r2_wih_t, 3 * H, H + A)) || (rc = put_t("rnn2.weight_hh_l0", o.r2_whh_t, 3 * H, H)) ||
        (rc = put_v("rnn2.bias_ih_l0", o.r2_bih, 3 * H)) || (rc = put_v("rnn2.bias_hh_l0", o.r2_bhh, 3 * H)) ||
        (rc = put_t("fc1.weight", o.fc1_t, FC, H + A)) || (rc = put_v("fc1.bias", o.fc1_b, FC)) ||
        (rc = put_t("fc2.weight", o.fc2_t, FC, FC + A)) || (rc = put_v("fc2.bias", o.fc2_b, FC)) ||
        (rc = put_t("fc3.weight", o.fc3_t, NC, FC)) || (rc = put_v("fc3.bias", o.fc3_b, NC)))
        return rc;
#undef NEED
    // hot-loop parameter bytes as the reference counts them: weights + biases, fp32
    h->loop_weight_bytes = 4LL * ((int64_t)H * IN_I + H + 2LL * 3 * H * H + 2LL * 3 * H + 3LL * H * (H + A) + 3LL * H * H + 2LL * 3 * H +
                                  (int64_t)FC * (H + A) + FC + (int64_t)FC * (FC + A) + FC + (int64_t)NC * FC + NC);
    if (h->team_dims && NC <= 1024) {
    // ---- team kernel layouts (loop_team2.hip): register-resident slices per (WG g, thread) -----
    // thread tid = wave*64 + r4*16 + q of WG g owns unit u = 16g + 4*wave + r4, columns 32q..32q+31
    const int TT = WRNN_TEAM_THREADS;
    std::vector<float> tw((size_t)32 * WRNN_TEAM_NWREG * TT), tf3((size_t)32 * 16384, 0.0f), vwI0(H), vu1(3 * H);
    {
        const float *whh1 = tv["rnn1.weight_hh_l0"].f(), *wih2 = tv["rnn2.weight_ih_l0"].f(), *whh2 = tv["rnn2.weight_hh_l0"].f();
        const float *wfc1 = tv["fc1.weight"].f(), *wfc2 = tv["fc2.weight"].f(), *wfc3 = tv["fc3.weight"].f();
        const float *wI = tv["I.weight"].f(), *wih1 = tv["rnn1.weight_ih_l0"].f();
        for (int g = 0; g < 32; ++g)
            for (int tid = 0; tid < TT; ++tid) {
                const int wave = tid >> 6, lane = tid & 63, r4 = lane >> 4, q = lane & 15;
                const int u = 16 * g + 4 * wave + r4;
                auto at = [&](int i) -> float & { return tw[((size_t)g * WRNN_TEAM_NWREG + i) * TT + tid]; };
                for (int gate = 0; gate < 3; ++gate)
                    for (int c = 0; c < 32; ++c) {
                        const int row = gate * H + u, col = 32 * q + c;
                        at(gate * 32 + c) = whh1[(size_t)row * H + col];
                        at(96 + gate * 32 + c) = wih2[(size_t)row * (H + A) + col];
                        at(192 + gate * 32 + c) = whh2[(size_t)row * H + col];
                    }
                for (int c = 0; c < 32; ++c) {
                    at(288 + c) = wfc2[(size_t)u * (FC + A) + 32 * q + c];
                    at(320 + c) = wfc1[(size_t)u * (H + A) + 32 * q + c];
                }
                // fc3 LDS image: [(wave*2 + rs)*8 + k][lane][e] <- W3[32g + 8 wave + 2 r4 + rs][32q + 4k + e]
                for (int rs = 0; rs < 2; ++rs) {
                    const int row = 32 * g + 8 * wave + 2 * r4 + rs;
                    if (row >= NC) continue;
                    for (int k = 0; k < 8; ++k)
                        for (int e = 0; e < 4; ++e)
                            tf3[(size_t)g * 16384 + ((size_t)((wave * 2 + rs) * 8 + k) * 64 + lane) * 4 + e] =
                                wfc3[(size_t)row * FC + 32 * q + 4 * k + e];
                }
            }
        for (int j = 0; j < H; ++j) vwI0[j] = wI[(size_t)j * IN_I];
        for (int r = 0; r < 3 * H; ++r) {   // u = W_ih1 . W_I[:,0]
            double acc = 0.0;
            for (int j = 0; j < H; ++j) acc += (double)wih1[(size_t)r * H + j] * (double)wI[(size_t)j * IN_I];
            vu1[r] = (float)acc;
        }
    }
    // ---- batch kernel layouts (loop_batch.hip): MFMA 4x4x1 A-operand images.  Lane (kp = lane>>2, i = lane&3) of wave
    // wl of WG g holds W[row(16g + 4wl + i)][k], k = 64 S + 16 e + kp for register slab s = 4 S + e.
    std::vector<float> bw((size_t)32 * 4 * 320 * 64, 0.0f), bf3((size_t)32 * 16384, 0.0f), bwn((size_t)32 * 8192, 0.0f);
    {
        const float *whh1 = tv["rnn1.weight_hh_l0"].f(), *wih2 = tv["rnn2.weight_ih_l0"].f(), *whh2 = tv["rnn2.weight_hh_l0"].f();
        const float *wfc1 = tv["fc1.weight"].f(), *wfc2 = tv["fc2.weight"].f(), *wfc3 = tv["fc3.weight"].f();
        for (int g = 0; g < 32; ++g)
            for (int wl = 0; wl < 4; ++wl)
                for (int lane = 0; lane < 64; ++lane) {
                    const int kp = lane >> 2, i = lane & 3;
                    const int u = 16 * g + 4 * wl + i;
                    // register r of the wave: W_ih2 r,z,n [0,96) | W_hh1 r,z,n [96,192) | W_hh2 r,z [192,256) | fc1 [256,288) | fc2 [288,320)
                    auto at = [&](int r) -> float & { return bw[(((size_t)g * 4 + wl) * 320 + r) * 64 + lane]; };
                    for (int s2 = 0; s2 < 32; ++s2) {
                        const int k = 64 * (s2 >> 2) + 16 * (s2 & 3) + kp;
                        for (int gate = 0; gate < 3; ++gate) {
                            const size_t row = (size_t)gate * H + u;
                            at(gate * 32 + s2) = wih2[row * (H + A) + k];
                            at(96 + gate * 32 + s2) = whh1[row * H + k];
                            if (gate < 2) at(192 + gate * 32 + s2) = whh2[row * H + k];
                            else bwn[(size_t)g * 8192 + ((size_t)(wl * 8 + (s2 >> 2)) * 64 + lane) * 4 + (s2 & 3)] = whh2[row * H + k];   // LDS image
                        }
                        at(256 + s2) = wfc1[(size_t)u * (H + A) + k];
                        at(288 + s2) = wfc2[(size_t)u * (FC + A) + k];
                        for (int set = 0; set < 2; ++set) {
                            const int row = 32 * g + 8 * wl + 4 * set + i;
                            if (row < NC)
                                bf3[(size_t)g * 16384 + ((size_t)((wl * 2 + set) * 8 + (s2 >> 2)) * 64 + lane) * 4 + (s2 & 3)] = wfc3[(size_t)row * FC + k];
                        }
                    }
                }
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    auto upload = [&](float *&dst, const std::vector<float> &src) -> int {
        if (dst) { (void)hipFree(dst); dst = nullptr; }
        HIP_TRY(h, hipMalloc(&dst, src.size() * sizeof(float)));
        HIP_TRY(h, hipMemcpy(dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
        return WRNN_OK;
    };
    if ((rc = upload(h->team_w, tw)) || (rc = upload(h->team_fc3, tf3)) || (rc = upload(h->wI0, vwI0)) || (rc = upload(h->u1, vu1)) ||
        (rc = upload(h->batch_w, bw)) || (rc = upload(h->batch_fc3, bf3)) || (rc = upload(h->batch_wn, bwn))) return rc;
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!h->mail) {
        HIP_TRY(h, hipMalloc(&h->mail, (size_t)8 * WRNN_MAIL_GRANULES_MAX * sizeof(unsigned long long)));
        HIP_TRY(h, hipMalloc(&h->ctl, 128));
    }
    if (h->wdev) { (void)hipFree(h->wdev); h->wdev = nullptr; }
    HIP_TRY(h, hipMalloc(&h->wdev, o.total * sizeof(float)));
    HIP_TRY(h, hipMemcpy(h->wdev, pk.data(), o.total * sizeof(float), hipMemcpyHostToDevice));
    h->loaded = true;
    return WRNN_OK;
}

static int ensure_aux(wrnn_handle *h, int B, int T) {
    const size_t need = (size_t)B * T * h->d.R;
    if (need > h->aux_cap) {
        if (h->aux_frames) (void)hipFree(h->aux_frames);
        h->aux_frames = nullptr; h->aux_cap = 0;
        HIP_TRY(h, hipMalloc(&h->aux_frames, need * sizeof(float)));
        h->aux_cap = need;
    }
    return WRNN_OK;
}

int wrnn_conditioning(wrnn_handle *h, const float *mels_dev, int32_t B, int32_t T, int32_t mels_padded, float *up_dev, float *aux_dev, void *stream) {
    if (!h || !mels_dev || B < 1 || T < 1) return fail(h, WRNN_ERR_INVALID, "wrnn_conditioning: bad arguments");
    if (!h->loaded) return fail(h, WRNN_ERR_STATE, "weights not loaded");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    if (int rc = ensure_aux(h, B, T)) return rc;
    const int mel_T = mels_padded ? T + 2 * h->d.P : T, mel_off = mels_padded ? h->d.P : 0;
    HIP_TRY(h, wrnn_launch_resnet(h, mels_dev, B, T, mel_T, mel_off, h->aux_frames, s));
    if (up_dev || aux_dev) HIP_TRY(h, wrnn_launch_materialize(h, mels_dev, h->aux_frames, B, T, mel_T, mel_off, up_dev, aux_dev, s));
    return WRNN_OK;
}

int wrnn_plan(wrnn_handle *h, int32_t B, int32_t T, int32_t batched, int32_t target, int32_t overlap, int32_t *rows_out, int64_t *steps_out) {
    if (!h || B < 1 || T < 1) return fail(h, WRNN_ERR_INVALID, "wrnn_plan: bad arguments");
    const int64_t total = (int64_t)T * h->d.HOP;
    if (!batched) {
        if (rows_out) *rows_out = B;
        if (steps_out) *steps_out = total;
        return WRNN_OK;
    }
    // fold_with_overlap (fatchord_version.py:293-340) indexes folded[i] = x[:, start:end, :] with a
    // batch-1 x; any other batch size raises in the reference.
    if (B != 1) return fail(h, WRNN_ERR_INVALID, "batched generation requires a single utterance (fold_with_overlap)");
    if (target < 1 || overlap < 0) return fail(h, WRNN_ERR_INVALID, "bad target/overlap");
    // Python floor division like fatchord_version.py:319 (a clip shorter than `overlap` gives -1 -> 0 folds -> error)
    const int64_t fold_den = (int64_t)target + overlap, fold_num = total - overlap;
    int64_t num_folds = fold_num / fold_den;
    if (fold_num % fold_den != 0 && fold_num < 0) --num_folds;
    const int64_t extended = num_folds * ((int64_t)overlap + target) + overlap;
    if (total - extended != 0) num_folds += 1;
    if (num_folds < 1) return fail(h, WRNN_ERR_INVALID, "sequence shorter than one fold");
    if (rows_out) *rows_out = (int32_t)num_folds;
    if (steps_out) *steps_out = (int64_t)target + 2LL * overlap;
    return WRNN_OK;
}

int wrnn_generate(wrnn_handle *h, const float *mels_dev, int32_t B, int32_t T, int32_t batched, int32_t target, int32_t overlap,
                  const wrnn_sample_opts *opts, int32_t *labels_out_dev, float *samples_out_dev, void *stream) {
    if (!h || !mels_dev || !opts || !samples_out_dev) return fail(h, WRNN_ERR_INVALID, "wrnn_generate: bad arguments");
    if (opts->struct_size != sizeof(wrnn_sample_opts))
        return fail(h, WRNN_ERR_INVALID, "wrnn_sample_opts.struct_size is %u, this library (ABI %d) expects %zu: caller built against another revision of wavernn_amd.h",
                    opts->struct_size, WRNN_ABI_VERSION, sizeof(wrnn_sample_opts));
    if (!h->loaded) return fail(h, WRNN_ERR_STATE, "weights not loaded");
    if (opts->frames_dev && batched) return fail(h, WRNN_ERR_INVALID, "frames_dev (ragged batch) is for unbatched calls: folds of one utterance have one length");
    if (opts->batch_rows < 0 || opts->batch_rows > WRNN_BATCH_MAX_ROWS) return fail(h, WRNN_ERR_INVALID, "batch_rows must be 0 (default) or 1..%d", WRNN_BATCH_MAX_ROWS);
    if (opts->team2_segment < 0) return fail(h, WRNN_ERR_INVALID, "team2_segment must be >= 0");
    const WrnnDims &d = h->d;
    int32_t rows = 0;
    int64_t steps = 0;
    if (int rc = wrnn_plan(h, B, T, batched, target, overlap, &rows, &steps)) return rc;
    if (opts->noise_mode == WRNN_NOISE_INJECTED && (!opts->noise1_dev || (d.mode == WRNN_MODE_MOL && !opts->noise2_dev)))
        return fail(h, WRNN_ERR_INVALID, "WRNN_NOISE_INJECTED needs noise pointers");
    if (opts->noise_mode == WRNN_NOISE_ARGMAX && d.mode != WRNN_MODE_RAW)
        return fail(h, WRNN_ERR_INVALID, "WRNN_NOISE_ARGMAX is RAW-only");
    if (opts->noise_mode < 0 || opts->noise_mode > 2) return fail(h, WRNN_ERR_INVALID, "bad noise_mode");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;

    // rows table, built on the device: nothing is staged on the host, the call never waits for the stream
    if ((size_t)rows > h->rows_cap) {
        if (h->rows_dev) (void)hipFree(h->rows_dev);
        h->rows_dev = nullptr; h->rows_cap = 0;
        if (h->order_dev) (void)hipFree(h->order_dev);
        if (h->sched_dev) (void)hipFree(h->sched_dev);
        h->order_dev = h->sched_dev = nullptr;
        HIP_TRY(h, hipMalloc(&h->rows_dev, (size_t)rows * sizeof(WrnnRow)));
        HIP_TRY(h, hipMalloc(&h->order_dev, (size_t)rows * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc(&h->sched_dev, ((size_t)rows + 64) * sizeof(int32_t)));
        h->rows_cap = rows;
    }
    const int sched_teams = h->n_teams < 1 ? 1 : h->n_teams;
    const int n_slots = (rows + sched_teams - 1) / sched_teams * sched_teams;
    HIP_TRY(h, wrnn_launch_rows(h->rows_dev, h->order_dev, h->sched_dev, rows, sched_teams, batched, (long)target + overlap, (long)steps,
                                opts->frames_dev, T, d.HOP, s));
    const int snake = opts->frames_dev ? 1 : 0;
    unsigned long long *const prof = h->prof_on ? h->prof : nullptr;
    if (int rc = ensure_aux(h, B, T)) return rc;
    HIP_TRY(h, hipMemsetAsync(h->err_dev, 0, 64, s));

    HIP_TRY(h, hipEventRecord(h->ev[0], s));
    const int mel_T = opts->mels_padded ? T + 2 * d.P : T, mel_off = opts->mels_padded ? d.P : 0;
    HIP_TRY(h, wrnn_launch_resnet(h, mels_dev, B, T, mel_T, mel_off, h->aux_frames, s));
    HIP_TRY(h, hipEventRecord(h->ev[1], s));

    WrnnLoopArgs a{};
    a.w = h->wdev; a.off = h->off; a.d = d; a.mels = mels_dev; a.mel_T = mel_T; a.mel_off = mel_off; a.aux_frames = h->aux_frames; a.rows = h->rows_dev;
    a.n_rows = rows; a.T = T; a.total_len = (int64_t)T * d.HOP; a.steps = steps;
    a.noise_mode = opts->noise_mode; a.seed = opts->seed; a.noise1 = opts->noise1_dev; a.noise2 = opts->noise2_dev;
    a.x_forced = opts->x_forced_dev; a.x_init = opts->x_init_dev; a.logits_out = opts->logits_out_dev; a.labels_out = labels_out_dev;
    a.samples_out = samples_out_dev; a.err = h->err_dev;
    int kernel = opts->kernel;
    int launches = 1;
    // what the team kernels (TEAM2, BATCH) need: co-residency (checked in wrnn_create), the 5-frame upsampling support
    // of pad = 2, hop <= 275, and 1024 or fewer classes.  AUTO falls back to the any-shape kernel otherwise;
    // an explicit request for a team kernel that cannot run is an error.
    const char *team_no = loop_team_obstacle(h);
    if (kernel == WRNN_KERNEL_AUTO) {
        if (team_no) kernel = WRNN_KERNEL_SIMPLE;
        // one row per XCD team: the latency kernel; more rows: the batch step with critical / shadow wave roles (round 4: 7.5 against 6.85
        // Msamples/s at RAW B = 64, 5.6 against 5.3 at MOL B = 32; WRNN_KERNEL_BATCH stays available by name)
        else kernel = rows <= h->n_teams ? WRNN_KERNEL_TEAM2 : (h->cs_ok ? WRNN_KERNEL_BATCH_CS : WRNN_KERNEL_BATCH);
    }
    if (kernel == WRNN_KERNEL_BATCH_CS && !team_no && !h->cs_ok) team_no = "loop_batch_cs_kernel cannot be resident (LDS/registers); WRNN_KERNEL_BATCH can";
    if (kernel == WRNN_KERNEL_SIMPLE) {
        HIP_TRY(h, wrnn_launch_loop_simple(a, s));
    } else if (kernel == WRNN_KERNEL_TEAM2 || kernel == WRNN_KERNEL_BATCH || kernel == WRNN_KERNEL_BATCH_CS) {
        const bool batch_family = kernel != WRNN_KERNEL_TEAM2;
        if (team_no) return fail(h, WRNN_ERR_INVALID, "%s", team_no);
        // conditioning pushed through the linear layers it feeds (once per call)
        const int H = d.H, FC = d.FC, F = d.F, A = d.A, R = d.R, P = d.P;
        const int TP = T + 2 * P, T1 = T + 1;
        const size_t nCM = (size_t)B * TP * H, nCA = (size_t)B * T1 * H, nVM = (size_t)B * TP * 3 * H, nVA = (size_t)B * T1 * 3 * H;
        const size_t nC2 = (size_t)B * T1 * 3 * H, nC3 = (size_t)B * T1 * FC, nC4 = (size_t)B * T1 * FC;
        const size_t nREC = (size_t)B * T1 * H * (batch_family ? 32 : 28);
        const size_t need = nCM + nCA + nVM + nVA + nC2 + nC3 + nC4 + nREC;
        if (need > h->tab_cap) {
            if (h->tab) (void)hipFree(h->tab);
            h->tab = nullptr; h->tab_cap = 0;
            HIP_TRY(h, hipMalloc(&h->tab, need * sizeof(float)));
            h->tab_cap = need;
        }
        float *tCM = h->tab, *tCA = tCM + nCM, *tVM = tCA + nCA, *tVA = tVM + nVM, *tC2 = tVA + nVA, *tC3 = tC2 + nC2, *tC4 = tC3 + nC3, *tREC = tC4 + nC4;
        const float *w = h->wdev;
        const WrnnPacked &o = h->off;
        HIP_TRY(h, wrnn_launch_frame_linear(1, mels_dev, (size_t)F * mel_T, 0, 0, w + o.I_t + (size_t)1 * H, H, nullptr, tCM, (size_t)TP * H, TP, F, H, B, mel_T, P - mel_off, s));
        HIP_TRY(h, wrnn_launch_frame_linear(0, h->aux_frames, (size_t)T * R, R, T, w + o.I_t + (size_t)(1 + F) * H, H, w + o.I_b, tCA, (size_t)T1 * H, T1, A, H, B, T, P, s));
        HIP_TRY(h, wrnn_launch_frame_linear(0, tCM, (size_t)TP * H, H, TP, w + o.r1_wih_t, 3 * H, nullptr, tVM, (size_t)TP * 3 * H, TP, H, 3 * H, B, T, P, s));
        HIP_TRY(h, wrnn_launch_frame_linear(0, tCA, (size_t)T1 * H, H, T1, w + o.r1_wih_t, 3 * H, w + o.r1_bih, tVA, (size_t)T1 * 3 * H, T1, H, 3 * H, B, T, P, s));
        HIP_TRY(h, wrnn_launch_frame_linear(0, h->aux_frames + A, (size_t)T * R, R, T, w + o.r2_wih_t + (size_t)H * 3 * H, 3 * H, w + o.r2_bih, tC2, (size_t)T1 * 3 * H, T1, A, 3 * H, B, T, P, s));
        HIP_TRY(h, wrnn_launch_frame_linear(0, h->aux_frames + 2 * A, (size_t)T * R, R, T, w + o.fc1_t + (size_t)H * FC, FC, w + o.fc1_b, tC3, (size_t)T1 * FC, T1, A, FC, B, T, P, s));
        HIP_TRY(h, wrnn_launch_frame_linear(0, h->aux_frames + 3 * A, (size_t)T * R, R, T, w + o.fc2_t + (size_t)FC * FC, FC, w + o.fc2_b, tC4, (size_t)T1 * FC, T1, A, FC, B, T, P, s));
        const size_t mail_bytes = (size_t)8 * WRNN_MAIL_GRANULES_MAX * sizeof(unsigned long long);
        if (batch_family) {
            // R = 4 * nq rows per XCD team in lock-step on the matrix cores (loop_batch.hip); the rows are spread evenly over
            // the teams first (rpb rows per batch), a team runs ceil(batches / n_teams) batches one after the other
            HIP_TRY(h, wrnn_launch_pack_records32(tCM, tCA, tVM, tVA, tC2, tC3, tC4, tREC, B, T, P, s));
            int rpb = (rows + h->n_teams - 1) / h->n_teams;
            if (rpb > WRNN_BATCH_MAX_ROWS) rpb = WRNN_BATCH_MAX_ROWS;
            if (opts->batch_rows > 0) rpb = opts->batch_rows;
            const bool cs = kernel == WRNN_KERNEL_BATCH_CS;   // critical / shadow wave roles (loop_batch_cs.hip)
            if (cs && rpb > 4 * wrnn_batch_cs_max_nq(d.mode)) rpb = 4 * wrnn_batch_cs_max_nq(d.mode);
            WrnnBatchArgs ba{};
            ba.w = w; ba.off = o; ba.d = d; ba.batch_w = h->batch_w; ba.batch_fc3 = h->batch_fc3; ba.batch_wn = h->batch_wn; ba.wI0 = h->wI0; ba.u1 = h->u1;
            ba.tabREC32 = tREC; ba.rows = h->rows_dev; ba.order = h->order_dev; ba.snake = snake; ba.n_rows = rows; ba.n_teams = h->n_teams; ba.nq = rpb <= 4 ? 1 : 2; ba.rpb = rpb;
            ba.T = T; ba.total_len = a.total_len; ba.steps = steps;
            ba.noise_mode = a.noise_mode; ba.seed = a.seed; ba.noise1 = a.noise1; ba.noise2 = a.noise2; ba.x_forced = a.x_forced; ba.x_init = a.x_init;
            ba.logits_out = a.logits_out; ba.labels_out = a.labels_out; ba.samples_out = a.samples_out;
            ba.mail = h->mail; ba.ctl = h->ctl; ba.err = h->err_dev; ba.prof = prof;
            if (prof) HIP_TRY(h, hipMemsetAsync(h->prof, 0, 8 * WRNN_PROF_SLOTS * sizeof(unsigned long long), s));
            HIP_TRY(h, hipEventRecord(h->ev[1], s));  // tables and records are prologue work
            // team kernels of one device run one after the other, whatever handle / stream launches them (wavernn_amd.h); the
            // mailbox reset belongs inside the gate: the handle's previous team kernel may still be reading it
            HIP_TRY(h, wrnn_team_gate_enter(h->cfg.device, s));
            hipError_t le = hipMemsetAsync(h->mail, 0, mail_bytes, s);
            if (le == hipSuccess) le = hipMemsetAsync(h->ctl, 0, 128, s);
            if (le == hipSuccess) le = cs ? wrnn_launch_loop_batch_cs(ba, s) : wrnn_launch_loop_batch(ba, s);
            const hipError_t ge = wrnn_team_gate_leave(h->cfg.device, s);
            HIP_TRY(h, le);
            HIP_TRY(h, ge);
            h->prof_div = (double)steps * ((((rows + rpb - 1) / rpb) + h->n_teams - 1) / h->n_teams);
        } else {
        HIP_TRY(h, wrnn_launch_pack_records(tCM, tCA, tVM, tVA, tREC, B, T, P, s));
        WrnnTeamArgs ta{};
        ta.w = w; ta.off = o; ta.d = d; ta.team_w = h->team_w; ta.team_fc3 = h->team_fc3; ta.wI0 = h->wI0; ta.u1 = h->u1;
        ta.tabREC = tREC; ta.tabCOND = nullptr; ta.tabC2 = tC2; ta.tabC3 = tC3; ta.tabC4 = tC4;
        ta.rows = h->rows_dev; ta.sched = h->sched_dev; ta.n_slots = n_slots; ta.ragged = snake; ta.n_rows = rows; ta.n_teams = h->n_teams; ta.T = T; ta.total_len = a.total_len; ta.steps = steps;
        ta.seg0 = 0; ta.seg_len = steps; ta.state = nullptr;
        ta.noise_mode = a.noise_mode; ta.seed = a.seed; ta.noise1 = a.noise1; ta.noise2 = a.noise2; ta.x_forced = a.x_forced; ta.x_init = a.x_init;
        ta.logits_out = a.logits_out; ta.labels_out = a.labels_out; ta.samples_out = a.samples_out;
        ta.mail = h->mail; ta.ctl = h->ctl; ta.err = h->err_dev; ta.prof = prof;
        {
            // The phase-A conditioning is streamed from HBM (8 KB per row and step).  A row is generated in segments,
            // one stream chunk + one loop launch each, sized so that the chunk (~64 MB over all rows) is still resident
            // in the memory-side cache when the loop reads it: against a stream written once for the whole clip
            // (903 MB for 5 s of audio, read back from DRAM) this is 6.5 % faster at B=1, bounds the scratch to
            // rows x seg x 8 KB, and costs one relaunch (~40 us) per segment.  Segment lengths are multiples of 32
            // steps (the shadow waves regenerate their Philox state on those boundaries).
            int64_t seg = ((int64_t)(64u << 20) / ((int64_t)rows * H * 4 * (int64_t)sizeof(float))) & ~(int64_t)31;
            if (seg > 16384) seg = 16384;
            if (seg < 2048) seg = 2048;
            if (opts->team2_segment > 0) { seg = (int64_t)opts->team2_segment & ~(int64_t)31; if (seg < 32) seg = 32; }
            if (seg > steps) seg = steps;
            const size_t nCOND = (size_t)rows * (size_t)seg * H * 4;
            if (nCOND > h->cond_cap) {
                if (h->cond) (void)hipFree(h->cond);
                h->cond = nullptr; h->cond_cap = 0;
                HIP_TRY(h, hipMalloc(&h->cond, nCOND * sizeof(float)));
                h->cond_cap = nCOND;
            }
            const size_t nST = (size_t)rows * WRNN_TEAM_STATE_FLOATS;
            if (nST > h->team_state_cap) {
                if (h->team_state) (void)hipFree(h->team_state);
                h->team_state = nullptr; h->team_state_cap = 0;
                HIP_TRY(h, hipMalloc(&h->team_state, nST * sizeof(float)));
                h->team_state_cap = nST;
            }
            HIP_TRY(h, hipEventRecord(h->ev[1], s));  // the tables are prologue work; the stream chunks are timed with the loop
            if (prof) HIP_TRY(h, hipMemsetAsync(h->prof, 0, 8 * WRNN_PROF_SLOTS * sizeof(unsigned long long), s));
            ta.team_w = h->team_w; ta.tabCOND = h->cond; ta.state = h->team_state;
            h->prof_div = (double)steps * ((rows + h->n_teams - 1) / h->n_teams);
            for (int64_t t0 = 0; t0 < steps; t0 += seg) {
                const int64_t len = steps - t0 < seg ? steps - t0 : seg;
                HIP_TRY(h, wrnn_launch_cond_stream(tREC, w + o.ktab, h->rows_dev, h->cond, rows, T, d.HOP, a.total_len, t0, len, s));
                ta.seg0 = t0; ta.seg_len = len;
                HIP_TRY(h, wrnn_team_gate_enter(h->cfg.device, s));   // see the BATCH branch
                hipError_t le = hipMemsetAsync(h->mail, 0, mail_bytes, s);
                if (le == hipSuccess) le = hipMemsetAsync(h->ctl, 0, 128, s);
                if (le == hipSuccess) le = wrnn_launch_loop_team2(ta, s);
                const hipError_t ge = wrnn_team_gate_leave(h->cfg.device, s);
                HIP_TRY(h, le);
                HIP_TRY(h, ge);
                launches = (int)(t0 / seg) + 1;
            }
        }
        }
    } else {
        return fail(h, WRNN_ERR_INVALID, "kernel %d not available", kernel);
    }
    HIP_TRY(h, hipEventRecord(h->ev[2], s));
    h->timing_valid = true;
    h->last.kernel = kernel; h->last.rows = rows; h->last.steps = steps; h->last.launches = launches;
    return WRNN_OK;
}

int wrnn_loss(wrnn_handle *h, const float *y_hat_dev, const void *y_dev, int64_t n_rows, float *loss_out_dev, void *stream) {
    if (!h || !y_hat_dev || !y_dev || !loss_out_dev || n_rows < 1) return fail(h, WRNN_ERR_INVALID, "wrnn_loss: bad arguments");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const size_t nblk = (size_t)(h->d.mode == WRNN_MODE_RAW ? (n_rows + 3) / 4 : (n_rows + 255) / 256);
    if (nblk + 1 > h->loss_cap) {
        if (h->loss_partial) (void)hipFree(h->loss_partial);
        h->loss_partial = nullptr; h->loss_cap = 0;
        HIP_TRY(h, hipMalloc(&h->loss_partial, (nblk + 1) * sizeof(double)));
        h->loss_cap = nblk + 1;
    }
    int *bad = (int *)(h->loss_partial + nblk);
    HIP_TRY(h, hipMemsetAsync(bad, 0, sizeof(double), s));
    HIP_TRY(h, wrnn_launch_loss(h->d.mode, y_hat_dev, y_dev, h->d.NC, (long)n_rows, h->loss_partial, bad, loss_out_dev, s));
    return WRNN_OK;
}

int wrnn_last_timing(wrnn_handle *h, wrnn_timing *out) {
    if (!h) return WRNN_ERR_INVALID;
    if (!h->timing_valid) return fail(h, WRNN_ERR_STATE, "no generate call to time");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipEventSynchronize(h->ev[2]));
    HIP_TRY(h, hipEventElapsedTime(&h->last.prologue_ms, h->ev[0], h->ev[1]));
    HIP_TRY(h, hipEventElapsedTime(&h->last.loop_ms, h->ev[1], h->ev[2]));
    unsigned errw = 0;
    HIP_TRY(h, hipMemcpy(&errw, h->err_dev, sizeof(errw), hipMemcpyDeviceToHost));
    if (out) *out = h->last;
    if (errw == WRNN_DEVERR_BUSY)
        return fail(h, WRNN_ERR_BUSY, "the team kernel's workgroups did not all become resident within its start-up wait: the GPU is shared with another "
                                      "kernel (another process?).  Retry, or use WRNN_KERNEL_SIMPLE");
    if (errw) return fail(h, WRNN_ERR_TIMEOUT, "device-side bounded spin gave up (code %u)", errw);
    return WRNN_OK;
}

int wrnn_phase_profile(wrnn_handle *h, int32_t enable) {
    if (!h) return WRNN_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (enable) {
        // the instrumented instantiations have their own register / LDS footprint: check their residency like wrnn_create does
        int blocks = 0;
        size_t lds = 0;
        if (h->team_ok) {
            hipError_t e = wrnn_team2_occupancy(h->cfg.mode, true, &blocks, &lds);
            for (int nq = 1; nq <= 2 && e == hipSuccess && blocks >= 1; ++nq) e = wrnn_batch_occupancy(h->cfg.mode, nq, true, &blocks, &lds);
            for (int nq = 1; nq <= wrnn_batch_cs_max_nq(h->cfg.mode) && e == hipSuccess && blocks >= 1; ++nq) e = wrnn_batch_cs_occupancy(h->cfg.mode, nq, true, &blocks, &lds);
            (void)hipGetLastError();
            if (e != hipSuccess || blocks < 1) return fail(h, WRNN_ERR_INVALID, "the instrumented team kernels cannot be resident on this device");
        }
        if (!h->prof) HIP_TRY(h, hipMalloc(&h->prof, 8 * WRNN_PROF_SLOTS * sizeof(unsigned long long)));
        HIP_TRY(h, hipMemset(h->prof, 0, 8 * WRNN_PROF_SLOTS * sizeof(unsigned long long)));
    }
    h->prof_on = enable != 0;
    return WRNN_OK;
}

int wrnn_phase_cycles(wrnn_handle *h, double *out) {
    if (!h || !out) return WRNN_ERR_INVALID;
    if (!h->prof_on || !h->prof || !h->timing_valid) return fail(h, WRNN_ERR_STATE, "no instrumented call to report (wrnn_phase_profile(h, 1), then a TEAM2 / BATCH call)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipEventSynchronize(h->ev[2]));
    unsigned long long pr[8 * WRNN_PROF_SLOTS];
    HIP_TRY(h, hipMemcpy(pr, h->prof, sizeof(pr), hipMemcpyDeviceToHost));
    const double n = h->prof_div > 0 ? h->prof_div : 1.0;   // steps x rows (or batches) team 0 ran
    for (int i = 0; i < 8 * WRNN_PROF_SLOTS; ++i) out[i] = (double)pr[i] / n;
    return WRNN_OK;
}

}  // extern "C"
