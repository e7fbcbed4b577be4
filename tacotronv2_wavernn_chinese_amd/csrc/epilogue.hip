// wrnn_epilogue: the float64 tail of generate() on the device (fatchord_version.py:243-258) --
// decode_mu_law (wavernn/utils/dsp.py:98-103, called with from_labels=False), xfade_and_unfold (:342-405),
// the trim to wave_len and the linear fade-out over the last 20 hops (:255-258).
//
// Everything transcendental is tabulated on the HOST in double, following NumPy's evaluation order (linspace =
// k * step + start with the end point pinned; sqrt; pow), so the device kernel is a pure gather: one thread per
// output sample, at most two fold contributions added in fold order, one multiply for the fade-out.  mu-law
// decode has only n_classes distinct inputs (the value fed back is 2k/(n_classes-1) - 1 in fp32), hence a table
// indexed by the label.
#include <cmath>

#include "wrnn_internal.h"

namespace {

__global__ void __launch_bounds__(256)
epilogue_kernel(const float *__restrict__ samples, const int32_t *__restrict__ labels, const double *__restrict__ dec,
                const double *__restrict__ fade_in, const double *__restrict__ fade_out, const double *__restrict__ tail,
                int rows, long steps, int batched, long target, long overlap, long wave_len, long tail_len,
                double *__restrict__ out) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= wave_len) return;
    auto val = [&](long r, long p) -> double {
        const size_t i = (size_t)r * (size_t)steps + (size_t)p;
        return dec ? dec[labels[i]] : (double)samples[i];   // (:243-248)
    };
    double v;
    if (!batched) {
        v = val(0, n);                                       // output = output[0]  (:253)
    } else {
        // unfolded = zeros; unfolded[i*(t+o) : i*(t+o) + t + 2o] += y[i] in fold order (:397-403); fold i's first
        // `overlap` samples carry fade_in, its last `overlap` samples fade_out (:387-388)
        const long st = target + overlap;
        const long i = n / st, p = n - i * st;
        v = 0.0;
        if (i >= 1 && p < overlap) v = __dadd_rn(v, __dmul_rn(val(i - 1, p + st), fade_out[p]));
        if (i < rows) {
            double a = val(i, p);
            if (p < overlap) a = __dmul_rn(a, fade_in[p]);
            v = __dadd_rn(v, a);
        }
    }
    if (n >= wave_len - tail_len) v = __dmul_rn(v, tail[n - (wave_len - tail_len)]);   // (:255-258)
    out[n] = v;
}

// wrnn_epilogue_rows: every row an independent unbatched utterance, grid (ceil(out_stride / 256), rows)
__global__ void __launch_bounds__(256)
epilogue_rows_kernel(const float *__restrict__ samples, const int32_t *__restrict__ labels, const double *__restrict__ dec,
                     const double *__restrict__ tail, long steps, long wave_len, long tail_len, const int32_t *__restrict__ frames,
                     int hop, double *__restrict__ out, long out_stride) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (n >= out_stride) return;
    long wl = wave_len;
    if (frames) {
        wl = ((long)frames[r] - 1) * hop;
        if (wl > wave_len) wl = wave_len;
    }
    double v = 0.0;
    if (wl >= tail_len && n < wl) {
        const size_t i = (size_t)r * (size_t)steps + (size_t)n;
        v = dec ? dec[labels[i]] : (double)samples[i];                                     // (:243-248), row r as :253 takes row 0
        if (n >= wl - tail_len) v = __dmul_rn(v, tail[n - (wl - tail_len)]);               // (:255-258)
    }
    out[(size_t)r * (size_t)out_stride + (size_t)n] = v;
}

// np.linspace(start, stop, num) in float64: arange(num) * step + start, last element = stop
#pragma clang fp contract(off)
void np_linspace(double start, double stop, long num, double *y) {
    if (num <= 0) return;
    const long div = num - 1;
    const double step = div > 0 ? (stop - start) / (double)div : 0.0;
    for (long k = 0; k < num; ++k) {
        const double m = (double)k * step;
        y[k] = m + start;
    }
    if (num > 1) y[num - 1] = stop;
}

}  // namespace

// Host only (no device needed): the float64 tables of the epilogue, in NumPy's evaluation order.
//   dec[k], k < n_classes : decode_mu_law of the fp32 value 2k/(n_classes-1) - 1 the loop feeds back   (dsp.py:98-103, :235)
//   fade_in / fade_out [overlap] : the equal-power crossfade of xfade_and_unfold                         (:374-385)
//   tail [20 * hop] : np.linspace(1, 0, 20 * hop_length)                                               (:256)
extern "C" int wrnn_epilogue_tables(int32_t n_classes, int32_t overlap, int32_t hop, double *dec, double *fade_in,
                                    double *fade_out, double *tail) {
    if (n_classes < 2 || overlap < 0 || hop < 1 || !dec || !tail || (overlap > 0 && (!fade_in || !fade_out))) return WRNN_ERR_INVALID;
    const double mu = (double)(n_classes - 1);
    for (int k = 0; k < n_classes; ++k) {
        const float xf = 2.0f * (float)k / ((float)n_classes - 1.0f) - 1.0f;   // the fp32 value the loop feeds back (:235)
        const double y = (double)xf;
        const double sg = y > 0.0 ? 1.0 : (y < 0.0 ? -1.0 : 0.0);
        dec[k] = sg / mu * (std::pow(1.0 + mu, std::fabs(y)) - 1.0);           // dsp.py:98-103
    }
    if (overlap > 0) {
        const long silence = overlap / 2, fl = overlap - silence;              // (:374-375)
        std::vector<double> t((size_t)fl);
        np_linspace(-1.0, 1.0, fl, t.data());
        for (long k = 0; k < silence; ++k) { fade_in[k] = 0.0; fade_out[k] = 1.0; }
        for (long k = 0; k < fl; ++k) {
            fade_in[silence + k] = std::sqrt(0.5 * (1.0 + t[(size_t)k]));      // (:378-379)
            fade_out[silence + k] = std::sqrt(0.5 * (1.0 - t[(size_t)k]));
        }
    }
    np_linspace(1.0, 0.0, 20L * hop, tail);                                    // (:256)
    return WRNN_OK;
}

// device copy of the host tables [dec NC | fade_in ov | fade_out ov | tail 20*hop] for overlap `ov`, rebuilt when ov changes
static const char *epilogue_tables_on_device(wrnn_handle *h, long ov) {
    const int NC = h->d.NC;
    const long tail_len = 20L * h->d.HOP;
    if (h->epi_overlap == ov && h->epi_tab) return nullptr;
    std::vector<double> tab((size_t)NC + 2 * (size_t)ov + (size_t)tail_len, 0.0);
    double *dec = tab.data(), *fin = dec + NC, *fout = fin + ov, *tail = fout + ov;
    wrnn_epilogue_tables(NC, (int32_t)ov, h->d.HOP, dec, fin, fout, tail);
    if (h->epi_tab) { (void)hipFree(h->epi_tab); h->epi_tab = nullptr; }
    if (hipMalloc(&h->epi_tab, tab.size() * sizeof(double)) != hipSuccess) return "wrnn_epilogue: hipMalloc failed";
    if (hipMemcpy(h->epi_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return "wrnn_epilogue: table upload failed";
    h->epi_overlap = ov;
    return nullptr;
}

extern "C" int wrnn_epilogue_rows(wrnn_handle *h, const float *samples_dev, const int32_t *labels_dev, int32_t rows, int64_t steps,
                                  int32_t mu_law, int64_t wave_len, const int32_t *frames_dev, double *wave_out_dev, int64_t out_stride,
                                  void *stream) {
    if (!h) return WRNN_ERR_INVALID;
    auto fail = [&](int code, const char *msg) { h->err = msg; return code; };
    if (!samples_dev || !wave_out_dev || rows < 1 || steps < 1) return fail(WRNN_ERR_INVALID, "wrnn_epilogue_rows: null buffer or empty input");
    const WrnnDims &d = h->d;
    const bool decode = mu_law && d.mode == WRNN_MODE_RAW;   // MOL forces mu_law off (:174)
    if (decode && !labels_dev) return fail(WRNN_ERR_INVALID, "wrnn_epilogue_rows: mu-law decode needs the labels of the RAW loop");
    const long tail_len = 20L * d.HOP;
    if (wave_len < tail_len) return fail(WRNN_ERR_INVALID, "wrnn_epilogue_rows: wave_len shorter than the 20-hop fade-out (the reference raises ValueError for T < 21)");
    if (wave_len > steps) return fail(WRNN_ERR_INVALID, "wrnn_epilogue_rows: wave_len exceeds the generated length");
    if (out_stride < wave_len) return fail(WRNN_ERR_INVALID, "wrnn_epilogue_rows: out_stride shorter than wave_len");
    if (hipSetDevice(h->cfg.device) != hipSuccess) return fail(WRNN_ERR_HIP, "wrnn_epilogue_rows: hipSetDevice failed");
    const long ov = h->epi_tab ? h->epi_overlap : 0;   // any overlap's table set holds dec and tail
    if (const char *e = epilogue_tables_on_device(h, ov)) return fail(WRNN_ERR_HIP, e);
    const double *dec = h->epi_tab, *tail = dec + d.NC + 2 * ov;
    (void)hipGetLastError();
    hipLaunchKernelGGL(epilogue_rows_kernel, dim3((unsigned)((out_stride + 255) / 256), (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                       samples_dev, labels_dev, decode ? dec : nullptr, tail, (long)steps, (long)wave_len, tail_len, frames_dev, d.HOP,
                       wave_out_dev, (long)out_stride);
    if (hipGetLastError() != hipSuccess) return fail(WRNN_ERR_HIP, "wrnn_epilogue_rows: launch failed");
    return WRNN_OK;
}

extern "C" int wrnn_epilogue(wrnn_handle *h, const float *samples_dev, const int32_t *labels_dev, int32_t rows, int64_t steps,
                             int32_t batched, int32_t target, int32_t overlap, int32_t mu_law, int64_t wave_len,
                             double *wave_out_dev, void *stream) {
    if (!h) return WRNN_ERR_INVALID;
    auto fail = [&](int code, const char *msg) { h->err = msg; return code; };
    if (!samples_dev || !wave_out_dev || rows < 1 || steps < 1) return fail(WRNN_ERR_INVALID, "wrnn_epilogue: null buffer or empty input");
    const WrnnDims &d = h->d;
    const bool decode = mu_law && d.mode == WRNN_MODE_RAW;   // MOL forces mu_law off (:174)
    if (decode && !labels_dev) return fail(WRNN_ERR_INVALID, "wrnn_epilogue: mu-law decode needs the labels of the RAW loop");
    const long tail_len = 20L * d.HOP;
    if (wave_len < tail_len)   // the reference dies on the fade-out broadcast (:258) for T < 21
        return fail(WRNN_ERR_INVALID, "wrnn_epilogue: wave_len shorter than the 20-hop fade-out (the reference raises ValueError for T < 21)");
    if (batched) {
        if (overlap < 0 || target < 1 || steps != (int64_t)target + 2 * (int64_t)overlap)
            return fail(WRNN_ERR_INVALID, "wrnn_epilogue: folds must be target + 2*overlap samples long");
        if (wave_len > (int64_t)rows * (target + overlap) + overlap) return fail(WRNN_ERR_INVALID, "wrnn_epilogue: wave_len exceeds the unfolded length");
    } else if (wave_len > steps) {
        return fail(WRNN_ERR_INVALID, "wrnn_epilogue: wave_len exceeds the generated length");
    }
    hipStream_t s = (hipStream_t)stream;
    if (hipSetDevice(h->cfg.device) != hipSuccess) return fail(WRNN_ERR_HIP, "wrnn_epilogue: hipSetDevice failed");
    const int NC = d.NC;
    const long ov = batched ? overlap : 0;
    if (const char *e = epilogue_tables_on_device(h, ov)) return fail(WRNN_ERR_HIP, e);
    const double *dec = h->epi_tab, *fin = dec + NC, *fout = fin + ov, *tail = fout + ov;
    (void)hipGetLastError();
    hipLaunchKernelGGL(epilogue_kernel, dim3((unsigned)((wave_len + 255) / 256)), dim3(256), 0, s, samples_dev, labels_dev,
                       decode ? dec : nullptr, fin, fout, tail, (int)rows, (long)steps, (int)(batched != 0), (long)target, (long)overlap,
                       (long)wave_len, tail_len, wave_out_dev);
    if (hipGetLastError() != hipSuccess) return fail(WRNN_ERR_HIP, "wrnn_epilogue: launch failed");
    return WRNN_OK;
}
