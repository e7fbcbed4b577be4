// Micro-benchmarks that decide the team-kernel design (DESIGN.md "hop cost"):
//   1. census: which XCD / CU every workgroup of a 1-WG-per-CU grid lands on
//   2. ping-pong: round-trip latency of an 8-byte {tag,value} granule between two
//      workgroups, same XCD vs different XCD, for store/load cache-policy variants
//   3. all-gather: N workgroups each publish their slice of a 512-float vector and
//      every workgroup collects the whole vector into LDS (one exchange of the
//      WaveRNN team kernel), same-XCD team of 32 vs whole-chip team of 256
// Every spin is bounded; a timeout sets an error word and the kernel exits.
//
// build: hipcc --offload-arch=gfx950 -O3 -o handoff handoff.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef unsigned long long u64;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned hw_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

// store / load flavours ------------------------------------------------------
enum { ST_PLAIN = 0, ST_SC1 = 1, ST_SC0SC1 = 2, ST_NT = 3 };
enum { LD_SC1 = 0, LD_SC0SC1 = 1, LD_SC0 = 2, LD_PLAIN = 3 };

template <int ST> __device__ __forceinline__ void st64(u64 *p, u64 v) {
    if (ST == ST_PLAIN) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (ST == ST_SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (ST == ST_SC0SC1) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (ST == ST_NT) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
template <int LD> __device__ __forceinline__ u64 ld64(const u64 *p) {
    u64 v;
    if (LD == LD_SC1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    if (LD == LD_SC0SC1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    if (LD == LD_SC0) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    if (LD == LD_PLAIN) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

// ---------------------------------------------------------------- 1. census
struct Census { unsigned xcc, hwid; };
__global__ void __launch_bounds__(512) census_kernel(Census *out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (threadIdx.x == 0) {
        smem[0] = 1;
        out[blockIdx.x].xcc = xcc_id();
        out[blockIdx.x].hwid = hw_id();
    }
    // stay resident a while so that all blocks co-reside (1 per CU via LDS size)
    u64 t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) { __builtin_amdgcn_s_sleep(10); }
}

// ------------------------------------------------------------- 2. ping-pong
// state[0] = ticket counter per xcc (8 words), roles decided at run time:
// the first WG to arrive on xcd A becomes "ping"; the first WG on xcd B (B may
// equal A -> the second arrival) becomes "pong".  Everyone else exits.
struct PingCtl {
    unsigned arrivals[8];
    unsigned err;
    unsigned pad[7];
    u64 cycles;   // s_memtime cycles for all rounds (ping side)
    u64 wall;     // wall_clock64 ticks (100 MHz)
    unsigned ping_hw, pong_hw;
};

template <int ST, int LD>
__global__ void __launch_bounds__(512) pingpong_kernel(PingCtl *ctl, u64 *box, int xcd_a, int xcd_b, int rounds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int role_s;
    if (threadIdx.x == 0) {
        smem[0] = 0;
        unsigned x = xcc_id();
        unsigned rank = atomicAdd(&ctl->arrivals[x], 1u);
        int role = -1;
        if ((int)x == xcd_a && rank == 0) role = 0;
        else if ((int)x == xcd_b && rank == (xcd_a == xcd_b ? 1u : 0u)) role = 1;
        role_s = role;
    }
    __syncthreads();
    const int role = role_s;
    if (role < 0 || threadIdx.x != 0) return;
    u64 *to_pong = box;       // written by ping
    u64 *to_ping = box + 64;  // written by pong (different 128B line... 512 B apart)
    if (role == 0) ctl->ping_hw = hw_id(); else ctl->pong_hw = hw_id();
    const unsigned SPIN_MAX = 4000000;
    if (role == 0) {
        // warm-up handshake so both sides are resident
        u64 c0 = 0, w0 = 0;
        for (int r = 1; r <= rounds + 16; ++r) {
            if (r == 17) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
            st64<ST>(to_pong, ((u64)r << 32) | (unsigned)r);
            unsigned spins = 0;
            while ((ld64<LD>(to_ping) >> 32) != (u64)r) {
                if (++spins > SPIN_MAX) { atomicExch(&ctl->err, 1u); return; }
            }
        }
        ctl->cycles = __builtin_readcyclecounter() - c0;
        ctl->wall = wall_clock64() - w0;
    } else {
        for (int r = 1; r <= rounds + 16; ++r) {
            unsigned spins = 0;
            while ((ld64<LD>(to_pong) >> 32) != (u64)r) {
                if (++spins > SPIN_MAX) { atomicExch(&ctl->err, 2u); return; }
            }
            st64<ST>(to_ping, ((u64)r << 32) | (unsigned)r);
        }
    }
}

// ------------------------------------------------------------ 3. all-gather
// Team formation: each WG reads its XCC id and takes a rank within that XCD.
// mode 0: team = the WGs of XCD `xcd_sel` (ranks 0..team-1), others exit.
// mode 1: team = all WGs (rank = global arrival order).
struct GatherCtl {
    unsigned arrivals[8];
    unsigned total;
    unsigned err;
    unsigned pad[6];
    u64 cycles[256];
    u64 wall[256];
    float checksum[256];
};

template <int ST, int LD, int VEC>
__global__ void __launch_bounds__(512) gather_kernel(GatherCtl *ctl, u64 *mail, int mode, int xcd_sel, int team,
                                                      int rounds, int work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = (float *)smem;  // VEC floats
    __shared__ int rank_s;
    if (threadIdx.x == 0) {
        unsigned x = xcc_id();
        unsigned r_local = atomicAdd(&ctl->arrivals[x], 1u);
        unsigned r_all = atomicAdd(&ctl->total, 1u);
        int rank = -1;
        if (mode == 0) { if ((int)x == xcd_sel && (int)r_local < team) rank = (int)r_local; }
        else { if ((int)r_all < team) rank = (int)r_all; }
        rank_s = rank;
    }
    __syncthreads();
    const int rank = rank_s;
    if (rank < 0) return;
    const int tid = threadIdx.x;
    const int per = VEC / team;  // values published per WG
    const unsigned SPIN_MAX = 4000000;
    float acc = 0.f;
    float myval = (float)rank;
    u64 c0 = 0, w0 = 0;
    for (int r = 1; r <= rounds + 16; ++r) {
        if (r == 17 && tid == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
        u64 *buf = mail + (size_t)(r & 1) * VEC;
        if (tid < per) st64<ST>(buf + rank * per + tid, ((u64)r << 32) | __float_as_uint(myval + tid));
        // collect: VEC granules by VEC threads (VEC <= 512)
        int fail = 0;
        if (tid < VEC) {
            unsigned spins = 0;
            u64 g;
            while (((g = ld64<LD>(buf + tid)) >> 32) != (u64)r) {
                if (++spins > SPIN_MAX) { atomicExch(&ctl->err, 3u); fail = 1; break; }
            }
            xs[tid] = __uint_as_float((unsigned)g);
        }
        if (__syncthreads_or(fail)) return;
        // consume: a little dependent work per round (emulates the layer's dot product)
        float s = 0.f;
        for (int k = 0; k < work; ++k) s += xs[(tid * 8 + k) & (VEC - 1)];
        acc += s;
        myval = s * 1e-9f + (float)rank;   // next value depends on this round's data
        __syncthreads();
    }
    if (tid == 0) {
        ctl->cycles[rank] = __builtin_readcyclecounter() - c0;
        ctl->wall[rank] = wall_clock64() - w0;
    }
    if (tid == 1) ctl->checksum[rank] = acc;
}


// ------------------------------------------------- 3b. all-gather, variants
// PW = number of polling waves (1, 2, 4 or 8); each polling lane owns
// 512/(64*PW) consecutive 8-byte granules and keeps all its loads in flight.
// G16 = 1: 16-byte granules {v0, v1, v2, tag}: 171 granules, 171 polling lanes.
__device__ __forceinline__ void ld128(const void *p, unsigned &a, unsigned &b, unsigned &c, unsigned &d) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    a = v.x; b = v.y; c = v.z; d = v.w;
}
__device__ __forceinline__ void st128(void *p, unsigned a, unsigned b, unsigned c, unsigned d) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

template <int PW, int G16, int SLEEP>
__global__ void __launch_bounds__(512) gather2_kernel(GatherCtl *ctl, u64 *mail, int xcd_sel, int rounds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = (float *)smem;
    __shared__ int rank_s;
    const int team = 32, VEC = 512;
    if (threadIdx.x == 0) {
        unsigned x = xcc_id();
        unsigned r_local = atomicAdd(&ctl->arrivals[x], 1u);
        atomicAdd(&ctl->total, 1u);
        rank_s = ((int)x == xcd_sel && (int)r_local < team) ? (int)r_local : -1;
    }
    __syncthreads();
    const int rank = rank_s;
    if (rank < 0) return;
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const unsigned SPIN_MAX = 2000000;
    float acc = 0.f, myval = (float)rank;
    u64 c0 = 0, w0 = 0;
    for (int r = 1; r <= rounds + 16; ++r) {
        if (r == 17 && tid == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
        int fail = 0;
        if (G16) {
            // 16 values per WG -> 6 granules of 3 (last one padded); layout: granule index = rank*6 + i
            unsigned *buf = (unsigned *)(mail + (size_t)(r & 1) * 1024);
            if (tid < 6) {
                float v0 = myval + 3 * tid, v1 = v0 + 1, v2 = v0 + 2;
                st128(buf + (rank * 6 + tid) * 4, __float_as_uint(v0), __float_as_uint(v1), __float_as_uint(v2), (unsigned)r);
            }
            if (tid < 192) {
                unsigned a, b, c, d, spins = 0;
                for (;;) {
                    ld128(buf + tid * 4, a, b, c, d);
                    if (d == (unsigned)r) break;
                    if (++spins > SPIN_MAX) { atomicExch(&ctl->err, 4u); fail = 1; break; }
                    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
                }
                xs[tid * 3 + 0] = __uint_as_float(a); xs[tid * 3 + 1] = __uint_as_float(b); xs[tid * 3 + 2] = __uint_as_float(c);
            }
        } else {
            u64 *buf = mail + (size_t)(r & 1) * VEC;
            if (tid < 16) st64<ST_PLAIN>(buf + rank * 16 + tid, ((u64)r << 32) | __float_as_uint(myval + tid));
            if (wave < PW) {
                constexpr int PER = VEC / (64 * PW);   // granules per polling lane: 8,4,2,1
                const int base = tid * PER;
                unsigned spins = 0;
                bool done[PER];
                #pragma unroll
                for (int i = 0; i < PER; ++i) done[i] = false;
                for (;;) {
                    bool all = true;
                    u64 g[PER];
                    #pragma unroll
                    for (int i = 0; i < PER; ++i)
                        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(g[i]) : "v"(buf + base + i) : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    #pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        asm volatile("" : "+v"(g[i]));
                        if ((g[i] >> 32) == (u64)r) { xs[base + i] = __uint_as_float((unsigned)g[i]); } else all = false;
                    }
                    if (all) break;
                    if (++spins > SPIN_MAX) { atomicExch(&ctl->err, 5u); fail = 1; break; }
                    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
                }
            }
        }
        if (__syncthreads_or(fail)) return;
        float s = xs[tid] + xs[(tid + 17) & 511];
        acc += s;
        myval = s * 1e-9f + (float)rank;
        __syncthreads();
    }
    if (tid == 0) { ctl->cycles[rank] = __builtin_readcyclecounter() - c0; ctl->wall[rank] = wall_clock64() - w0; }
    if (tid == 1) ctl->checksum[rank] = acc;
}

template <int PW, int G16, int SLEEP>
static void run_gather2(GatherCtl *ctl, u64 *mail, int rounds) {
    CHECK(hipMemset(ctl, 0, sizeof(GatherCtl)));
    CHECK(hipMemset(mail, 0, 2 * 1024 * sizeof(u64)));
    (void)hipFuncSetAttribute((const void *)gather2_kernel<PW, G16, SLEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024));
    gather2_kernel<PW, G16, SLEEP><<<256, 512, 96 * 1024>>>(ctl, mail, 0, rounds);
    CHECK(hipDeviceSynchronize());
    static GatherCtl h;
    CHECK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
    double wmax = 0, cmax = 0;
    for (int i = 0; i < 32; ++i) { if (h.wall[i] > wmax) wmax = (double)h.wall[i]; if (h.cycles[i] > cmax) cmax = (double)h.cycles[i]; }
    printf("gather2 pollwaves=%d g16=%d sleep=%d err=%u  per-round %.3f us (%.0f cycles)\n", PW, G16, SLEEP, h.err,
           wmax * 0.01 / rounds, cmax / rounds);
}

// gather3: NO LDS staging and no workgroup barrier -- each of PW waves of a workgroup polls the whole 512-granule
// vector itself (8 granules per lane: granules 4l..4l+3 and 256+4l..256+4l+3, four dwordx4 loads per look), wave 0
// publishes the workgroup's 16 values of the next round once its own poll completed.  Measures whether the XCD's L2
// sustains 32 WGs x PW waves re-reading the same 4 KB (the "column-sliced direct poll" exchange).
template <int PW>
__global__ void __launch_bounds__(512) gather3_kernel(GatherCtl *ctl, u64 *mail, int xcd_sel, int rounds) {
    __shared__ int rank_s;
    if (threadIdx.x == 0) {
        unsigned x = xcc_id();
        unsigned r_local = atomicAdd(&ctl->arrivals[x], 1u);
        atomicAdd(&ctl->total, 1u);
        rank_s = ((int)x == xcd_sel && (int)r_local < 32) ? (int)r_local : -1;
    }
    __syncthreads();
    const int rank = rank_s;
    if (rank < 0) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (wave >= PW) return;
    const unsigned SPIN_MAX = 2000000;
    float acc = 0.f, myval = (float)rank;
    u64 c0 = 0, w0 = 0;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    for (int r = 1; r <= rounds + 16; ++r) {
        if (r == 17 && tid == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
        u64 *buf = mail + (size_t)(r & 1) * 512;
        if (wave == 0 && lane < 16) st64<ST_PLAIN>(buf + rank * 16 + lane, ((u64)r << 32) | __float_as_uint(myval + lane));
        const u4 *p0 = (const u4 *)(buf + 4 * lane), *p1 = (const u4 *)(buf + 256 + 4 * lane);
        unsigned spins = 0;
        u4 a, b, c, d;
        for (;;) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(a) : "v"(p0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16 sc1" : "=&v"(b) : "v"(p0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(c) : "v"(p1) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16 sc1" : "=&v"(d) : "v"(p1) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
            const bool ok = a.y == (unsigned)r && a.w == (unsigned)r && b.y == (unsigned)r && b.w == (unsigned)r &&
                            c.y == (unsigned)r && c.w == (unsigned)r && d.y == (unsigned)r && d.w == (unsigned)r;
            if (__all(ok)) break;
            if (++spins > SPIN_MAX) { atomicExch(&ctl->err, 6u); return; }
        }
        float s = __uint_as_float(a.x) + __uint_as_float(a.z) + __uint_as_float(b.x) + __uint_as_float(b.z) +
                  __uint_as_float(c.x) + __uint_as_float(c.z) + __uint_as_float(d.x) + __uint_as_float(d.z);
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
        acc += s;
        myval = s * 1e-9f + (float)rank;
    }
    if (tid == 0) { ctl->cycles[rank] = __builtin_readcyclecounter() - c0; ctl->wall[rank] = wall_clock64() - w0; }
    if (lane == 1) ctl->checksum[rank] = acc;
}

template <int PW>
static void run_gather3(GatherCtl *ctl, u64 *mail, int rounds) {
    CHECK(hipMemset(ctl, 0, sizeof(GatherCtl)));
    CHECK(hipMemset(mail, 0, 2 * 1024 * sizeof(u64)));
    gather3_kernel<PW><<<256, 512, 0>>>(ctl, mail, 0, rounds);
    CHECK(hipDeviceSynchronize());
    static GatherCtl h;
    CHECK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
    double wmax = 0, cmax = 0;
    for (int i = 0; i < 32; ++i) { if (h.wall[i] > wmax) wmax = (double)h.wall[i]; if (h.cycles[i] > cmax) cmax = (double)h.cycles[i]; }
    printf("gather3 (direct poll, no LDS/barrier) waves/WG=%d err=%u  per-round %.3f us (%.0f cycles)\n", PW, h.err,
           wmax * 0.01 / rounds, cmax / rounds);
}

// ----------------------------------------------------------------- driver
static const size_t LDS_BIG = 96 * 1024;  // > 80 KiB: one workgroup per CU

template <int ST, int LD>
static void run_pingpong(const char *label, PingCtl *ctl, u64 *box, int xa, int xb, int rounds) {
    CHECK(hipMemset(ctl, 0, sizeof(PingCtl)));
    CHECK(hipMemset(box, 0, 4096));
    (void)hipFuncSetAttribute((const void *)pingpong_kernel<ST, LD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BIG);
    pingpong_kernel<ST, LD><<<256, 512, LDS_BIG>>>(ctl, box, xa, xb, rounds);
    CHECK(hipDeviceSynchronize());
    PingCtl h;
    CHECK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
    printf("pingpong %-22s xcd %d->%d err=%u  round-trip %.1f cycles  %.3f us  (one-way %.3f us) hw %08x/%08x\n", label,
           xa, xb, h.err, (double)h.cycles / rounds, (double)h.wall * 0.01 / rounds, (double)h.wall * 0.005 / rounds,
           h.ping_hw, h.pong_hw);
}

template <int ST, int LD, int VEC>
static void run_gather(const char *label, GatherCtl *ctl, u64 *mail, int mode, int xcd, int team, int rounds, int work) {
    CHECK(hipMemset(ctl, 0, sizeof(GatherCtl)));
    CHECK(hipMemset(mail, 0, 2 * 512 * sizeof(u64)));
    (void)hipFuncSetAttribute((const void *)gather_kernel<ST, LD, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BIG);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    gather_kernel<ST, LD, VEC><<<256, 512, LDS_BIG>>>(ctl, mail, mode, xcd, team, rounds, work);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    static GatherCtl h;
    CHECK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
    double wmax = 0, cmax = 0;
    for (int i = 0; i < team; ++i) { if (h.wall[i] > wmax) wmax = (double)h.wall[i]; if (h.cycles[i] > cmax) cmax = (double)h.cycles[i]; }
    printf("gather %-20s mode=%d xcd=%d team=%3d vec=%d work=%2d err=%u  per-round %.3f us (%.0f cycles)  kernel %.3f ms arrivals=%u\n",
           label, mode, xcd, team, VEC, work, h.err, wmax * 0.01 / rounds, cmax / rounds, ms, h.total);
}

int main(int argc, char **argv) {
    int rounds = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz LDS/block=%zu\n", prop.name, prop.multiProcessorCount, prop.clockRate,
           prop.sharedMemPerBlock);

    // 1. census
    Census *cd;
    CHECK(hipMalloc(&cd, 256 * sizeof(Census)));
    (void)hipFuncSetAttribute((const void *)census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BIG);
    census_kernel<<<256, 512, LDS_BIG>>>(cd);
    CHECK(hipDeviceSynchronize());
    std::vector<Census> ch(256);
    CHECK(hipMemcpy(ch.data(), cd, 256 * sizeof(Census), hipMemcpyDeviceToHost));
    int per_xcc[16] = {0};
    int mismatch = 0;
    for (int b = 0; b < 256; ++b) { per_xcc[ch[b].xcc & 15]++; if ((int)ch[b].xcc != (b % 8)) mismatch++; }
    printf("census: WGs per XCC:");
    for (int i = 0; i < 8; ++i) printf(" %d", per_xcc[i]);
    printf("  (block b on xcc != b%%8: %d of 256)\n", mismatch);
    // distinct (xcc, hwid CU fields) check
    {
        int dup = 0;
        for (int a = 0; a < 256; ++a)
            for (int b = a + 1; b < 256; ++b)
                if (ch[a].xcc == ch[b].xcc && (ch[a].hwid & 0xffff00) == (ch[b].hwid & 0xffff00)) dup++;
        printf("census: pairs of WGs sharing (xcc, SE/SH/CU bits of HW_ID): %d ; sample hwid %08x %08x %08x\n", dup,
               ch[0].hwid, ch[1].hwid, ch[8].hwid);
    }

    // 2. ping-pong
    PingCtl *pc; u64 *box;
    CHECK(hipMalloc(&pc, sizeof(PingCtl)));
    CHECK(hipMalloc(&box, 4096));
    run_pingpong<ST_PLAIN, LD_SC1>("st=plain ld=sc1", pc, box, 0, 0, rounds);
    run_pingpong<ST_PLAIN, LD_SC0SC1>("st=plain ld=sc0sc1", pc, box, 0, 0, rounds);
    run_pingpong<ST_SC1, LD_SC1>("st=sc1 ld=sc1", pc, box, 0, 0, rounds);
    run_pingpong<ST_SC0SC1, LD_SC0SC1>("st=sc0sc1 ld=sc0sc1", pc, box, 0, 0, rounds);
    run_pingpong<ST_NT, LD_SC1>("st=nt ld=sc1", pc, box, 0, 0, rounds);
    run_pingpong<ST_SC1, LD_SC1>("st=sc1 ld=sc1", pc, box, 0, 1, rounds);
    run_pingpong<ST_SC0SC1, LD_SC0SC1>("st=sc0sc1 ld=sc0sc1", pc, box, 0, 1, rounds);
    run_pingpong<ST_SC1, LD_SC1>("st=sc1 ld=sc1", pc, box, 0, 4, rounds);
    // expected to FAIL (stale) cross-XCD: plain store is not visible to another XCD's L2 -> bounded spin reports err
    run_pingpong<ST_PLAIN, LD_SC1>("st=plain ld=sc1 (X)", pc, box, 0, 1, 200);

    // 3. all-gather of a 512-float vector
    GatherCtl *gc; u64 *mail;
    CHECK(hipMalloc(&gc, sizeof(GatherCtl)));
    CHECK(hipMalloc(&mail, 2 * 512 * sizeof(u64)));
    for (int work = 0; work <= 16; work += 16) {
        run_gather<ST_PLAIN, LD_SC1, 512>("st=plain ld=sc1", gc, mail, 0, 0, 32, rounds, work);
        run_gather<ST_SC1, LD_SC1, 512>("st=sc1 ld=sc1", gc, mail, 0, 0, 32, rounds, work);
        run_gather<ST_PLAIN, LD_SC1, 512>("st=plain ld=sc1", gc, mail, 0, 3, 32, rounds, work);
        run_gather<ST_PLAIN, LD_SC1, 512>("st=plain ld=sc1", gc, mail, 0, 0, 16, rounds, work);
        run_gather<ST_PLAIN, LD_SC1, 512>("st=plain ld=sc1", gc, mail, 0, 0, 8, rounds, work);
        run_gather<ST_SC1, LD_SC1, 512>("st=sc1 ld=sc1", gc, mail, 1, 0, 256, rounds, work);
        run_gather<ST_SC0SC1, LD_SC0SC1, 512>("st=sc0sc1 ld=sc0sc1", gc, mail, 1, 0, 256, rounds, work);
        run_gather<ST_SC1, LD_SC1, 512>("st=sc1 ld=sc1", gc, mail, 1, 0, 64, rounds, work);
    }
    // 32-value exchange (the sampler's (value,index) pairs)
    run_gather<ST_PLAIN, LD_SC1, 32>("st=plain ld=sc1", gc, mail, 0, 0, 32, rounds, 0);
    run_gather<ST_SC1, LD_SC1, 32>("st=sc1 ld=sc1", gc, mail, 0, 0, 32, rounds, 0);
    CHECK(hipFree(mail));
    CHECK(hipMalloc(&mail, 2 * 1024 * sizeof(u64)));
    run_gather2<8, 0, 0>(gc, mail, rounds);
    run_gather2<4, 0, 0>(gc, mail, rounds);
    run_gather2<2, 0, 0>(gc, mail, rounds);
    run_gather2<1, 0, 0>(gc, mail, rounds);
    run_gather2<1, 0, 1>(gc, mail, rounds);
    run_gather2<2, 0, 1>(gc, mail, rounds);
    run_gather2<8, 0, 1>(gc, mail, rounds);
    run_gather2<1, 1, 0>(gc, mail, rounds);
    run_gather2<1, 1, 1>(gc, mail, rounds);
    run_gather3<1>(gc, mail, rounds);
    run_gather3<4>(gc, mail, rounds);
    run_gather3<8>(gc, mail, rounds);
    printf("done\n");
    return 0;
}
