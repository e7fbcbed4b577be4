// WRNN_KERNEL_BATCH: the per-sample loop (fatchord_version.py:194-241) for MANY rows per team, in lock-step, on the
// matrix cores.  The reference advances all B rows of a batch together (h1/h2 are (B, 512), :194-196; every layer is one
// batched matmul, :209-223); this kernel does the same per XCD team: R = 4*NQ rows step together, so every weight that
// sits in a register is used R times per step instead of once (loop_team2.hip), and one exchange / barrier / reduction
// tree serves R rows.
//
// Team, residency and exchange protocol are those of loop_team2.hip: one team = the 32 workgroups of one XCD (formed from
// HW_REG_XCC_ID at run time), fp32 weights resident in registers / LDS, 8-byte {tag, value} granules through the XCD's L2
// (plain store, sc1 load, the data is the flag), parity double-buffering, bounded spins.  What is different:
//
//  * Arithmetic on v_mfma_f32_4x4x1_16b_f32 (exact fp32, the vector FMA rate from ONE instruction per 256 MACs).  The
//    instruction is 16 independent 4x4 outer products: block b = lane>>2 multiplies A_b[i = lane&3] by B_b[j = lane&3]
//    and accumulates D_b[i][j] (vgpr i of lane (b, j)).  Here a block is one K phase kp = lane>>2, i = one of the 4
//    weight rows (hidden units / fc rows) the wave owns, j = one of 4 rows of the batch:
//        A lane (kp, i) = W[unit_i][k],   B lane (kp, j) = x[k][row_j],   k = 64 S + 16 e + kp   (slab s = 4 S + e)
//    so one A register holds 64 DISTINCT weights (no replication: the register file is ~70 % weights) and serves 4 batch
//    rows per issue; R = 8 rows = two issues with the same A.  The K sum runs through the accumulator (32 slabs), the
//    16 K phases are folded once per phase with 2 v_permlane32_swap + 1 v_permlane16_swap + 2 DPP adds -- instead
//    of a 4-step DPP reduction per row and output.  The B operand is one conflict-free ds_read_b128 per 4 issues.
//  * 4 waves per workgroup, one per SIMD, 512 registers each (VGPR + AGPR; MFMA reads A operands from either): a wave
//    holds ALL weight rows of its 4 hidden units -- W_hh1, W_ih2, W_hh2 (3 gates each), fc1, fc2: 352 registers -- and its
//    8 fc3 rows sit in LDS.  (Two waves per SIMD at 256 registers each, the team2 split, spills ~100-220 registers to
//    scratch here: measured with -Rpass-analysis.)  What team2 gave to "shadow" waves runs in the same instruction stream,
//    right after the publish of each phase and before the poll of its exchange: W_hh1.h1' (window 2), W_hh2.h2' (window 3),
//    the next step's conditioning (window 1, behind the x2 | h1' publish), sampling noise (window 4).  MFMA issue is asynchronous to the wave's VALU / LDS
//    stream, and nothing needs a hand-off through LDS: gh1, gh2, conditioning, noise stay in the owning thread's registers.
//  * Phase A (I + GRU1, elementwise) is no longer replicated in every workgroup: each workgroup evaluates its own 16 hidden
//    units for the R rows and publishes x2 and h1' (one more exchange than team2, but no 8 KB/step/row conditioning stream:
//    a workgroup needs the conditioning of 16 units only, evaluated from the per-frame records with the 5-tap form; HBM
//    traffic per sample ~ the 836 B of SURVEY 8d instead of 8 KB).
//  * Thread (wl, lane) is "unit 16 g + 4 wl + U(lane>>4), batch row 4 (kp2 % NQ) + (lane & 3)", kp2 = (lane>>2)&3,
//    U = {0, 2, 1, 3} (where the fold leaves the units); lanes with kp2 >= NQ hold duplicates and publish nothing.
//
// Per step: 5 exchanges (x2|h1', x3, fc1, fc2, race winners), 5 workgroup barriers.  Every exchange is an all-gather of
// R x 512 granules in the order the B operand wants them in LDS, read with 16-byte sc1 loads (two granules each).
#include "batch_common.h"

template <int MODE, int NQ, bool PROF>
__global__ void __launch_bounds__(TB_THREADS) loop_batch_kernel(WrnnBatchArgs a) {
    typedef Lay<NQ> L;
    constexpr int R = L::R, NM = L::NM;
    constexpr int DG = NQ == 1 ? 2 : 1, DS = NQ == 1 ? 4 : 2, D3 = NQ == 1 ? 2 : 1;   // prefetch depths (slabs) of the MFMA loops
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = (float *)smem;
    int *misc_i = (int *)(lds + L::L_MISC);
    float *xn = lds + L::L_XN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wl = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 3, kp2 = (lane >> 2) & 3, rho = lane >> 4;
    const int iu = ((rho & 1) << 1) | (rho >> 1);          // unit the fold leaves in this row of 16 lanes: {0, 2, 1, 3}
    const int my_rq = kp2 % NQ;
    const bool primary = kp2 < NQ;                         // the other lanes of four hold duplicates
    const int rb = 4 * my_rq + j;                          // batch row (0..R-1) of this thread
    const WrnnDims d = a.d;
    const int NC = d.NC, HOP = d.HOP, T = a.T;

    // ---- team formation: by the XCD this workgroup actually runs on (see loop_team2.hip) ------------
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
        // co-residency checked, not assumed (see loop_team2.hip): all 32 workgroups of the XCD must have arrived
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

    const int unit = 16 * g + 4 * wl + iu;                 // hidden unit / fc1 / fc2 row of this thread (fold layout)
    const int cls0 = 32 * g + 8 * wl + iu;                 // fc3 rows (classes) of this thread: cls0 and cls0 + 4
    const bool wg_has_fc3 = 32 * g < NC;
    // mailbox index of this thread's (unit, batch row) in a gathered vector: [rq][wl][S = g>>2][iu][j][e = g&3]
    const unsigned mb_own = ((((unsigned)my_rq * 4u + (unsigned)wl) * 8u + (unsigned)(g >> 2)) * 4u + (unsigned)iu) * 16u + (unsigned)j * 4u + (unsigned)(g & 3);
    const unsigned gvoff = (unsigned)tid * 16u;   // this thread's 16 bytes inside a 4 KB slice

    // ---- resident weights: batch_w [32 WG][4 waves][320][64 lanes] (A-operand images, api.hip):
    //      W_ih2 r,z,n [0,96) -> VGPRs (the serial chain) | W_hh1 r,z,n [96,192) | W_hh2 r,z [192,256) | fc1 [256,288) |
    //      fc2 [288,320) -> parked in AGPRs.  Gate n of W_hh2 (shadow path: tolerant of LDS latency) and the fc3 slice are
    //      A-operand images in LDS.
    float wv[96], wa[224];
    {
        const float *src = a.batch_w + (((size_t)g * 4 + wl) * 320) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 96; ++i) wv[i] = src[(size_t)i * 64];
#pragma unroll
        for (int i = 0; i < 224; ++i) apark(wa[i], src[(size_t)(96 + i) * 64]);
        const float4 *f3 = (const float4 *)(a.batch_fc3 + (size_t)g * 16384);
        float4 *dst = (float4 *)(lds + L::L_FC3);
        for (int i = tid; i < 4096; i += TB_THREADS) dst[i] = f3[i];
        const float4 *wn = (const float4 *)(a.batch_wn + (size_t)g * 8192);
        dst = (float4 *)(lds + L::L_WN);
        for (int i = tid; i < 2048; i += TB_THREADS) dst[i] = wn[i];
        for (int i = tid; i < L::L_TOTAL - L::L_P; i += TB_THREADS) lds[L::L_P + i] = 0.0f;
        // per-thread constants: W_I[:,0] and u = W_ih1.W_I[:,0] of `unit`, fc3 biases, recurrent biases
        float *cs = lds + L::L_CST + tid;
        cs[C_A0 * 256] = a.wI0[unit]; cs[C_A1 * 256] = a.u1[unit]; cs[C_A2 * 256] = a.u1[512 + unit]; cs[C_A3 * 256] = a.u1[1024 + unit];
        cs[C_B30 * 256] = cls0 < NC ? a.w[a.off.fc3_b + cls0] : 0.0f;
        cs[C_B31 * 256] = cls0 + 4 < NC ? a.w[a.off.fc3_b + cls0 + 4] : 0.0f;
        cs[C_H1R * 256] = a.w[a.off.r1_bhh + unit]; cs[C_H1Z * 256] = a.w[a.off.r1_bhh + 512 + unit]; cs[C_H1N * 256] = a.w[a.off.r1_bhh + 1024 + unit];
        cs[C_H2R * 256] = a.w[a.off.r2_bhh + unit]; cs[C_H2Z * 256] = a.w[a.off.r2_bhh + 512 + unit]; cs[C_H2N * 256] = a.w[a.off.r2_bhh + 1024 + unit];
    }
    __syncthreads();
    // opaque per-thread LDS bases (see lds_cf4p): activation vectors as B operands (P | Q | H1 are contiguous), the LDS-resident
    // A operands, the constants, the destination of gathered pairs
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const lds_cf4p vP = (lds_cf4p)(size_t)launder(smem_base + (unsigned)L::L_P * 4u + (unsigned)lane * 16u);
    const lds_cf4p vQ = vP + L::VEC / 4, vH1 = vP + 2 * (L::VEC / 4);
    const lds_cf4p w3 = (lds_cf4p)(size_t)launder(smem_base + (unsigned)L::L_FC3 * 4u + ((unsigned)(wl * 2) * 8u * 64u + (unsigned)lane) * 16u);
    const lds_cf4p wnl = (lds_cf4p)(size_t)launder(smem_base + (unsigned)L::L_WN * 4u + ((unsigned)wl * 8u * 64u + (unsigned)lane) * 16u);
    const lds_cfp cst = (lds_cfp)(size_t)launder(smem_base + (unsigned)L::L_CST * 4u + (unsigned)tid * 4u);
    // gathered pair m = (rq, wl') of this thread: tid = S*32 + iu*8 + j*2 + e/2 -> [rq][S = tid>>5][kp = 4 wl' + iu][j][e], i.e. float
    // index ((rq*8 + S)*16 + 4 wl')*16 + 2*(tid & 31) = thread part + (m>>2)*2048 + (m&3)*64
    const lds_f2p gdst = (lds_f2p)(size_t)launder(smem_base + (unsigned)L::L_P * 4u + ((unsigned)(tid >> 5) * 256u + 2u * (unsigned)(tid & 31)) * 4u);

    bool dead = false;
    unsigned epoch = 0;
    unsigned *prof_lds = (unsigned *)(lds + L::L_PROF);   // 32-bit: a phase accumulates < 2^32 cycles per launch (3 000 x 110 275 = 3.3e8)
    unsigned prof_last = 0;

    // schedule: batch b = slots [b * rpb, (b + 1) * rpb) of a.order; the batches are dealt to the teams round-robin, or in snake
    // order on a ragged batch (a.order is longest-first, so batches hold rows of similar length and the teams' sums even out)
    for (int pass = 0; pass * a.n_teams < n_batches; ++pass) {
        const int batch = pass * a.n_teams + ((a.snake && (pass & 1)) ? a.n_teams - 1 - team : team);
        if (batch >= n_batches) continue;
        const int slot_raw = batch * a.rpb + rb;
        const bool row_ok = rb < a.rpb && slot_raw < a.n_rows;
        const int row = a.order[row_ok ? slot_raw : a.n_rows - 1];     // spare slots of a batch re-run the last row (outputs masked)
        const WrnnRow rw = a.rows[row];
        const int64_t bsteps = a.rows[a.order[batch * a.rpb]].steps;   // the batch runs for its first (longest) row's steps
        const float *recb = a.tabREC32 + (size_t)rw.utt * (T + 1) * 512 * 32 + (size_t)unit * 32;
        const float *ktab = a.w + a.off.ktab;

        // h1 = h2 = 0, x = 0 (:194-196)  =>  gh1 = b_hh1, gh2 = b_hh2
        float h1 = 0.0f, h2 = 0.0f, x2own = 0.0f;
        float gh1r = cst[C_H1R * 256], gh1z = cst[C_H1Z * 256], gh1n = cst[C_H1N * 256];
        float gh2r = cst[C_H2R * 256], gh2z = cst[C_H2Z * 256], gh2n = cst[C_H2N * 256];
        float4 cd = make_float4(0.f, 0.f, 0.f, 0.f);      // {cI, v_r, v_z, v_n} of the coming step
        float4 c2 = make_float4(0.f, 0.f, 0.f, 0.f);      // {c2_r, c2_z, c2_n, c3} of the current frame
        float c4 = 0.0f;
        float nz0 = 0.f, nz1 = 0.f, pz0 = 0.f, pz1 = 0.f;  // -log q of classes cls0 / cls0+4: this step | the odd step of the Philox block
        int nfi = (int)(rw.start / HOP), nph = (int)(rw.start - (int64_t)nfi * HOP);   // frame / phase of the step being prepared
        int cst_frame = -1000000, pend_frame = -1;
        if (tid < R) {
            const int s0 = batch * a.rpb + tid;
            xn[tid] = (a.x_init && tid < a.rpb && s0 < a.n_rows) ? a.x_init[a.order[s0]] : 0.0f;
        }
        // rows this wave finishes (exchange 5): wave wl owns batch rows wl, wl + 4
        int frow[NQ];
        int fsteps[NQ];
#pragma unroll
        for (int bi = 0; bi < NQ; ++bi) {
            const int brow = wl + 4 * bi, s0 = batch * a.rpb + brow;
            const bool rok = brow < a.rpb && s0 < a.n_rows;
            frow[bi] = a.order[rok ? s0 : a.n_rows - 1];
            fsteps[bi] = rok ? a.rows[frow[bi]].steps : 0;   // 0 = masked
        }

        // conditioning {cI, v_r, v_z, v_n} of step ts for (unit, row): record + 5-tap upsampling (prologue.hip)
        // The loads are issued one step ahead: the record (24 floats) stays in registers and is re-read only when the frame
        // changes, the 5 taps of step t + 2 are requested while the values of step t + 1 are combined, so no load latency is
        // waited for (measured: 2 150 - 2 400 cycles per step when the loads are waited for where they are issued).
        float4 ra0, ra1, ra2, ra3, ra4, ra5;
        float rk0 = 0.f, rk1 = 0.f, rk2 = 0.f, rk3 = 0.f, rk4 = 0.f;
        ra0 = ra1 = ra2 = ra3 = ra4 = ra5 = make_float4(0.f, 0.f, 0.f, 0.f);
        int rec_frame = -1000000, fetched_frame = -1;
        auto cond_fetch = [&](int64_t ts) {
            const int64_t pos = rw.start + ts;
            const bool live = pos < a.total_len;
            const int fi = live ? nfi : T;
            const int ph = live ? nph : 0;
            if (++nph == HOP) { nph = 0; ++nfi; }
            fetched_frame = fi;
            if (fi != rec_frame) {
                const float4 *r = (const float4 *)(recb + (size_t)fi * 512 * 32);
                ra0 = r[0]; ra1 = r[1]; ra2 = r[2]; ra3 = r[3]; ra4 = r[4]; ra5 = r[5];
                rec_frame = fi;
            }
            const float *kt = ktab + ph * 5;
            rk0 = kt[0]; rk1 = kt[1]; rk2 = kt[2]; rk3 = kt[3]; rk4 = kt[4];
        };
        auto cond_combine = [&]() {
            pend_frame = fetched_frame;
            cd.x = fmaf(rk4, ra2.x, fmaf(rk3, ra1.w, fmaf(rk2, ra1.z, fmaf(rk1, ra1.y, fmaf(rk0, ra1.x, ra0.x)))));
            cd.y = fmaf(rk4, ra5.y, fmaf(rk3, ra4.z, fmaf(rk2, ra3.w, fmaf(rk1, ra3.x, fmaf(rk0, ra2.y, ra0.y)))));
            cd.z = fmaf(rk4, ra5.z, fmaf(rk3, ra4.w, fmaf(rk2, ra4.x, fmaf(rk1, ra3.y, fmaf(rk0, ra2.z, ra0.z)))));
            cd.w = fmaf(rk4, ra5.w, fmaf(rk3, ra5.x, fmaf(rk2, ra4.y, fmaf(rk1, ra3.z, fmaf(rk0, ra2.w, ra0.w)))));
        };
        // per-frame constants of (unit, row) once the frame changed: c2 (r,z,n), c3, c4 (record slots 24..28)
        auto frame_consts = [&]() {
            if (pend_frame != cst_frame) {
                const float4 *r = (const float4 *)(recb + (size_t)pend_frame * 512 * 32);
                c2 = r[6];
                c4 = recb[(size_t)pend_frame * 512 * 32 + 28];
                cst_frame = pend_frame;
            }
        };
        // -log q of this thread's two classes for step ts (RAW)
        auto prep_noise = [&](int64_t ts) {
            if (MODE != WRNN_MODE_RAW) return;
            if (a.noise_mode == WRNN_NOISE_INJECTED) {
                const float *qp = a.noise1 + ((size_t)ts * a.n_rows + row) * NC;
                nz0 = cls0 < NC ? -logf(qp[cls0]) : 0.0f;
                nz1 = cls0 + 4 < NC ? -logf(qp[cls0 + 4]) : 0.0f;
            } else if (a.noise_mode == WRNN_NOISE_PHILOX) {
                // one Philox block = classes (2c, 2c+1) x steps (2s, 2s+1) (device_util.h): evaluated on even steps.  Lane l (units 0, 2 of
                // the wave: even classes c, c+4) and lane l+32 (units 1, 3: classes c+1, c+5) need the same two blocks: the lower lane
                // evaluates block(c), the upper one block(c+4), and each hands the partner its half with one v_permlane32_swap.
                // -log q = -log(-log u): the inner log exactly (q is tiny for u -> 1, where v_log_f32 is not accurate relative to the
                // result and such a draw tends to win the race), the outer one fast.
                // The block on even steps, its two halves TRANSFORMED one per step (pz0 / pz1 carry the odd step's raw bits).  The first
                // version transformed all four draws on the even step: ~2 000 cycles there -- more than the f2 exchange this sits under
                // hides -- and nothing on the odd one.  Same values, the work spread evenly: B = 64 6 624 -> 6 824 ksamples/s, and the
                // kernel's 12 bytes of scratch are gone.
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
        };
        cond_fetch(0);
        cond_combine();
        if (bsteps > 1) cond_fetch(1);
        __syncthreads();

        for (int64_t t = 0; t < bsteps; ++t) {
            ++epoch;
            const unsigned par = epoch & 1u;
            if (PROF) prof_last = (unsigned)__builtin_readcyclecounter();

            // ================= window 1: phase A | publish x2, h1' | conditioning of the next step | gather both =================
            {
                // I + GRU1 for (unit, row) (:208-212); gi = u * x_{t-1} + v[t] (algebra: DESIGN.md 3.2)
                const float xprev = xn[rb];
                const float xin = fmaf(cst[C_A0 * 256], xprev, cd.x);
                const float rg = sigmoid_fast(fmaf(cst[C_A1 * 256], xprev, cd.y) + gh1r);
                const float zg = sigmoid_fast(fmaf(cst[C_A2 * 256], xprev, cd.z) + gh1z);
                const float ng = tanh_fast(fmaf(cst[C_A3 * 256], xprev, cd.w) + rg * gh1n);
                h1 = (1.0f - zg) * ng + zg * h1;
                x2own = xin + h1;
                if (primary) {
                    st_granule(mail, L::G_X2 + par * L::RG + mb_own, epoch, __float_as_uint(x2own));
                    st_granule(mail, L::G_H1 + par * L::RG + mb_own, epoch, __float_as_uint(h1));
                }
                frame_consts();   // constants of this step's frame (needed from phase B on)
                // conditioning of the next step, behind the publish: the wave would otherwise only wait for the x2 exchange here
                // (round 3: it sat at the start of window 5, where nothing is in flight: 740 serial cycles; B = 64 6 273 -> 6 634
                // ksamples/s, MOL B = 32 5 008 -> 5 142, A/B in one session).  cond_combine() uses what was requested a step ago --
                // loads OLDER than the granule stores above, so its wait does not include the stores' acknowledgement; the
                // loads of cond_fetch(t + 2) land during the rest of the step.
                if (t + 1 < bsteps) {
                    cond_combine();
                    if (t + 2 < bsteps) cond_fetch(t + 2);
                }
            }
            PB(0);   // phase A + publish + conditioning of the next step
            {
                if (NQ == 1) {
                    // R = 4: both vectors in one round trip (2 x 4 loads in flight)
                    u4v gx[2][NM];
                    const unsigned offs[2] = {(L::G_X2 + par * L::RG) * 8u, (L::G_H1 + par * L::RG) * 8u};
                    gather_vecs<NM, 2>(mrs, gvoff, offs, epoch, gx, dead, a.err, 21u);
                    PB(1);   // x2 and h1' arrived
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        gdst[(0 * L::VEC + (m >> 2) * 2048 + (m & 3) * 64) / 2] = (f2v){__uint_as_float(gx[0][m].x), __uint_as_float(gx[0][m].z)};
                        gdst[(2 * L::VEC + (m >> 2) * 2048 + (m & 3) * 64) / 2] = (f2v){__uint_as_float(gx[1][m].x), __uint_as_float(gx[1][m].z)};
                    }
                } else {
                    // R = 8: both vectors in flight at once (2 x 8 x 4 registers) push the kernel into scratch spills (measured with
                    // -Rpass-analysis: 59 vs 3 registers): two round trips
                    u4v gx[1][NM];
                    const unsigned offs[1] = {(L::G_X2 + par * L::RG) * 8u};
                    gather_vecs<NM, 1>(mrs, gvoff, offs, epoch, gx, dead, a.err, 21u);
                    PB(1);   // x2 arrived
#pragma unroll
                    for (int m = 0; m < NM; ++m) gdst[(0 * L::VEC + (m >> 2) * 2048 + (m & 3) * 64) / 2] = (f2v){__uint_as_float(gx[0][m].x), __uint_as_float(gx[0][m].z)};
                    // (requesting h1' -- needed by W_hh1 only -- behind phase B's MFMAs instead, with one more barrier: +0.1 %, not kept)
                    const unsigned offs2[1] = {(L::G_H1 + par * L::RG) * 8u};
                    gather_vecs<NM, 1>(mrs, gvoff, offs2, epoch, gx, dead, a.err, 22u);
#pragma unroll
                    for (int m = 0; m < NM; ++m) gdst[(2 * L::VEC + (m >> 2) * 2048 + (m & 3) * 64) / 2] = (f2v){__uint_as_float(gx[0][m].x), __uint_as_float(gx[0][m].z)};
                }
            }
            PB(2);   // h1' gathered, both written
            __syncthreads();   // B1
            PB(3);

            // ================= window 2: phase B (GRU2, :213-216) | gh1' = W_hh1 . h1' | gather x3 =================
            {
                f4 acc[3][NQ];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[gt][q] = (f4){0.f, 0.f, 0.f, 0.f};
                mfma_gates<NQ, 3, false, DG>(wv, vP, acc, NoMid());
                PB(4);   // phase B MFMAs issued
                float tr = 0.f, tz = 0.f, tn = 0.f;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float fr = fold_kp(acc[0][q]), fz = fold_kp(acc[1][q]), fn = fold_kp(acc[2][q]);
                    if (q == 0 || my_rq == q) { tr = fr; tz = fz; tn = fn; }
                }
                PB(5);   // folded
                const float rg = sigmoid_fast((tr + c2.x) + gh2r);
                const float zg = sigmoid_fast((tz + c2.y) + gh2z);
                const float ng = tanh_fast((tn + c2.z) + rg * gh2n);
                h2 = (1.0f - zg) * ng + zg * h2;
                const float x3 = x2own + h2;
                if (primary) st_granule(mail, L::G_X3 + par * L::RG + mb_own, epoch, __float_as_uint(x3));
            }
            PB(6);   // gates + publish x3
            {
                // off the serial chain, under the x3 exchange: gh1 of the next step
                f4 acc[3][NQ];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[gt][q] = (f4){0.f, 0.f, 0.f, 0.f};
                u4v gx[1][NM];
                const unsigned offs[1] = {(L::G_X3 + par * L::RG) * 8u};
                // the x3 slices are requested part of the way through the shadow MFMAs (the producers are normally done by then), so their
                // L2 round trip runs under the rest instead of after it.  R = 8: before slab 5 of 8 (3: -1.7 %, stale looks; 4: -0.7 %; 6, 7:
                // -0.3 %; the same for the fc1 slices in window 3).  R = 4: before the LAST slab -- the phase is half
                // as long, and requested half way 36 % of the early looks came back stale (counted with -DWRNN_COUNT_SLOW) and took the
                // slow path: sentinel polls + a second full look (MOL B = 32: 5 120 -> 5 280 ksamples/s, A/B in one session)
                {
                    auto req = [&]() { gather_issue<NM, 1>(mrs, gvoff, offs, gx); };
                    mfma_gates<NQ, 3, true, DG, decltype(req), (NQ == 1 ? 7 : 5)>(wa, vH1, acc, req);
                }
                PB(7);   // W_hh1 MFMAs issued
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float fr = fold_kp(acc[0][q]), fz = fold_kp(acc[1][q]), fn = fold_kp(acc[2][q]);
                    if (q == 0 || my_rq == q) { gh1r = fr + cst[C_H1R * 256]; gh1z = fz + cst[C_H1Z * 256]; gh1n = fn + cst[C_H1N * 256]; }
                }
                PB(8);   // W_hh1 folded
                gather_vecs<NM, 1, true>(mrs, gvoff, offs, epoch, gx, dead, a.err, 23u);
#pragma unroll
                for (int m = 0; m < NM; ++m) gdst[(1 * L::VEC + (m >> 2) * 2048 + (m & 3) * 64) / 2] = (f2v){__uint_as_float(gx[0][m].x), __uint_as_float(gx[0][m].z)};
            }
            PB(9);   // x3 gathered
            __syncthreads();   // B2
            PB(10);

            // ================= window 3: phase C (fc1, :217-218) | gh2' = W_hh2 . h2' | gather fc1 outputs =================
            {
                f4 sum[NQ];
                mfma_single<NQ, (NQ == 1 ? 4 : 2), true, DS>(wa + 160, vQ, sum);
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float f = fold_kp(sum[q]);
                    if (q == 0 || my_rq == q) s = f;
                }
                if (primary) st_granule(mail, L::G_F1 + par * L::RG + mb_own, epoch, __float_as_uint(fmaxf(s + c2.w, 0.0f)));
            }
            PB(11);  // fc1 + publish
            {
                // off the serial chain: gh2 of the next step = W_hh2 . h2' + b_hh2, h2' = x3 - x2 formed on the fly from the two
                // gathered vectors (the same subtraction team2 does when x3 arrives); gate n's weights come from LDS
                f4 acc[3][NQ];
#pragma unroll
                for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[gt][q] = (f4){0.f, 0.f, 0.f, 0.f};
                u4v gx[1][NM];
                const unsigned offs[1] = {(L::G_F1 + par * L::RG) * 8u};
                // software pipeline: the LDS operands of slab S + 1 (x3, x2, gate-n weights) are requested before the MFMAs of slab S
                // are issued (the scheduling barrier keeps hipcc from sinking the reads next to their use, which exposes the LDS
                // latency once per slab: measured 3 900 cycles for this phase against 2 340 for W_hh1)
                f4 xq[NQ], xp[NQ], wn = wnl[0];
#pragma unroll
                for (int q = 0; q < NQ; ++q) { xq[q] = vQ[(q * 8) * 64]; xp[q] = vP[(q * 8) * 64]; }
#pragma unroll
                for (int S = 0; S < 8; ++S) {
                    if (S == (NQ == 1 ? 4 : 5)) gather_issue<NM, 1>(mrs, gvoff, offs, gx);   // fc1 slices requested part of the way (see window 2)
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
                        const float wr = aget(wa[96 + 4 * S + e]), wz = aget(wa[128 + 4 * S + e]);
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            acc[0][q] = mfma4(wr, b[q][e], acc[0][q]);
                            acc[1][q] = mfma4(wz, b[q][e], acc[1][q]);
                            acc[2][q] = mfma4(wcur[e], b[q][e], acc[2][q]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float fr = fold_kp(acc[0][q]), fz = fold_kp(acc[1][q]), fn = fold_kp(acc[2][q]);
                    if (q == 0 || my_rq == q) { gh2r = fr + cst[C_H2R * 256]; gh2z = fz + cst[C_H2Z * 256]; gh2n = fn + cst[C_H2N * 256]; }
                }
                PB(12);  // W_hh2
                gather_vecs<NM, 1, true>(mrs, gvoff, offs, epoch, gx, dead, a.err, 24u);
#pragma unroll
                for (int m = 0; m < NM; ++m) gdst[(2 * L::VEC + (m >> 2) * 2048 + (m & 3) * 64) / 2] = (f2v){__uint_as_float(gx[0][m].x), __uint_as_float(gx[0][m].z)};
            }
            PB(13);  // f1 gathered
            __syncthreads();   // B3
            PB(14);

            // ================= window 4: phase D (fc2, :220-221) | noise of this step | gather fc2 =================
            {
                f4 sum[NQ];
                mfma_single<NQ, (NQ == 1 ? 4 : 2), true, DS>(wa + 192, vH1, sum);
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float f = fold_kp(sum[q]);
                    if (q == 0 || my_rq == q) s = f;
                }
                if (primary) st_granule(mail, L::G_F2 + par * L::RG + mb_own, epoch, __float_as_uint(fmaxf(s + c4, 0.0f)));
            }
            PB(15);  // fc2 + publish
            prep_noise(t);
            PB(16);  // noise
            {
                u4v gx[1][NM];
                const unsigned offs[1] = {(L::G_F2 + par * L::RG) * 8u};
                gather_vecs<NM, 1>(mrs, gvoff, offs, epoch, gx, dead, a.err, 25u);   // (requesting the slices before the noise: 3 % slower,
                                                                                     //  the early look comes back stale and takes the slow path)
#pragma unroll
                for (int m = 0; m < NM; ++m) gdst[(0 * L::VEC + (m >> 2) * 2048 + (m & 3) * 64) / 2] = (f2v){__uint_as_float(gx[0][m].x), __uint_as_float(gx[0][m].z)};
            }
            PB(18);  // f2 gathered
            __syncthreads();   // B4
            PB(19);

            // ================= window 5: phase E (fc3 :223 + sampler :225-237) | race =================
            PB(17);
            {
                float lg0 = 0.f, lg1 = 0.f;
                if (wg_has_fc3) {
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
                    lg0 += cst[C_B30 * 256]; lg1 += cst[C_B31 * 256];
                    if (a.logits_out && primary && row_ok && t < rw.steps) {
                        float *lo = a.logits_out + ((size_t)t * a.n_rows + row) * NC;
                        if (cls0 < NC) lo[cls0] = lg0;
                        if (cls0 + 4 < NC) lo[cls0 + 4] = lg1;
                    }
                }
                if (MODE == WRNN_MODE_RAW) {
                    // the race argmax_k logit_k - log q_k (:231-235): winner of this thread's 2 classes, then of the 8 classes
                    // of the wave for the thread's batch row (fold over the 4 rows of 16 lanes), one granule per wave and row
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
                        st_granule(mail, L::G_PR + par * L::PRG + (unsigned)rb * 128u + (unsigned)(g * 4 + wl),
                                   (epoch << 10) | (unsigned)(k & 1023), __float_as_uint(v));
                } else {
                    // MOL: the 30 fc3 outputs of every batch row are published by workgroup 0
                    if (wg_has_fc3 && primary) {
                        if (cls0 < NC) st_granule(mail, L::G_PR + par * L::PRG + (unsigned)rb * 128u + (unsigned)cls0, epoch, __float_as_uint(lg0));
                        if (cls0 + 4 < NC) st_granule(mail, L::G_PR + par * L::PRG + (unsigned)rb * 128u + (unsigned)(cls0 + 4), epoch, __float_as_uint(lg1));
                    }
                }
            }
            PB(20);  // fc3 + race in the wave + publish
            // ---- exchange 5: wave w finishes batch rows w, w + 4 (the candidates of both rows are fetched together) ----
            u4v gqa[NQ];
            if (MODE == WRNN_MODE_RAW) {
                const unsigned tg = epoch & 0x3fffffu;
#pragma unroll
                for (int i = 0; i < NQ; ++i) gqa[i] = ld_pair(mrs, (unsigned)lane * 16u, (L::G_PR + par * L::PRG + (unsigned)(wl + 4 * i) * 128u) * 8u);
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < NQ; ++i) ok = ok && (gqa[i].y >> 10) == tg && (gqa[i].w >> 10) == tg;
                    if (__all(ok) || dead) break;
                    if (++spins > TB_SPIN_MAX) { dead = true; if (lane == 0) atomicExch(a.err, 26u); break; }
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int i = 0; i < NQ; ++i)
                        if (!__all((gqa[i].y >> 10) == tg && (gqa[i].w >> 10) == tg))
                            gqa[i] = ld_pair(mrs, (unsigned)lane * 16u, (L::G_PR + par * L::PRG + (unsigned)(wl + 4 * i) * 128u) * 8u);
                }
            }
            PB(21);  // race candidates arrived
#pragma unroll
            for (int bi = 0; bi < NQ; ++bi) {
                const int brow = wl + 4 * bi;
                const int rrow = frow[bi];
                const bool rok = t < fsteps[bi];   // a real row that has not reached its own length (ragged batch)
                float x_new;
                int lab;
                if (MODE == WRNN_MODE_RAW) {
                    const u4v gq = gqa[bi];
                    const float va = __uint_as_float(gq.x), vb = __uint_as_float(gq.z);
                    const bool pb = vb > va;   // equal scores: the lower slot = the lower class range wins
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
                    float nzv = 0.0f;
                    if (lane <= nr) {
                        float u;
                        if (a.noise_mode == WRNN_NOISE_INJECTED)
                            u = lane < nr ? a.noise1[((size_t)t * a.n_rows + rrow) * nr + lane] : a.noise2[(size_t)t * a.n_rows + rrow];
                        else
                            u = 1e-5f + wrnn_uniform(a.seed, (uint64_t)t, (uint32_t)rrow, (uint32_t)lane) * (1.0f - 2e-5f);
                        nzv = lane < nr ? -logf(-logf(u)) : logf(u) - logf(1.0f - u);
                    }
                    float mylg = 0.0f;
                    {
                        const u64 *gp = mail + L::G_PR + par * L::PRG + (unsigned)brow * 128u + (unsigned)(lane < NC ? lane : 0);
                        u64 gq = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        unsigned spins = 0;
                        while (!dead && !__all((unsigned)(gq >> 32) == epoch)) {
                            if (++spins > TB_SPIN_MAX) { dead = true; if (lane == 0) atomicExch(a.err, 27u); break; }
                            __builtin_amdgcn_s_sleep(1);
                            gq = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        mylg = __uint_as_float((unsigned)gq);
                    }
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
                if (lane == 0) {
                    xn[brow] = a.x_forced ? a.x_forced[(size_t)t * a.n_rows + rrow] : x_new;   // (:237)
                    if (g == 0 && rok) {
                        if (a.labels_out) a.labels_out[(size_t)rrow * a.steps + t] = lab;
                        a.samples_out[(size_t)rrow * a.steps + t] = x_new;
                    }
                }
            }
            PB(22);  // winners reduced
            __syncthreads();   // B5
            PB(23);
            if ((t & 63) == 63) {   // bounded-spin bail-out, checked workgroup-wide every 64 steps
                if (dead && lane == 0) misc_i[M_DEAD] = 1;
                __syncthreads();
                if (misc_i[M_DEAD]) return;
            }
        }
        __syncthreads();
    }
#ifdef WRNN_COUNT_SLOW
    if (blockIdx.x == 0 && tid == 0)
        printf("stale first looks / looks: x2 %u/%u  h1 %u/%u  x3 %u/%u  f1 %u/%u  f2 %u/%u\n", wrnn_dbg_slow[21], wrnn_dbg_slow[53], wrnn_dbg_slow[22],
               wrnn_dbg_slow[54], wrnn_dbg_slow[23], wrnn_dbg_slow[55], wrnn_dbg_slow[24], wrnn_dbg_slow[56], wrnn_dbg_slow[25], wrnn_dbg_slow[57]);
#endif
    if (PROF && a.prof && lane == 0 && g == 0 && team == 0) {
        for (int i = 0; i < 24; ++i) a.prof[wl * WRNN_PROF_SLOTS + i] += prof_lds[wl * 24 + i];
    }
}

template <int MODE, int NQ>
static hipError_t launch_one(const WrnnBatchArgs &a, hipStream_t s) {
    const size_t lds = (size_t)Lay<NQ>::L_TOTAL * sizeof(float);
    hipError_t e;
    if (a.prof) {
        e = hipFuncSetAttribute((const void *)loop_batch_kernel<MODE, NQ, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((loop_batch_kernel<MODE, NQ, true>), dim3(a.n_teams * TB_WGS), dim3(TB_THREADS), lds, s, a);
    } else {
        e = hipFuncSetAttribute((const void *)loop_batch_kernel<MODE, NQ, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((loop_batch_kernel<MODE, NQ, false>), dim3(a.n_teams * TB_WGS), dim3(TB_THREADS), lds, s, a);
    }
    return hipGetLastError();
}

// a.nq = 1 (4 rows per team) or 2 (8 rows)
hipError_t wrnn_launch_loop_batch(const WrnnBatchArgs &a, hipStream_t s) {
    (void)hipGetLastError();  // the runtime is shared with PyTorch: drop any stale sticky error of this thread
    if (a.d.mode == WRNN_MODE_RAW) return a.nq == 2 ? launch_one<WRNN_MODE_RAW, 2>(a, s) : launch_one<WRNN_MODE_RAW, 1>(a, s);
    return a.nq == 2 ? launch_one<WRNN_MODE_MOL, 2>(a, s) : launch_one<WRNN_MODE_MOL, 1>(a, s);
}

// co-residency facts for wrnn_create's check: LDS bytes and the occupancy the runtime reports for the instantiation that
// (mode, nq, prof) launches
template <int MODE, int NQ>
static hipError_t occ_one(bool prof, int *blocks_per_cu, size_t *lds_bytes) {
    const size_t lds = (size_t)Lay<NQ>::L_TOTAL * sizeof(float);
    *lds_bytes = lds;
    const void *fn = prof ? (const void *)loop_batch_kernel<MODE, NQ, true> : (const void *)loop_batch_kernel<MODE, NQ, false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, TB_THREADS, lds);
}
hipError_t wrnn_batch_occupancy(int mode, int nq, bool prof, int *blocks_per_cu, size_t *lds_bytes) {
    if (mode == WRNN_MODE_RAW) return nq == 2 ? occ_one<WRNN_MODE_RAW, 2>(prof, blocks_per_cu, lds_bytes) : occ_one<WRNN_MODE_RAW, 1>(prof, blocks_per_cu, lds_bytes);
    return nq == 2 ? occ_one<WRNN_MODE_MOL, 2>(prof, blocks_per_cu, lds_bytes) : occ_one<WRNN_MODE_MOL, 1>(prof, blocks_per_cu, lds_bytes);
}
