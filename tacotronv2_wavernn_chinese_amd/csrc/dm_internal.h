// Shared by the two kernels of the secondary dual-softmax model (loop_deepmind.hip, loop_dm_team.hip).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

struct WrnnDmArgs {
    const float *w;        // packed: RT [H][3H] | O1T [S][S] | O1b | O2T [S][Q] | O2b | O3T | O3b | O4T | O4b | Ic [3S][2] | If [3S][3] | bu | br | be
    size_t oRT, oO1T, oO1b, oO2T, oO2b, oO3T, oO3b, oO4T, oO4b, oIc, oIf, obu, obr, obe;
    int H, Q;
    long seq_len;
    int noise_mode;
    unsigned long long seed;
    const float *noise;    // (seq_len, 2, Q) Exp(1) draws or null
    int *coarse, *fine;
};

// team kernel (loop_dm_team.hip): 32 workgroups of one XCD, weights on chip
#define WRNN_DM_MAIL_GRANULES 5120   // x2 parities: h_c, t1_c, h_f, t1_f (512 each) + coarse / fine class values (256 each)
struct WrnnDmTeamArgs {
    WrnnDmArgs base;
    const float *team_w;          // [32 WGs][3 * H/16 registers][512 threads]   R rows of the thread's hidden unit
    const float *team_lds;        // [32 WGs][O1 U*S | O3 U*S | O2 QW*S | O4 QW*S]  LDS images (quarter-wave row, plane, lane)
    unsigned long long *mail;     // [WRNN_DM_MAIL_GRANULES]
    unsigned *ctl;                // [32] team formation counters
    unsigned *err;                // device error word
};
bool wrnn_dm_team_supported(int H, int Q);
hipError_t wrnn_launch_dm_team(const WrnnDmTeamArgs &a, hipStream_t s);
