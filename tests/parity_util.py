"""Comparison rules for the GPU path vs the oracle (see DESIGN.md "parity protocol").

The sampler is a race: argmax_k p_k / q_k over 1024 classes (torch.multinomial's n=1 path).  Two
correct fp32 implementations that sum in a different order can legitimately disagree when the two
best race scores are (nearly) tied, and in a free-running generation everything after such a step is
a different -- equally valid -- trajectory.  So:
  * teacher-forced (P0): at EVERY step the GPU label must equal the oracle label, or the step must be a
    near-tie (oracle's relative margin < NEAR_TIE) and the GPU must have picked the oracle's runner-up;
  * free-running (P1): labels must be identical up to the first such near-tie (if any) -- and the check does NOT stop
    there: the oracle is re-run teacher-forced on the GPU's OWN trajectory (`x_forced` = the samples the GPU fed back), so
    every later step is still compared under the per-step rule.  `check_on_gpu_trajectory_*` do exactly that in one oracle
    pass: zero mismatches on it means (by induction over the steps) the GPU run IS the oracle's free run, bit for bit;
    a mismatch must be a near-tie with the GPU on the runner-up.  Every step of every row is compared (`compared ==
    L * rows`, asserted by the BASELINE-size tests).
With identical labels the fed-back value is bit-identical, which is stricter than the +-1 LSB the
north star asks for.
"""
from __future__ import annotations

import os

import numpy as np

NEAR_TIE = 2e-5       # RAW: relative gap of the two best p/q scores (round 5: tightened from 1e-4; the six near-ties observed so far -- see below -- have
                      # fp32 margins of 2.0e-7 ... 6.0e-6)
# How often a near-tie may happen: at most 1 + ceil(NEAR_TIE_RATE x compared steps) per check.  A near-tie picks the oracle's RUNNER-UP class, which need
# not be adjacent: |dlabel| at those steps is recorded (bound_near_ties, parity_report) so that the distance from the north star's literal "+-1 LSB" is a
# tracked number.  Observed on the shipped kernels (profiles/r05_parity.txt): configs[2] at T = 401, all 64 rows, Philox: 3 in 7 057 600 (4.3e-7 per step);
# the same size on injected reference noise, 16 rows: 2 in 1 764 400; configs[3] at world 1: 1 in 180 400 (margin 6.0e-6); 0 in every other check
# (configs[1] both noise modes, every T = 41 run, all MOL runs).
NEAR_TIE_RATE = 2e-6   # round 5: tightened from 1e-5
# The BASELINE-size tests check a subset of the rows of configs[2] / [4] over all 110 275 steps; the subset rotates with this
# number (bumped every round) so that over the rounds every (team, position) pair is covered.  (Round 5: the BASELINE-size tests check ALL rows;
# the rotation is kept for the quick subsets of developer sessions, PARITY_ROWS=subset.)
ROW_ROTATION = 5
_REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'parity_report.txt')


def parity_report(line: str) -> None:
    """One line per parity check into gpurun_out/parity_report.txt (merged back from the GPU box; copied to profiles/ per round)."""
    print('\n[parity] ' + line)
    try:
        os.makedirs(os.path.dirname(_REPORT), exist_ok=True)
        with open(_REPORT, 'a') as f:
            f.write(line + '\n')
    except OSError:
        pass


def bound_near_ties(tag: str, compared: int, near_ties) -> None:
    """Assert the near-tie RATE and record count + max |dlabel| of the near-tie steps.  near_ties: [(t, row, |dlabel|)]."""
    # 1 + ceil(rate x N) (round-5 advisor): with int() a 180 400-step check was allowed exactly the ONE near-tie it has (configs[3] at world 1, margin
    # 6.0e-6) -- seed-deterministic, but any legitimate re-ordering of an fp32 sum in a kernel may flip a second tie at that size.  For N >= 500 k steps the
    # two rules differ by at most one.
    allowed = 1 + int(np.ceil(compared * NEAR_TIE_RATE))
    worst = max((d for _, _, d in near_ties), default=0)
    parity_report(f'{tag}: steps compared {compared}, near-tie divergences {len(near_ties)} (allowed {allowed}), max |dlabel| at a near-tie '
                  f'{worst}, identical everywhere else; near-tie steps (t, row, |dlabel|): {list(near_ties)[:8]}')
    assert len(near_ties) <= allowed, f'{tag}: {len(near_ties)} near-ties in {compared} steps (allowed {allowed})'


def near_tie_truth(tag, sd, mel, x_fed, t, q_t, gpu_label, ref, row):
    """Which side is right?  At a near-tie step the race is re-evaluated in float64 (oracle/torch_ref.float64_logits_at: the whole history of the
    row in double precision along the fed-back trajectory) on the same draws q_t (n_classes,), and the report line says where the exact arithmetic
    falls: {fp32-oracle margin, float64 winner + its float64 margin, GPU label, oracle label}.  Recorded, not asserted: both fp32 results are
    legitimate roundings of a race that is tied to ~1e-5."""
    from oracle import torch_ref
    lg = torch_ref.float64_logits_at(sd, mel, x_fed, [t])[0]
    score = lg - np.log(np.asarray(q_t, np.float64))            # argmax p / q == argmax logit - log q
    order = np.argsort(-score)
    w, ru = int(order[0]), int(order[1])
    m64 = float(1.0 - np.exp(score[ru] - score[w]))               # relative gap of the two best p / q, as the oracle's margin
    side = 'the GPU' if w == int(gpu_label) else ('the fp32 oracle' if w == int(ref['labels'][t, row]) else 'NEITHER')
    parity_report(f'{tag}: near-tie at step {t}: gpu label {int(gpu_label)}, fp32-oracle label {int(ref["labels"][t, row])} (margin {float(ref["margin"][t, row]):.3e}), '
                  f'float64 winner {w} (runner-up {ru}, margin {m64:.3e}) -> exact arithmetic sides with {side}')
    return w


NEAR_TIE_MOL = 1e-4   # MOL: absolute gap of the two best Gumbel scores
MOL_LSB = 2.0 / (2 ** 9 - 1)


def check_teacher_forced_raw(got, ref):
    """got (L, rows) int; ref: oracle dict with labels/margin/runner (L, rows)."""
    bad = np.argwhere(got != ref['labels'])
    for t, r in bad:
        assert ref['margin'][t, r] < NEAR_TIE and got[t, r] == ref['runner'][t, r], \
            f'step {t} row {r}: gpu {got[t, r]} oracle {ref["labels"][t, r]} margin {ref["margin"][t, r]:.3e}'
    return len(bad)


def check_free_run_raw(got, ref):
    """Identical until the first near-tie.  Returns list of first-divergence steps per row (None = none)."""
    first = []
    for r in range(got.shape[1]):
        mism = np.flatnonzero(got[:, r] != ref['labels'][:, r])
        if mism.size == 0:
            first.append(None)
            continue
        t = int(mism[0])
        assert ref['margin'][t, r] < NEAR_TIE and got[t, r] == ref['runner'][t, r], \
            f'row {r}: first divergence at step {t} is not a near-tie (margin {ref["margin"][t, r]:.3e})'
        first.append(t)
    return first


def check_on_gpu_trajectory_raw(got_labels, got_samples, oracle_loop):
    """RAW, free-running GPU run checked at EVERY step.  `got_labels`, `got_samples` (L, rows); `oracle_loop(x_forced)` runs
    the oracle on the same conditioning and noise with the fed-back value forced to `x_forced` (L, rows).  The oracle is
    driven along the GPU's own trajectory, so a near-tie divergence (allowed: GPU on the oracle's runner-up, margin <
    NEAR_TIE) does not end the comparison.  Returns dict(compared, near_ties [(t, row, |dlabel|)], max_abs_other): steps
    compared (= L * rows), the near-tie steps, and max |dlabel| over all OTHER steps (0 by construction when the asserts
    pass).  near_ties == [] means the GPU run equals the oracle's free run (induction over t)."""
    ref = oracle_loop(np.ascontiguousarray(got_samples, dtype=np.float32))
    check_teacher_forced_raw(got_labels, ref)
    bad = np.argwhere(got_labels != ref['labels'])
    near = [(int(t), int(r), int(abs(int(got_labels[t, r]) - int(ref['labels'][t, r])))) for t, r in bad]
    return dict(compared=int(got_labels.size), near_ties=near, max_abs_other=0, ref=ref)


def check_on_gpu_trajectory_mol(got_samples, got_mix, oracle_loop):
    """MOL analogue: the oracle is forced onto the GPU's fed-back samples, so at EVERY step both sides start from the same
    state: mixture index equal (or a near-tie on the runner-up) and the continuous sample within 2e-5."""
    ref = oracle_loop(np.ascontiguousarray(got_samples, dtype=np.float32))
    check_mol(got_samples, got_mix, ref, teacher_forced=True)
    same = got_mix == ref['labels']
    err = np.abs(got_samples - ref['samples'])[same]
    return dict(compared=int(got_mix.size), index_mismatches=int((~same).sum()), max_err=float(err.max()) if err.size else 0.0, ref=ref)


def check_mol(got_samples, got_mix, ref, teacher_forced):
    """MOL: continuous sample within one 9-bit LSB (in practice ~1e-6) wherever the mixture index agrees;
    a mixture-index disagreement must be a near-tie."""
    L, rows = got_samples.shape
    for r in range(rows):
        mism = np.flatnonzero(got_mix[:, r] != ref['labels'][:, r])
        for t in mism:
            assert ref['margin'][t, r] < NEAR_TIE_MOL and got_mix[t, r] == ref['runner'][t, r], \
                f'row {r} step {t}: mixture index differs and is not a near-tie'
        end = L if (teacher_forced or mism.size == 0) else int(mism[0])
        ok = np.ones(L, bool)
        ok[mism] = False
        ok[end:] = False
        err = np.abs(got_samples[ok, r] - ref['samples'][ok, r])
        # teacher-forced: every step starts from the oracle's own state -> fp32 round-off only.
        # free-running: the fed-back value is continuous (no quantiser re-synchronises the two
        # trajectories), so round-off differences accumulate over thousands of steps; they must stay
        # well inside one 9-bit LSB.
        tol = 2e-5 if teacher_forced else MOL_LSB / 4
        assert err.size == 0 or err.max() <= tol, f'row {r}: max sample error {err.max():.3e} (tol {tol:.1e})'


def label_stats(got, ref_labels, first):
    """What the north star's "+-1 LSB at 10 bit" contract looks like on a run: `got`, `ref_labels` (L, rows); `first` =
    check_free_run_raw's list of first-divergence steps.  Returns the number of steps compared (every step of a row up
    to its first near-tie divergence, exclusive; everything if it never diverges), the mismatches among them, their
    max |label difference| (0 = the fed-back values are bit-identical), and |label difference| AT the near-tie steps
    (a near-tie picks the runner-up class of the race, which need not be an adjacent class)."""
    compared = mism = max_abs = 0
    near = []
    for r in range(got.shape[1]):
        end = got.shape[0] if first[r] is None else int(first[r])
        d = np.abs(got[:end, r].astype(np.int64) - ref_labels[:end, r].astype(np.int64))
        compared += end
        mism += int(np.count_nonzero(d))
        if d.size:
            max_abs = max(max_abs, int(d.max()))
        if first[r] is not None:
            t = int(first[r])
            near.append(int(abs(int(got[t, r]) - int(ref_labels[t, r]))))
    return dict(compared=compared, mismatches=mism, max_abs=max_abs, near_tie_abs=near)
