"""The half-window table of the ping-pong variant of the batch kernel (loop_batch.hip, `if constexpr (PP)`), checked for LDS
hazards and exchange ordering on the CPU.  The table restates the kernel's comment block; a barrier ends every half-window.

Rules a schedule has to meet:
  * inside one half-window nothing reads an LDS buffer that the gather of the same half-window writes (the gather's writes are
    not ordered against the compute phase's reads: different waves are at different points between two barriers);
  * a buffer is written (gathered) in an earlier half-window than any read of that content, and not overwritten before its
    last reader ran;
  * a gather follows the publish of its content by at least one half-window (that is the point of the schedule: the exchange is
    looked at one compute phase after it was published), and precedes the publish that reuses the same mailbox parity by more
    than a step.
"""

# (compute phase, quad, LDS buffers it reads, what it publishes) | (gathered content, quad, LDS buffer written)
# buffers: P (x2, later fc2 outputs), Q (x3), H (h1', later fc1 outputs), X (x_{t-1} of the quad's rows); suffix = quad
HW = [
    (('A', 'a', ['Xa'], 'x2h1'), ('race', 'b', 'Xb')),         # 1: race(b) belongs to the previous step
    (('A', 'b', ['Xb'], 'x2h1'), ('x2h1', 'a', ['Pa', 'Ha'])),  # 2
    (('B', 'a', ['Pa', 'Ha'], 'x3'), ('x2h1', 'b', ['Pb', 'Hb'])),
    (('B', 'b', ['Pb', 'Hb'], 'x3'), ('x3', 'a', ['Qa'])),
    (('C', 'a', ['Qa', 'Pa'], 'f1'), ('x3', 'b', ['Qb'])),
    (('C', 'b', ['Qb', 'Pb'], 'f1'), ('f1', 'a', ['Ha'])),
    (('D', 'a', ['Ha'], 'f2'), ('f1', 'b', ['Hb'])),
    (('D', 'b', ['Hb'], 'f2'), ('f2', 'a', ['Pa'])),
    (('E', 'a', ['Pa'], 'race'), ('f2', 'b', ['Pb'])),
    (('E', 'b', ['Pb'], 'race'), ('race', 'a', 'Xa')),
]
# which content a compute phase expects in the buffers it reads
NEEDS = {'A': {'X': 'race'}, 'B': {'P': 'x2h1', 'H': 'x2h1'}, 'C': {'Q': 'x3', 'P': 'x2h1'}, 'D': {'H': 'f1'}, 'E': {'P': 'f2'}}


def _as_list(x):
    return x if isinstance(x, list) else [x]


def test_no_gather_writes_what_its_half_window_reads():
    for comp, gat in HW:
        assert not set(comp[2]) & set(_as_list(gat[2])), (comp, gat)


def test_every_read_sees_the_content_it_expects_over_three_steps():
    content = {}          # buffer -> (what, step it belongs to)
    published = {}        # (what, quad) -> (step, half-window index) of the last publish
    for step in range(3):
        for h, (comp, gat) in enumerate(HW):
            phase, q, reads, pub = comp
            if step > 0 or phase != 'A':   # step 0 starts from the initial x (zeros / x_init), not from a race
                for b in reads:
                    what, st = content[b]
                    assert what == NEEDS[phase][b[0]], (step, h, comp, content[b])
                    # phase A of step s reads the race result of step s - 1, everything else content of its own step
                    assert st == (step - 1 if phase == 'A' else step), (step, h, comp, content[b])
            published[(pub, q)] = (step, h)
            what, gq, writes = gat
            gstep = step - 1 if (what == 'race' and h == 0) else step
            if gstep < 0:
                continue
            # the gathered content was published by every workgroup at least one half-window (one barrier) earlier ...
            ps, ph = published[(what, gq)]
            assert (ps, ph) < (step, h) and ps == gstep, (step, h, gat, published[(what, gq)])
            for b in _as_list(writes):
                content[b] = (what, gstep)


def test_mailbox_parity_is_not_reused_before_it_was_read():
    """Region (content, quad, parity) is published at step s and again at step s + 2: the gather of step s has to come before
    the publish of step s + 1 of the SAME content and quad in program order -- every workgroup then is past its read when any
    workgroup can be two steps ahead (a workgroup cannot pass a gather of step s + 1 before all published step s + 1)."""
    pub_at, gat_at = {}, {}
    for h, (comp, gat) in enumerate(HW):
        pub_at[(comp[3], comp[1])] = h
        gat_at[(gat[0], gat[1])] = h + (len(HW) if (gat[0] == 'race' and h == 0) else 0)   # race(b) is read in the next step
    for key, g in gat_at.items():
        assert pub_at[key] < g < pub_at[key] + len(HW), (key, pub_at[key], g)
