"""Ring-protocol models of the r4 experimental conv kernels (moved out of tests/ with their sources; run: pytest tools/experiments)"""
"""Model check (CPU) of the LDS-DMA ring protocol of conv3x3_halo3_kernel (genpercept_amd/csrc/conv_halo.hip).

The kernel never drains its DMA ring: every K-step ends with a COUNTED `s_waitcnt vmcnt(N)` and one raw `s_barrier`, and which
operations may stay in flight depends on the tap, on tile / workgroup ends and on the role split (half of the waves issue their DMA
before the MFMAs, half after).  A wrong count does not fail a test run reliably -- the DMA usually lands in time anyway -- so the
table is checked here against an adversarial model instead:

  * a wave's loads complete in issue order, as late as its waits allow: after `vmcnt(N)` only the loads older than its N youngest are
    guaranteed to have landed (stores also occupy the counter but complete in any order, so they can only make a wait stronger);
  * data is visible to OTHER waves only through a barrier that follows the issuing wave's covering wait;
  * every LDS read (fragment prefetch of step s+1 during step s, the in-place input transform, the prologue) must find its operands
    certified that way, and every DMA into a ring slot / halo buffer must be issued after a barrier that follows the last read of the
    previous content (and, for the halo buffer the epilogue uses as its staging window, after the epilogue).

The schedule below is a transcription of kstep() / the prologue; test_model_matches_source pins the transcribed lines.
"""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NW = 8


# ---- conv3x3_halo5_kernel (conv_halo5.hip): same ring, no fragment prefetch -- step s reads tile s, issues tile s+2 ------------------------------

def simulate_halo5(cpt, ntiles, a_it=3, b_it=1, extra_barrier=True, S=9, transform_steps=(), single_halo_buffer=False):
    """S steps per chunk (9 taps: conv_halo5.hip; 6 (ky, position pair) steps: conv_halo6.hip, whose raw halo is a single buffer that a
    transform reads in `transform_steps` of the chunk before)"""
    nchunks = cpt * ntiles
    nsteps = S * nchunks
    fifo = [[] for _ in range(NW)]
    done = [set() for _ in range(NW)]
    certified = set()
    last_read_step = {}

    def issue(w, res, n, step):
        fifo[w].extend([res] * n)
        prev = ("W", res[1] - 3) if res[0] == "W" else ("H", res[1] - (1 if single_halo_buffer else 2))
        if prev in last_read_step:   # a read in step t is over for EVERY wave only at the barrier that ends step t
            assert last_read_step[prev] < step, f"{res} issued in step {step} while {prev} is still read in step {last_read_step[prev]}"

    def wait(w, n):
        keep = fifo[w][len(fifo[w]) - n:] if n else []
        for r in fifo[w][:len(fifo[w]) - n] if n else fifo[w]:
            if r not in keep:
                done[w].add(r)
        fifo[w] = list(keep)

    def barrier():
        for r in set.intersection(*done):
            certified.add(r)

    def read(res, step, what):
        assert res in certified, f"step {step}: {what} reads {res} before it is certified (cpt {cpt}, tiles {ntiles})"
        last_read_step[res] = max(last_read_step.get(res, -1), step)

    for w in range(NW):  # prologue: halo 0, tiles 0 and 1; vmcnt(B_IT); barrier
        issue(w, ("H", 0), a_it, -1)
        issue(w, ("W", 0), b_it, -1)
        issue(w, ("W", 1), b_it, -1)
        wait(w, b_it)
    barrier()
    if transform_steps:
        read(("H", 0), -1, "prologue transform")
    for s in range(nsteps):
        c, tap = divmod(s, S)
        cc = c % cpt
        tile_end = cc == cpt - 1
        final = tile_end and c == nchunks - 1
        issue_w = not (final and tap >= S - 2)
        issue_h = tap == 0 and not final

        def dma(w):
            if issue_w:
                issue(w, ("W", s + 2), b_it, s)
            if issue_h:
                issue(w, ("H", c + 1), a_it, s)
        dma_first = [w >= NW // 2 and not (tap == S - 1 and tile_end) for w in range(NW)]
        for w in range(NW):
            if dma_first[w]:
                dma(w)
        read(("W", s), s, "fragments")
        if not transform_steps:
            read(("H", c), s, "fragments")
        elif tap in transform_steps and not final:
            read(("H", c + 1), s, "input transform")
        if tap == S - 1 and tile_end:
            # the epilogue's staging window is this chunk's halo buffer: WITHOUT the extra barrier a fast wave would write it while a slow wave
            # (held up issuing its DMA) has not read its last fragments yet -- modelled as a read one step later than any write may begin
            if not extra_barrier:
                raise AssertionError("epilogue staging may overwrite halo rows a slower wave still has to read")
            for w in range(NW):
                wait(w, 0)
        for w in range(NW):
            if not dma_first[w]:
                dma(w)
        if tap == S - 1 and final:
            break
        for w in range(NW):
            if tap <= 1:
                wait(w, a_it + b_it if not final else b_it)
            elif tap < S - 2:
                wait(w, b_it)
            elif tap == S - 2:
                wait(w, 0 if final else b_it)
            elif not tile_end:
                wait(w, b_it)
        barrier()
    return nsteps


@pytest.mark.parametrize("cpt", [2, 3, 4, 8, 16])
@pytest.mark.parametrize("ntiles", [1, 2, 3])
def test_halo5_ring_protocol_is_safe(cpt, ntiles):
    simulate_halo5(cpt, ntiles)


def simulate_halo6(cpt, ntiles, a_it=3, prologue_barrier=True, transform_steps=(3, 4)):
    """conv_halo6.hip: six steps per chunk, weight PLANES (two per step) in a ring of six slots; step s reads plane 2 s + 1 at its start and plane
    2 s + 2 under its second half (fragment prefetch), issues planes 2 s + 5 and 2 s + 6; single raw-halo buffer, transformed in steps 3 and 4 of the
    chunk before; the workgroup's last chunk stops issuing at plane 11 and drains."""
    nchunks = cpt * ntiles
    nsteps = 6 * nchunks
    nplanes = 12 * nchunks
    fifo = [[] for _ in range(NW)]
    done = [set() for _ in range(NW)]
    certified = set()
    last_read = {}      # resource -> (step, before_barrier): the last read and whether a barrier has been passed since

    def issue(w, res, n, step):
        fifo[w].extend([res] * n)
        prev = ("P", res[1] - 6) if res[0] == "P" else ("H", res[1] - 1)
        if prev in last_read:
            assert last_read[prev] < step, f"{res} issued in step {step} while {prev} may still be read (step {last_read[prev]})"

    def wait(w, n):
        keep = fifo[w][len(fifo[w]) - n:] if n else []
        for r in fifo[w][:len(fifo[w]) - n] if n else fifo[w]:
            if r not in keep:
                done[w].add(r)
        fifo[w] = list(keep)

    def barrier():
        for r in set.intersection(*done):
            certified.add(r)

    def read(res, step, what):
        assert res in certified, f"step {step}: {what} reads {res} before it is certified (cpt {cpt}, tiles {ntiles})"
        last_read[res] = max(last_read.get(res, -10), step)

    for w in range(NW):
        issue(w, ("H", 0), a_it, -2)
        for pl in range(5):
            issue(w, ("P", pl), 1, -2)
        wait(w, 2)
    barrier()
    read(("H", 0), -2, "prologue transform")
    barrier()
    read(("P", 0), -1 if prologue_barrier else 0, "prologue fragments")   # without the barrier a fast wave is in step 0 while a slow one still reads
    if prologue_barrier:
        barrier()
    for s in range(nsteps):
        c, t = divmod(s, 6)
        cc = c % cpt
        tile_end = cc == cpt - 1
        final = tile_end and c == nchunks - 1
        issue_a = not (final and t >= 4)
        issue_b = not (final and t >= 3)
        issue_h = t == 0 and not final

        def dma(w):
            if issue_a:
                issue(w, ("P", 2 * s + 5), 1, s)
            if issue_b:
                issue(w, ("P", 2 * s + 6), 1, s)
            if issue_h:
                issue(w, ("H", c + 1), a_it, s)
        dma_first = [w >= NW // 2 and not (t == 5 and tile_end) for w in range(NW)]
        for w in range(NW):
            if dma_first[w]:
                dma(w)
        read(("P", 2 * s + 1), s, "second-half fragments")
        if t in transform_steps and not final:
            read(("H", c + 1), s, "input transform")
        if not (t == 5 and tile_end):
            read(("P", 2 * s + 2), s, "prefetch")
        if t == 5 and tile_end:
            barrier()
            for w in range(NW):
                wait(w, 0)
            if not final:
                for w in range(NW):
                    done[w] |= set()   # (vmcnt(0): everything this wave issued has landed)
                # plane 0 of the next tile is read by each wave after ITS vmcnt(0) -- but landed for everybody only after a barrier: it was
                # issued in step 3 and certified by the barrier that ended step 4
                read(("P", 2 * s + 2), s, "next tile's first fragments")
        for w in range(NW):
            if not dma_first[w]:
                dma(w)
        if t == 5 and final:
            break
        for w in range(NW):
            if final and t >= 3:
                wait(w, 0)
            elif t <= 1:
                wait(w, a_it + 2 if not final else 2)
            elif not (t == 5 and tile_end):
                wait(w, 2)
        barrier()
    return nsteps


@pytest.mark.parametrize("cpt", [2, 3, 4, 8, 16])
@pytest.mark.parametrize("ntiles", [1, 2, 3])
def test_halo6_ring_protocol_is_safe(cpt, ntiles):
    simulate_halo6(cpt, ntiles)
    with pytest.raises(AssertionError):  # the transform one step earlier would read a halo that is not certified yet
        simulate_halo6(cpt, ntiles, transform_steps=(2, 3))
    with pytest.raises(AssertionError):  # without the barrier behind the prologue's fragment read, step 0 may refill plane 0's slot under a slow wave
        simulate_halo6(cpt, ntiles, prologue_barrier=False)
    s6 = open(os.path.join(ROOT, "tools", "experiments", "conv_halo6.hip")).read()
    for line in ["const bool issue_a = !(final_ && T >= 4), issue_b = !(final_ && T >= 3), issue_h = T == 0 && !final_;",
                 "if (issue_a) stage_plane((2 * T + 5) % 6, adv_a);",
                 "if (issue_b) stage_plane((2 * T + 6) % 6, w_step);",
                 "if (final_ && T >= 3) wait_vm<0>();",
                 "else if (T <= 1) { if (!final_) wait_vm<A_IT + B_IT>(); else wait_vm<B_IT>(); }",
                 "else if (!(T == 5 && tile_end)) wait_vm<B_IT>();",
                 "load_half(f1, IC<2 * T + 1>{}, parc);",
                 "if (prefetch) load_half(f0, IC<2 * T + 2>{}, parc);",
                 "for (int pl = 0; pl < 5; ++pl) stage_plane(pl, w_step);",
                 "__builtin_amdgcn_s_barrier();  // every wave holds plane 0's fragments: step 0 refills that slot (plane 6)"]:
        assert line in s6, line


def test_halo5_model_detects_a_weaker_wait_and_matches_source():
    code = open(__file__).read().split("def simulate_halo5(")[1].split("\n@pytest")[0]
    ns = {}
    exec("NW = 8\ndef simulate_halo5(" + code.replace("            elif tap < S - 2:\n                wait(w, b_it)", "            elif tap < S - 2:\n                wait(w, 2 * b_it)"), ns)
    with pytest.raises(AssertionError):
        ns["simulate_halo5"](2, 2)
    s = open(os.path.join(ROOT, "tools", "experiments", "conv_halo5.hip")).read()
    for line in ["const bool issue_w = !(final_ && TAP >= 7), issue_h = TAP == 0 && !final_;",
                 "const bool dma_first = second_half && !(TAP == 8 && tile_end);",
                 "if (issue_w) stage_w((TAP + 2) % 3, adv);",
                 "if (TAP <= 1) { if (!final_) wait_vm<A_IT + B_IT>(); else wait_vm<B_IT>(); }",
                 "else if (TAP < 7) wait_vm<B_IT>();",
                 "else if (TAP == 7) { if (final_) wait_vm<0>(); else wait_vm<B_IT>(); }",
                 "else if (!tile_end) wait_vm<B_IT>();",
                 "__builtin_amdgcn_s_barrier();  // every wave holds its last fragments",
                 "stage_w(1, w_step);\n    wait_vm<B_IT>();"]:
        assert line in s, line


