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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NW = 8


def simulate(cpt, ntiles, a_it=6, b_it=2, fused=True, nt=9, nb=3):
    """Returns nothing; raises AssertionError with a description on the first protocol violation."""
    # nt taps (K-steps) per chunk, nb weight-ring slots: 9 / 3 = the nine-tap kernels; 4 / 4 = the phase mode of the x2-upsample conv (r5: PH)
    nchunks = cpt * ntiles
    nsteps = nt * nchunks
    fifo = [[] for _ in range(NW)]          # per wave: issued load ops, oldest first: ("W", step) or ("H", chunk), one entry per DMA instruction
    done = [set() for _ in range(NW)]       # per wave: resources guaranteed landed (all its parts)
    certified = set()                       # resources every wave has had certified before a barrier that was passed
    last_read_step = {}                     # resource -> last step in which somebody reads it from LDS
    issue_step = {}

    def issue(w, res, n, step):
        fifo[w].extend([res] * n)
        issue_step.setdefault(res, step)
        # WAR: the slot / buffer being overwritten
        prev = ("W", res[1] - nb) if res[0] == "W" else ("H", res[1] - 2)
        if prev in last_read_step:
            assert last_read_step[prev] < step, f"{res} issued in step {step} while {prev} is still read in step {last_read_step[prev]}"

    def wait(w, n):
        keep = fifo[w][len(fifo[w]) - n:] if n else []
        for r in fifo[w][:len(fifo[w]) - n] if n else fifo[w]:
            if r not in keep:       # a resource is complete when none of its instructions is among the n youngest
                done[w].add(r)
        fifo[w] = list(keep)

    def barrier():
        for r in set.intersection(*done):
            certified.add(r)

    def read(res, step, what):
        assert res in certified, f"step {step}: {what} reads {res} before it is certified (cpt {cpt}, tiles {ntiles})"
        last_read_step[res] = max(last_read_step.get(res, -1), step)

    # ---- prologue ----
    for w in range(NW):
        issue(w, ("H", 0), a_it, -1)
        for t in range(3):
            issue(w, ("W", t), b_it, -1)
        wait(w, b_it)
    barrier()
    if fused:
        read(("H", 0), -1, "prologue transform")
    read(("W", 0), -1, "prologue fragments")
    read(("H", 0), -1, "prologue fragments")
    barrier()

    # ---- main loop ----
    for s in range(nsteps):
        c, tap = divmod(s, nt)
        cc = c % cpt
        tile_end = cc == cpt - 1
        final = tile_end and c == nchunks - 1
        issue_w = not (final and tap >= nt - 3)
        issue_h = tap == 0 and not final

        def dma(w):
            if issue_w:
                issue(w, ("W", s + 3), b_it, s)
            if issue_h:
                issue(w, ("H", c + 1), a_it, s)
        dma_first = [w >= NW // 2 and not (tap == nt - 1 and tile_end) for w in range(NW)]
        for w in range(NW):
            if dma_first[w]:
                dma(w)
        # fragment prefetch of step s+1 (tap 8: after the epilogue, unless this is the workgroup's last step)
        if not (tap == nt - 1 and final):
            read(("W", s + 1), s, "fragment prefetch")
            read(("H", (s + 1) // nt), s, "fragment prefetch")
        if fused and not final and 3 <= tap <= 7:
            read(("H", c + 1), s, "input transform")
        if tap == nt - 1 and tile_end:
            for w in range(NW):
                wait(w, 0)                                  # vmcnt(0) ahead of the epilogue
            last_read_step[("H", c)] = max(last_read_step.get(("H", c), -1), s)   # epilogue staging window = this chunk's halo buffer
        for w in range(NW):
            if not dma_first[w]:
                dma(w)
        if tap == nt - 1 and final:
            break
        for w in range(NW):
            if nt == 4:   # phase mode: the halo issued in tap 0 is read from tap 3 on; the workgroup's last three steps issue no tile
                if tap <= 1:
                    wait(w, a_it + b_it if not final else (b_it if tap == 0 else 0))
                elif tap == 2:
                    wait(w, 0 if final else b_it)
                elif not tile_end:
                    wait(w, b_it)
            elif tap <= 1:
                wait(w, a_it + b_it if not final else b_it)
            elif tap < 6:
                wait(w, b_it)
            elif tap < 8:
                wait(w, 0 if final else b_it)
            elif not tile_end:
                wait(w, b_it)
        barrier()
    return nsteps


@pytest.mark.parametrize("cpt", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("ntiles", [1, 2, 3])
@pytest.mark.parametrize("geom", [(6, 2), (2, 2), (4, 2)])  # (halo DMA instructions per wave, weight DMA instructions per wave): 18x18 / 10x10 / 14x18 (12-row tiles) halo
def test_ring_protocol_is_safe(cpt, ntiles, geom):
    simulate(cpt, ntiles, a_it=geom[0], b_it=geom[1], fused=True)
    simulate(cpt, ntiles, a_it=geom[0], b_it=geom[1], fused=False)


@pytest.mark.parametrize("cpt", [1, 2, 3, 4, 8, 20])
@pytest.mark.parametrize("ntiles", [1, 2, 3, 5])
def test_phase_mode_ring_protocol_is_safe(cpt, ntiles):
    """conv3x3_halo3_kernel<..., PH = true> (x2-upsample conv as four phase convolutions): four steps per chunk, 4-deep weight ring"""
    simulate(cpt, ntiles, a_it=6, b_it=2, fused=False, nt=4, nb=4)
    with pytest.raises(AssertionError):   # with the nine-tap kernel's 3-deep ring the tile issued in step s would overwrite one still being read
        simulate(2, 2, a_it=6, b_it=2, fused=False, nt=4, nb=2)


def test_model_detects_a_weaker_wait():
    """The model is not vacuous: allowing one more weight tile in flight at taps 2..5 must be flagged."""
    import types
    src = simulate.__code__
    ns = {}
    code = open(__file__).read().split("def simulate(")[1].split("\n@pytest")[0]
    code = "def simulate(" + code.replace("            elif tap < 6:\n                wait(w, b_it)", "            elif tap < 6:\n                wait(w, 2 * b_it)")
    exec("NW = 8\n" + code, ns)
    with pytest.raises(AssertionError):
        ns["simulate"](2, 2)


def test_model_matches_source():
    s = open(os.path.join(ROOT, "genpercept_amd", "csrc", "conv_halo.hip")).read()
    for line in ["const bool issue_w = !(final_ && TAP >= NT - 3), issue_h = TAP == 0 && !final_;",
                 "const bool dma_first = ((ABL & 64) ? true : (ABL & 128) ? false : second_half) && !(TAP == NT - 1 && tile_end);",
                 "if (TAP <= 1) { if (!final_) halo_wait_vm<A_IT + B_IT>(); else halo_wait_vm<B_IT>(); }",
                 "else if (TAP < 6) halo_wait_vm<B_IT>();",
                 "else if (TAP < 8) { if (final_) halo_wait_vm<0>(); else halo_wait_vm<B_IT>(); }",
                 "else if (!tile_end) halo_wait_vm<B_IT>();",
                 "if (TAP <= 1) { if (!final_) halo_wait_vm<A_IT + B_IT>(); else if (TAP == 0) halo_wait_vm<B_IT>(); else halo_wait_vm<0>(); }",
                 "else if (TAP == 2) { if (final_) halo_wait_vm<0>(); else halo_wait_vm<B_IT>(); }",
                 "constexpr int WSLOT = PH ? (TAP + 3) % 4 : TAP % 3;",
                 "halo_wait_vm<0>();  // everything this wave has in flight has landed",
                 "halo_wait_vm<B_IT>();\n    __builtin_amdgcn_s_barrier();"]:
        assert line in s, line


# ---- persistent GEMM (pgemm.hip): 3-deep ring over the (tile, k) step stream ----------------------------------------------------------
def simulate_pgemm(nk, ntiles, lps=6, nb=3):
    total = nk * ntiles
    fifo = [[] for _ in range(NW)]
    done = [set() for _ in range(NW)]
    certified = set()
    last_read = {}

    def issue(w, stage, step, after_own_epilogue):
        prev = stage - nb
        if prev in last_read:
            assert last_read[prev] < step, f"stage {stage} issued in step {step} while stage {prev} is still read in step {last_read[prev]}"
        fifo[w].extend([stage] * lps)

    def wait(w, n):
        keep = fifo[w][len(fifo[w]) - n:] if n else []
        for r in (fifo[w][:len(fifo[w]) - n] if n else fifo[w]):
            if r not in keep:
                done[w].add(r)
        fifo[w] = list(keep)

    def barrier():
        certified.update(set.intersection(*done))

    def read(stage, step):
        assert stage in certified, f"step {step}: fragments of stage {stage} read before certification (nk {nk}, tiles {ntiles})"
        last_read[stage] = max(last_read.get(stage, -1), step)

    for w in range(NW):
        for st in range(min(nb, total)):
            issue(w, st, -1, False)
        wait(w, 2 * lps if (nb > 3 and total > 3) else lps if total > 2 else 0)
    barrier()
    read(0, -1)
    barrier()
    for gs in range(total):
        kt = gs % nk
        tile_end = kt == nk - 1
        issue_ok, more = gs + nb < total, gs + 1 < total
        dma_first = [w >= 4 and not tile_end for w in range(NW)]
        for w in range(NW):
            if dma_first[w] and issue_ok:
                issue(w, gs + nb, gs, False)
        if more:
            read(gs + 1, gs)          # (past the end the kernel reads stale bytes nobody uses)
        if tile_end:
            for w in range(NW):
                wait(w, 0)
        for w in range(NW):
            if not dma_first[w] and issue_ok:
                issue(w, gs + nb, gs, tile_end)   # at a tile end: after this wave's own epilogue (its window = its own DMA pieces of the slot)
        if not more:
            break
        for w in range(NW):
            if not tile_end:
                if nb == 3:
                    wait(w, lps if issue_ok else 0)
                else:  # (the kernel computes this after ++gs: behind = total - 2 - gs_new)
                    behind = total - 2 - (gs + 1)
                    wait(w, 2 * lps if behind >= 2 else lps if behind == 1 else 0)
        barrier()


@pytest.mark.parametrize("nk", [1, 2, 3, 5, 10, 40])
@pytest.mark.parametrize("ntiles", [1, 2, 4])
@pytest.mark.parametrize("lps", [6, 4])
@pytest.mark.parametrize("nb", [3, 4])
def test_pgemm_ring_protocol_is_safe(nk, ntiles, lps, nb):
    simulate_pgemm(nk, ntiles, lps, nb)


def test_pgemm_model_matches_source():
    s = open(os.path.join(ROOT, "genpercept_amd", "csrc", "pgemm.hip")).read()
    for line in ["const bool issue = gs + NB < total, more = gs + 1 < total;",
                 "const bool dma_first = second_half && !tile_end;",
                 "if (NB == 3) { if (issue) wait_vm<LPS>(); else wait_vm<0>(); }",
                 "const int behind = total - 2 - gs;",
                 "if (behind >= 2) wait_vm<2 * LPS>(); else if (behind == 1) wait_vm<LPS>(); else wait_vm<0>();",
                 "if (NB > 3 && total > 3) wait_vm<2 * LPS>(); else if (total > 2) wait_vm<LPS>(); else wait_vm<0>();",
                 'asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");']:
        assert line in s, line


# ---- generic implicit GEMM (igemm.hip): NSTAGE-deep ring, wait + barrier at the TOP of a step, operands read in the step itself ----------
def simulate_igemm(nk, nstage, lps=6):
    fifo = [[] for _ in range(NW)]
    done = [set() for _ in range(NW)]
    certified = set()
    last_read = {}

    def issue(w, stage, step):
        prev = stage - nstage
        if prev in last_read:
            assert last_read[prev] < step, f"stage {stage} overwrites stage {prev} in step {step}, still read in step {last_read[prev]}"
        fifo[w].extend([stage] * lps)

    def wait(w, n):
        keep = fifo[w][len(fifo[w]) - n:] if n else []
        for r in (fifo[w][:len(fifo[w]) - n] if n else fifo[w]):
            if r not in keep:
                done[w].add(r)
        fifo[w] = list(keep)

    for w in range(NW):
        for s0 in range(min(nstage - 1, nk)):
            issue(w, s0, -1)
    for kt in range(nk):
        ahead = min(nstage - 2, nk - 1 - kt)
        for w in range(NW):
            wait(w, 2 * lps if ahead >= 2 else lps if ahead == 1 else 0)
        certified.update(set.intersection(*done))          # s_barrier
        do_stage = kt + nstage - 1 < nk
        for w in range(NW // 2, NW):                        # second half: DMA before the MFMAs
            if do_stage:
                issue(w, kt + nstage - 1, kt)
        assert kt in certified, f"step {kt} reads its operands before they are certified (nk {nk}, nstage {nstage})"
        last_read[kt] = kt
        for w in range(NW // 2):
            if do_stage:
                issue(w, kt + nstage - 1, kt)


@pytest.mark.parametrize("nstage", [2, 3, 4])
@pytest.mark.parametrize("nk", [1, 2, 3, 5, 9, 45, 180])
def test_igemm_ring_protocol_is_safe(nk, nstage):
    simulate_igemm(nk, nstage)


def test_igemm_model_matches_source():
    s = open(os.path.join(ROOT, "genpercept_amd", "csrc", "igemm.hip")).read()
    for line in ["const int ahead = min(NSTAGE - 2, nk - 1 - kt);  // stages issued after step kt's",
                 "if (ahead >= 2) wait_vm_n<2 * LPS>();", "else if (ahead == 1) wait_vm_n<LPS>();", "else wait_vm_n<0>();",
                 "const bool do_stage = kt + NSTAGE - 1 < nk && !(p.dbg & 1);",
                 "if (second_half && do_stage) stage(nxt);", "if (!second_half && do_stage) stage(nxt);"]:
        assert line in s, line


# ---- flash_attn512_kernel (attention.hip): two K / V^T slots, one full wait + one barrier per 32-key tile --------------------------------
def simulate_flash512(tiles_per_item, nw=4):
    """Work items of a workgroup in sequence; per tile: wait vmcnt(0), barrier, then the S^T phase (reads K of the tile's slot) with the
    NEXT tile's sixteen DMA pieces issued between its MFMAs into the other slot, then the P.V phase (reads V^T of the tile's slot).  Every
    item starts with a workgroup barrier (it re-parks its Q rows in LDS and restarts the slots at 0)."""
    certified, done, fifo = set(), [set() for _ in range(nw)], [[] for _ in range(nw)]
    last_read = {}                              # resource -> event index of its last LDS read
    slot_content = {0: None, 1: None}
    clock = [0]

    def tick():
        clock[0] += 1
        return clock[0]

    def barrier():
        for r in set.intersection(*done):
            certified.add(r)
        return tick()

    last_barrier = [0]
    for item, ntiles in enumerate(tiles_per_item):
        last_barrier[0] = barrier()             # __syncthreads() at the top of the item (nothing new is certified by it)
        for t in range(ntiles):
            res = (item, t)
            if t == 0:                          # `if (t0 < t1) stage(0, t0)`: all sixteen pieces of the first tile, before the loop
                prev = slot_content[0]
                assert prev is None or last_read.get(prev, 0) < last_barrier[0], f"{res} overwrites {prev} before a barrier after its last read"
                for w in range(nw):
                    fifo[w].append(res)
                slot_content[0] = res
            for w in range(nw):                 # wait_vm<0>()
                done[w].update(fifo[w])
                fifo[w] = []
            last_barrier[0] = barrier()
            slot = t & 1
            assert slot_content[slot] == res and res in certified, f"tile {res}: S^T reads a slot that is not certified"
            if t + 1 < ntiles:                  # stage_piece(slot_next, kt + 1, ...) between the S^T MFMAs
                nxt = (item, t + 1)
                prev = slot_content[slot ^ 1]
                assert prev is None or last_read.get(prev, 0) < last_barrier[0], f"{nxt} overwrites {prev} before a barrier after its last read"
                for w in range(nw):
                    fifo[w].append(nxt)
                slot_content[slot ^ 1] = nxt
            last_read[res] = tick()             # S^T phase reads K, then (possibly after the cold rescale path) P.V reads V^T
            last_read[res] = tick()


@pytest.mark.parametrize("tiles", [[1], [2], [3, 1], [288, 36], [5, 4, 7], [1, 1, 1]])
def test_flash512_slot_protocol_is_safe(tiles):
    simulate_flash512(tiles)


def test_flash512_model_detects_a_missing_item_barrier():
    """Without the barrier at the top of an item, the first tile of the next item would be staged into slot 0 while slower waves may
    still read the previous item's last tile from it (odd tile counts end in slot 0)."""
    def broken(tiles_per_item):  # the same bookkeeping for slot 0, the item barrier left out
        clock, last_read, slot0, last_barrier = [0], {}, None, 0
        for item, ntiles in enumerate(tiles_per_item):
            for t in range(ntiles):
                res = (item, t)
                if t == 0:
                    assert slot0 is None or last_read.get(slot0, 0) < last_barrier, "overwrite before barrier"
                if (t & 1) == 0:
                    slot0 = res
                clock[0] += 1
                last_barrier = clock[0]         # per-tile barrier
                clock[0] += 1
                last_read[res] = clock[0]
    with pytest.raises(AssertionError):
        broken([3, 1])
    broken([2, 1])  # (an even tile count ends in slot 1: the per-tile barrier of the last tile already covers slot 0)


def test_flash512_model_matches_source():
    src = open(os.path.join(ROOT, "genpercept_amd", "csrc", "attention.hip")).read()
    k = src[src.index("void flash_attn512_kernel"):src.index("void flash512_combine_kernel")]
    loop = k[k.index("for (; kt < t1; ++kt) {"):]
    assert loop.index("wait_vm<0>();") < loop.index("__builtin_amdgcn_s_barrier();") < loop.index("s_phase(sb, kt, nx, rel ^ 1") < loop.index("pv_phase(sb);")
    assert "stage_piece(slot_next, kt + 1, ks >> 1)" in k and "if (t0 < t1) stage(0, t0);" in k
    assert k.index("__syncthreads();") < k.index("if (t0 < t1) stage(0, t0);")   # the item barrier precedes the first staging of the item
