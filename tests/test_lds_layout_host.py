"""CPU checks of the LDS slot layouts of csrc/conv_wino4_wgrad.hip: the raw-operand ring slots are filled by LDS-direct
16-byte loads (lane i of a request lands at base + 16 i, so the layout IS the request's lane -> (row, channel, group)
map) and read back by the transform with ds_read_b128.  A ds_read_b128 is served in four 16-lane groups, 64 banks of 4
bytes = 16 slots of 16 bytes per cycle (MI355X_MICROARCH.md, LDS table): a group is conflict-free when its 16 slot
indices are distinct modulo 16.  The formulas below restate the kernel's (kept in sync by hand; the kernel's results are
checked on the GPU by tests/kernel_checks.py: wino4_wgrad*)."""
import numpy as np

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def conflict_free(slot_of_lane):
    return all(len({slot_of_lane(l) % 16 for l in g}) == 16 for g in B128_GROUPS)


def test_groups_partition_the_wave():
    assert sorted(l for g in B128_GROUPS for l in g) == list(range(64))


def test_dy_slot_layout():
    """dY slot: [row 4][co 64][group ^ ((co >> 2) & 3)] — request lane -> slot is the identity, reader = (tile, co = lane)"""
    # every (row, co, true group) is fetched exactly once by the 16 requests of 64 lanes
    seen = set()
    for pos in range(16 * 64):
        row, co, p = pos >> 8, (pos & 255) >> 2, (pos & 3) ^ (((pos & 255) >> 2 >> 2) & 3)
        seen.add((row, co, p))
    assert len(seen) == 4 * 64 * 4
    # four consecutive lanes of a request cover one 64-byte row segment of one channel (coalesced quads)
    for pos in range(0, 16 * 64, 4):
        assert len({(q >> 8, (q & 255) >> 2) for q in range(pos, pos + 4)}) == 1
    for tile in range(4):
        assert conflict_free(lambda lane: lane * 4 + (tile ^ ((lane >> 2) & 3)))
    # without the swizzle the same read is a 4-way conflict
    assert not conflict_free(lambda lane: lane * 4 + 1)


def test_x_slot_layout():
    """x slot: [row 6][ci 32][(group + ((ci >> 3) & 1)) % 6]; reader lanes = (tile parity hh = lane >> 5, ci = lane & 31)
    read group tile + 1 (patch columns 1..4), the (0,5) role also groups tile and tile + 2"""
    seen = set()
    for pos in range(18 * 64):
        row, rem = divmod(pos, 192)
        ci = rem // 6
        p = (rem - ci * 6 + 6 - ((ci >> 3) & 1)) % 6
        seen.add((row, ci, p))
    assert len(seen) == 6 * 32 * 6
    for tbase in (0, 2):          # k-step B: tiles 0,1; k-step A: tiles 2,3
        for delta in (0, 1, 2):   # groups tile, tile + 1, tile + 2
            def slot(lane, tbase=tbase, delta=delta):
                ci, t = lane & 31, tbase + (lane >> 5)
                return ci * 6 + (t + delta + ((ci >> 3) & 1)) % 6
            assert conflict_free(slot), (tbase, delta)
    # the plain stride of 6 slots per channel is a 2-way conflict
    assert not conflict_free(lambda lane: (lane & 31) * 6 + (lane >> 5) + 1)


def test_prologue_group_mapping():
    """the fused prologue rewrites group tid (rows 0-3) and 768 + tid (rows 4,5; first six waves): a wave covers 64
    consecutive groups of ONE row, so row and channel base are wave-uniform and both groups share channel / group-of-row"""
    groups = []
    for tid in range(768):
        wave, lane = tid >> 6, tid & 63
        rem = (wave % 3) * 64 + lane
        g0 = (wave // 3) * 192 + rem
        groups.append(g0)
        assert g0 // 192 == wave // 3 and g0 % 192 == rem
        if wave < 6:
            g1 = g0 + 768
            groups.append(g1)
            assert g1 // 192 == 4 + wave // 3 and g1 % 192 == rem
    assert sorted(groups) == list(range(1152))


def test_transformed_operand_reads_are_lane_linear():
    """V[36][2][32] / Mg[36][2][64] of a k-step: a dword read serves 32 lanes per cycle on 32 banks — the MFMA lanes'
    addresses are consecutive dwords within each half wave"""
    for wj in range(6):
        for wsb in range(2):
            mg = [wj * 128 + (lane >> 5) * 64 + wsb * 32 + (lane & 31) for lane in range(64)]
            v = [wj * 64 + lane for lane in range(64)]
            for half in (mg[:32], mg[32:], v[:32], v[32:]):
                assert len({a % 32 for a in half}) == 32
