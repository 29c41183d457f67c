"""CPU: the LDS bank model behind the round-5 plane layouts (tools/lds_conflicts.py; measured on the device by tools/probe_ldsfrag.hip:
127 B/clk/CU for rows of columns + 16 bytes, 224-235 for the layouts below) and the swizzles' address maps as the kernels compute them
(rd_msgpass_fused.hip `pofs`, rd_attnfuse.hip `tofs`): bijections onto the plane, conflict-free for the reads they were chosen for."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import lds_conflicts as L  # noqa: E402


def pofs(row, col, ldx=256):          # rd_msgpass_fused.hip: element offset of (row, col) in a split plane (rows of 512 bytes)
    return row * ldx + (col & 128) + ((((col >> 3) ^ row) & 15) << 3) + (col & 7)


def tofs(row, col, ldt=64):           # rd_attnfuse.hip: [feature][step] / score planes (rows of 128 bytes)
    tsw = (row & 3) | (((row >> 3) & 1) << 2)
    return row * ldt + ((((col >> 3) ^ tsw) & 7) << 3) + (col & 7)


def test_swizzles_are_bijections_that_keep_16_byte_chunks_whole():
    for f, rows, cols in ((pofs, 48, 256), (tofs, 96, 64)):
        seen = set()
        for r in range(rows):
            for c in range(cols):
                o = f(r, c)
                assert r * cols <= o < (r + 1) * cols            # a row stays inside its own bytes
                assert o % 8 == c % 8                            # the position inside a 16-byte chunk is untouched
                seen.add(o)
        assert len(seen) == rows * cols


def test_fragment_read_is_two_way_conflicted_with_16_byte_padding_and_free_with_32():
    for nkc in (5, 8, 9):                                        # encoder x planes, K1 planes, encoder hidden planes
        n16 = 4 * nkc
        assert L.frag_cycles(lambda r, c: r * (n16 * 16 + 16) + c * 16, nkc) == 8.0      # rounds 1-4: "+ 8 elements"
        assert L.frag_cycles(lambda r, c: r * (n16 * 16 + 32) + c * 16, nkc) == 4.0      # round 5: + 32 bytes
    assert L.frag_cycles(lambda r, c: r * 512 + c * 16, 8) == 32.0                       # plain 512-byte rows: every lane on one quad
    assert L.frag_cycles(lambda r, c: pofs(r, 8 * c) * 2, 8) == 4.0                      # K1: 512-byte rows, chunk ^ (row & 15)


def _tr_cycles(addr_of_lane):
    """ds_read_b64_tr_b16: two 32-lane groups, 8 bytes per lane over 64 banks"""
    tot = 0
    for g in (range(0, 32), range(32, 64)):
        busy = {}
        for l in g:
            a = addr_of_lane(l)
            for q in range(2):
                busy.setdefault(((a // 4) + q) % 64, set()).add(a // 8)
        tot += max(len(v) for v in busy.values())
    return tot


def test_transposed_planes_reads_are_conflict_free_with_the_swizzle():
    def tr(addr, k0, col0, second):
        return _tr_cycles(lambda l: addr(k0 + 8 * (l >> 4) + ((l & 15) >> 2) + 4 * second, col0 + 4 * (l & 3)))
    old = lambda r, c: (r * 72 + c) * 2                          # round 4: rows of TS + 8 elements
    new = lambda r, c: tofs(r, c) * 2
    for k0 in (0, 32, 64):
        for col0 in (0, 16, 32, 48):
            for sec in (0, 1):
                assert tr(old, k0, col0, sec) == 4               # 2-way on both halves
                assert tr(new, k0, col0, sec) == 2               # free
    # the same planes read as 16-byte fragments (lane: row l & 15, chunk l >> 4)
    for row0 in (0, 16, 32, 80):
        for k0 in (0, 32):
            assert L.cycles(lambda l: new(row0 + (l & 15), k0 + 8 * (l >> 4))) == 4
            assert L.cycles(lambda l: old(row0 + (l & 15), k0 + 8 * (l >> 4))) == 8


def test_chain_stage_rows_take_the_accumulator_stores_without_conflicts():
    """rd_encfuse.hip `to_stage` (round 5, second half): an accumulator of the swapped-operand product is four consecutive columns of
    one row -- lane (row l & 15, chunk 4 j + (l >> 4)) stores 16 bytes into the fp32 stage.  Rows of KPD + 8 floats (672 bytes) are
    conflict-free in the 16-lane group model; KPD + 4 (the first version of that change) was 2-way, plain KPD rows 4-way."""
    KPD = 160

    def store(stg, j):
        return L.cycles(lambda l: (l & 15) * stg * 4 + (4 * j + (l >> 4)) * 16)
    for j in range(10):
        assert store(KPD + 8, j) == 4
        assert store(KPD + 4, j) == 8
        assert store(KPD, j) == 16
