"""What the compiler made of the hot kernels, checked at build time (no GPU): `raindrop_amd.build` compiles with
-Rpass-analysis=kernel-resource-usage and keeps the per-kernel numbers next to the objects.

Why these are tests and not notes: each of them was a measured loss once (rules 11 and 21) --
  * register spills in the two fused encoder chains were 9 % of the training step,
  * the K = 3D row-block product at 178 registers ran one workgroup per CU and therefore TWO rounds on 256 CUs (8 us per call),
and both are one careless edit away from coming back without any functional test noticing."""
import pytest

from raindrop_amd import build


@pytest.fixture(scope="module")
def usage():
    build.build(verbose=False)
    u = build.resource_usage()
    assert len(u) > 100, "no resource-usage records: was the library built by raindrop_amd.build?"
    return u


_CACHE = {}


def _sizes_and_offsets():
    """kernel code sizes and s_getpc offsets of the built library, once per test session (the disassembly takes seconds)"""
    if "v" not in _CACHE:
        build.build(verbose=False)
        _CACHE["v"] = (build.kernel_code_sizes(), build.getpc_offsets())
    return _CACHE["v"]


def _find(usage, *parts):
    hits = [(k, v) for k, v in usage.items() if all(p in k for p in parts)]
    assert hits, "no kernel matching %r" % (parts,)
    return hits


# (name fragments, max VGPRs or None): the kernels of the P19 training step and of the P12 / PAM paths
HOT = [
    (("k_msg_fwd_fusedILi3ELi34ELi60E",), 128), (("k_msg_bwd_fusedILi3ELi34ELi60E",), 128), (("4k_dwE",), None), (("k_wprep",), None),
    (("k_enc_post_fwdILi152ELi272E",), 128), (("k_enc_pre_bwdILi152ELi272E",), 128),          # 16 waves per workgroup: 128 is the cap
    (("k_enc_post_fwdILi160ELi288E",), 128), (("k_enc_pre_bwdILi160ELi288E",), 128),
    (("k_rowgemmILi5ELi32ELi2ELb0ELb0ELi8E",), 128),                                         # two 8-wave workgroups per CU
    (("k_rowgemmILi15ELi32ELi1ELb0ELb0ELi8E",), 128),                                        # two per CU: ONE round of 266 workgroups
    (("5k_twgILb0E",), None), (("5k_twgILb1E",), None), (("k_twg_reduce",), None),
    (("k_attn_fwd_fusedILi5ELi5E",), 256), (("k_attn_bwd_fusedILi5ELi5E",), 256),             # 8 waves per workgroup, one workgroup per CU
    (("k_attn_fwd_one_b16wILi5ELb1E",), 128), (("k_attn_bwd_one_b16wILi5ELb1E",), 128),
    (("k_attn_fwd_b16ILi5ELb1ELb0E",), 256), (("k_attn_bwd_dq_b16ILi5ELb1ELb0E",), 256), (("k_attn_bwd_dkv_b16ILi5ELb1ELb0E",), 256),
    (("k_head_rowsILi1E",), 128), (("k_head_wgrad",), None), (("k_gemm_panelILi2E",), 256), (("k_gemm_panel_wideILi2E",), 256), (("k_gemm_panel_pcILi2E",), 256), (("k_adam",), None), (("k_wsplit",), None),
]


@pytest.mark.parametrize("parts,max_vgprs", HOT, ids=[p[0][0] for p in HOT])
def test_hot_kernel_has_no_scratch_and_fits_its_occupancy(usage, parts, max_vgprs):
    for name, u in _find(usage, *parts):
        assert u["scratch"] == 0 and u["vgpr_spill"] == 0, (name, u)          # (SGPR spills go to VGPR lanes, not to memory)
        if max_vgprs is not None:
            assert u["vgprs"] + u.get("agprs", 0) <= max_vgprs, (name, u)


def test_scratch_users_are_the_known_ones(usage):
    """Every kernel with a non-zero scratch size is on this list (the runtime-width backward chain, launched only under
    RD_ENC_SPECIALIZE=0 -- production shapes outside the two compiled-in width pairs take the row-block launches --, the
    preprocessing kernels' local arrays, one 12-byte spill at head_dim 81..96 without 16-byte rows): a new entry is a regression.
    (Round 4 removed the F > 48 instantiations of the fused message passing, which spilled 68-88 bytes per lane.)"""
    known = ("k_enc_pre_bwdILi0ELi0E", "k_pw_leaves", "k_pw_combine", "k_attn_bwd_dkv_b16ILi6ELb0ELb0E")
    new = [k for k, u in usage.items() if u.get("scratch") and not any(n in k for n in known)]
    assert not new, new


def test_resource_remark_parser_keeps_real_warnings():
    """The build filters hipcc's kernel-resource-usage remarks (and the source excerpts under them) out of what it echoes, and must
    not swallow a real diagnostic that follows."""
    err = "\n".join([
        "a.hip:11:1: remark: Function Name: _Z1kPf [-Rpass-analysis=kernel-resource-usage]",
        "   11 |   float b1) {",
        "      | ^",
        "a.hip:11:1: remark:     VGPRs: 37 [-Rpass-analysis=kernel-resource-usage]",
        "a.hip:11:1: remark:     AGPRs: 0 [-Rpass-analysis=kernel-resource-usage]",
        "a.hip:11:1: remark:     ScratchSize [bytes/lane]: 12 [-Rpass-analysis=kernel-resource-usage]",
        "a.hip:11:1: remark:     Occupancy [waves/SIMD]: 8 [-Rpass-analysis=kernel-resource-usage]",
        "a.hip:11:1: remark:     VGPRs Spill: 3 [-Rpass-analysis=kernel-resource-usage]",
        "a.hip:11:1: remark:     LDS Size [bytes/block]: 4096 [-Rpass-analysis=kernel-resource-usage]",
        "a.hip:20:7: warning: unused variable 'x' [-Wunused-variable]",
        "   20 |   int x;",
        "      |       ^",
    ])
    usage, rest = build._parse_resource_remarks(err)
    assert usage == {"_Z1kPf": {"vgprs": 37, "agprs": 0, "scratch": 12, "occupancy": 8, "vgpr_spill": 3, "lds": 4096}}
    assert "unused variable" in rest and "int x;" in rest and "remark" not in rest and "float b1" not in rest


# Own-code touch lengths (rd_common.h touch_own_code): GENERATED by the build from the linked code objects (raindrop_amd/build.py
# TOUCH_SITES -> csrc/rd_touch_gen.h), checked before the library is moved into place.
SITES = build.TOUCH_SITES


def test_build_checks_code_touch_lengths():
    """The check the build runs on a freshly linked library fails when a touched range would leave its kernel, and the generated
    table IS the fixed point for the library that is in place."""
    build.build(verbose=False)
    table = build.read_touch_table()
    assert set(table) == {s for s, _, _ in SITES}
    build.check_code_touch()
    with pytest.raises(RuntimeError, match="own-code touch"):
        build.check_code_touch(table=dict(table, DW=1 << 20))
    assert build.ideal_touch_table(build.LIB, *_sizes_and_offsets()) == table


@pytest.mark.parametrize("site,pat,kind", SITES, ids=[c[0] for c in SITES])
def test_own_code_touch_stays_inside_the_kernel(site, pat, kind):
    """touch_own_code reads [pc, pc + bytes) behind the kernel's s_getpc_b64: the range must end inside EVERY instantiation the site's
    pattern matches (the s_getpc's offset from the disassembly + 8 bytes, or 384 bytes of slack without the disassembler -- the rule
    of build.check_code_touch), and the source must take its length from the generated macro, not from a literal."""
    import os
    import re
    all_sizes, offs = _sizes_and_offsets()
    touch = build.read_touch_table()[site]
    sizes = {k: v for k, v in all_sizes.items() if re.search(pat, k)}
    assert sizes, site
    for k, v in sizes.items():
        slack = offs[k] + build.CODE_TOUCH_MARGIN if k in offs else build.CODE_TOUCH_SLACK
        assert v >= touch + slack, (k, v, touch, slack)
    src = "".join(open(os.path.join(build.CSRC, f)).read() for f in sorted(os.listdir(build.CSRC)) if f.endswith(".hip"))
    assert re.search(r"\bRD_TL_%s\b" % site, src), site
    assert not re.search(r"RD_TOUCH_CODE(_FIRST|_X)?\(\d", src), "a literal touch length is back in the sources"


def test_step_kernels_touch_all_of_their_code():
    """Not more than 256 bytes of any kernel of the P19 step lie behind the touched range (round 5: an uncovered tail is what a box
    without instruction look-ahead fetches cold, line by line -- the tall chain body's LayerNorm2 ran 9 k cycles instead of 5 k).
    The generic-shape instantiations of the fused message passing share one length (the smallest of three)."""
    import re
    sizes, offs = _sizes_and_offsets()
    if not offs:
        pytest.skip("llvm-objdump not installed")
    table = build.read_touch_table()
    for site, pat, kind in SITES:
        if kind != "step" or site in ("K1_FWD", "K1_BWD"):
            continue
        for k, v in sizes.items():
            if re.search(pat, k):
                assert v - offs[k] - table[site] <= 256, (k, v, offs[k], table[site])


def test_own_code_touch_starts_right_behind_the_entry():
    """The touched range is [address behind s_getpc_b64, + bytes): the s_getpc must sit close to the kernel's entry in EVERY
    instantiation -- the compiler is free to schedule the prologue in front of it, and what lies in front is not touched."""
    import re
    sizes, offs = _sizes_and_offsets()
    if not offs:
        pytest.skip("llvm-objdump not installed")
    table = build.read_touch_table()
    n = 0
    for site, pat, kind in SITES:
        for k in sizes:
            if re.search(pat, k) and k in offs:
                n += 1
                assert offs[k] <= 384 and offs[k] + 8 + table[site] <= sizes[k], (k, offs[k], table[site], sizes[k])
    assert n >= 40, n


def test_stale_or_unchecked_library_is_relinked(tmp_path, monkeypatch):
    """the library in place is always one that passed the touch check -- it is linked under a temporary name and
    renamed afterwards, and its stamp (hash of the objects it was linked from) makes build() relink when it does not match."""
    import os
    build.build(verbose=False)
    stamp = build.LIB + ".stamp"
    good = open(stamp).read()
    try:
        with open(stamp, "w") as fh:
            fh.write("not the objects' hash")
        build.build(verbose=False)
        assert open(stamp).read() == good and not os.path.exists(build.LIB + ".tmp")
    finally:
        with open(stamp, "w") as fh:
            fh.write(good)
