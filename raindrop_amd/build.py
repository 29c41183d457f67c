"""Build recipe for libraindrop_hip.so (gfx950 only, in-tree so the .so travels with gpurun).

    python -m raindrop_amd.build            # incremental
    python -m raindrop_amd.build --force
    python -m raindrop_amd.build --touch-table    # code size / own-code touch length / uncovered tail of every kernel

hipcc cross-compiles without a GPU; each translation unit becomes an object under
`raindrop_amd/csrc/_build/` and the objects are linked into `raindrop_amd/libraindrop_hip.so`.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(PKG, "libraindrop_hip.so")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         "-Rpass-analysis=kernel-resource-usage",       # per-kernel registers / scratch / LDS -> <object>.usage.json (tests/test_kernel_resources.py)
         "-I", INCLUDE]


# Own-code touch lengths (rd_common.h touch_own_code): every kernel starts by requesting `RD_TL_<site>` bytes of its OWN code behind
# its s_getpc_b64 into L2.  The loads must stay inside the kernel in EVERY instantiation -- a load past the end of the last kernel of
# a code object is a memory fault -- and should cover all of it (an uncovered tail is fetched cold, line by line, on the pool's boxes
# without instruction look-ahead).  Until round 5 the lengths were hand-kept literals next to a build-time check, i.e. any compiler
# point release or code edit broke the build.  Now they are GENERATED from the linked code objects: `csrc/rd_touch_gen.h` holds one
# `#define RD_TL_<site> <bytes>` per site below; build() links to a temporary file, measures every kernel (ELF symbol sizes +
# s_getpc offsets from the disassembly), rewrites the header where a length is off, recompiles the translation units that use a
# changed macro, and repeats until the table is a fixed point (the literal's encoding can move the code size by 4 bytes once);
# only a library whose table passes check_code_touch is moved into place.
#   (site, regex over the mangled kernel name, "step" | "x")   step: kernels of the P19 training step (RD_TOUCH_CODE: the WHOLE kernel);
#   x: the kernels of the other configurations (RD_TOUCH_CODE_X: one length per template = its smallest instantiation).
TOUCH_SITES = [
    ("K1_FWD_P19", r"k_msg_fwd_fusedILi3ELi34ELi60E", "step"), ("K1_BWD_P19", r"k_msg_bwd_fusedILi3ELi34ELi60E", "step"),
    ("K1_FWD", r"k_msg_fwd_fusedILi\dELi0E", "step"), ("K1_BWD", r"k_msg_bwd_fusedILi\dELi0E", "step"),
    ("ATTN_FWD", r"k_attn_fwd_fusedILi\d+ELi\d+ELb0E", "step"), ("ATTN_BWD", r"k_attn_bwd_fusedILi\d+ELi\d+ELb0E", "step"),
    ("ATTN_FWD_B", r"k_attn_fwd_fusedILi\d+ELi\d+ELb1E", "step"), ("ATTN_BWD_B", r"k_attn_bwd_fusedILi\d+ELi\d+ELb1E", "step"),
    ("EF_POST_P19L", r"k_enc_post_fwdILi152ELi272ELb1ELb0E", "step"), ("EF_POST_P19L_B", r"k_enc_post_fwdILi152ELi272ELb1ELb1E", "step"), ("EF_POST_P19", r"k_enc_post_fwdILi152ELi272ELb0ELb0E", "step"), ("EF_POST_P19_B", r"k_enc_post_fwdILi152ELi272ELb0ELb1E", "step"),
    ("EF_POST_P12L", r"k_enc_post_fwdILi160ELi288ELb1ELb0E", "step"), ("EF_POST_P12L_B", r"k_enc_post_fwdILi160ELi288ELb1ELb1E", "step"), ("EF_POST_P12", r"k_enc_post_fwdILi160ELi288ELb0ELb0E", "step"), ("EF_POST_P12_B", r"k_enc_post_fwdILi160ELi288ELb0ELb1E", "step"),
    ("EF_POST_RTL", r"k_enc_post_fwdILi0ELi0ELb1ELb0E", "step"), ("EF_POST_RTL_B", r"k_enc_post_fwdILi0ELi0ELb1ELb1E", "step"), ("EF_POST_RT", r"k_enc_post_fwdILi0ELi0ELb0ELb0E", "step"), ("EF_POST_RT_B", r"k_enc_post_fwdILi0ELi0ELb0ELb1E", "step"),
    ("EF_PRE_P19L", r"k_enc_pre_bwdILi152ELi272ELb1ELb0E", "step"), ("EF_PRE_P19L_B", r"k_enc_pre_bwdILi152ELi272ELb1ELb1E", "step"), ("EF_PRE_P19", r"k_enc_pre_bwdILi152ELi272ELb0ELb0E", "step"), ("EF_PRE_P19_B", r"k_enc_pre_bwdILi152ELi272ELb0ELb1E", "step"),
    ("EF_PRE_P12L", r"k_enc_pre_bwdILi160ELi288ELb1ELb0E", "step"), ("EF_PRE_P12L_B", r"k_enc_pre_bwdILi160ELi288ELb1ELb1E", "step"), ("EF_PRE_P12", r"k_enc_pre_bwdILi160ELi288ELb0ELb0E", "step"), ("EF_PRE_P12_B", r"k_enc_pre_bwdILi160ELi288ELb0ELb1E", "step"),
    ("EF_PRE_RTL", r"k_enc_pre_bwdILi0ELi0ELb1ELb0E", "step"), ("EF_PRE_RTL_B", r"k_enc_pre_bwdILi0ELi0ELb1ELb1E", "step"), ("EF_PRE_RT", r"k_enc_pre_bwdILi0ELi0ELb0ELb0E", "step"), ("EF_PRE_RT_B", r"k_enc_pre_bwdILi0ELi0ELb0ELb1E", "step"),
    ("DW", r"4k_dwE", "step"), ("DW_REDUCE", r"k_dw_reduce", "step"), ("ADAM", r"6k_adamE", "step"), ("ADAM_DEV", r"10k_adam_devE", "step"),
    ("WSPLIT", r"k_wsplit", "step"), ("TWG", r"5k_twgILb0E", "step"), ("TWG_ONE", r"5k_twgILb1E", "step"),
    ("HEAD_P19", r"k_head_rowsILi1ELi12ELi3E", "step"), ("HEAD", r"k_head_rowsILi1ELi16ELi4E", "step"),
    ("GEMM", r"6k_gemmI", "x"), ("GEMM_X3", r"13k_gemm_bf16x3I", "x"), ("GEMM_PANEL", r"12k_gemm_panelI", "x"), ("GEMM_PANEL_WIDE", r"17k_gemm_panel_wideI", "x"), ("GEMM_PANEL_PC", r"15k_gemm_panel_pcI", "x"), ("ROWGEMM", r"9k_rowgemmI", "x"),
    ("ATTN_FWD_ONE", r"19k_attn_fwd_one_b16wI", "x"), ("ATTN_BWD_ONE", r"19k_attn_bwd_one_b16wI", "x"), ("ATTN_FWD_B16", r"14k_attn_fwd_b16I", "x"),
    ("ATTN_BWD_DQ", r"17k_attn_bwd_dq_b16I", "x"), ("ATTN_BWD_DKV", r"18k_attn_bwd_dkv_b16I", "x"), ("ADD_LN_FWD", r"14k_add_ln_fwd_vE", "x"),
    ("LN_BWD_R", r"10k_ln_bwd_rI", "x"), ("LN_BWD_V", r"10k_ln_bwd_vE", "x"),
]
TOUCH_GEN = os.path.join(CSRC, "rd_touch_gen.h")
CODE_TOUCH_SLACK = 384          # bytes allowed for the prologue in front of the s_getpc_b64 when the disassembler is not there to say
CODE_TOUCH_MARGIN = 8           # bytes kept free behind the last touched line (the s_getpc's own length + the last dword)
TOUCH_MAX_PASSES = 4


def read_touch_table(path=None):
    """{site: bytes} of csrc/rd_touch_gen.h ({} when the file is missing)."""
    path = path or TOUCH_GEN
    if not os.path.exists(path):
        return {}
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define RD_TL_(\w+) (\d+)\s*$", open(path).read(), re.M)}


def _write_touch_table(table):
    lines = ["// GENERATED by raindrop_amd/build.py from the linked code objects -- do not edit (see TOUCH_SITES there).",
             "// Bytes of its own code each kernel requests into L2 at its start (rd_common.h touch_own_code).", "#pragma once"]
    lines += ["#define RD_TL_%s %d" % (site, table.get(site, 0)) for site, _, _ in TOUCH_SITES]
    with open(TOUCH_GEN, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def ideal_touch_table(lib, sizes=None, offs=None):
    """{site: the longest touch that stays inside every kernel the site's pattern matches} for the library `lib`: the largest multiple
    of 128 not above size - (s_getpc offset + margin) (without the disassembler: size - CODE_TOUCH_SLACK)."""
    sizes = kernel_code_sizes(lib) if sizes is None else sizes
    offs = getpc_offsets(lib) if offs is None else offs
    out = {}
    for site, pat, _ in TOUCH_SITES:
        rx = re.compile(pat)
        room = [v - (offs[k] + CODE_TOUCH_MARGIN if k in offs else CODE_TOUCH_SLACK) for k, v in sizes.items() if rx.search(k)]
        if not room:
            raise RuntimeError("own-code touch site %s: no kernel matches %r in %s" % (site, pat, lib))
        out[site] = max(0, min(room) // 128 * 128)
    return out


def check_code_touch(lib=None, table=None):
    """Raise if any touched range could leave its kernel: [s_getpc offset, + touch) must lie inside the kernel's code for every
    kernel a site's pattern matches.  `table`: {site: bytes}, default the generated header the library was compiled with."""
    table = read_touch_table() if table is None else table
    sizes = kernel_code_sizes(lib)
    offs = getpc_offsets(lib)
    bad = []
    for site, pat, _ in TOUCH_SITES:
        rx = re.compile(pat)
        ks = {k: v for k, v in sizes.items() if rx.search(k)}
        if not ks:
            bad.append("%s: no kernel matches %r" % (site, pat))
        if site not in table:
            bad.append("%s: not in %s" % (site, TOUCH_GEN))
            continue
        for k, v in ks.items():
            slack = offs[k] + CODE_TOUCH_MARGIN if k in offs else CODE_TOUCH_SLACK
            if v < table[site] + slack:
                bad.append("%s %s: %d bytes of code, touches %d (+%d slack)" % (site, k, v, table[site], slack))
    if bad:
        raise RuntimeError("own-code touch lengths exceed the built kernels (rd_common.h touch_own_code; raindrop_amd/build.py "
                           "TOUCH_SITES):\n  " + "\n  ".join(bad))


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(src, touch=None):
    """sha1 over the source, the hand-written headers and the flags; of the GENERATED touch table only the macros this source
    names enter (a changed length recompiles the translation units that use it, not the library)."""
    h = hashlib.sha1()
    for path in [src] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h") and f != "rd_touch_gen.h"] \
            + [os.path.join(INCLUDE, "raindrop_hip.h")]:
        with open(path, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(f for f in FLAGS if not f.startswith("/")).encode())   # path-independent
    touch = read_touch_table() if touch is None else touch
    used = sorted(set(re.findall(r"RD_TL_(\w+)", open(src).read())))
    h.update(" ".join("%s=%s" % (u, touch.get(u)) for u in used).encode())
    return h.hexdigest()


def _compile(src, force, objdir=None, extra=()):
    obj = os.path.join(objdir or OBJ, os.path.basename(src)[:-4] + ".o")
    stamp_file = obj + ".stamp"
    stamp = _stamp(src) + " ".join(extra)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file) \
            and open(stamp_file).read() == stamp:
        return obj, False
    cmd = [HIPCC] + FLAGS + list(extra) + ["-c", src, "-o", obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, res.stdout, res.stderr))
    usage, rest = _parse_resource_remarks(res.stderr)
    with open(obj + ".usage.json", "w") as fh:
        json.dump(usage, fh, indent=0, sort_keys=True)
    if rest.strip():
        sys.stderr.write(rest)
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return obj, True


_FN = re.compile(r"remark:\s+Function Name: (\S+)\s+\[-Rpass-analysis=kernel-resource-usage\]")
_KV = re.compile(r"remark:\s+([A-Za-z][A-Za-z \[\]/]*?): (\S+)\s+\[-Rpass-analysis=kernel-resource-usage\]")
_CTX = re.compile(r"^\s*\d*\s*\|")                 # the source excerpt / caret lines clang prints under a diagnostic
_KEYS = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
         "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill", "LDS Size [bytes/block]": "lds"}


def _parse_resource_remarks(stderr):
    """hipcc's kernel-resource-usage remarks -> {mangled kernel name: {vgprs, agprs, scratch, occupancy, lds, ..}}; returns that and
    the rest of stderr (real warnings)."""
    usage, rest, cur, after_remark = {}, [], None, False
    for line in stderr.splitlines():
        if "-Rpass-analysis=kernel-resource-usage" in line:
            after_remark = True
            m = _FN.search(line)
            if m:
                cur = usage.setdefault(m.group(1), {})
                continue
            m = _KV.search(line)
            if m and cur is not None and m.group(1) in _KEYS:
                v = m.group(2)
                cur[_KEYS[m.group(1)]] = int(v) if v.lstrip("-").isdigit() else v
            continue
        if after_remark and _CTX.match(line):
            continue
        after_remark = False
        rest.append(line)
    return usage, "\n".join(rest) + ("\n" if rest else "")


def resource_usage():
    """{mangled kernel name: usage} over every object of the last build (raindrop_amd/csrc/_build/*.usage.json)."""
    out = {}
    for f in sorted(os.listdir(OBJ)) if os.path.isdir(OBJ) else []:
        if f.endswith(".usage.json"):
            with open(os.path.join(OBJ, f)) as fh:
                out.update(json.load(fh))
    return out


def _objdump():
    """llvm-objdump of the toolchain HIPCC belongs to (or the default ROCm location); None when there is none."""
    cands = [os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin", "llvm-objdump"),
             os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "llvm-objdump"), "/opt/rocm/lib/llvm/bin/llvm-objdump"]
    for c in cands:
        if os.path.exists(c):
            return os.path.realpath(c)
    import shutil
    return shutil.which("llvm-objdump")


def _link(objs, out):
    res = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (res.stdout, res.stderr))


def _lib_stamp(objs):
    h = hashlib.sha1()
    for o in objs:
        with open(o + ".stamp") as fh:
            h.update(fh.read().encode())
    return h.hexdigest()


def _build_checked(srcs, force, objdir, extra, lib, touch_on=True):
    """Compile + link `lib` with the own-code touch table brought to a fixed point (see TOUCH_SITES).  The library is linked under a
    temporary name and renamed into place only after check_code_touch passed: a failed build never leaves an unchecked library
    behind (and `lib`.stamp -- the hash of the objects it was linked from -- makes a stale or unchecked one get relinked)."""
    if not os.path.exists(TOUCH_GEN):
        _write_touch_table({})                               # first pass without any touch; the loop below fills it in
    elif touch_on and any(site not in read_touch_table() for site, _, _ in TOUCH_SITES):
        _write_touch_table(read_touch_table())               # a new site starts at 0 (no touch) and is measured like the rest
    tmp = lib + ".tmp"
    rebuilt_any = False
    for it in range(TOUCH_MAX_PASSES):
        with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
            results = list(ex.map(lambda s: _compile(s, force and it == 0, objdir, tuple(extra)), srcs))
        objs = [o for o, _ in results]
        rebuilt = any(r for _, r in results)
        rebuilt_any = rebuilt_any or rebuilt
        stamp = _lib_stamp(objs)
        fresh = os.path.exists(lib) and os.path.exists(lib + ".stamp") and open(lib + ".stamp").read() == stamp
        if fresh and not rebuilt:
            return lib, rebuilt_any
        _link(objs, tmp)
        try:
            if touch_on:
                want, have = ideal_touch_table(tmp), read_touch_table()
                if want != have:
                    if it == TOUCH_MAX_PASSES - 1:
                        raise RuntimeError("own-code touch table did not reach a fixed point in %d passes" % TOUCH_MAX_PASSES)
                    _write_touch_table(want)
                    continue                                  # recompile what uses a changed macro, link and measure again
                check_code_touch(tmp, have)
            os.replace(tmp, lib)
            with open(lib + ".stamp", "w") as fh:
                fh.write(stamp)
            return lib, rebuilt_any
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    raise RuntimeError("unreachable")


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    _, rebuilt = _build_checked(srcs, force, None, (), LIB)
    if verbose:
        print("libraindrop_hip.so: %s (%d sources, %s)" % (
            LIB, len(srcs), "rebuilt" if rebuilt else "up to date"))
    return LIB


def build_variant(name, extra):
    """A/B build of the WHOLE library with extra hipcc flags -> raindrop_amd/_ab/lib_<name>.so (git-ignored, travels with gpurun:
    delete raindrop_amd/_ab/ when the A/B is done); use as RD_LIB_PATH=raindrop_amd/_ab/lib_<name>.so.
    The variant is compiled against the main library's touch table and CHECKED against it (flags that shrink a kernel below its
    touch length fail here); -DRD_NO_CODE_TOUCH variants skip the check."""
    objdir = os.path.join(PKG, "_ab", name)
    os.makedirs(objdir, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = [o for o, _ in ex.map(lambda s: _compile(s, False, objdir, tuple(extra)), srcs)]
    lib = os.path.join(PKG, "_ab", "lib_%s.so" % name)
    tmp = lib + ".tmp"
    _link(objs, tmp)
    try:
        if "-DRD_NO_CODE_TOUCH" not in extra:
            check_code_touch(tmp)
        os.replace(tmp, lib)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    print(lib)
    return lib


def kernel_code_sizes(lib=None):
    """{mangled kernel name: code bytes} of every gfx950 kernel in the built library: the clang offload bundles inside the shared
    object ("__CLANG_OFFLOAD_BUNDLE__", one per translation unit) each hold an AMDGPU ELF; its FUNC symbols' sizes are the kernels'
    code lengths.  tests/test_kernel_resources.py keeps the own-code touch lengths (rd_common.h touch_own_code) below them."""
    import struct
    out = {}
    for e in _code_objects(lib):
        shoff, = struct.unpack_from("<Q", e, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", e, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", e, shoff + k * shentsize) for k in range(shnum)]
        for s in secs:
            if s[1] != 2:                                        # SHT_SYMTAB
                continue
            stroff = secs[s[6]][4]
            for k in range(s[5] // 24):
                name, info, _, _, _, sz = struct.unpack_from("<IBBHQQ", e, s[4] + 24 * k)
                if (info & 15) == 2 and sz:                      # STT_FUNC
                    end = e.index(b"\0", stroff + name)
                    out[e[stroff + name:end].decode()] = sz
    return out


def _code_objects(lib=None):
    """The gfx950 ELF images inside the built library (one per translation unit)."""
    import struct
    d = open(lib or LIB, "rb").read()
    pos = 0
    while True:
        i = d.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
        if i < 0:
            return
        pos = i + 24
        nb = struct.unpack_from("<Q", d, i + 24)[0]
        o = i + 32
        for _ in range(nb):
            off, size, tl = struct.unpack_from("<QQQ", d, o); o += 24
            triple = d[o:o + tl].decode(errors="replace"); o += tl
            if "gfx950" in triple and size and d[i + off:i + off + 4] == b"\x7fELF":
                yield d[i + off:i + off + size]


def getpc_offsets(lib=None):
    """{mangled kernel name: byte offset of its (first) s_getpc_b64 from the kernel's entry} by disassembly (llvm-objdump of the
    ROCm LLVM): where touch_own_code's range starts.  Empty when the disassembler is not installed."""
    import tempfile
    objdump = _objdump()
    out = {}
    if not objdump:
        return out
    with tempfile.TemporaryDirectory() as tmp:
        for n, e in enumerate(_code_objects(lib)):
            f = os.path.join(tmp, "co%d.elf" % n)
            with open(f, "wb") as fh:
                fh.write(e)
            txt = subprocess.run([objdump, "-d", "--no-show-raw-insn", f], capture_output=True, text=True).stdout
            cur, start = None, 0
            for line in txt.splitlines():
                m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
                if m:
                    cur, start = m.group(2), int(m.group(1), 16)
                elif cur and "s_getpc_b64" in line and cur not in out:
                    a = re.search(r"//\s*([0-9A-Fa-f]+):", line)
                    if a:
                        out[cur] = int(a.group(1), 16) - start
    return out


def touch_table(lib=None):
    """[(site, kernel, code bytes, s_getpc offset, touched bytes, bytes of the kernel's tail NOT touched)] for every kernel with an
    own-code touch -- `python -m raindrop_amd.build --touch-table`.  The uncovered tail is what a box without instruction look-ahead
    fetches cold, one 64-byte line per trip to memory."""
    sizes, offs, table = kernel_code_sizes(lib), getpc_offsets(lib), read_touch_table()
    rows = []
    for site, pat, _ in TOUCH_SITES:
        rx = re.compile(pat)
        for k, v in sorted(sizes.items()):
            if rx.search(k):
                rows.append((site, k, v, offs.get(k), table.get(site, 0), v - table.get(site, 0) - (offs.get(k) or 0)))
    return rows


if __name__ == "__main__":
    if "--touch-table" in sys.argv:
        build()
        for site, k, v, o, t, u in touch_table():
            print("%-14s %-80s size %6d  s_getpc @%4s  touch %6d  uncovered %6d" % (site, k[:80], v, o, t, u))
    elif "--variant" in sys.argv:                     # python -m raindrop_amd.build --variant philox -DRD_RNG_PHILOX
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    else:
        build(force="--force" in sys.argv)
