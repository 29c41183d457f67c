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


# Own-code touch lengths (rd_common.h touch_own_code): (kernel name fragment, bytes the kernel requests behind its s_getpc_b64).  The
# loads must stay inside the kernel's code in EVERY instantiation -- a different compiler version or flag set can shrink a kernel, and
# a load past the end of the last kernel of a code object is a memory fault -- so the build itself checks them against the linked
# library (check_code_touch, called by build() and build_variant()) and FAILS on violation; tests/test_kernel_resources.py re-checks.
CODE_TOUCH = [("k_msg_fwd_fusedILi3ELi34ELi60E", 13056), ("k_msg_bwd_fusedILi3ELi34ELi60E", 10368), ("k_msg_fwd_fusedILi1ELi0", 12544), ("k_msg_fwd_fusedILi2ELi0", 12544),
              ("k_msg_fwd_fusedILi3ELi0", 12544), ("k_msg_bwd_fusedILi1ELi0", 9728), ("k_msg_bwd_fusedILi2ELi0", 9728), ("k_msg_bwd_fusedILi3ELi0", 9728), ("k_attn_fwd_fused", 13952), ("k_attn_bwd_fused", 31872),
              ("k_enc_post_fwdILi152ELi272ELb1", 44160), ("k_enc_post_fwdILi152ELi272ELb0", 43776), ("k_enc_post_fwdILi160ELi288ELb1", 43648),
              ("k_enc_post_fwdILi160ELi288ELb0", 43136), ("k_enc_post_fwdILi0ELi0ELb1", 55680), ("k_enc_post_fwdILi0ELi0ELb0", 58752),
              ("k_enc_pre_bwdILi152ELi272ELb1", 46592), ("k_enc_pre_bwdILi152ELi272ELb0", 46208), ("k_enc_pre_bwdILi160ELi288ELb1", 46080),
              ("k_enc_pre_bwdILi160ELi288ELb0", 45696), ("k_enc_pre_bwdILi0ELi0ELb1", 63104), ("k_enc_pre_bwdILi0ELi0ELb0", 62976),
              ("4k_dwE", 7424), ("k_dw_reduce", 9984), ("6k_adam", 2816), ("10k_adam_dev", 3200),
              ("k_wsplit", 4992), ("5k_twgILb0E", 10496), ("5k_twgILb1E", 8704), ("k_head_rowsILi1ELi12ELi3E", 15616), ("k_head_rowsILi1ELi16ELi4E", 17024)]
# the kernels outside the P19 step (RD_TOUCH_CODE_X)
CODE_TOUCH_X = [("6k_gemmI", 25088), ("13k_gemm_bf16x3I", 28160), ("12k_gemm_panelI", 28160), ("9k_rowgemmI", 4096),
                ("19k_attn_fwd_one_b16wI", 5632), ("19k_attn_bwd_one_b16wI", 6144), ("14k_attn_fwd_b16I", 6912),
                ("17k_attn_bwd_dq_b16I", 7040), ("18k_attn_bwd_dkv_b16I", 6016), ("14k_add_ln_fwd_vE", 5120), ("10k_ln_bwd_rI", 5120),
                ("10k_ln_bwd_vE", 6144)]
CODE_TOUCH_SLACK = 384          # bytes allowed for the prologue in front of the s_getpc_b64 when the disassembler is not there to say
CODE_TOUCH_MARGIN = 8           # bytes kept free behind the last touched line (the s_getpc's own length + the last dword)


def check_code_touch(lib=None):
    """Raise if any touched range could leave its kernel (see CODE_TOUCH): [s_getpc offset, + touch) must lie inside the kernel's code.
    The s_getpc offsets come from the disassembly (getpc_offsets); without the disassembler CODE_TOUCH_SLACK bytes are assumed."""
    sizes = kernel_code_sizes(lib)
    offs = getpc_offsets(lib)
    bad = []
    for frag, touch in CODE_TOUCH + CODE_TOUCH_X:
        ks = {k: v for k, v in sizes.items() if frag in k}
        if not ks:
            bad.append("%s: no such kernel in the library" % frag)
        for k, v in ks.items():
            slack = offs[k] + CODE_TOUCH_MARGIN if k in offs else CODE_TOUCH_SLACK
            if v < touch + slack:
                bad.append("%s: %d bytes of code, touches %d (+%d slack)" % (k, v, touch, slack))
    if bad:
        raise RuntimeError("own-code touch lengths exceed the built kernels (rd_common.h touch_own_code; raindrop_amd/build.py "
                           "CODE_TOUCH):\n  " + "\n  ".join(bad))


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(src):
    h = hashlib.sha1()
    for path in [src] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] \
            + [os.path.join(INCLUDE, "raindrop_hip.h")]:
        with open(path, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(f for f in FLAGS if not f.startswith("/")).encode())   # path-independent
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


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (res.stdout, res.stderr))
    if rebuilt:
        check_code_touch(LIB)
    if verbose:
        print("libraindrop_hip.so: %s (%d sources, %s)" % (
            LIB, len(srcs), "rebuilt" if rebuilt else "up to date"))
    return LIB


def build_variant(name, extra):
    """A/B build of the WHOLE library with extra hipcc flags -> raindrop_amd/_ab/lib_<name>.so (git-ignored, travels with gpurun);
    use as RD_LIB_PATH=raindrop_amd/_ab/lib_<name>.so.  `tools/ab_build.sh` is the one-file form."""
    objdir = os.path.join(PKG, "_ab", name)
    os.makedirs(objdir, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = [o for o, _ in ex.map(lambda s: _compile(s, False, objdir, tuple(extra)), srcs)]
    lib = os.path.join(PKG, "_ab", "lib_%s.so" % name)
    res = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (res.stdout, res.stderr))
    if "-DRD_NO_CODE_TOUCH" not in extra:
        check_code_touch(lib)
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
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    out = {}
    if not os.path.exists(objdump):
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
    """[(kernel, code bytes, s_getpc offset, touched bytes, bytes of the kernel's tail NOT touched)] for every kernel with an own-code
    touch -- `python -m raindrop_amd.build --touch-table`.  The uncovered tail is what a box without instruction look-ahead fetches
    cold, one 64-byte line per trip to memory (round 5: 1-3 KB per kernel of the step had been left uncovered by lengths kept below
    the smallest instantiation of each template; DESIGN.md "Round 5 in ten lines", item 10)."""
    sizes, offs = kernel_code_sizes(lib), getpc_offsets(lib)
    rows = []
    for frag, touch in CODE_TOUCH + CODE_TOUCH_X:
        for k, v in sorted(sizes.items()):
            if frag in k:
                rows.append((k, v, offs.get(k), touch, v - touch - (offs.get(k) or 0)))
    return rows


if __name__ == "__main__":
    if "--touch-table" in sys.argv:
        build()
        for k, v, o, t, u in touch_table():
            print("%-90s size %6d  s_getpc @%4s  touch %6d  uncovered %6d" % (k[:90], v, o, t, u))
    elif "--variant" in sys.argv:                     # python -m raindrop_amd.build --variant philox -DRD_RNG_PHILOX
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    else:
        build(force="--force" in sys.argv)
