"""Static checks on the gfx950 ISA of the kernels (no GPU needed): what the compiler really emitted.

    python tools/isa_scan.py [source.hip ...] [--filter REGEX]

For every kernel of every given source (default: all of raindrop_amd/csrc/*.hip) prints
  * registers / scratch / spills / static LDS (a non-zero scratch size is a bug here: DESIGN.md rule 11),
  * the widths of its global loads and stores (rule 16: a runtime-selected pair of load forms had been merged into 4-byte loads
    in every attention kernel; `dwordx4` must show up where the source says float4),
  * `load -> full wait` pairs: global loads followed within a dozen instructions by `s_waitcnt vmcnt(0)` (rules 2, 17, 19: masked
    loads, look-ahead behind a branch, per-phase parameter loads -- each pair is a dependent round trip of the workgroup),
  * resident workgroups per CU from registers and static LDS (rule 21; dynamic LDS is not visible here).
Compiles with `hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps` into a temporary directory."""
import argparse, glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_to_asm(src, tmp):
    base = os.path.splitext(os.path.basename(src))[0]
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-save-temps", "-c", src, "-o", base + ".o"], cwd=tmp, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(tmp, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def demangle_short(name):
    m = re.match(r"_ZN2rd12_GLOBAL__N_1\d+(k_\w+?)(I.*)?E?v", name)
    return (m.group(1) + (" <" + m.group(2)[1:40] + ">" if m.group(2) else "")) if m else name[:70]


def scan(asm_path, flt):
    s = open(asm_path).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", s, re.S):
        pass
    # metadata block: one YAML-ish record per kernel
    for rec in re.split(r"\n\s+- \.agpr_count", s)[1:]:
        nm = re.search(r"\.name:\s+(\S+)", rec)
        if not nm:
            continue
        g = lambda k: int((re.search(r"\.%s:\s+(\d+)" % k, rec) or [0, 0])[1])
        meta[nm.group(1)] = dict(vgpr=g("vgpr_count"), scratch=g("private_segment_fixed_size"), spills=g("vgpr_spill_count"),
                                 lds=g("group_segment_fixed_size"), wg=g("max_flat_workgroup_size"))
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if flt and not re.search(flt, name):
            continue
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        widths = {}
        for l in lines:
            op = l.split()[0]
            if op.startswith("global_load") or op.startswith("global_store"):
                widths[op.replace("global_", "")] = widths.get(op.replace("global_", ""), 0) + 1
        pairs = 0
        for i, l in enumerate(lines):
            if l.startswith("global_load"):
                for j in range(i + 1, min(i + 14, len(lines))):
                    if lines[j].startswith("global_load"):
                        break
                    if lines[j].startswith("s_waitcnt") and "vmcnt(0)" in lines[j]:
                        pairs += 1
                        break
        md = meta.get(name, {})
        occ = ""
        if md.get("vgpr") and md.get("wg"):
            waves = max(1, md["wg"] // 64)
            per_simd = 512 // max(md["vgpr"], 1)                     # waves per SIMD by registers (512 VGPRs per lane and SIMD)
            by_regs = (per_simd * 4) // waves
            by_lds = (160 * 1024) // md["lds"] if md.get("lds") else 99
            occ = "  wg/CU <= %d (regs) / %s (static LDS)" % (by_regs, by_lds if by_lds < 99 else "-")
        flag = "  !! SCRATCH" if md.get("scratch") else ""
        print("%-58s vgpr %3d scratch %3d lds %6d%s%s" % (demangle_short(name), md.get("vgpr", -1), md.get("scratch", -1), md.get("lds", -1), occ, flag))
        print("    %s | load->full-wait pairs: %d" % (" ".join("%s:%d" % kv for kv in sorted(widths.items())), pairs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sources", nargs="*")
    ap.add_argument("--filter", default=None, help="regex on the mangled kernel name")
    a = ap.parse_args()
    srcs = a.sources or sorted(glob.glob(os.path.join(ROOT, "raindrop_amd", "csrc", "*.hip")))
    with tempfile.TemporaryDirectory() as tmp:
        for src in srcs:
            print("== " + os.path.relpath(os.path.abspath(src), ROOT))
            scan(compile_to_asm(os.path.abspath(src), tmp), a.filter)


if __name__ == "__main__":
    main()
