"""Static instruction counts of a kernel BETWEEN its barriers, from the compiler's assembly (`hipcc --save-temps` -> *.s).

The fused kernels of the step are bound by instruction issue per wave (DESIGN "Round 5: K1", item 3): a phase's duration follows
its static instruction count times the waves per SIMD that run it.  This prints, per barrier-delimited segment in program order:
instructions by class (valu / salu / lds / vmem / mfma / other), the branches inside, and the waitcnt's.  Loops are counted once.

usage: isa_phases.py file.s <kernel name fragment> [--labels]
"""
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, frag = sys.argv[1], sys.argv[2]
    labels = "--labels" in sys.argv
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and frag in l.split(":")[0]:
            start = i
            break
    if start is None:
        sys.exit("no kernel with %r" % frag)
    print(lines[start])
    seg, segs = {"n": 0}, []
    opre = re.compile(r"^\s+([a-z_0-9]+)")
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        s = l.strip()
        if s.endswith(":") and s.startswith(".LBB"):
            seg.setdefault("labels", []).append(s[:-1])
            continue
        m = opre.match(l)
        if not m or s.startswith((";", ".")):
            continue
        op = m.group(1)
        seg["n"] += 1
        c = classify(op)
        seg[c] = seg.get(c, 0) + 1
        if op.startswith("s_cbranch") or op == "s_branch":
            seg["br"] = seg.get("br", 0) + 1
        if op == "s_waitcnt" and "vmcnt" in s:
            seg.setdefault("vmcnt", []).append(re.search(r"vmcnt\((\d+)\)", s).group(1))
        if op == "s_barrier":
            segs.append(seg)
            seg = {"n": 0}
        if op == "s_endpgm":
            seg["end"] = 1
            segs.append(seg)
            seg = {"n": 0}
    if seg["n"]:
        segs.append(seg)
    print("%4s %6s %6s %6s %5s %5s %5s %4s  vmcnt waits" % ("seg", "total", "valu", "salu", "lds", "vmem", "mfma", "br"))
    for i, g in enumerate(segs):
        print("%4d %6d %6d %6d %5d %5d %5d %4d  %s%s" % (i, g["n"], g.get("valu", 0), g.get("salu", 0), g.get("lds", 0), g.get("vmem", 0),
                                                        g.get("mfma", 0), g.get("br", 0), ",".join(g.get("vmcnt", [])),
                                                        "  <end>" if g.get("end") else ""))
        if labels and g.get("labels"):
            print("       labels:", " ".join(g["labels"]))


if __name__ == "__main__":
    main()
