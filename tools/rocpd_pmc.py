#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd sqlite database.
    python tools/rocpd_pmc.py gpurun_out/pmc/p_results.db [kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, filt=""):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    def tab(prefix):
        return [t for t in tabs if t.startswith(prefix)][0]
    disp, sym, pmc, info = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    symcols = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in symcols else "kernel_name"
    q = ("select s.%s, i.name, e.value, d.id, d.end - d.start from %s e join %s i on e.pmc_id = i.id "
         "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id" % (name_col, pmc, info, disp, sym))
    acc = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    dur = defaultdict(dict)
    for name, cname, val, did, dt in c.execute(q):
        if filt and filt not in name:
            continue
        acc[name][cname] += val
        ndisp[name].add(did)
        dur[name][did] = dt
    for name in sorted(acc, key=lambda n: -sum(dur[n].values())):
        n = len(ndisp[name])
        print("%s  (dispatches %d, avg %.2f us)" % (name[:100], n, sum(dur[name].values()) / n / 1e3))
        for cname, v in sorted(acc[name].items()):
            print("    %-28s %16.1f per dispatch" % (cname, v / n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
