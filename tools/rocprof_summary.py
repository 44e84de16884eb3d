"""Turn a rocprofv3 rocpd results.db (what `rocprofv3 --kernel-trace --stats` writes on this
image) into the plain-text per-kernel summary committed under profiles/."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    lines = ["rocprofv3 --kernel-trace --stats summary (from %s)" % db_path.split("/")[-1],
             "%-72s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        short = name.replace("mpcx::(anonymous namespace)::", "").replace("mpcx::", "")
        lines.append("%-72s %8d %14.3f %12.3f %7.2f%%" % (short[:72], calls, total, avg, pct))
    # launches of one kernel that do different work (nlmpc_sqp_wg: the solve, then the second pass over the instances whose working set outgrew
    # a cut capacity -- normally none, a few microseconds): the longest launches on their own
    try:
        rows = list(cur.execute("select name, count(*), min(end - start) / 1e3, max(end - start) / 1e3 from kernels group by name"))
        for name, n, mn, mx in rows:
            if n > 1 and mn < 0.05 * mx:
                big = [d for (d,) in cur.execute("select (end - start) / 1e3 from kernels where name = ? and (end - start) / 1e3 > ?", (name, 0.25 * mx))]
                short = name.replace("mpcx::(anonymous namespace)::", "").replace("mpcx::", "")
                lines.append("  %s: %d of the %d launches are longer than a quarter of the longest: their average %.3f us (the others: the second pass)"
                             % (short[:72], len(big), n, sum(big) / len(big)))
    except sqlite3.Error:
        pass
    lines.append("")
    lines.append("per-kernel launch geometry / resources (first dispatch of each):")
    seen = set()
    q = "select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels order by start"
    for name, gx, wx, lds, scr, vg, ag, sg in cur.execute(q):
        if name in seen:
            continue
        seen.add(name)
        short = name.replace("mpcx::(anonymous namespace)::", "").replace("mpcx::", "")
        lines.append("  %-64s grid %-8d wg %-5d lds %-7d scratch %-5d vgpr %-4d agpr %-4d sgpr %d" % (short[:64], gx, wx, lds, scr, vg, ag, sg))
    txt = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
