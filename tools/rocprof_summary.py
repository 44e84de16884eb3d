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
