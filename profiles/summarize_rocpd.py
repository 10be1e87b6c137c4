"""Turn a rocprofv3 results .db (rocpd sqlite) into the per-kernel summary committed under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    lines = []
    if note:
        lines.append("# " + note)
    lines.append("# source: rocprofv3 --kernel-trace --stats  (rocpd view top_kernels; durations in microseconds)")
    lines.append("%-72s %6s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append("%-72s %6d %14.1f %12.1f %8.2f" % (name[:72], calls, total, avg, pct))
    lines.append("")
    lines.append("# per-dispatch resources (first dispatch of each of our kernels)")
    lines.append("%-72s %10s %10s %8s %8s %8s %10s" % ("kernel", "grid", "wg", "vgpr", "sgpr", "lds", "scratch"))
    seen = set()
    for r in cur.execute("select name,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size from kernels order by id"):
        if r[0] in seen or "ctc::" not in r[0]:
            continue
        seen.add(r[0])
        lines.append("%-72s %10d %10d %8d %8d %8d %10d" % ((r[0][:72],) + tuple(r[1:])))
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
