"""Per-kernel timeline of the last training step in a rocprofv3 kernel trace (rocpd database):
usage: python scratch/timeline.py <dir with the .db> [kernels per step, default: detected from the Adam kernels]"""
import os, sys, sqlite3, re, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles"))
d = sys.argv[1]
f = [os.path.join(r, x) for r, _, fs in os.walk(d) for x in fs if x.endswith(".db")][0]
con = sqlite3.connect(f)
cur = con.execute("select * from kernels limit 1")
cols = [c[0] for c in cur.description]
print("# columns:", cols)
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
q = "select name, start, end%s%s from kernels order by start" % ((", " + gx) if gx else "", (", " + wx) if wx else "")
rows = list(con.execute(q))
def short(n):
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if m:
        k = int(m.group(1)); rest = n[m.end() + k:]
        args = [("bf16" if t == "DF16b" else "f16" if t == "DF16_" else t[2:-1]) for t in re.findall(r"DF16b|DF16_|Li\d+E|Lb[01]E", rest.split("EEv")[0])]
        return n[m.end():m.end() + k] + "<" + ",".join(args) + ">"
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:int(os.environ.get("TL_NAME", "70"))]
# step boundaries: the fused Adam kernels end a step
adam = [i for i, r in enumerate(rows) if "FusedAdam" in r[0] or "adam_multi" in r[0]]
ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] - i > 20]
segs_min = 150
if len(ends) < 3:
    print("cannot find steps"); sys.exit(1)
segs = [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1)]
segs = [sg for sg in segs if sg[1] - sg[0] > segs_min]
lo, hi = min(segs, key=lambda sg: rows[sg[1] - 1][2] - rows[sg[0]][1])      # the fastest step = a graph replay
print("# steps found: %d, walls ms: %s" % (len(segs), " ".join("%.2f" % ((rows[b - 1][2] - rows[a][1]) / 1e6) for a, b in segs)))
step = rows[lo:hi]
print("# kernels in step: %d  wall %.3f ms" % (len(step), (step[-1][2] - step[0][1]) / 1e6))
busy = sum(r[2] - r[1] for r in step)
gaps = sum(max(0, step[i][1] - step[i - 1][2]) for i in range(1, len(step)))
print("# sum of kernel durations %.3f ms, sum of gaps %.3f ms" % (busy / 1e6, gaps / 1e6))
agg = collections.OrderedDict()
for i, r in enumerate(step):
    gap = (r[1] - step[i - 1][2]) if i else 0
    extra = ("grid %6d x %4d" % (r[3] // max(1, r[4]) if wx else r[3], r[4] if wx else 0)) if gx else ""
    print("%4d %-64s %s  %8.1f us  gap %6.1f us" % (i, short(r[0]), extra, (r[2] - r[1]) / 1e3, gap / 1e3))
    a = agg.setdefault(short(r[0]), [0, 0, 0]); a[0] += 1; a[1] += r[2] - r[1]; a[2] += max(0, gap)
print("# by kernel: calls, total us, gap before us")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-64s %4d %9.1f %8.1f" % (k, a[0], a[1] / 1e3, a[2] / 1e3))
