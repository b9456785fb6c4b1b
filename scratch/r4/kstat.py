"""per-kernel table from a rocprofv3 --kernel-trace database: python kstat.py <dir> [min_share]"""
import os, re, sqlite3, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "profiles"))
d = sys.argv[1]
dbf = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")][0]
con = sqlite3.connect(dbf)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = con.execute("select s.kernel_name, d.end - d.start from %s d join %s s on d.kernel_id = s.id" % (kd, ks)).fetchall()
def short(n):
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if m:
        k = int(m.group(1)); base = n[m.end():m.end() + k]; rest = n[m.end() + k:]
        args = []
        for mm in re.finditer(r"DF16b|DF16_|Li\d+E|Lb[01]E|f(?=[A-Z]|$)", rest.split("EEv")[0]):
            t = mm.group(0)
            args.append("bf16" if t == "DF16b" else "f16" if t == "DF16_" else "f32" if t == "f" else t[2:-1] if t[1] == "i" else ("T" if t[2] == "1" else "F"))
        return base + "<" + ",".join(args) + ">"
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:100]
agg = collections.defaultdict(lambda: [0, 0])
for n, dt in rows:
    a = agg[short(n)]; a[0] += 1; a[1] += dt
tot = sum(a[1] for a in agg.values())
print("total GPU time %.3f ms over %d launches" % (tot / 1e6, len(rows)))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-78s calls %6d avg %9.1f us  %5.2f%%" % (n, c, t / c / 1e3, 100.0 * t / tot))
