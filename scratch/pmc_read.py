import sqlite3, sys, collections
for path in sys.argv[1:]:
    db = sqlite3.connect(path); c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if not view:
        print([t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()]); continue
    cols = [d[1] for d in c.execute("pragma table_info(%s)" % view)]
    rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from %s group by kernel_name, counter_name" % view))
    by = collections.defaultdict(dict)
    for k, cn, v, n in rows: by[k][cn] = (v, n)
    for k, d in by.items():
        if "conv" not in k: continue
        print(k[:90])
        for cn, (v, n) in sorted(d.items()): print("   %-28s %16.0f  (n=%d)" % (cn, v, n))
