#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/names
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/db -o t -- python3 $R/scratch/r4/c16bench.py child > $OUT/run.log 2>&1
python3 - <<PY
import sqlite3, os
d="$OUT/db"
f=[os.path.join(r,x) for r,_,fs in os.walk(d) for x in fs if x.endswith(".db")][0]
con=sqlite3.connect(f)
tabs=[r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'kernel' in t.lower()][:10])
for row in con.execute("select distinct name from kernels limit 40"): print("KERNELS.name:", row[0][:200])
ks=[t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols=[r[1] for r in con.execute("pragma table_info(%s)"%ks)]
print(cols)
for row in con.execute("select kernel_name, display_name from %s limit 12"%ks) if 'display_name' in cols else []: print("SYM:", row[0][:150], "|", row[1][:150])
PY
rm -rf $OUT/db
