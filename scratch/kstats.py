import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
rows = list(c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("GPU busy per step %.2f ms" % (tot / steps / 1e3))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{r[2]/steps:9.1f} us/step {100*r[2]/tot:5.1f}%  n/step={r[1]/steps:6.1f} avg={r[3]:8.1f}  {r[0][:110]}")
