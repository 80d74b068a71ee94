# Prints the kernels of consecutive dispatches of a rocprofv3 --kernel-trace database: start, duration, gap to the one before.
# usage: python3 tools/timeline.py <results.db> [first dispatch] [how many]
import sqlite3,sys
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=list(db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%drb%' order by d.start"))
print(len(rows))
at=int(sys.argv[2]) if len(sys.argv)>2 else 600
base=rows[at][1]
prev=None
for name,st,en in rows[at:at+int(sys.argv[3]) if len(sys.argv)>3 else at+16]:
    short=name.split('drb_')[1][:18]
    print(f"{short:20s} start {(st-base)/1e3:8.1f} dur {(en-st)/1e3:6.1f} gap {((st-prev)/1e3 if prev else 0):6.1f}")
    prev=en
