import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
pat=sys.argv[2:] 
for r in c.execute("select name,total_calls,average from top_kernels"):
    if not pat or any(p in r[0] for p in pat): print("%10.1f us x%-4d %s"%(r[2],r[1],r[0][:110]))
