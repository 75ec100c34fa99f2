import sqlite3, sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
q="""select name, grid_x, grid_y, grid_z, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, lds_size, vgpr_count from kernels where name like '%chatts%' group by name, grid_x, grid_y, grid_z order by name, grid_x"""
for r in cur.execute(q): print([ (x[:70] if isinstance(x,str) else (round(x,2) if isinstance(x,float) else x)) for x in r])
