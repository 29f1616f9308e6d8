import glob, sqlite3, sys
dbs = sorted(glob.glob(sys.argv[1] + '/**/*.db', recursive=True))
for db in dbs:
    con = sqlite3.connect(db)
    rows = con.execute("select name, (end-start)/1000.0, grid_x, grid_y from kernels where name like ? order by start", ('%'+sys.argv[2]+'%',)).fetchall()
    for name, d, gx, gy in rows: print('%8.1f us  grid %dx%d  %s' % (d, gx, gy, name[:60]))
