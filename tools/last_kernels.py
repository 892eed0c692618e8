"""Print the last N kernels of a rocprofv3 kernel-trace CSV (start offset, duration, grid, name)."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1], recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sub = rows[-int(sys.argv[2]):]
t0 = int(sub[0]['Start_Timestamp'])
for r in sub:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("t=%8.1f dur=%7.1f grid=%7s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', '?')),
                                            r['Kernel_Name'].replace('ffgp_', '').replace('(GemmArgs)', '').replace('void ', '')[:60]))
