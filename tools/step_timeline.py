"""Print the per-kernel timeline of one training step from a rocprofv3 --kernel-trace CSV directory (measurement helper).
usage: python tools/step_timeline.py gpurun_out/prof_dir"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'head_fused_kernel' in r['Kernel_Name'] or 'head_x3_kernel' in r['Kernel_Name']]
j = idx[-2]
while 'adam' not in rows[j]['Kernel_Name']:
    j -= 1
t0 = int(rows[j]['End_Timestamp']); k = j + 1
while True:
    r = rows[k]; s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print("%8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:70]))
    if 'adam' in r['Kernel_Name']:
        break
    k += 1
