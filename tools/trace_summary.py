"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid size) count / avg / min duration."""
import csv, sys, collections
agg = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    name = row['Kernel_Name'].split('(')[0].replace('void ', '')
    agg[(name, int(row['Grid_Size_X']) if 'Grid_Size_X' in row else int(row.get('Grid_Size', 0)))].append(
        (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for (name, grid), v in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    if pat in name:
        print("%-34s grid %9d  n %4d  avg %9.1f us  min %9.1f us" % (name[:34], grid, len(v), sum(v) / len(v), min(v)))
