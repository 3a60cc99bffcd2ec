import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r['Kernel_Name'].startswith(('void nq::', 'k_budget'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows)//2:]          # steady state
dur = collections.defaultdict(list); gaps = []
for a, b in zip(rows, rows[1:]):
    n = re.sub(r'\(.*', '', a['Kernel_Name']).replace('void nq::', '')
    dur[n].append(int(a['End_Timestamp']) - int(a['Start_Timestamp']))
    gaps.append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
for n, v in dur.items():
    v.sort(); print('%-40s n=%5d median %6.2f us' % (n[:40], len(v), v[len(v)//2] / 1e3))
gaps.sort(); print('gap between consecutive kernels: median %.2f us, mean %.2f us' % (gaps[len(gaps)//2] / 1e3, sum(gaps) / len(gaps) / 1e3))
tot = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('kernels per us-span: %d kernels in %.1f us -> %.2f us per kernel slot' % (len(rows), tot / 1e3, tot / 1e3 / len(rows)))
