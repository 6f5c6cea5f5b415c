import csv, sys, statistics as st, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
solves=[]; cur=None
for r in rows:
    n=r['Kernel_Name']
    if 'k_solve_init' in n: cur=[r]
    elif cur is not None:
        cur.append(r)
        if 'k_finish' in n: solves.append(cur); cur=None
print('solves', len(solves))
for s in solves[3:6]:
    t0=int(s[0]['Start_Timestamp'])
    print(' | '.join('%s %.1f-%.1f'%(r['Kernel_Name'].replace('void ','')[:10], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3) for r in s))
