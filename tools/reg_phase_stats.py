"""Per-level summary of the [reg] lines the tracker prints with PCS_TICK_PROFILE=1 (resident ICP schedules: phase
microseconds per unit-iteration):  python tools/reg_phase_stats.py <stderr file>"""
import re, sys, numpy as np
rows=[]
for l in open(sys.argv[1]):
    m=re.match(r"\[reg\] slot (\d+) lv (\d+) C (\d+) Mn (\d+) Ra (\d+) Rn (\d+) iters (\d+) loop (\d+) us (\d+) (\d+) (\d+) (\d+)",l)
    if m: rows.append(list(map(int,m.groups())))
a=np.array(rows)
print(len(a), "regs; total loop iters", a[:,7].sum(), " total CU-ms", a[:,8:12].sum()/1e3)
for lv in range(3):
    for big in (False,True):
        s=a[(a[:,1]==lv)&((a[:,2]>5000)==big)]
        it=s[:,7].sum()
        print("lv",lv,"C>5000" if big else "C<5000","regs",len(s),"loop iters",it,"us/iter: budget %.0f search %.0f solve %.0f apply %.0f"%tuple(s[:,8+q].sum()/it for q in range(4)), "Mn+Rn %.0f"%(s[:,3]+s[:,5]).mean())
