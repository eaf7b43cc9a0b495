"""Hand-over analysis of a pipe-engine trace (diagnostic): joins the per-step store times of tools/pipe_trace.py (stamped kernel,
subdomain 0 of the 216^3 / 2x2x2 problem) with the per-step producer requirements of tools/pipe_schedule_probe.py 216 0 and
prints the store-to-store lag of dependent steps, step durations, and a back-trace of the critical path of each sweep."""
import numpy as np, struct
z = np.load("gpurun_out/pipe_trace216.npz")
st, meta = z["st"], z["meta"]
t0 = st[:,0].min()
g0 = np.where(meta[:,0]==0)[0]
nt = len(g0)
tstep = (st[g0,16:]-t0)/100.   # us, time of each step's store issue
first = (st[g0,1]-t0)/100.
data = open("gpurun_out/pipe_needs.bin","rb").read()
off=0; prods=[]; needs=[]
for i in range(nt):
    npd, ns = struct.unpack_from("ii", data, off); off+=8
    pr = np.frombuffer(data, dtype=np.int32, count=npd, offset=off); off+=4*npd
    nd = np.frombuffer(data, dtype=np.uint16, count=(npd+2)*ns, offset=off).reshape(ns,npd+2)[:, :npd]; off+=2*(npd+2)*ns   # (+ the step's W and active rows)
    prods.append(pr); needs.append(nd)
steps = st[g0,6].astype(int)
sweep = meta[g0,1]
# per step: own time, ready time = max over producers time of their step need-1
res=[]
for i in range(nt):
    if len(prods[i])==0: continue
    ns=steps[i]
    rdy=np.full(ns,-1e9); who=np.zeros(ns,dtype=int)
    for j,p in enumerate(prods[i]):
        nd=needs[i][:,j].astype(int)
        tp=np.where(nd>0, tstep[p][np.maximum(nd-1,0)], -1e9)
        upd=tp>rdy; who[upd]=p; rdy=np.maximum(rdy,tp)
    own=tstep[i][:ns]
    prev=np.concatenate([[first[i]],own[:-1]])
    res.append((i,own,rdy,prev,who))
for sw in (0,1):
    lat=[];bind=[];dt=[];nbind=0;ntot=0
    for (i,own,rdy,prev,who) in res:
        if sweep[i]!=sw: continue
        l=own-rdy            # store time of the step minus store time of the last needed producer step
        d=own-prev           # step duration
        lat.append(l); dt.append(d)
    lat=np.concatenate(lat); dt=np.concatenate(dt)
    print(f"sweep {sw}: steps {len(lat)}; own_store - producer_store percentiles 1/5/25/50/75: {np.percentile(lat,[1,5,25,50,75]).round(2)}")
    print(f"   step duration percentiles 5/25/50/75/95/99: {np.percentile(dt,[5,25,50,75,95,99]).round(2)}  mean {dt.mean():.3f}")
    tight = lat < 4.0
    print(f"   steps with lag<4us: {tight.mean():.2%}; their duration mean {dt[tight].mean():.3f}; others {dt[~tight].mean():.3f}")
    for lo,hi in [(0,1.5),(1.5,2),(2,2.5),(2.5,3),(3,4),(4,6),(6,1e9)]:
        m=(lat>=lo)&(lat<hi); print(f"     lag [{lo},{hi}): {m.mean():.2%} mean duration {dt[m].mean() if m.any() else 0:.3f}")
# timeline of a few consecutive dependent tasks: print for task 30..33 the step times around the start
for i in (30,31,32):
    (ii,own,rdy,prev,who)=[r for r in res if r[0]==i][0]
    print("task",i,"first",first[i].round(1),"steps",steps[i])
    for t in list(range(0,12))+list(range(60,66)):
        print(f"   t={t:3d} store {own[t]:8.2f} dur {own[t]-prev[t]:5.2f} ready {rdy[t]:8.2f} (task {who[t]}) lag {own[t]-rdy[t]:5.2f}")
print("---- critical path back-trace (L sweep of group 0)")
R={r[0]:r for r in res}
def backtrace(i,t,verbose=False):
    hops=0;intra=0;thop=0.;tintra=0.;path=[]
    while True:
        if i not in R: # no producers: head task
            intra+=t; tintra+=tstep[i][t]-first[i]; break
        (ii,own,rdy,prev,who)=R[i]
        if t==0 or (own[t]-rdy[t] < own[t]-prev[t]+2.3 and rdy[t]>prev[t]-2.3):   # producer's store later than (own previous store - 2.3us): the hop was binding
            # find needed step of producer
            p=who[t]; j=list(prods[i]).index(p); s=int(needs[i][t,j])-1
            hops+=1; thop+=own[t]-tstep[p][s]; path.append((i,t,p,s,own[t]-tstep[p][s]))
            i,t=p,s
        else:
            intra+=1; tintra+=own[t]-prev[t]; t-=1
    return hops,intra,thop,tintra,path
for sw in (0,1):
    ids=[i for i in range(nt) if sweep[i]==sw]
    last=max(ids,key=lambda i:tstep[i][steps[i]-1])
    h,n,th,tn,path=backtrace(last,steps[last]-1)
    print(f"sweep {sw}: last task {last}; path: {h} hops taking {th:.0f} us ({th/max(h,1):.2f} each), {n} intra steps taking {tn:.0f} us ({tn/max(n,1):.2f} each)")
    print("   sample hops:", [(a,b,c,d,round(e,2)) for (a,b,c,d,e) in path[:40:4]])
print("---- follow the binding producer chain back from task 31 step 6")
i,t=31,6
for k in range(40):
    if i not in R: print("head task",i,"step",t, "store",tstep[i][t].round(2), "dur", (tstep[i][t]-tstep[i][t-1]).round(2) if t>0 else 0); break
    (ii,own,rdy,prev,who)=R[i]
    p=who[t]; j=list(prods[i]).index(p); s=int(needs[i][t,j])-1
    print(f"task {i:3d} step {t:3d} store {own[t]:8.2f} dur {own[t]-prev[t]:5.2f} | latest producer task {p} step {s} stored {tstep[p][s]:8.2f} lag {own[t]-tstep[p][s]:5.2f}")
    if own[t]-tstep[p][s] < (own[t]-prev[t])+0.3: i,t=p,s
    else: t-=1
print("---- largest hand-overs on the critical path of each sweep: (consumer task, step) <- (producer task, step): us")
for sw in (0, 1):
    ids = [i for i in range(nt) if sweep[i] == sw]
    last = max(ids, key=lambda i: tstep[i][steps[i] - 1])
    h, n, th, tn, path = backtrace(last, steps[last] - 1)
    big = sorted(path, key=lambda q: -q[4])[:12]
    print(f"sweep {sw}:", [(int(a), int(b), int(c), int(d), round(float(e), 1)) for (a, b, c, d, e) in big])
    hc = np.array([q[4] for q in path])
    print(f"   hand-over cost percentiles 10/50/90: {np.percentile(hc, [10, 50, 90]).round(2)}; sum of those above 6 us: {hc[hc > 6].sum():.0f} us in {int((hc > 6).sum())} hops")
