import numpy as np, sys
def mask_for(i):
    m=i
    for s in (1,2,4,8,16): m|=m>>s
    return m
def coalesce_time(k, rng, trials=300, limit=None):
    masks=np.array([0]+[mask_for(i) for i in range(1,k)],dtype=np.int64)
    out=[]
    for _ in range(trials):
        ia=int(rng.integers(1,k)); ib=int(rng.integers(1,k))
        if ia==ib: ib = ib%(k-1)+1
        t=0
        lim = limit or 200*k
        raws=rng.integers(0,2**32,size=lim,dtype=np.uint64)
        while ia!=ib and t<lim:
            r=int(raws[t]); t+=1
            if (r & masks[ia]) <= ia:
                ia-=1
                if ia==0: ia=k-1
            if (r & masks[ib]) <= ib:
                ib-=1
                if ib==0: ib=k-1
        out.append(t if ia==ib else -1)
    return np.array(out)
rng=np.random.default_rng(0)
for k in (64,83,100,128,129,200,257,500,1000,2049,3789):
    tr=200 if k<=1000 else 60
    t=coalesce_time(k,rng,trials=tr)
    E=sum((mask_for(i)+1)/(i+1) for i in range(1,k))
    ok=t[t>=0]
    print(k,'E=%.0f'%E,'fail',int((t<0).sum()),'median %.1f perms'%(np.median(ok)/E),'p90 %.1f'%(np.quantile(ok,0.9)/E),'max %.1f perms'%(ok.max()/E), 'max draws',ok.max())
