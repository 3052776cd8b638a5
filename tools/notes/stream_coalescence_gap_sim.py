import numpy as np
def mask_for(i):
    m=i
    for s in (1,2,4,8,16): m|=m>>s
    return m
def run(k, gap, rng, trials=100):
    masks=np.array([0]+[mask_for(i) for i in range(1,k)],dtype=np.int64)
    E=sum((mask_for(i)+1)/(i+1) for i in range(1,k))
    res=[]
    for _ in range(trials):
        ia=int(rng.integers(1,k)); ib=((ia-1-gap)%(k-1))+1
        t=0; lim=int(400*E)
        raws=rng.integers(0,2**32,size=lim,dtype=np.uint64)
        while ia!=ib and t<lim:
            r=int(raws[t]); t+=1
            if (r & masks[ia]) <= ia:
                ia-=1
                if ia==0: ia=k-1
            if (r & masks[ib]) <= ib:
                ib-=1
                if ib==0: ib=k-1
        res.append(t/E if ia==ib else np.inf)
    res=np.array(res)
    return np.median(res), np.quantile(res,0.9), res.max()
rng=np.random.default_rng(1)
for k in (500,3789):
    for gap in (1,3,10,30,100,300):
        print(k,gap,'median/p90/max perms: %.1f %.1f %.1f'%run(k,gap,rng,trials=60 if k>1000 else 150))
