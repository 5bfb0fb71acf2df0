import itertools, random, sys
RD_GROUPS = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
             list(range(32,36))+list(range(44,48))+list(range(52,60)), list(range(36,44))+list(range(48,52))+list(range(60,64))]
WR_GROUPS = [list(range(g*8, g*8+8)) for g in range(8)]
def plan(lg, mx=3):
    r=[]
    while lg>=mx: r.append(mx); lg-=mx
    if lg: r.append(lg)
    return r
def cost(addrs, groups, mod):
    tot=0
    for g in groups:
        slots={}
        for l in g:
            slots.setdefault(addrs[l]%mod, set()).add(addrs[l])
        tot += max(len(v) for v in slots.values())
    return tot
def make_swz(ms):   # ms[h-4] = 4-bit mask XORed into low bits when bit h of i is set
    def f(i):
        x=i; h=4; lo=0
        v=i>>4; k=0
        while v:
            if v&1: lo^=ms[k] if k < len(ms) else 0
            v>>=1; k+=1
        return i^lo
    return f
def fft_slot(pl, lg, k, swz):
    pos=0; rem=lg
    for rl in pl:
        rem-=rl; pos+=(k&((1<<rl)-1))<<rem; k>>=rl
    return swz(pos)
def evaluate(swz, lg, nt, detail=False):
    N=1<<lg; pl=plan(lg); lev=lg; tot=0.0; res=[]
    for si,rl in enumerate(pl):
        R=1<<rl; q=1<<(lev-rl); L=1<<lev
        rd=wr=0; n=0
        nbf=N//R
        per=nbf//nt if nbf>nt else 1
        for w0 in range(0, nt, 64):
            for k in range(per):
                lanes=[ (w0//64)*64*per + l + 64*k for l in range(64)] if per>1 else [w0+l for l in range(64)]
                for r in range(R):
                    addrs=[swz(((b//q)*L + b%q) + r*q) for b in lanes]
                    rd+=cost(addrs, RD_GROUPS, 16); wr+=cost(addrs, WR_GROUPS, 8); n+=1
        res.append((R,q, rd/(4*n), wr/(8*n)))
        tot += 4*rd/(4*n)*R + 8*wr/(8*n)*R      # cycles per thread-stage
        lev-=rl
    # natural-order pair reads of the merge (k and h-k), lanes = consecutive k
    h=N; rdm=0; n=0
    for w0 in range(0, min(nt, N//2), 64):
        ks=[w0+l for l in range(64)]
        a1=[fft_slot(pl,lg,k,swz) for k in ks]; a2=[fft_slot(pl,lg,(h-k)%h,swz) for k in ks]
        rdm+=cost(a1,RD_GROUPS,16)+cost(a2,RD_GROUPS,16); n+=2
    res.append(('merge',0,rdm/(4*n),0))
    tot += 4*rdm/(4*n)*2*( (N//4)//nt + 1)
    return (tot,res) if detail else tot
CFG=((10,128),(11,256),(12,512),(9,64),(8,64))
def total(ms):
    f=make_swz(ms); return sum(evaluate(f,lg,nt) for lg,nt in CFG)
cur=[1,2,4,8,0,0,0,0]
print('current', total(cur))
for lg,nt in CFG: print(lg, evaluate(make_swz(cur),lg,nt,True))
best=cur[:]; bc=total(best)
random.seed(1)
for restart in range(6):
    ms = best[:] if restart==0 else [random.randrange(16) for _ in range(8)]
    c=total(ms); improved=True
    while improved:
        improved=False
        for h in range(8):
            for v in range(16):
                if v==ms[h]: continue
                t=ms[:]; t[h]=v; ct=total(t)
                if ct<c-1e-9: ms,c=t,ct; improved=True
    print('restart',restart,ms,c, flush=True)
    if c<bc: best,bc=ms,c
print('best',best,bc)
for lg,nt in CFG: print(lg, evaluate(make_swz(best),lg,nt,True))
