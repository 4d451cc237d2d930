import itertools
G = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
     list(range(4,12))+list(range(16,20))+list(range(28,32)),
     list(range(32,36))+list(range(44,48))+list(range(52,60)),
     list(range(36,44))+list(range(48,52))+list(range(60,64))]
def conflicts(rowf, f, k32):
    tot=0
    for g in G:
        slots={}
        for l in g:
            r=rowf(l); c=(l>>4)+4*k32
            a=r*128+((c^f(r))&7)*16
            s=(a//16)%16
            slots.setdefault(s,set()).add(a)
        tot+=max(len(v) for v in slots.values())-1
    return tot
# A rows: r = l&15 (+16*mb)
rowA=lambda l: l&15
# W rows with permutation: 8*(l15>>2)+4*b+(l15&3)
def rowW(b): return lambda l: 8*((l&15)>>2)+4*b+(l&3)
cands={}
for bits in itertools.product(range(6),repeat=3):
    f=lambda r,bits=bits: ((r>>bits[0])&1)|(((r>>bits[1])&1)<<1)|(((r>>bits[2])&1)<<2)
    ca=sum(conflicts(rowA,f,k) for k in (0,1))
    cw=sum(conflicts(rowW(b),f,k) for k in (0,1) for b in (0,1))
    cands[bits]=(ca,cw)
print("A free:",[b for b,v in cands.items() if v[0]==0][:10])
print("W free:",[b for b,v in cands.items() if v[1]==0][:10])
print("both:",[b for b,v in cands.items() if v==(0,0)][:10])
print("linear:", sum(conflicts(rowA,lambda r:0,k) for k in (0,1)))
