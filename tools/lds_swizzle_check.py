import itertools
def conflicts(f, log_t, s, stride=9, banks=32):
    # returns worst-case and average conflict degree over radix-4 steps
    Tn=1<<log_t; elems=1<<(s+log_t)
    worst=1; tot=0; cnt=0
    lvs=list(range(1,s,2))
    for lv in lvs:
        half=1<<(lv-1)
        for q0 in range(0, elems>>2, 32):
            for off in range(4):
                bs={}
                for q in range(q0,min(q0+32,elems>>2)):
                    t=q&(Tn-1); pi=q>>log_t; k=pi&(half-1); blk=pi>>(lv-1)
                    ia=(((blk<<(lv+1))+k)<<log_t)+t; st=half<<log_t
                    e=f(ia+off*st)
                    b=(stride*e)%banks
                    bs.setdefault(b,set()).add(e)
                d=max(len(v) for v in bs.values())
                worst=max(worst,d); tot+=d; cnt+=1
    return worst, tot/max(cnt,1)
cands={
 'id':lambda e:e,
 'pad5':lambda e:e+(e>>5),
 'x2':lambda e:e^((e>>2)&0x18)^0, 
 'xor_hi':lambda e: e ^ ((e>>5)&31),
 'xor_hi2':lambda e: e ^ (((e>>5)^(e>>7)^(e>>9))&31),
 'rot':lambda e: (e&~31)|(((e&31)+(e>>5)*1+(e>>7)*0)&31),
 'rot5':lambda e: (e&~31)|(((e&31)+(e>>5)*5)&31),
 'rot9':lambda e: (e&~31)|(((e&31)+(e>>5)*9)&31),
}
import sys
for name,f in cands.items():
    for stride in (1,9):
        res=[]
        for s in (5,6,7,8,9):
            log_t=10-s
            res.append(conflicts(f,log_t,s,stride))
        print(name,stride,[(w,round(a,2)) for w,a in res])
print('---')
for name,f in {'x2_1c':lambda e:e^((e>>2)&0x1c),'x2_1e':lambda e:e^((e>>2)&0x1e),'x2_1f':lambda e:e^((e>>2)&0x1f), 'x2x4':lambda e:e^((e>>2)&0x18)^((e>>4)&0x6), 'x2x4b':lambda e:e^((e>>2)&0x18)^((e>>5)&0x6),'x2x4c':lambda e:e^((e>>2)&0x18)^((e>>6)&0x6),'x2x4d':lambda e:e^((e>>2)&0x18)^((e>>4)&0x7)}.items():
    assert len(set(f(e) for e in range(1024)))==1024
    print(name,[(w,round(a,2)) for w,a in [conflicts(f,10-s,s,9) for s in (4,5,6,7,8,9)]])
