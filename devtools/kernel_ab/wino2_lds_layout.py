import itertools
groups=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
groups+= [[l+32 for l in g] for g in groups]
def check(posf, swf, Mt=0):
    worst=1
    for hh in (0,1):
      for cr in range(0,4):      # patch row offset incl. group offset (0..3)
        for cc in range(0,4):
          for g in groups:
            seen={}
            for l in g:
                m=l>>1; p1=l&1; ty=m&7; tx=m>>3
                lr=2*ty+cr; lc=2*(4*Mt+tx)+cc
                part=2*hh+p1
                s=(posf(lr,lc)*4 + (part ^ swf(lr,lc)))%16
                seen[s]=seen.get(s,0)+1
            worst=max(worst,max(seen.values()))
    return worst
best=[]
for RS in range(18,26):
  for a in range(0,4):
    for md in (1,2,4):
      for swk in range(0,6):
        def posf(lr,lc,RS=RS,a=a,md=md): return lr*RS+lc+a*((lr>>1)%md)
        def swf(lr,lc,swk=swk):
            t=lr>>1
            return [0,t&3,(t>>1)&3,((t&1)<<1),((t>>1)&1)<<1, ((t>>2)&1)<<1|((t>>1)&1)][swk]
        w=max(check(posf,swf,0),check(posf,swf,1))
        best.append((w,RS,a,md,swk))
best.sort()
print(best[:10])
