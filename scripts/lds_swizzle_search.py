import itertools, sys
# lane groups of ds_read_b128 (guide)
groups = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
          list(range(4,12))+list(range(16,20))+list(range(28,32)),
          [32+x for x in list(range(0,4))+list(range(12,16))+list(range(20,28))],
          [32+x for x in list(range(4,12))+list(range(16,20))+list(range(28,32))]]
def conflicts(f, PPP=8):
    tot=0; worst=0
    for s3 in range(3):
        for hf in range(2):
            for lo in (0,4):
                for g in groups:
                    banks={}
                    for lane in g:
                        lr=lane&15; kg=lane>>4
                        hx=16*hf+lr+s3
                        p=hx*PPP+((kg^f(hx))^lo)
                        b=p%16
                        banks[b]=banks.get(b,0)+1
                    extra=max(banks.values())-1
                    tot+=extra; worst=max(worst,extra)
    return tot,worst
print("current", conflicts(lambda hx:(hx>>1)&7))
# search g over h=hx>>1 with period 8: g = permutation-ish table of 8 entries values 0..7
best=None
import random
for P in (8,16):
    cnt=0
    if P==8:
        it=itertools.product(range(8),repeat=8)
    else:
        it=None
    if it:
        for tab in it:
            t,w=conflicts(lambda hx:tab[(hx>>1)%8])
            if best is None or t<best[0]:
                best=(t,w,tab); print(best)
            if t==0: break
print("best",best)
# formula search (round 6): g(h) = (a h ^ b (h >> 1) ^ c (h >> 2) ^ d) & 7 or the same with +; the simplest hit is a = 2: f(hx) = hx & 6
found = []
for a, b, c, d in itertools.product(range(8), repeat=4):
    for name, fn in (("xor", lambda h: ((h * a) ^ ((h >> 1) * b) ^ ((h >> 2) * c) ^ d) & 7), ("add", lambda h: ((h * a) + ((h >> 1) * b) + ((h >> 2) * c) + d) & 7)):
        if conflicts(lambda hx: fn(hx >> 1))[0] == 0:
            found.append((name, a, b, c, d))
print(len(found), "conflict-free formulas; first:", found[:6], " hx & 6:", conflicts(lambda hx: hx & 6))
