import itertools
GROUPS_B128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31],
               [32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59],[36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63]]
def nat(k):
    worst = 1
    for off in (0, 32):
      for s in range(4):
        for grp in GROUPS_B128:
            banks = {}
            for l in grp:
                r = off + (l & 31); c = 2*s + (l >> 5)
                b = (r & 1)*32 + ((c ^ k(r)) & 7)*4
                banks.setdefault(b, set()).add((r, c))
            worst = max(worst, max(len(v) for v in banks.values()))
    return worst
def tr(k):
    worst = 1
    for off in range(0, 64, 8):
      for cb in (0, 4):
        for half in (0, 1):
            banks = {}
            for l in range(32):
                L, G = l & 15, (l >> 4) & 1
                r = off + 4*half + (L >> 2); col = 8*cb + 16*G + 4*(L & 3)
                c = col >> 3; h8 = (col & 7)//4
                b = (r & 1)*32 + ((c ^ k(r)) & 7)*4 + h8*2
                banks.setdefault(b, set()).add((r, c, h8))
            worst = max(worst, max(len(v) for v in banks.values()))
    return worst
cur = lambda r: (r >> 1) & 7
print("current: nat", nat(cur), "tr", tr(cur))
good = []
for m in itertools.product(range(32), repeat=3):
    def k(r, m=m):
        x = (r >> 1) & 31
        return (bin(x & m[0]).count("1") & 1) | ((bin(x & m[1]).count("1") & 1) << 1) | ((bin(x & m[2]).count("1") & 1) << 2)
    if nat(k) == 1 and tr(k) == 1: good.append(m)
print(len(good), good[:12])
