import itertools
# LDS tile: row r (0..63), 128 B rows, chunk c (0..7) stored at position c ^ k(r); bank(dword) = (r&1)*32 + pos*4 + w
GROUPS_B128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31],
               [32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59],[36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63]]
def conflicts_nat(k):
    worst = 1
    for kb in range(4):
      for s in range(2):
        for grp in GROUPS_B128:
            banks = {}
            for l in grp:
                i, g = l & 15, l >> 4
                r = 16*kb + i; c = 4*s + g
                b = ((r & 1)*32 + ((c ^ k(r)) & 7)*4)
                banks.setdefault(b, set()).add((r, c))
            worst = max(worst, max(len(v) for v in banks.values()))
    return worst
def conflicts_tr(k):
    worst = 1
    for off in (0, 16, 32, 48):
      for db in range(4):
        for grp in (range(0,32), range(32,64)):
            banks = {}
            for l in grp:
                i, g = l & 15, l >> 4
                r = off + 4*g + (i >> 2); col = 16*db + 4*(i & 3)
                c = col >> 3; half = (col & 7)//4   # 8-byte half of the chunk
                b = (r & 1)*32 + ((c ^ k(r)) & 7)*4 + half*2
                banks.setdefault(b, set()).add((r, c, half))
            worst = max(worst, max(len(v) for v in banks.values()))
    return worst
cur = lambda r: (r >> 1) & 7
print("current: nat", conflicts_nat(cur), "tr", conflicts_tr(cur))
best = []
# k(r) = GF(2)-linear in row bits 1..5 -> 3 bits: matrix 3x5
for m in itertools.product(range(32), repeat=3):
    def k(r, m=m):
        x = (r >> 1) & 31
        return (bin(x & m[0]).count("1") & 1) | ((bin(x & m[1]).count("1") & 1) << 1) | ((bin(x & m[2]).count("1") & 1) << 2)
    a, b = conflicts_nat(k), conflicts_tr(k)
    if a == 1 and b == 1:
        best.append(m)
print(len(best), best[:10])
